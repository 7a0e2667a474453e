timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/t18.log 2>&1; tail -2 gpurun_out/t18.log | cut -c1-250
tools/ab_lib.sh tiaozhanbei_unet_amd/libunet_hip_base.so "conv 32 128 128 128 128 --op fwd" "conv 32 512 512 32 32 --op fwd" "conv 32 256 256 64 64 --op dgrad"
for i in 1 2; do
echo -n "base: "; UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_base.so python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
echo -n "new : "; python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
done
