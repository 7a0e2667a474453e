timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/t17.log 2>&1; tail -3 gpurun_out/t17.log | cut -c1-250
tools/ab_lib.sh tiaozhanbei_unet_amd/libunet_hip_base.so "conv 32 128 128 128 128 --op fwdstats" "conv 32 256 256 64 64 --op fwdstats" "conv 32 64 64 256 256 --op fwdstats"
for i in 1 2; do
echo -n "base: "; UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_base.so python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
echo -n "new : "; python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
done
