timeout -k 10 120 python tools/bench_layer.py convt 2 512 256 8 8 --iters 2 > gpurun_out/cg_first.log 2>&1 || { echo "first run failed/hung"; tail -5 gpurun_out/cg_first.log; exit 1; }
tail -1 gpurun_out/cg_first.log
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "convt" > gpurun_out/cg_tests.log 2>&1; tail -5 gpurun_out/cg_tests.log | cut -c1-250
for cfg in "32 1024 512 16 16" "32 512 256 32 32"; do
  for op in fwd dgrad; do
    timeout -k 10 120 python tools/bench_layer.py convt $cfg --op $op --ab 2,1 --abvar UNET_CONVT_IMPL 2>&1 | grep -E "check|TFLOP"
  done
done
