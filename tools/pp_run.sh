timeout -k 10 120 python tools/bench_layer.py conv 2 64 64 32 32 --iters 2 > gpurun_out/w16_first.log 2>&1 || { echo "first run failed/hung"; tail -5 gpurun_out/w16_first.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/w16_tests.log 2>&1; tail -12 gpurun_out/w16_tests.log | cut -c1-300
for op in fwd fwdstats dgrad; do
  timeout -k 10 120 python tools/bench_layer.py conv 32 64 64 256 256 --op $op --ab 3,1 --abvar UNET_WS_MFMA 2>&1 | grep -E "check|TFLOP"
done
timeout -k 10 120 python tools/bench_layer.py conv 32 64 64 256 256 --op dgrad --acc 1 --ab 3,1 --abvar UNET_WS_MFMA 2>&1 | grep -E "check|TFLOP"
timeout -k 10 120 python tools/bench_layer.py conv 32 64 128 128 128 --op fwdstats --ab 3,1 --abvar UNET_WS_MFMA 2>&1 | grep -E "check|TFLOP"
