timeout -k 10 120 env UNET_WS_SPREAD=1 python tools/bench_layer.py conv 2 64 64 32 32 --iters 2 > gpurun_out/ws_first.log 2>&1 || { echo "first run failed/hung"; tail -5 gpurun_out/ws_first.log; exit 1; }
UNET_WS_SPREAD=1 UNET_WS_ST=0 timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3 or dgrad or big or stat or two" > gpurun_out/ws_tests.log 2>&1; tail -2 gpurun_out/ws_tests.log | cut -c1-250
for sp in 0 1; do
for op in fwd fwdstats dgrad; do
  echo "spread=$sp"; UNET_WS_SPREAD=$sp timeout -k 10 120 python tools/bench_layer.py conv 32 64 64 256 256 --op $op --ab 0,1 --abvar UNET_WS_ST 2>&1 | grep -E "TFLOP|check"
done
done
