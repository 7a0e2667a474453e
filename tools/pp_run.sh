timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "convt" > gpurun_out/cw_tests.log 2>&1; tail -5 gpurun_out/cw_tests.log | cut -c1-250
for cfg in "32 1024 512 16 16" "32 512 256 32 32"; do
  timeout -k 10 120 python tools/bench_layer.py convt $cfg --op wgrad --ab 3,1 --abvar UNET_CONVT_IMPL 2>&1 | grep -E "check|TFLOP"
done
