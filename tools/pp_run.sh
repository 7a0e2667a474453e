timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "convt_dgrad_fused" > gpurun_out/cb_tests.log 2>&1; tail -5 gpurun_out/cb_tests.log | cut -c1-250
tools/ab_bench.sh "UNET_FUSE_BN_CONVT=0" "UNET_FUSE_BN_CONVT=1" | tail -4 | cut -c1-330
