# A/B of conv3_dma_kernel (default) against the register-staged kernels (UNET_CONV_IMPL=3), same process
set -e
for shape in "32 128 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16" "32 256 128 128 128" "32 512 256 64 64"; do
  for op in fwd dgrad; do
    python3 tools/bench_layer.py conv $shape --iters 20 --op $op --ab 3,1 --abvar UNET_CONV_IMPL
  done
done
