# persistent LDS-DMA conv (default) vs the register-staged kernels (UNET_CONV_IMPL=3), same process
set -e
for shape in "32 128 64 256 256" "32 128 128 128 128" "32 256 128 128 128" "32 128 256 64 64" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16"; do
  for op in fwd dgrad; do
    python3 tools/bench_layer.py conv $shape --iters 20 --op $op --ab 3,1 --abvar UNET_CONV_IMPL 2>/dev/null
  done
done
