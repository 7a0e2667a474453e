# A/B of the wgrad kernel variants (UNET_WGRAD_IMPL), same process
set -e
for shape in "32 64 64 256 256" "32 128 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16" "32 128 64 256 256"; do
  python3 tools/bench_layer.py conv $shape --iters 20 --op wgrad --ab ${WG_AB:-2,1} --abvar UNET_WGRAD_IMPL
done
