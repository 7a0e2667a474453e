#!/bin/bash
# stamps of the weight-gradient kernels on the benchmark's layer shapes (needs `make stamps`)
export UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_stamps.so
for shape in "32 128 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16" "32 128 64 256 256"; do
  for impl in 1 3; do UNET_WGRAD_IMPL=$impl timeout -k 10 120 python3 tools/wgrad_stamps.py $shape || exit 1; done
done
