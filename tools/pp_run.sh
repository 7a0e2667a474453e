export UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_stamps.so
UNET_PDMA_PP=1 timeout -k 10 120 python tools/pdma_stamps.py 32 1024 1024 16 16 2>&1 | grep -v amdgpu.ids
unset UNET_HIP_LIB
timeout -k 10 200 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "adam or freshness" 2>&1 | tail -3 | cut -c1-200
