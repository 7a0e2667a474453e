for cfg in "32 256 256 64 64" "32 512 256 64 64" "32 256 512 32 32" "32 128 256 64 64"; do
  for op in fwdstats dgrad; do
    timeout -k 10 120 python tools/bench_layer.py conv $cfg --op $op --ab 0,1 --abvar UNET_PDMA_PP 2>&1 | grep -E "TFLOP"
  done
done
