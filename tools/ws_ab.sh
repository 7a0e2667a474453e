#!/bin/bash
# same-process A/B of UNET_WS_STG (0 lock-step, 1 opposite-ends DMA, 2 + deferred stores) on the 64-input-channel layers
vals=${1:-1,2}
for shape in "32 64 64 256 256" "32 64 128 256 256"; do
  for op in fwd fwdstats dgrad; do
    timeout -k 10 120 python3 tools/bench_layer.py conv $shape --iters 20 --op $op --ab $vals --abvar UNET_WS_STG 2>&1 | grep -E "TFLOP|check" || exit 1
  done
done
