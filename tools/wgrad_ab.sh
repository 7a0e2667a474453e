#!/bin/bash
# same-process A/B of a tuning variable on the conv3x3 weight gradient of every layer shape of the benchmark:
#   tools/wgrad_ab.sh [VAR] [values]        (default UNET_WGRAD_XCD 0,1)
var=${1:-UNET_WGRAD_XCD}; vals=${2:-0,1}; op=${3:-wgrad}
for shape in "32 64 64 256 256" "32 128 64 256 256" "32 64 128 128 128" "32 128 128 128 128" "32 256 128 128 128" "32 128 256 64 64" "32 256 256 64 64" "32 512 256 64 64" "32 256 512 32 32" "32 512 512 32 32" "32 1024 512 32 32" "32 512 1024 16 16" "32 1024 1024 16 16"; do
  timeout -k 10 120 python3 tools/bench_layer.py conv $shape --iters 20 --op $op --ab $vals --abvar $var 2>&1 | grep -E "TFLOP|check" || exit 1
done
