#!/bin/bash
# like wgrad_ab.sh on a representative subset of the shapes
var=${1:-UNET_WGRAD_IMPL}; vals=${2:-1,3}; op=${3:-wgrad}
for shape in "32 128 64 256 256" "32 128 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16"; do
  timeout -k 10 120 python3 tools/bench_layer.py conv $shape --iters 20 --op $op --ab $vals --abvar $var 2>&1 | grep -E "TFLOP|check" || exit 1
done
