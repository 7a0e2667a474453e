#!/bin/bash
# same-process A/B of a tuning variable on conv forward / dgrad of representative layer shapes:  tools/conv_ab.sh VAR vals [ops]
var=${1:-UNET_PDMA_PRE}; vals=${2:-0,1}; ops=${3:-"fwd fwdstats dgrad"}
for shape in "32 128 128 128 128" "32 256 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 128 64 256 256"; do
  for op in $ops; do
    timeout -k 10 120 python3 tools/bench_layer.py conv $shape --iters 20 --op $op --ab $vals --abvar $var 2>&1 | grep -E "TFLOP|check" || exit 1
  done
done
