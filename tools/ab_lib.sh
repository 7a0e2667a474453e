#!/bin/bash
# same-box A/B of two builds of the library on single layers: tools/ab_lib.sh <base.so> "<bench_layer args>" ...
base=$1; shift
for cfg in "$@"; do
  for rnd in 1 2; do
    echo -n "base: "; UNET_HIP_LIB=$base timeout -k 10 120 python tools/bench_layer.py $cfg 2>&1 | grep TFLOP
    echo -n "new : "; timeout -k 10 120 python tools/bench_layer.py $cfg 2>&1 | grep TFLOP
  done
done
