#!/bin/bash
# same-box A/B of several builds of the library on single layers: tools/ab_libs.sh "<lib1> <lib2> ..." "<bench_layer args>" ...
libs=$1; shift
for cfg in "$@"; do
  for rnd in 1 2; do
    for lib in $libs; do
      echo -n "$(basename $lib): "; UNET_HIP_LIB=$lib timeout -k 10 120 python tools/bench_layer.py $cfg 2>&1 | grep TFLOP
    done
  done
done
