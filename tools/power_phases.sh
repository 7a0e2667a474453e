#!/bin/bash
# What does each PHASE of the step draw?  rocm-smi (power, sclk) sampled while one kernel family loops (tools/power_phase.py).
#   tools/power_phases.sh   -> gpurun_out/power_phases.log
out=gpurun_out/power_phases.log
: > $out
rocm-smi --showmaxpower 2>&1 | grep -E "Max Graphics" >> $out
for phase in ${PHASES:-idle bn_apply bn_bwd conv64 conv128 conv512 wgrad512}; do
  echo "=== $phase" >> $out
  python3 tools/power_phase.py $phase --seconds 7 >> $out 2>/dev/null &
  pid=$!
  sleep 3.5
  for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed -E 's/GPU\[0\]\s*: //' | tr '\n' ' ' >> $out
    echo >> $out
    sleep 0.3
  done
  wait $pid
done
cat $out
