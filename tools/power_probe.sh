#!/bin/bash
# Is the training step power-limited?  Samples rocm-smi (power, sclk, temperature) while bench.py runs.
#   tools/power_probe.sh            -> gpurun_out/power_probe.log
out=gpurun_out/power_probe.log
: > $out
rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>&1 | grep -v "^$" | head -40 >> $out
python3 bench.py --steps 400 --warmup 5 --blocks 3 --no-cpu-baseline --no-roofline > gpurun_out/power_probe_bench.json 2>/dev/null &
pid=$!
sleep 9
for i in $(seq 1 10); do
  echo "--- sample $i" >> $out
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -8 >> $out
  sleep 0.4
done
wait $pid
head -c 200 gpurun_out/power_probe_bench.json >> $out
cat $out
