#!/bin/bash
# same-box A/B of bench.py under env toggles: tools/ab_bench.sh "<VAR=VAL ...>" "<VAR=VAL ...>" ...   (interleaved, 2 rounds)
out=gpurun_out/ab_bench.log
: > $out
for round in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    echo "== round $round cfg[$i]: $cfg" >> $out
    env $cfg python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], 'img/s', d['ms_per_step'], 'ms', json.dumps(r['per_class_ms_per_step']))
" >> $out
  done
done
cat $out
