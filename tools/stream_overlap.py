#!/usr/bin/env python3
"""How much do the two decoder streams of AnomalyUNet overlap, and is the GPU ever idle?  Reads a rocprofv3 --kernel-trace CSV of a
few bench.py steps: time with >= 1 / >= 2 kernels resident, kernel time per queue, idle gaps (step boundary = adam_multi_kernel).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktrace -- python3 bench.py --steps 4 --warmup 2 --blocks 1 --no-cpu-baseline --no-roofline
    python tools/stream_overlap.py gpurun_out/ktrace
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_multi" in r[2]]
    if len(adam) < 3:
        print("need >= 3 steps in the trace")
        return
    lo, hi = adam[-3], adam[-1]                       # two whole steps
    seg = rows[lo + 1:hi + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    ev = []
    for s, e, _, _ in seg:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy1 = busy2 = 0
    depth, prev = 0, t0
    gaps = []
    for t, d in ev:
        if depth >= 1: busy1 += t - prev
        if depth >= 2: busy2 += t - prev
        if depth == 0 and t > prev: gaps.append(t - prev)
        depth += d; prev = t
    per_q = defaultdict(int)
    for s, e, _, q in seg:
        per_q[q] += e - s
    wall = t1 - t0
    print(f"two steps: wall {wall / 2e6:.3f} ms/step, >=1 kernel resident {busy1 / 2e6:.3f} ms/step ({100 * busy1 / wall:.1f} %), "
          f">=2 resident {busy2 / 2e6:.3f} ms/step ({100 * busy2 / wall:.1f} %), idle {100 * (wall - busy1) / wall:.1f} %")
    for q, t in sorted(per_q.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q}: kernel time {t / 2e6:.3f} ms/step")
    gaps.sort(reverse=True)
    print("  largest idle gaps (us):", [round(g / 1e3, 1) for g in gaps[:8]], f"; all gaps {sum(gaps) / 2e6:.3f} ms/step, {len(gaps) // 2} per step")


if __name__ == "__main__":
    main()
