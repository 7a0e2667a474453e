"""Where a training step's wall time goes that is NOT kernel execution: reads a `rocprofv3 --kernel-trace` CSV
(Start_Timestamp / End_Timestamp per dispatch), cuts it into steps at the `adam_multi_kernel` launches and reports, for the
steady-state steps, the step time, the union of busy intervals (at least one kernel running), the idle time, the number of
launches and the largest gaps with the kernels on either side.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 3 --blocks 1 \
        --no-cpu-baseline --no-roofline
    python tools/trace_gaps.py gpurun_out/trace
"""
import csv
import glob
import os
import sys
from collections import Counter


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for cut in ("(", "<"):
        i = name.find(cut)
        if i > 0 and not name.startswith("at::"):
            name = name[:i]
    return name[:60]


def main(path, top=12):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no kernel_trace.csv under {path}")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "adam_multi_kernel" in r[2]]
    if len(cuts) < 4:
        raise SystemExit("fewer than 4 optimiser launches in the trace")
    steps = [(cuts[i] + 1, cuts[i + 1] + 1) for i in range(1, len(cuts) - 1)]        # skip the first (warm-up) step
    tot = busy_tot = 0.0
    gaps = Counter()
    gap_n = Counter()
    launches = 0
    for a, b in steps:
        seg = rows[a:b]
        t0, t1 = rows[a - 1][1], max(r[1] for r in seg)      # from the end of the previous optimiser launch
        busy, (cur_s, cur_e, prev_name) = 0, (t0, t0, rows[a - 1][2])
        for s, e, name in seg:
            if s > cur_e:                           # nothing was running between cur_e and s
                busy += cur_e - cur_s
                key = (short(prev_name), short(name))
                gaps[key] += s - cur_e
                gap_n[key] += 1
                cur_s, cur_e, prev_name = s, e, name
            elif e > cur_e:
                cur_e, prev_name = e, name
        busy += cur_e - cur_s
        tot += t1 - t0
        busy_tot += busy
        launches += len(seg)
    n = len(steps)
    print(f"{n} steps: {tot / n / 1e6:.3f} ms per step, busy {busy_tot / n / 1e6:.3f} ms, idle {(tot - busy_tot) / n / 1e6:.3f} ms, "
          f"{launches / n:.0f} launches per step")
    print("largest idle gaps per step (us total, count per step, avg us): kernel before -> kernel after")
    for key, v in gaps.most_common(top):
        print(f"  {v / n / 1e3:8.1f} us  x{gap_n[key] / n:5.1f}  {v / gap_n[key] / 1e3:6.2f}   {key[0]}  ->  {key[1]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
