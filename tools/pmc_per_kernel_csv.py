#!/usr/bin/env python3
"""Per-kernel sums of the rocprofv3 --pmc passes (the committed form of the raw counter_collection CSVs, which are too big to keep):
    python tools/pmc_per_kernel_csv.py gpurun_out/pmc_r03_FETCH_SIZE gpurun_out/pmc_r03_WRITE_SIZE gpurun_out/pmc_r03_MFMA profiles/r03_pmc_csv
-> <out>/<pass>_per_kernel.csv with columns kernel, counter, dispatches, sum, mean_per_dispatch."""
import collections
import csv
import glob
import os
import sys


def main():
    *dirs, out = sys.argv[1:]
    os.makedirs(out, exist_ok=True)
    for d in dirs:
        acc = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"][:90], r["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
        name = os.path.basename(d.rstrip("/")).split("_")[-1]
        if name == "SIZE":
            name = "_".join(os.path.basename(d.rstrip("/")).split("_")[-2:])
        with open(os.path.join(out, f"{name}_per_kernel.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
            for (kern, ctr), (n, s) in sorted(acc.items()):
                w.writerow([kern, ctr, n, round(s, 1), round(s / n, 1)])


if __name__ == "__main__":
    main()
