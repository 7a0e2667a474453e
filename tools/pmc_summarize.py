#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel HBM traffic (profiles/rNN_pmc_traffic.json).

    tools/pmc_summarize.py <FETCH_SIZE dir> <WRITE_SIZE dir> <MFMA dir|-> <out.json>

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE
are reported in KiB; FETCH_SIZE counts 128-byte requests at 64 bytes for wide (16 B/lane) streaming reads, so it is
DOUBLED; WRITE_SIZE is exact for 16-byte-per-lane stores.  traffic = 2 * FETCH_SIZE + WRITE_SIZE, averaged per launch.
bench.py reads the file to fill roofline.traffic for its dominant kernel."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I", name)
    if m:
        return m.group(1)
    return name.split("(")[0].strip()


def load(d, counters):
    out = {c: defaultdict(lambda: [0.0, 0]) for c in counters}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c in out:
                e = out[c][short(r["Kernel_Name"])]
                e[0] += float(r["Counter_Value"])
                e[1] += 1
    return out


def main():
    fdir, wdir, mdir, dst = sys.argv[1:5]
    fetch = load(fdir, ["FETCH_SIZE"])["FETCH_SIZE"]
    write = load(wdir, ["WRITE_SIZE"])["WRITE_SIZE"]
    mf = load(mdir, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]) if mdir != "-" else None
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(k, [0.0, 0])
        ws, wn = write.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        rec = {"launches": n, "fetch_kib_per_launch_raw": round(fs / max(fn, 1), 1),
               "write_kib_per_launch": round(ws / max(wn, 1), 1),
               "traffic_bytes_per_launch": int((2.0 * fs / max(fn, 1) + ws / max(wn, 1)) * 1024)}
        if mf is not None and k in mf["SQ_VALU_MFMA_BUSY_CYCLES"] and mf["GRBM_GUI_ACTIVE"].get(k, [0, 0])[0] > 0:
            busy = mf["SQ_VALU_MFMA_BUSY_CYCLES"][k][0]
            act = mf["GRBM_GUI_ACTIVE"][k][0]
            rec["mfma_busy_frac"] = round(busy / (act / 8.0 * 1024.0), 4)      # 1024 SIMDs; GRBM summed over 8 XCDs
        kernels[k] = rec
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha16
    json.dump({"csrc_sha16": csrc_sha16(), "commit": os.environ.get("UNET_COMMIT", "unknown"),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes of `python3 bench.py "
                         "--steps 3 --warmup 2 --blocks 1 --no-cpu-baseline --no-roofline` (tools/pmc_traffic.sh), "
                         "UNET_TWO_STREAMS=0; traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per launch",
               "kernels": dict(top)}, open(dst, "w"), indent=1)
    for k, v in top[:14]:
        print(f"{k[:44]:44s} launches {v['launches']:5d}  traffic/launch {v['traffic_bytes_per_launch'] / 1e6:8.1f} MB"
              + (f"  mfma busy {100 * v['mfma_busy_frac']:.1f}%" if "mfma_busy_frac" in v else ""))


if __name__ == "__main__":
    main()
