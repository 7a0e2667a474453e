#!/bin/bash
# memory-side counters of the 64-channel weight-stationary conv (and a pure streaming pass for comparison)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -E "TCC_EA0?_(WRREQ|RDREQ)|TCC_(HIT|MISS|REQ)|TCP_TCC|WRREQ_64B|RDREQ_32B|TCC_EA0_WR_UNCACHED|TCC_BUSY" | head -40 > gpurun_out/pmc_avail.txt
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum"; do
  tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcm_$tag -- python3 tools/bench_layer.py conv 32 64 64 256 256 --iters 5 --op fwdstats > gpurun_out/pmcm_$tag.log 2>&1 || echo "group failed: $grp"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmcm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]
        if "ws16" in k or "bn_" in k:
            e = acc[(k, r["Counter_Name"])]; e[0] += 1; e[1] += float(r["Counter_Value"])
for (k, c), (n, s) in sorted(acc.items()):
    print(f"{k:42s} {c:28s} {s / n:14.0f} per launch ({n})")
PY
cat gpurun_out/pmc_avail.txt | head -30
