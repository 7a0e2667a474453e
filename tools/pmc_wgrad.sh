#!/bin/bash
# SQ counters of the weight-gradient kernels (old 32x32x16 LDS-DMA kernel vs wgrad16_kernel) on one layer shape:
#   tools/pmc_wgrad.sh "32 512 512 32 32"
set -e
shape=${1:-"32 512 512 32 32"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for impl in 1 3; do
  UNET_WGRAD_IMPL=$impl rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmcw_$impl -- python3 tools/bench_layer.py conv $shape --iters 5 --op wgrad > gpurun_out/pmcw_$impl.log 2>&1
  UNET_WGRAD_IMPL=$impl rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d gpurun_out/pmcw2_$impl -- python3 tools/bench_layer.py conv $shape --iters 5 --op wgrad > gpurun_out/pmcw2_$impl.log 2>&1 || true
done
python3 - <<'PY'
import csv, glob, collections
for impl in (1, 3):
    for grp in ("pmcw", "pmcw2"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
        for f in glob.glob(f"gpurun_out/{grp}_{impl}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "wgrad" not in k or "reduce" in k: continue
                k = "wgrad16" if "wgrad16" in k else "wgrad_dma"
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(f"impl={impl} {k}: " + "  ".join(f"{c}={v / max(cnt[(k, c)], 1):.4g}" for c, v in sorted(d.items())))
PY
