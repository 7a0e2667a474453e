#!/bin/bash
# SQ instruction counters of one conv layer under several builds of the library (evidence for the address-arithmetic diet):
#   tools/pmc_valu_diet.sh "<lib1> <lib2>" "<bench_layer args>" ...
libs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1))
  for lib in $libs; do
    tag=$(basename $lib .so)_$i
    UNET_HIP_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcd_$tag -- python3 tools/bench_layer.py $cfg --iters 5 > gpurun_out/pmcd_$tag.log 2>&1 || echo "failed: $tag"
  done
done
python3 - "$libs" "$@" <<'PY'
import csv, glob, collections, os, sys
libs = sys.argv[1].split(); cfgs = sys.argv[2:]
for i, cfg in enumerate(cfgs, 1):
    print("##", cfg)
    for lib in libs:
        tag = os.path.basename(lib)[:-3] + f"_{i}"
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
        for f in glob.glob(f"gpurun_out/pmcd_{tag}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                if not k.startswith("conv3_"): continue
                k = k.split("(")[0]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            v = {c: x / max(cnt[(k, c)], 1) for c, x in d.items()}
            extra = f"  (VALU-MFMA)/MFMA={(v.get('SQ_INSTS_VALU', 0) - v.get('SQ_INSTS_MFMA', 0)) / max(v.get('SQ_INSTS_MFMA', 1), 1):.2f}"
            print(f"{os.path.basename(lib):28s} {k:28s} " + "  ".join(f"{c[3:]}={x:.4g}" for c, x in sorted(v.items())) + extra)
PY
