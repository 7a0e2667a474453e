#!/bin/bash
# HBM traffic of every kernel of the benchmark step from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace domains beside them.
#   tools/pmc_traffic.sh [extra bench.py args]      -> gpurun_out/pmc_${TAG}_{FETCH_SIZE,WRITE_SIZE,MFMA}/ + profiles/${TAG}_pmc_traffic.json
set -e
TAG=${PMC_TAG:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  UNET_TWO_STREAMS=0 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}_$ctr -- python3 bench.py --steps 3 --warmup 2 --blocks 1 --no-cpu-baseline --no-roofline "$@" > gpurun_out/pmc_${TAG}_$ctr.log 2>&1
done
UNET_TWO_STREAMS=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_${TAG}_MFMA -- python3 bench.py --steps 3 --warmup 2 --blocks 1 --no-cpu-baseline --no-roofline "$@" > gpurun_out/pmc_${TAG}_MFMA.log 2>&1
python3 tools/pmc_summarize.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/pmc_${TAG}_MFMA profiles/${TAG}_pmc_traffic.json
cp profiles/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json     # (only gpurun_out/ travels back from the GPU box)
