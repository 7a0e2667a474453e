# PMC counters of the end-of-round kernels: persistent LDS-DMA conv (512->512 @32x32) and the wgrad kernel.
# Separate passes per counter group (rocprofv3 --pmc, no trace domains), as MI355X_MICROARCH.md prescribes.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for tag in "f:fwd" "w:wgrad"; do
    t=${tag%%:*}; op=${tag#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc3_${t}_$i -- python3 tools/bench_layer.py conv 32 512 512 32 32 --iters 5 --op $op > gpurun_out/pmc3_${t}_$i.log 2>&1
  done
done
ls gpurun_out | grep pmc3
