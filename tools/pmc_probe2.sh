set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  for tag in "a:32 512 512 32 32" "b:32 64 64 256 256"; do
    t=${tag%%:*}; shape=${tag#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc2_${t}_$i -- python3 tools/bench_layer.py conv $shape --iters 5 > gpurun_out/pmc2_${t}_$i.log 2>&1 || true
  done
done
