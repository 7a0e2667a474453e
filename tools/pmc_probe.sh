set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in "32 512 512 32 32" "32 64 64 256 256" "32 1024 512 32 32"; do
  python3 tools/bench_layer.py conv $shape --iters 20
  python3 tools/bench_layer.py conv $shape --iters 20 --op wgrad
done
python3 tools/bench_layer.py convt 32 128 64 128 128 --iters 20
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  for tag in "a:32 512 512 32 32" "b:32 64 64 256 256"; do
    t=${tag%%:*}; shape=${tag#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_${t}_$i -- python3 tools/bench_layer.py conv $shape --iters 5 > gpurun_out/pmc_${t}_$i.log 2>&1
  done
done
ls gpurun_out
