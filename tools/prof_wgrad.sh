cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "a:32 64 64 256 256" "b:32 512 512 32 32" "c:32 1024 512 32 32" "d:32 128 128 128 128"; do
  t=${tag%%:*}; shape=${tag#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pw_$t -- python3 tools/bench_layer.py conv $shape --iters 10 --op wgrad > /dev/null 2>&1
  echo "== $shape"; cat gpurun_out/pw_$t/*/*kernel_stats.csv | cut -c1-60,60-200 | awk -F, '{print substr($1,1,50), $2, $4}' | head -6
done
