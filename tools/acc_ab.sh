for shape in "32 128 64 256 256" "32 256 128 128 128" "32 512 256 64 64" "32 1024 512 32 32" "32 64 64 256 256"; do
  for acc in 0 1; do python3 tools/bench_layer.py conv $shape --iters 20 --op dgrad --acc $acc 2>/dev/null; done
done
