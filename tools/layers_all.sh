# every distinct conv3x3 layer shape of AnomalyUNet bs=32 256x256, all three operators (bf16)
for shape in "32 64 64 256 256" "32 128 64 256 256" "32 64 128 128 128" "32 128 128 128 128" "32 256 128 128 128" "32 128 256 64 64" "32 256 256 64 64" "32 512 256 64 64" "32 256 512 32 32" "32 512 512 32 32" "32 1024 512 32 32" "32 512 1024 16 16" "32 1024 1024 16 16"; do
  for op in fwd dgrad wgrad; do python3 tools/bench_layer.py conv $shape --iters 20 --op $op 2>/dev/null; done
done
