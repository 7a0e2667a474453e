# per-layer micro-benchmarks of the MFMA kernels (bf16), AnomalyUNet bs=32 256x256 shapes
for shape in "32 64 64 256 256" "32 128 64 256 256" "32 128 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16"; do
  for op in fwd dgrad wgrad; do python3 tools/bench_layer.py conv $shape --iters 20 --op $op; done
done
for shape in "32 1024 512 16 16" "32 512 256 32 32" "32 256 128 64 64" "32 128 64 128 128"; do
  python3 tools/bench_layer.py convt $shape --iters 20
done
