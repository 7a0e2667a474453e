# A/B of the streaming transposed-conv kernels against the generic igemm path (same box, same process)
set -e
for shape in "32 256 128 64 64" "32 128 64 128 128"; do
  for op in fwd dgrad; do
    python3 tools/bench_layer.py convt $shape --iters 20 --op $op --ab 0,1 --abvar UNET_CONVT_IMPL
  done
  python3 tools/bench_layer.py convt $shape --iters 20 --op wgrad
done
