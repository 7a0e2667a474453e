for shape in "32 256 128 64 64" "32 128 64 128 128"; do
  python3 tools/bench_layer.py convt $shape --iters 10 --ab 0,1 --abvar UNET_CONVT_IMPL
done
