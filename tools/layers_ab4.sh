for shape in "32 128 128 128 128" "32 256 256 64 64" "32 512 512 32 32" "32 1024 512 32 32" "32 1024 1024 16 16"; do
  python3 tools/bench_layer.py conv $shape --iters 10 --op fwd --ab 0,1 --abvar UNET_CONV_VAR
done
