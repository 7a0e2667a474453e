python3 tools/bench_layer.py conv 32 64 64 256 256 --iters 10 --op fwd --ab 0,2,1
python3 tools/bench_layer.py conv 32 64 128 256 256 --iters 10 --op fwd --ab 0,2,1
python3 tools/bench_layer.py conv 32 64 128 128 128 --iters 10 --op fwd --ab 0,2,1
