# fixed cost of a small LDS-DMA GEMM launch (back-to-back launches of the same shape: a lower bound of the in-network cost)
run() { python tools/bench_conv.py --iters 200 --ring 2 --taps 1 --c16 "$@" | grep TFLOP; }
run --B 1 --H 16 --W 16 --cin 64 --cout 64
run --B 1 --H 16 --W 16 --cin 512 --cout 512
run --B 32 --H 8 --W 8 --cin 512 --cout 512
run --B 32 --H 8 --W 8 --cin 512 --cout 1024
run --B 32 --H 8 --W 8 --cin 2048 --cout 512
run --B 32 --H 16 --W 16 --cin 512 --cout 512
run --B 32 --H 4 --W 4 --cin 512 --cout 512
python tools/bench_conv.py --iters 200 --ring 2 --B 32 --H 8 --W 8 --cin 512 --cout 512 | grep TFLOP
python tools/bench_conv.py --iters 200 --ring 2 --B 32 --H 4 --W 4 --cin 512 --cout 512 | grep TFLOP
python tools/bench_conv.py --iters 200 --ring 2 --B 32 --H 16 --W 16 --cin 512 --cout 512 | grep TFLOP
