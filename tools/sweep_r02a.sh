# epilogue / main-loop ablations of the LDS-DMA conv kernel on the shapes that dominate cfg3 and cfg2 (ablation library)
run() { python tools/bench_conv.py --ablation --iters 30 --ring 2 "$@" | grep TFLOP; }
echo "== cfg3 dominant: conv 128->128 @128x128 (512x128 tiles, 18 K-steps)"
for d in 0 1 1024 2048 3072 2 4 64; do run --cin 128 --cout 128 --dbg $d; done
echo "== same, forced 256x128 geometry"
for d in 0 1; do run --cin 128 --cout 128 --geom 1 --dbg $d; done
echo "== cfg3: conv 128->128 @64x64"
for d in 0 1; do run --cin 128 --cout 128 --H 64 --W 64 --dbg $d; done
echo "== cfg2 dominant: conv 256->256 @128x128"
for d in 0 1 1024 2048 3072 2 4 64; do run --dbg $d; done
