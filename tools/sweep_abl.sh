run() { python tools/bench_conv.py --iters 30 --ring 2 "$@" | grep TFLOP; }
echo "== conv 256->256 @128x128 (2048 tiles)"
for d in 0 1 3 5 7; do run --dbg $d; done
echo "== conv 512->256 @128x128"
for d in 0 1 3 5; do run --cin 512 --dbg $d; done
echo "== conv 512->512 @32x32 (256 tiles)"
for d in 0 1 3 5; do run --cin 512 --cout 512 --H 32 --W 32 --dbg $d; done
