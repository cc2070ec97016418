run() { python tools/bench_conv.py --iters 30 --ring 2 "$@" | grep TFLOP; }
echo "== ff1-like M=32768 N=4096 K=512 c16"
for d in 0 1 1024 2048 3072; do run --taps 1 --B 32 --H 32 --W 32 --cin 512 --cout 4096 --c16 --dbg $d; done
echo "== ff1-like f32 out"
for d in 0 1024; do run --taps 1 --B 32 --H 32 --W 32 --cin 512 --cout 4096 --dbg $d; done
echo "== conv 256->256 128x128"
for d in 0 1 1024 3072; do run --dbg $d; done
echo "== proj M=131072 N=256 K=256"
for d in 0 1 1024 3072; do run --taps 1 --B 32 --H 64 --W 64 --cin 256 --cout 256 --dbg $d; done
