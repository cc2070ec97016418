run() { python tools/bench_conv.py --iters 30 --taps 1 --ring 2 "$@" | grep TFLOP; }
echo "== ff1-like M=32768 N=4096 K=512 c16"
for d in 0 1 2 4; do run --B 32 --H 32 --W 32 --cin 512 --cout 4096 --c16 --dbg $d; done
for g in 1 2; do run --B 32 --H 32 --W 32 --cin 512 --cout 4096 --c16 --geom $g; done
echo "== proj M=32768 N=512 K=512 f32 out"
for d in 0 1 2 4; do run --B 32 --H 32 --W 32 --cin 512 --cout 512 --dbg $d; done
for g in 1 2; do run --B 32 --H 32 --W 32 --cin 512 --cout 512 --geom $g; done
echo "== proj M=8192 N=512 K=512 f32 out"
for d in 0 1 2 4; do run --B 32 --H 16 --W 16 --cin 512 --cout 512 --dbg $d; done
for g in 1 3; do run --B 32 --H 16 --W 16 --cin 512 --cout 512 --geom $g; done
echo "== ff2 M=32768 N=512 K=2048"
for d in 0 1 2 4; do run --B 32 --H 32 --W 32 --cin 2048 --cout 512 --dbg $d; done
echo "== proj M=131072 N=256 K=256"
for d in 0 1; do run --B 32 --H 64 --W 64 --cin 256 --cout 256 --dbg $d; done
for g in 1 2; do run --B 32 --H 64 --W 64 --cin 256 --cout 256 --geom $g; done
