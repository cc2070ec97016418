# implicit-GEMM vs LDS-halo convolution on cfg3's dominant shape, 16-bit output as inside the network (ablation library)
run() { python tools/bench_conv.py --ablation --iters 30 --ring 2 --c16 "$@" | grep TFLOP; }
for h in 0 1; do
  echo "== conv 128->128 @128x128 halo=$h: full / no epilogue / stores dropped / no DMA in loop / DMA only"
  for d in 0 1 1024 2 4; do run --cin 128 --cout 128 --halo $h --dbg $d; done
done
for h in 0 1; do
  echo "== conv 256->256 @128x128 halo=$h"
  for d in 0 1 1024 2 4; do run --halo $h --dbg $d; done
done
