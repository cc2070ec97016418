# does a second resident workgroup hide the epilogue?  128x128 tiles with the 2-stage ring (64 KiB: two workgroups per CU) against
# the 512x128 plan on cfg3's dominant shape, 16-bit output (ablation library)
run() { python tools/bench_conv.py --ablation --iters 30 --ring 2 --c16 "$@" 2>/dev/null | grep TFLOP; }
for g in 0 2 1; do
  echo "== conv 128->128 @128x128 geom=$g: full / no epilogue / no DMA in loop / DMA only"
  for d in 0 1 2 4; do run --cin 128 --cout 128 --geom $g --dbg $d; done
done
