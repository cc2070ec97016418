# short-K GEMMs of the transformer blocks at the 32x32 level of cfg2 (M = 32768): tile geometry sweep
run() { python tools/bench_conv.py --iters 50 --ring 2 --taps 1 --B 32 --H 32 --W 32 --c16 "$@" | grep TFLOP; }
for shape in "--cin 512 --cout 512" "--cin 512 --cout 1024" "--cin 512 --cout 4096" "--cin 2048 --cout 512"; do
  for g in 0 1 2 3 4; do run $shape --geom $g; done
done
echo "== 16x16 level (M = 8192)"
run2() { python tools/bench_conv.py --iters 50 --ring 2 --taps 1 --B 32 --H 16 --W 16 --c16 "$@" | grep TFLOP; }
for shape in "--cin 512 --cout 512" "--cin 512 --cout 1024" "--cin 512 --cout 4096" "--cin 2048 --cout 512"; do
  for g in 0 1 2 3; do run2 $shape --geom $g; done
done
