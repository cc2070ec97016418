cd "$GRAFT_REPO_ROOT"
for shape in "--cin 256 --cout 256" "--cin 512 --cout 256" "--cin 128 --cout 256" "--cin 256 --cout 256 --H 64 --W 64" "--cin 512 --cout 256 --H 64 --W 64"; do
  for rep in 1 2; do
    for g in 0 4; do
      python tools/bench_conv.py --B 32 --H 128 --W 128 $shape --c16 --iters 30 --geom $g 2>/dev/null | grep TFLOP
    done
  done
done
