# dx-shared-stage convolution kernel (plan switch 47) against the implicit-GEMM kernel on the dominant shapes, 16-bit output
cd "$GRAFT_REPO_ROOT"
for shape in "--cin 128 --cout 128" "--cin 256 --cout 256" "--cin 256 --cout 128" "--cin 512 --cout 256" "--cin 128 --cout 128 --H 64 --W 64" "--cin 256 --cout 256 --H 64 --W 64"; do
  for rep in 1 2; do
    for v in "--plan 47=0" "--plan 47=1"; do
      python tools/bench_conv.py --B 32 --H 128 --W 128 $shape --c16 --iters 30 $v 2>/dev/null | grep TFLOP
    done
  done
done
