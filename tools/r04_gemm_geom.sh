# 1x1 (attention / MLP projection) GEMM shapes of cfg2 at C = 512 under each tile geometry (plan switch 2: 0 planner, 1 256x128, 2 128x128, 3 256x256, 4 512x128)
cd "$GRAFT_REPO_ROOT"
for shape in "--H 32 --W 32 --cin 512 --cout 512" "--H 32 --W 32 --cin 512 --cout 1536" "--H 32 --W 32 --cin 512 --cout 4096" "--H 32 --W 32 --cin 2560 --cout 512" "--H 16 --W 16 --cin 512 --cout 512" "--H 16 --W 16 --cin 512 --cout 4096" "--H 8 --W 8 --cin 512 --cout 512"; do
  for g in 0 1 2 3; do
    python tools/bench_conv.py --B 32 $shape --taps 1 --c16 --iters 50 --geom $g 2>/dev/null | grep TFLOP
  done
done
