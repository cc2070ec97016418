# register epilogue specialised for the network's common case (A/B: debug bit 8192 = the general form), per launch and in the network
cd "$GRAFT_REPO_ROOT"
for shape in "--cin 128 --cout 128" "--cin 256 --cout 256" "--cin 256 --cout 128" "--cin 128 --cout 128 --H 64 --W 64"; do
  for rep in 1 2; do
    for d in 8192 0; do
      python tools/bench_conv.py --B 32 --H 128 --W 128 $shape --c16 --iters 30 --dbg $d 2>/dev/null | grep TFLOP
    done
  done
done
for w in cfg3 cfg5 cfg2; do
  for rep in 1 2; do
    for v in 8192 0; do
      echo "$w dbg=$v: $(python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-train --plan 1=$v 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["frac"])')"
    done
  done
done
