# register epilogue specialised for the network's common case (A/B: debug bit 8192 = the general form), in the network
cd "$GRAFT_REPO_ROOT"
for w in cfg2 cfg3 cfg4; do
  for rep in 1 2; do
    for v in 8192 0; do
      echo "$w dbg=$v: $(python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-train --plan 1=$v 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["other_gemm"]["frac"])')"
    done
  done
done
