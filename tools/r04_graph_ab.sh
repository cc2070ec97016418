# eager launches against hipGraph replay, no profiler attached, alternating inside one call: bash tools/r04_graph_ab.sh cfg3 [cfg2 ...]
cd "$GRAFT_REPO_ROOT"
for w in ${@:-cfg3}; do
  for rep in 1 2 3; do
    for g in 0 1; do
      python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline --graph $g > gpurun_out/ab_tmp.json 2>/dev/null
      echo "$w graph=$g rep$rep: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step' % d['ms_per_step'])")"
    done
  done
done
