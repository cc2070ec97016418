# per-block time breakdown of one PC step at cfg2 / cfg3 / cfg5 (bench.py --layers), plus the default bench line
set -e
cd "$GRAFT_REPO_ROOT"
for w in cfg2 cfg3 cfg5; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --layers gpurun_out/layers_$w.csv --shapes gpurun_out/shapes_$w.csv > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  python tools/layer_table.py gpurun_out/layers_$w.csv > gpurun_out/layers_$w.md
  echo "$w: $(python -c "import json;d=json.load(open('gpurun_out/bench_$w.json'));print(d['ms_per_step'], d['value'])")"
done
