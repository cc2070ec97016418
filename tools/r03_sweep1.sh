set -e
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline --plan "$2" > gpurun_out/ab_tmp.json 2>/dev/null; echo "$1 plan[$2]: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step, %d dispatches' % (d['ms_per_step'], d.get('dispatches_per_step', -1)))")"; }
for w in cfg2 cfg3; do
  run $w ""
  run $w "12=1"
  run $w "30=384,31=512"
  run $w "30=300,31=512"
  run $w "30=192,31=384"
  run $w "30=192,31=512"
  run $w ""
done
