# A/B of one plan switch inside one call: bash tools/r03_ab.sh "28=0" "28=1" [workloads...]
set -e
cd "$GRAFT_REPO_ROOT"
A="$1"; B="$2"; shift 2
for w in ${@:-cfg2 cfg3}; do
  for rep in 1 2; do
    for plan in "$A" "$B"; do
      python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline --plan "$plan" > gpurun_out/ab_tmp.json 2>/dev/null
      echo "$w plan[$plan] rep$rep: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step, %d dispatches' % (d['ms_per_step'], d.get('dispatches_per_step', -1)))")"
    done
  done
done
