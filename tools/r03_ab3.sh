# A/B/C of plan settings inside one call: bash tools/r03_ab3.sh "37=0" "37=1" "37=2" -- [workloads...]
set -e
cd "$GRAFT_REPO_ROOT"
PLANS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do PLANS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for w in ${@:-cfg2 cfg3}; do
  for rep in 1 2; do
    for plan in "${PLANS[@]}"; do
      python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline --plan "$plan" > gpurun_out/ab_tmp.json 2>/dev/null
      echo "$w plan[$plan] rep$rep: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step, %d dispatches' % (d['ms_per_step'], d.get('dispatches_per_step', -1)))")"
    done
  done
done
