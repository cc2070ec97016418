# A/B of two library builds inside one call: bash tools/r03_lib_ab.sh <old.so> [workloads]   (the new one is the in-tree library)
set -e
cd "$GRAFT_REPO_ROOT"
OLD="$1"; shift
for w in ${@:-cfg2 cfg3}; do
  for rep in 1 2; do
    for lib in "$OLD" ""; do
      python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline ${lib:+--lib $lib} > gpurun_out/ab_tmp.json 2>/dev/null
      echo "$w lib[${lib:-new}] rep$rep: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step, %d dispatches' % (d['ms_per_step'], d.get('dispatches_per_step', -1)))")"
    done
  done
done
