# several library builds inside one call: bash tools/r03_libs_ab.sh "<lib1> <lib2> ..." [workloads]   ("-" = the in-tree library)
set -e
cd "$GRAFT_REPO_ROOT"
LIBS="$1"; shift
for w in ${@:-cfg2 cfg3}; do
  for rep in 1 2; do
    for lib in $LIBS; do
      L=""; [ "$lib" != "-" ] && L="--lib $lib"
      python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline $L > gpurun_out/ab_tmp.json 2>/dev/null
      echo "$w lib[$lib] rep$rep: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step' % d['ms_per_step'])")"
    done
  done
done
