# usage: tools/ab_bench.sh "<T2P_DEBUG a>" "<T2P_DEBUG b>" [workload] -- alternates a/b/a/b inside one call
A="$1"; B="$2"; W="${3:-cfg2}"
for d in "$A" "$B" "$A" "$B"; do
  T2P_DEBUG="$d" python bench.py --workload $W --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json; [print('bench [$d] $W', round(json.loads(l)['ms_per_step'],2)) for l in sys.stdin if l.startswith('{')]"
done
