# usage: tools/ab_bench.sh "<plan a>" "<plan b>" [workload] -- alternates a/b/a/b inside one call (bench.py --plan "key=value,...")
A="$1"; B="$2"; W=${3:-cfg2}
for d in "$A" "$B" "$A" "$B"; do
  python bench.py --workload $W --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --plan "$d" 2>&1 | python -c "import sys,json; [print('bench [$d] $W', round(json.loads(l)['ms_per_step'],2)) for l in sys.stdin if l.startswith('{')]"
done
