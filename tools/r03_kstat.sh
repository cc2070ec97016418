# kernel-trace statistics of a short bench run for kernels matching a pattern: bash tools/r03_kstat.sh <workload> "<plan>" "<grep -E pattern>"
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kstat && mkdir -p gpurun_out/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstat -- python3 bench.py --workload $1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --plan "$2" > gpurun_out/kstat.log 2>&1
f=$(find gpurun_out/kstat -name "*kernel_stats.csv" | head -1)
python - "$f" "$3" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY
find gpurun_out/kstat -name "*_kernel_trace.csv" -delete
