# kernel-trace statistics of the training step: bash tools/r04_train_prof.sh <config.yml> <batch>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_train_prof && mkdir -p gpurun_out/r04_train_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_train_prof -- python3 tools/bench_train.py --config $1 --batch $2 --steps 3 > gpurun_out/r04_train_prof.log 2>&1
f=$(find gpurun_out/r04_train_prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r04_train_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
find gpurun_out/r04_train_prof -name "*_kernel_trace.csv" -delete
