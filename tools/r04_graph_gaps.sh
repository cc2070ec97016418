# eager launches against hipGraph replay at one workload: inter-kernel gaps from rocprofv3 kernel traces (bash tools/r04_graph_gaps.sh cfg3)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-cfg3}
for g in 0 1; do
  rm -rf gpurun_out/gaps_$g && mkdir -p gpurun_out/gaps_$g
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps_$g -- python3 bench.py --workload $W --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline --graph $g > gpurun_out/gaps_$g.log 2>&1
  f=$(find gpurun_out/gaps_$g -name "*kernel_trace.csv" | head -1)
  python tools/gap_hist.py "$f" "$W graph=$g"
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/gaps_$g.log | head -1
  rm -rf gpurun_out/gaps_$g
done
