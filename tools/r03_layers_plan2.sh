# per-block rows matching a pattern under several plan settings: bash tools/r03_layers_plan2.sh "<grep -E pattern>" <workload> "<k=v>" ...
set -e
cd "$GRAFT_REPO_ROOT"
PAT="$1"; W="$2"; shift 2
for plan in "$@"; do
  python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --plan "$plan" --layers gpurun_out/layers_ab.csv > /dev/null 2>&1
  echo "== $W plan $plan"; python tools/layer_table.py gpurun_out/layers_ab.csv | grep -E "$PAT"
done
