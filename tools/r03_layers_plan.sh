# per-block tables under two plan settings: bash tools/r03_layers_plan.sh "<k=v>" "<k=v>" [workload] [grep pattern]
set -e
cd "$GRAFT_REPO_ROOT"
W=${3:-cfg2}; PAT=${4:-"res"}
for plan in "$1" "$2"; do
  python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --plan "$plan" --layers gpurun_out/layers_ab.csv > /dev/null 2>&1
  echo "== plan $plan"; python tools/layer_table.py gpurun_out/layers_ab.csv | grep -E "^\| ($PAT)[a-z_]* \| (4|8) \|"
done
