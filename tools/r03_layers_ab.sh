set -e
cd "$GRAFT_REPO_ROOT"
OLD="$1"
for lib in "$OLD" ""; do
  python bench.py --workload cfg2 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 ${lib:+--lib $lib} --layers gpurun_out/layers_ab.csv > /dev/null 2>&1
  echo "== lib ${lib:-new}"; python tools/layer_table.py gpurun_out/layers_ab.csv | grep -E "^\| (attn|st) \|"
done
