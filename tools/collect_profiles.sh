# Official per-round measurement pass (run on the GPU box through gpurun; outputs under gpurun_out/).
#   bash tools/collect_profiles.sh r04
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32"
echo "[1/10] default bench"; python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "[2/10] sustained: one complete 1000-step run"; python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-f32 > gpurun_out/bench_1000.json 2> gpurun_out/bench_1000.err
echo "[3/10] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- $B > gpurun_out/prof_$TAG.log 2>&1
echo "[4/10] FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1
echo "[5/10] WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -- $B > gpurun_out/pmc_write.log 2>&1
echo "[6/10] SQ pass 1"; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_sq1 -- $B > gpurun_out/pmc_sq1.log 2>&1
echo "[7/10] SQ pass 2"; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_sq2 -- $B > gpurun_out/pmc_sq2.log 2>&1
echo "[8/10] TCC"; rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${TAG}_tcc -- $B > gpurun_out/pmc_tcc.log 2>&1
echo "[9/10] cfg3 kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_cfg3 -- $B --workload cfg3 > gpurun_out/prof_${TAG}_cfg3.log 2>&1
echo "[10/10] other workloads, shapes, layers"
for w in cfg3 cfg4 cfg5; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-f32 > gpurun_out/bench_$w.json 2>/dev/null; done
for w in cfg2 cfg3; do
  python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --shapes gpurun_out/shapes_$w.csv --layers gpurun_out/layers_$w.csv > /dev/null 2>&1
  python tools/shapes_table.py gpurun_out/shapes_$w.csv --top 80 > gpurun_out/${TAG}_shapes_$w.md
  python tools/layer_table.py gpurun_out/layers_$w.csv > gpurun_out/${TAG}_layers_$w.md
done
python tools/summarize_profiles.py --tag $TAG --stats gpurun_out/prof_$TAG --fetch gpurun_out/pmc_${TAG}_fetch --write gpurun_out/pmc_${TAG}_write --steps-in-trace 8 --out gpurun_out/profiles_$TAG
python tools/summarize_profiles.py --tag $TAG --stats gpurun_out/prof_${TAG}_cfg3 --steps-in-trace 8 --workload cfg3 --out gpurun_out/profiles_$TAG
python tools/summarize_pmc.py --tag $TAG --out gpurun_out/profiles_$TAG gpurun_out/pmc_${TAG}_sq1 gpurun_out/pmc_${TAG}_sq2 gpurun_out/pmc_${TAG}_tcc
find gpurun_out -name "*_kernel_trace.csv" -delete    # keep the merge-back under the size cap
find gpurun_out -name "*counter_collection.csv" -delete
echo done
