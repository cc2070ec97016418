# Official per-round measurement pass (run on the GPU box through gpurun; outputs under gpurun_out/).
#   bash tools/collect_profiles.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
echo "[1/7] default bench"; python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "[2/7] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- $B > gpurun_out/prof_$TAG.log 2>&1
echo "[3/7] FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1
echo "[4/7] WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -- $B > gpurun_out/pmc_write.log 2>&1
echo "[5/7] SQ pass 1"; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_sq1 -- $B > gpurun_out/pmc_sq1.log 2>&1
echo "[6/7] SQ pass 2"; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${TAG}_sq2 -- $B > gpurun_out/pmc_sq2.log 2>&1
echo "[7/7] TCC"; rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${TAG}_tcc -- $B > gpurun_out/pmc_tcc.log 2>&1
find gpurun_out -name "*_kernel_trace.csv" -path "*pmc_*" -delete    # keep the merge-back under the size cap
echo done
