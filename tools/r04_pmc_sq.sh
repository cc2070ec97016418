# SQ / TCC counter passes of bench.py at cfg2 and cfg3 (separate --pmc passes with --kernel-trace only): bash tools/r04_pmc_sq.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in cfg2 cfg3; do
  B="python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-f32"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r04b_${w}_sq1 -- $B > gpurun_out/pmc_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_r04b_${w}_sq2 -- $B > gpurun_out/pmc_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_r04b_${w}_tcc -- $B > gpurun_out/pmc_tcc.log 2>&1
  tag=r04; [ $w = cfg3 ] && tag=r04_cfg3
  python tools/summarize_pmc.py --tag $tag --out gpurun_out/profiles_r04 gpurun_out/pmc_r04b_${w}_sq1 gpurun_out/pmc_r04b_${w}_sq2 gpurun_out/pmc_r04b_${w}_tcc
done
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
echo done
