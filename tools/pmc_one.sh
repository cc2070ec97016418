# LDS counters of one bench_conv.py shape: bash tools/pmc_one.sh --cin 256 --cout 256 [...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_tmp -- python3 tools/bench_conv.py --iters 5 --c16 "$@" > gpurun_out/pmc_tmp.log 2>&1
grep TFLOP gpurun_out/pmc_tmp.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_tmp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-60:]
    if "gemm_d" in k:
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()): print(k, c, round(v / n[(k, c)] / 1e6, 2))
PY
rm -rf gpurun_out/pmc_tmp
