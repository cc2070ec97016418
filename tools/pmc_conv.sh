# SQ counters of one conv shape, LDS-halo kernel on / off:  bash tools/pmc_conv.sh [bench_conv args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 0 1; do
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
rm -rf gpurun_out/pmc_tmp
rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -- python3 tools/bench_conv.py --iters 5 --ring 2 --halo $h "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_tmp/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void t2p::","")[:48]
    if "conv_halo" in k or "gemm_dma" in k:
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    print("halo$h", k, {c: round(x/n[(k,c)]/1e6,2) for c,x in v.items()}, "(x1e6 per launch)")
PY
done; done
rm -rf gpurun_out/pmc_tmp
