# LDS-halo conv kernel against the implicit-GEMM kernel on the dominant shapes (+ LDS bank-conflict counters)
run() { python tools/bench_conv.py --iters 30 --ring 2 "$@" | grep TFLOP; }
for h in 0 1 0 1; do run --halo $h; done
for h in 0 1 0 1; do run --halo $h --cin 128 --cout 128; done
for h in 0 1; do run --halo $h --cin 512 --cout 256; done
for h in 0 1; do run --halo $h --cin 256 --cout 256 --H 64 --W 64; done
for h in 0 1; do run --halo $h --cin 512 --cout 512 --H 32 --W 32; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 0 1; do
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_halo$h -- python3 tools/bench_conv.py --iters 5 --halo $h > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_halo$h/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:60]
    if "conv_halo" in k or "gemm_dma" in k:
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    print("halo$h", k, {c: round(x/n[(k,c)]) for c,x in v.items()})
PY
done
