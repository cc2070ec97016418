# extra evidence of round 4: sustained cfg3 run, cfg3 counters of its dominant convolution kernel, cfg3 at 64 chains, 2-rank rehearsal on one card
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --workload cfg3 --steps 1000 --warmup 5 --no-cpu-baseline --no-f32 > gpurun_out/bench_cfg3_1000.json 2>/dev/null
# (optional workload of the round-3 review: cfg3 at 64 chains per GPU -- a DIFFERENT workload than BASELINE configs[2], never the headline: how much of
# an evaluation is batch-independent latency)
python bench.py --workload cfg3 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-f32 > gpurun_out/bench_cfg3_b64.json 2>/dev/null
# two ranks on this one card over gloo: the N > 1 control flow of bench.py with its cfg3 / cfg5 keys and per-rank times (NOT a scaling measurement)
T2P_FORCE_DEVICE=0 T2P_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err || echo "2-rank rehearsal failed"
B="python3 bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-f32"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r04c3_sq1 -- $B > gpurun_out/pmc_c3_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_r04c3_tcc -- $B > gpurun_out/pmc_c3_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r04c3_fetch -- $B > gpurun_out/pmc_c3_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r04c3_write -- $B > gpurun_out/pmc_c3_w.log 2>&1
python tools/summarize_pmc.py --tag r04_cfg3 --out gpurun_out/profiles_r04 gpurun_out/pmc_r04c3_sq1 gpurun_out/pmc_r04c3_tcc
python - <<'PY'
import csv, glob, collections, json
out = {}
for kind, d in (("fetch", "gpurun_out/pmc_r04c3_fetch"), ("write", "gpurun_out/pmc_r04c3_write")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].split("(")[0].replace("void t2p::", "").replace("t2p::", "")
            acc[nm][0] += 1; acc[nm][1] += float(r["Counter_Value"])
    for nm, (c, v) in acc.items():
        out.setdefault(nm, {})[kind + "_kb_per_launch"] = v / c
        out[nm]["launches_" + kind] = c
for nm, t in out.items():
    t["hbm_bytes_per_launch"] = (2.0 * t.get("fetch_kb_per_launch", 0.0) + t.get("write_kb_per_launch", 0.0)) * 1024.0
json.dump(out, open("gpurun_out/profiles_r04/r04_cfg3_pmc_traffic.json", "w"), indent=1, sort_keys=True)
PY
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
echo done
