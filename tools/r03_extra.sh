# extra evidence of round 3: sustained cfg3 run, hipGraph replay A/B, cfg3 counters of its dominant convolution kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --workload cfg3 --steps 1000 --warmup 5 --no-cpu-baseline --no-f32 > gpurun_out/bench_cfg3_1000.json 2>/dev/null
for w in cfg2 cfg3; do for g in 0 1 0 1; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-roofline --graph $g > gpurun_out/ab_tmp.json 2>/dev/null
  echo "$w graph=$g: $(python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('%.3f ms/step' % d['ms_per_step'])")"
done; done
B="python3 bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-f32"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r03c3_sq1 -- $B > gpurun_out/pmc_c3_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_r03c3_tcc -- $B > gpurun_out/pmc_c3_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r03c3_fetch -- $B > gpurun_out/pmc_c3_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r03c3_write -- $B > gpurun_out/pmc_c3_w.log 2>&1
python tools/summarize_pmc.py --tag r03_cfg3 --out gpurun_out/profiles_r03 gpurun_out/pmc_r03c3_sq1 gpurun_out/pmc_r03c3_tcc
python - <<'PY'
import csv, glob, collections, json
out = {}
for kind, d in (("fetch", "gpurun_out/pmc_r03c3_fetch"), ("write", "gpurun_out/pmc_r03c3_write")):
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
json.dump(out, open("gpurun_out/profiles_r03/r03_cfg3_pmc_traffic.json", "w"), indent=1, sort_keys=True)
PY
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
echo done
