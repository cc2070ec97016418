# incremental issue state + shortcut-in-convolution: tests, then in-call A/B (head library / fused / unfused) at cfg2 and cfg3
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r02e_ops.log 2>&1 || { tail -30 gpurun_out/r02e_ops.log; exit 1; }
tail -2 gpurun_out/r02e_ops.log
python -m pytest tests/test_gpu_baseline.py tests/test_gpu_forward.py -x -q -s -k "not thousand and not hundred" > gpurun_out/r02e_base.log 2>&1 || { tail -30 gpurun_out/r02e_base.log; exit 1; }
grep "rel-L2\|passed\|failed" gpurun_out/r02e_base.log | tail -12
b() { python bench.py --workload $1 --steps 8 --warmup 3 --no-cpu-baseline --no-f32 $2 $3 $4 $5 2>gpurun_out/r02e_err.log | python -c "import sys,json; [print('bench $1 $2 $3 $4 $5', round(json.loads(l)['ms_per_step'],2), json.loads(l)['roofline']['achieved']) for l in sys.stdin if l.startswith('{')]"; }
for r in 1 2; do
  for w in cfg2 cfg3; do
    b $w --lib tools/_ab/libt2p_head.so
    b $w --plan 23=1
    b $w --plan 23=0
  done
done
