set -e
run() { python tools/bench_conv.py --iters 30 "$@" | grep TFLOP; }
echo "== M=8192 N=512 cin512"
for cfg in "2 0" "2 2" "1 2" "1 3" "3 4" "3 6" "3 3"; do set -- $cfg; run --B 32 --H 16 --W 16 --cin 512 --cout 512 --geom $1 --nsplit $2 --ring 2; done
echo "== M=8192 N=512 cin1024"
for cfg in "2 0" "2 2" "1 2" "1 3" "3 4" "3 6"; do set -- $cfg; run --B 32 --H 16 --W 16 --cin 1024 --cout 512 --geom $1 --nsplit $2 --ring 2; done
echo "== M=2048 N=512 cin512"
for cfg in "2 6" "2 4" "1 8" "1 12" "3 16" "3 12"; do set -- $cfg; run --B 32 --H 8 --W 8 --cin 512 --cout 512 --geom $1 --nsplit $2 --ring 2; done
echo "== M=32768 N=512 cin512"
for cfg in "3 0" "3 2" "1 0"; do set -- $cfg; run --B 32 --H 32 --W 32 --cin 512 --cout 512 --geom $1 --nsplit $2 --ring 2; done
