# tile geometry of the mid-size convolutions of cfg3 (16-bit output; --geom: 0 plan, 1 256x128x3, 2 128x128, 3 256x256, 4 512x128)
run() { python tools/bench_conv.py --iters 50 --ring 2 --c16 "$@" 2>/dev/null | grep TFLOP; }
echo "== 128->128 @64x64 (cfg3 64^2 level)"; for g in 0 1 2 4; do run --H 64 --W 64 --cin 128 --cout 128 --geom $g; done
echo "== 256->256 @32x32 (cfg3 32^2 level)"; for g in 0 1 2 3; do run --H 32 --W 32 --cin 256 --cout 256 --geom $g; done
echo "== 256->256 @64x64 (cfg2 64^2 level)"; for g in 0 1 2 3; do run --H 64 --W 64 --cin 256 --cout 256 --geom $g; done
echo "== 512->512 @32x32 (cfg2 32^2 level)"; for g in 0 1 2 3; do run --H 32 --W 32 --cin 512 --cout 512 --geom $g; done
