"""In-kernel timeline of the dx-shared convolution kernel (the -DT2P_ABLATION build stamps s_memtime per tile):
where a tile's time goes -- set-up, first data, K loop, epilogue issue, store drain -- and how the tiles of a launch overlap.

    python tools/dxs_timeline.py --cin 128 --cout 128 [--B 32 --H 128 --W 128]
"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2protein_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32); ap.add_argument("--H", type=int, default=128); ap.add_argument("--W", type=int, default=128)
    ap.add_argument("--cin", type=int, default=128); ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--plan", default="")
    ap.add_argument("--tile", type=int, default=0, help="rows of a tile when the plan selects another kernel than the default (256: gemm_dxs2_kernel)")
    ap.add_argument("--dbg", type=int, default=0, help="ablation bits (1024: the epilogue's stores go nowhere)")
    a = ap.parse_args()
    lib = _lib.load_ablation()
    assert lib.t2p_debug_set(1, a.dbg) == 0
    for kv in filter(None, a.plan.split(",")):
        k, v = kv.split("="); assert lib.t2p_debug_set(int(k), int(v)) == 0
    td = torch.float16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(a.B, a.H, a.W, a.cin, device="cuda", generator=g).to(td)
    w = (torch.randn(a.cout, 9 * a.cin, device="cuda", generator=g) / (9 * a.cin) ** 0.5).to(td)
    b = torch.zeros(a.cout, device="cuda"); out = torch.empty(a.B, a.H, a.W, a.cout, device="cuda", dtype=td)
    P = lambda t: C.c_void_p(t.data_ptr())
    run = lambda: lib.t2p_op_conv3x3_shortcut(2, P(x), P(w), P(b), None, 0, None, 0, C.c_float(1.0), P(out), 0, a.B, a.H, a.W, a.cin, a.cout, None)
    for _ in range(5):
        assert run() == 0, lib.t2p_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    n = 4096 * 8
    buf = (C.c_ulonglong * n)()
    lib.t2p_ablation_dxs_stamps.restype = C.c_int
    got = lib.t2p_ablation_dxs_stamps(buf, n)
    assert got == n
    st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    st = st[st[:, 0] > 0]
    ntile = (a.B * a.H * a.W + 511) // 512 if a.cout <= 128 else (a.B * a.H * a.W // 256) * ((a.cout + 255) // 256)
    if a.tile:
        ntile = a.B * a.H * a.W // a.tile
    st = st[:min(ntile, len(st))]         # every launch of this process had the same grid: the stamps are the last launch's
    clk = np.median((st[:, 5] - st[:, 0]) / np.maximum(st[:, 7] - st[:, 6], 1)) * 100e6      # shader Hz
    names = ["set-up (entry -> first DMA issue)", "first stage lands + first K-tile", "K loop (rest)", "epilogue: compute + issue stores", "store drain"]
    d = np.diff(st[:, :6], axis=1)
    print(f"shape B{a.B} {a.H}x{a.W} {a.cin}->{a.cout} dbg {a.dbg}: {len(st)} tiles, launch {e0.elapsed_time(e1) * 1e3:.1f} us by events, shader clock {clk / 1e9:.2f} GHz")
    tot = (st[:, 5] - st[:, 0])
    for k, nm in enumerate(names):
        print(f"  {nm:38s} median {np.median(d[:, k]) / clk * 1e6:7.2f} us  ({np.median(d[:, k]) / np.median(tot) * 100:4.1f} % of a tile)   p90 {np.percentile(d[:, k], 90) / clk * 1e6:7.2f}")
    print(f"  tile total median {np.median(tot) / clk * 1e6:.2f} us; tiles per CU {len(st) / 256:.2f}; sum of medians x rounds {np.median(tot) / clk * 1e6 * np.ceil(len(st) / 256):.1f} us")


if __name__ == "__main__":
    main()
