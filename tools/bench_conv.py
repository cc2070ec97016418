#!/usr/bin/env python3
"""Micro-benchmark of one 3x3 convolution / GEMM shape through the C ABI (for rocprofv3 runs).
    python tools/bench_conv.py --B 32 --H 128 --W 128 --cin 256 --cout 256 --dtype f16 --iters 20
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--H", type=int, default=128)
    ap.add_argument("--W", type=int, default=128)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--taps", type=int, default=9)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--geom", type=int, default=0)
    ap.add_argument("--ring", type=int, default=2, help="t2p_debug_set(8, .): 2 = the 16x16x32 kernels with the register epilogue (what the engine runs), 1 = 32x32x16")
    ap.add_argument("--no-dma", action="store_true")
    ap.add_argument("--c16", action="store_true", help="compute-dtype output instead of fp32")
    ap.add_argument("--nsplit", type=int, default=0, help="force this split-K factor (conv only; attaches a workspace)")
    ap.add_argument("--plan", default="", help="further plan switches 'key=value,...' (t2p_debug_set)")
    ap.add_argument("--ablation", action="store_true", help="use the -DT2P_ABLATION library (needed for --dbg bits that skip work)")
    a = ap.parse_args()
    from text2protein_amd import _lib
    lib = _lib.load_ablation() if a.ablation else _lib.load()
    dt = _lib.DTYPE_NAMES[a.dtype]
    lib.t2p_debug_set(1, a.dbg)
    lib.t2p_debug_set(2, a.geom)
    lib.t2p_debug_set(8, a.ring)
    lib.t2p_debug_set(0, 0 if a.no_dma else 1)
    for kv in filter(None, a.plan.split(",")):
        k, v = kv.split("=")
        assert lib.t2p_debug_set(int(k), int(v)) == 0, kv
    if a.nsplit:
        lib.t2p_debug_set(10, 1024)
        lib.t2p_debug_set(11, a.nsplit)
    td = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(a.B, a.H, a.W, a.cin, device="cuda", generator=g).to(td)
    w = (torch.randn(a.cout, a.taps * a.cin, device="cuda", generator=g) / (a.taps * a.cin) ** 0.5).to(td)
    b = torch.zeros(a.cout, device="cuda")
    out = torch.empty(a.B, a.H, a.W, a.cout, device="cuda", dtype=td if a.c16 else torch.float32)
    P = lambda t: C.c_void_p(t.data_ptr())

    def run():
        if a.taps == 9 and a.c16:     # 16-bit output, as inside the network (the shortcut operator without a shortcut segment)
            rc = lib.t2p_op_conv3x3_shortcut(dt, P(x), P(w), P(b), None, 0, None, 0, C.c_float(1.0), P(out), 0, a.B, a.H, a.W, a.cin, a.cout, None)
        elif a.taps == 9:
            rc = lib.t2p_op_conv3x3(dt, P(x), int(dt == 0), P(w), P(b), P(out), a.B, a.H, a.W, a.cin, a.cout, 0, None)
        else:
            M = a.B * a.H * a.W
            rc = lib.t2p_op_gemm(dt, P(x), int(dt == 0), P(w), P(out), 0 if a.c16 else 1, M, a.cout, a.cin, a.cin, a.cin, a.cout, P(b), None, 1.0, None)
        assert rc == 0, lib.t2p_last_error()

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    c16 = " c16" if a.c16 else ""
    fl = 2.0 * a.B * a.H * a.W * a.cout * a.taps * a.cin
    print(f"{a.dtype} B{a.B} {a.H}x{a.W} cin{a.cin} cout{a.cout} taps{a.taps} dbg{a.dbg} geom{a.geom} ring{a.ring} nsplit{a.nsplit}{c16} plan[{a.plan}]: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
