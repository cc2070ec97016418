#!/usr/bin/env python3
"""Micro-benchmark of t2p_op_st_entry (GroupNorm -> proj_in -> LayerNorm -> q|k|v in one launch): python tools/bench_st_entry.py --B 32 --n 256"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--stats", type=int, default=1)
    ap.add_argument("--form", default="entry", choices=["entry", "tail3"])
    ap.add_argument("--lib", default="")
    ap.add_argument("--timing", action="store_true", help="library built with -DSF_TIMING: print workgroup 0's phase stamps")
    a = ap.parse_args()
    from text2protein_amd import _lib
    lib = _lib.load_path(os.path.abspath(a.lib)) if a.lib else _lib.load()
    Cc, G = 256, 32
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(a.B, a.n, Cc, device="cuda", generator=g).half()
    cs = torch.randn(a.B * a.n // 64, Cc, 2, device="cuda", generator=g).abs() * 64 + 100
    gamma = torch.ones(Cc, device="cuda"); beta = torch.zeros(Cc, device="cuda")
    w_in = (torch.randn(Cc, Cc, device="cuda", generator=g) / 16).half()
    w_qkv = (torch.randn(3 * Cc, Cc, device="cuda", generator=g) / 16).half()
    b_in = torch.zeros(Cc, device="cuda")
    t = torch.empty(a.B, a.n, Cc, device="cuda", dtype=torch.float16)
    qkv = torch.empty(a.B, a.n, 3 * Cc, device="cuda", dtype=torch.float16)
    P = lambda v: C.c_void_p(v.data_ptr())

    w_ff1 = (torch.randn(8 * Cc, Cc, device="cuda", generator=g) / 16).half()
    b_ff1 = torch.zeros(8 * Cc, device="cuda")
    w_3 = (torch.randn(Cc, 5 * Cc, device="cuda", generator=g) / 36).half()
    y = torch.empty(a.B, a.n, Cc, device="cuda", dtype=torch.float16)
    ys = torch.empty(a.B * a.n // 64, Cc, 2, device="cuda")

    def run():
        if a.form == "tail3":
            rc = lib.t2p_op_st_entry(2, P(x), None, G, None, None, 1e-6, P(w_in), P(b_in), P(t), P(gamma), P(beta), 1e-5, P(w_ff1), 8 * Cc, P(b_ff1), 1,
                                     P(t), P(qkv), P(w_3), P(b_in), P(x), P(y), P(ys), a.B, a.n, Cc, None)
            assert rc == 0, lib.t2p_last_error()
            return
        rc = lib.t2p_op_st_entry(2, P(x), P(cs) if a.stats else None, G, P(gamma), P(beta), 1e-6, P(w_in), P(b_in), None, P(gamma), P(beta), 1e-5,
                                 P(w_qkv), 3 * Cc, None, 0, P(t), P(qkv), None, None, None, None, None, a.B, a.n, Cc, None)
        assert rc == 0, lib.t2p_last_error()

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    if a.timing:
        st = qkv.view(torch.int64).flatten()[:8].cpu().tolist()
        print("phase stamps (us from start): " + ", ".join(f"{(v - st[0]) / 100.0:.2f}" for v in st[1:8]) + "  (the last: end of the second product)")
    print(f"st_entry B{a.B} n{a.n} stats{a.stats}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per launch (back to back)", flush=True)


if __name__ == "__main__":
    main()
