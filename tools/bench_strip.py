#!/usr/bin/env python3
"""attn_strip_kernel in isolation (default: cfg2's 32 x 32 AttnBlockpp shape) against the unfused GEMM -> softmax -> GEMM path:
time per launch.  (The round-3 ablation figures quoted in profiles/README.md came from a development build of the kernel with
timing-only switches: no pass 1 / no pass 2 / loads without MFMAs.)

    python tools/bench_strip.py [--n 1024] [--d 512] [--B 32]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--lib", default="", help="a measurement build (tools/build_variant.py)")
    a = ap.parse_args()
    from text2protein_amd import _lib
    lib = _lib.load_path(os.path.abspath(a.lib)) if a.lib else _lib.load()
    B, n, d = a.B, a.n, a.d
    qk = torch.randn(B, n, 2 * d, device="cuda").half()
    vt = torch.randn(B, d, n, device="cuda").half()
    out = torch.empty(B, n, d, device="cuda", dtype=torch.float16)
    ws = torch.empty(lib.t2p_op_attention_ws(2, B, 1, n, n), dtype=torch.uint8, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731

    def run():
        _lib.check(lib.t2p_op_attention(2, P(qk), 2 * d, C.c_void_p(qk.data_ptr() + 2 * d), 2 * d, P(vt), n, P(out), B, 1, n, n, d,
                                        d ** -0.5, P(ws), None))
    for strip in (1, 0):
        lib.t2p_debug_set(29, strip)
        for m in [0]:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"{'strip kernel' if strip else 'GEMM + softmax + GEMM'}: {us:.1f} us per launch "
                  f"({4.0 * n * n * d * B / us / 1e6:.0f} TFLOP/s)", flush=True)
    lib.t2p_debug_set(29, 1)


if __name__ == "__main__":
    main()
