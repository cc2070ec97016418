#!/usr/bin/env python3
"""Short-K GEMMs of the transformer blocks in isolation, with warm and with cold weights.
A chain of `n` launches y_i = x W_i^T (each launch its own weight matrix, as in the network); timed with events around the
whole chain.  warm: the chain repeated (weights in L2 / MALL from the pass before); cold: a 2 GiB buffer is written between
passes, so every W_i comes from HBM.
    python tools/bench_small_gemm.py [--plan k=v,...]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", default="")
    ap.add_argument("--n", type=int, default=40)
    a = ap.parse_args()
    from text2protein_amd import _lib
    lib = _lib.load()
    if a.plan:
        _lib.set_plan_switches(a.plan)
    td = torch.float16
    P = lambda t: C.c_void_p(t.data_ptr())
    flush = torch.empty(1 << 29, device="cuda", dtype=torch.float32)
    print("M N K | warm us/launch | cold us/launch | warm TFLOP/s")
    for (M, N, K) in [(32768, 512, 512), (8192, 512, 512), (2048, 512, 512), (512, 512, 512), (32768, 1024, 512), (8192, 1024, 512),
                      (2048, 1024, 512), (32768, 4096, 512), (8192, 4096, 512), (2048, 4096, 512), (32768, 512, 2048), (8192, 512, 2048),
                      (2048, 512, 2048)]:
        n = a.n
        x = torch.randn(M, K, device="cuda").to(td)
        ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(td) for _ in range(n)]
        out = torch.empty(M, N, device="cuda", dtype=td)

        def chain():
            for w in ws:
                rc = lib.t2p_op_gemm(2, P(x), 0, P(w), P(out), 0, M, N, K, K, K, N, None, None, 1.0, None)
                assert rc == 0, lib.t2p_last_error()

        def timed(cold):
            best = 1e9
            for _ in range(4):
                if cold:
                    flush.fill_(1.0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); chain(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / n)
            return best

        chain()
        warm, cold = timed(False), timed(True)
        print(f"{M} {N} {K} | {warm:.1f} | {cold:.1f} | {2.0 * M * N * K / warm / 1e6:.0f}")


if __name__ == "__main__":
    main()
