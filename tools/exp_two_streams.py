#!/usr/bin/env python3
"""Experiment (not used by the product): do two half-batches on two HIP streams overlap?

Two engines (own activation pools, replicated weights), 16 chains each, one score evaluation per stream:
sequential on one stream vs concurrent on two streams vs one engine with all 32 chains."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from text2protein_amd import synth
    from text2protein_amd.config import load_config
    from text2protein_amd.model import HipScoreModel
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    fname = {"cfg2": "test_config.yml", "cfg3": "cond_length.yml"}[wl]
    cfg = load_config(os.path.join(ROOT, "configs", fname), **{"data.max_res_num": 128, "model.num_scales": 1000})
    cfg.device = "cuda:0"
    sd = synth.synth_state_dict(cfg, 0)
    C = cfg.data.num_channels
    def mk(B, seed):
        m = HipScoreModel(cfg, dtype="f16")
        m.load_state_dict(sd)
        ctx = synth.synth_context(B, 512, cfg.model.context_dim, seed).cuda()
        m.set_context(ctx)
        x = torch.randn(B, C, 128, 128, device="cuda") * 50
        lab = torch.full((B,), 500, device="cuda", dtype=torch.long)
        return m, x, lab
    full = mk(32, 1)
    ha, hb = mk(16, 2), mk(16, 3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(fn, n=6):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def one32():
        full[0](full[1], full[2])

    def seq16():
        with torch.cuda.stream(s1):
            ha[0](ha[1], ha[2])
            hb[0](hb[1], hb[2])

    def par16():
        with torch.cuda.stream(s1):
            ha[0](ha[1], ha[2])
        with torch.cuda.stream(s2):
            hb[0](hb[1], hb[2])

    print(f"{wl}: one engine, 32 chains      : {run(one32):7.2f} ms per score evaluation", flush=True)
    print(f"{wl}: 2 x 16 chains, one stream  : {run(seq16):7.2f} ms", flush=True)
    print(f"{wl}: 2 x 16 chains, two streams : {run(par16):7.2f} ms", flush=True)


if __name__ == "__main__":
    main()
