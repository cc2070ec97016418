#!/usr/bin/env python3
"""Experiment (not used by the product): two half-batches out of phase.

One captured graph: fork; stream 1 = score evaluation of chains 0..15 (engine A); stream 2 = a sleep of
`delay` GPU cycles, then the evaluation of chains 16..31 (engine B); join.  Replayed and timed against the
single-engine 32-chain evaluation, itself graph-replayed for a like-for-like comparison."""
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
        lab = torch.full((B,), 500, device="cuda", dtype=torch.int32)
        lab_i = lab
        out = torch.empty_like(x)
        return m, x, lab_i, out

    import ctypes as Ct
    from text2protein_amd._lib import check, ptr, stream_ptr

    def score(mo):       # raw C call: no tensor allocation inside the capture
        m, x, lab, out = mo
        check(m.lib.t2p_engine_score_ex(m._h, ptr(x), ptr(lab), None, ptr(out), x.shape[0], stream_ptr()))

    full, ha, hb = mk(32, 1), mk(16, 2), mk(16, 3)
    for mo in (full, ha, hb):                         # eager warm-up: fills the engines' activation pools
        score(mo)
    torch.cuda.synchronize()

    def timed(g, n=8):
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    cap = torch.cuda.Stream()
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, stream=cap):
        score(full)
    print(f"{wl}: one engine, 32 chains, graph replay: {timed(g0):7.2f} ms per score evaluation", flush=True)
    s2 = torch.cuda.Stream()
    for delay_ms in (0.0, 1.0, 2.0, 3.0, 4.0, 6.0):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            s2.wait_stream(torch.cuda.current_stream())
            score(ha)
            with torch.cuda.stream(s2):
                if delay_ms > 0:
                    torch.cuda._sleep(int(delay_ms * 2.0e6))      # ~cycles of a ~2 GHz clock
                score(hb)
            torch.cuda.current_stream().wait_stream(s2)
        print(f"{wl}: 2 x 16 chains, second stream delayed ~{delay_ms:.0f} ms: {timed(g):7.2f} ms", flush=True)
        del g


if __name__ == "__main__":
    main()
