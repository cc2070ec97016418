#!/usr/bin/env python3
"""Error of the f16 engine against the exact-f32 engine after every block of ONE score evaluation at full size (t2p_debug_tap):
where along the network the per-evaluation error of the benchmarked precision builds up.

    python tools/exp_f16_layers.py cond_length > gpurun_out/f16_layers_cond_length.md
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import FULL, full_inputs, load_golden, rel_l2   # noqa: E402


def main():
    from text2protein_amd import _lib, synth
    from text2protein_amd.arch import build_arch
    from text2protein_amd.config import load_config
    from text2protein_amd.model import HipScoreModel
    lib = _lib.load()
    stem = sys.argv[1] if len(sys.argv) > 1 else "cond_length"
    fname, L, N, B, T, _ = FULL[stem]
    cfg = load_config(os.path.join(ROOT, "configs", fname), **{"data.max_res_num": L, "model.num_scales": N})
    cfg.device = "cuda:0"
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = (t.cuda() for t in full_inputs(cfg, B, T))
    layers = list(build_arch(cfg).all_layers())
    cap = B * L * L * max(l.out_ch for l in layers)
    bufs = {dt: torch.zeros(cap, device="cuda") for dt in ("f32", "f16")}
    models = {}
    for dt in ("f32", "f16"):
        models[dt] = HipScoreModel(cfg, dtype=dt)
        models[dt].load_state_dict(sd)
    ref = torch.from_numpy(load_golden("full_" + stem)["score"])
    print(f"# f16 engine vs exact-f32 engine after every block: {stem}, labels {labels.tolist()} (sigma ~ 97 / ~ 0.16)\n")
    print("| # | block | kind | side | C | sample 0 (label 3) | sample 1 (label 700) |\n|---|---|---|---|---|---|---|")
    sh = (C.c_int64 * 4)()
    for i, l in enumerate(layers):
        taps = {}
        for dt in ("f32", "f16"):
            _lib.check(lib.t2p_debug_tap(i, C.c_void_p(bufs[dt].data_ptr()), cap, None))
            out = models[dt](x, labels, ctx)
            _lib.check(lib.t2p_debug_tap(-1, None, 0, sh))
            Cc, H, W = int(sh[0]), int(sh[1]), int(sh[2])
            taps[dt] = bufs[dt][: B * H * W * Cc].reshape(B, H * W, Cc).cpu()
        e = [rel_l2(taps["f16"][b], taps["f32"][b]) for b in range(B)]
        print(f"| {i} | {l.prefix} | {l.kind}{' up' if l.up else ' down' if l.down else ''} | {H} | {Cc} | {e[0]:.2e} | {e[1]:.2e} |")
    for dt in ("f32", "f16"):
        out = models[dt](x, labels, ctx).cpu()
        print(f"\nscore, {dt} engine vs the reference: " + ", ".join(f"{rel_l2(out[b], ref[b]):.3e}" for b in range(B)))


if __name__ == "__main__":
    main()
