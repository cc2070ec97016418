#!/usr/bin/env python3
"""Where does the f16 engine's per-evaluation error come from?  One score evaluation of the full-size fixtures
(tests/golden/full_<stem>.npz, outputs of the reference) under individual plan switches, error per fixture sample
(label 3: sigma ~ 97; label 700: sigma ~ 0.16).

    python tools/exp_f16_error.py [cond_length test_config]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import FULL, full_inputs, load_golden, rel_l2   # noqa: E402


def main():
    from text2protein_amd import _lib, synth
    from text2protein_amd.config import load_config
    from text2protein_amd.model import HipScoreModel
    lib = _lib.load()
    stems = sys.argv[1:] or ["cond_length", "test_config"]
    plans = [("default", []), ("fp32 residual stream (14=0)", [(14, 0)]), ("fp32 h1 (9=0)", [(9, 0)]),
             ("fp32 stream + h1", [(14, 0), (9, 0)]), ("input conv on FMA (26=0)", [(26, 0)]), ("unfused attention (5=0)", [(5, 0)]),
             ("GEGLU unfused (7=0)", [(7, 0)]), ("head conv on GEMM tile (15=0)", [(15, 0)]), ("separate shortcut (23=0)", [(23, 0)]),
             ("gather up-conv (21=0)", [(21, 0)]), ("no GN stats fusion (6=0)", [(6, 0)]),
             ("fp32 stream + h1 + no stats fusion", [(14, 0), (9, 0), (6, 0)])]
    for stem in stems:
        fname, L, N, B, T, _ = FULL[stem]
        cfg = load_config(os.path.join(ROOT, "configs", fname), **{"data.max_res_num": L, "model.num_scales": N})
        cfg.device = "cuda:0"
        g = load_golden("full_" + stem)
        ref = torch.from_numpy(g["score"])
        sd = synth.synth_state_dict(cfg, 0)
        x, labels, ctx = (t.cuda() for t in full_inputs(cfg, B, T))
        print(f"== {stem}: labels {labels.tolist()}, |score| rms per sample {[float(ref[i].pow(2).mean().sqrt()) for i in range(B)]}")
        for dt in ("f32", "bf16"):
            m = HipScoreModel(cfg, dtype=dt)
            m.load_state_dict(sd)
            out = m(x, labels, ctx).cpu()
            print(f"{dt:>6} engine: " + "  ".join(f"sample {i}: {rel_l2(out[i], ref[i]):.3e}" for i in range(B)))
            del m
        for name, sw in plans:
            for k, v in sw:
                _lib.check(lib.t2p_debug_set(k, v))
            try:
                m = HipScoreModel(cfg, dtype="f16")
                m.load_state_dict(sd)
                out = m(x, labels, ctx).cpu()
                print(f"   f16 {name:<40}: " + "  ".join(f"sample {i}: {rel_l2(out[i], ref[i]):.3e}" for i in range(B)) +
                      f"  all: {rel_l2(out, ref):.3e}")
                del m
            finally:
                for k, _ in sw:
                    lib.t2p_debug_set(k, 1)


if __name__ == "__main__":
    main()
