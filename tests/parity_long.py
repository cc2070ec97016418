"""Long-horizon parity experiment (run by hand on the GPU box; too slow for the unit suites):
BASELINE configs[0] -- test_config.yml, B=2, L=64, N=100 PC steps (200 score evaluations) -- HIP
engine in each compute dtype against the CPU oracle on identical weights, text and noise.

    python tests/parity_long.py [--steps 100] [--tokens 128] [--out gpurun_out/parity_long.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--dtypes", default="f32,f16,bf16")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_long.json"))
    ap.add_argument("--mode", default="all", choices=["all", "hip", "oracle", "compare"],
                    help="hip: run the HIP engine and save outputs (GPU box); oracle: run the CPU oracle and save "
                         "(any box: the noise comes from a seeded torch CPU generator); compare: read both")
    a = ap.parse_args()
    from oracle import t2p_oracle as O
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.config import load_config
    from text2protein_amd.model import HipScoreModel

    N = 100
    cfg = load_config(os.path.join(ROOT, "configs", "test_config.yml"), **{"data.max_res_num": a.res, "model.num_scales": N})
    cfg.device = "cuda:0"
    B, C_, L = 2, cfg.data.num_channels, cfg.data.max_res_num
    sd = synth.synth_state_dict(cfg, 0)
    ctx = synth.synth_context(B, a.tokens, cfg.model.context_dim, 5)
    g = torch.Generator().manual_seed(2024)
    draws = [torch.randn(B, C_, L, L, generator=g) for _ in range(1 + 2 * a.steps)]
    res = {"config": f"test_config.yml L={L} N={N} B={B} T={a.tokens} steps={a.steps}", "dtypes": {}}
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, C_, L, L), 1e-5)
    hip_file = os.path.join(ROOT, "gpurun_out", f"parity_long_hip_{a.res}_{a.steps}.pt")
    ora_file = os.path.join(ROOT, "gpurun_out", f"parity_long_oracle_{a.res}_{a.steps}.pt")
    os.makedirs(os.path.dirname(hip_file), exist_ok=True)
    got = {}
    if a.mode in ("oracle", "compare"):
        a.dtypes = ""
    if a.mode == "compare":
        got = torch.load(hip_file)
    for dt in [d for d in a.dtypes.split(",") if d]:
        model = HipScoreModel(cfg, dtype=dt)
        model.load_state_dict(sd)
        it = iter(draws)
        t0 = time.time()
        out, _ = fn(model, context=ctx, noise_fn=lambda s: next(it), n_iter=a.steps)
        torch.cuda.synchronize()
        got[dt] = out.cpu()
        print(f"[{dt}] HIP run {time.time() - t0:.1f}s finite={bool(torch.isfinite(out).all())}", flush=True)
        del model
    if a.mode == "hip":
        torch.save(got, hip_file)
        print("saved", hip_file)
        return
    if a.mode == "compare":
        want = torch.load(ora_file)
    else:
        t0 = time.time()
        it = iter(draws)

        class _Trace(list):
            def append(self, v):
                super().append(v)
                if len(self) % 10 == 0:
                    print(f"oracle step {len(self)} ({time.time() - t0:.0f}s)", flush=True)

        want, _ = O.pc_sampler_ve(sd, cfg, (B, C_, L, L), ctx, noise_fn=lambda s: next(it), n_steps_limit=a.steps, trace=_Trace())
        res["oracle_seconds"] = time.time() - t0
        print(f"oracle {res['oracle_seconds']:.1f}s on {torch.get_num_threads()} threads", flush=True)
        if a.mode == "oracle":
            torch.save(want, ora_file)
            print("saved", ora_file)
            return
    for dt, o in got.items():
        e = float((o.double() - want.double()).norm() / want.double().norm())
        res["dtypes"][dt] = e
        print(f"[{dt}] final x_mean rel-L2 vs oracle after {a.steps} PC steps = {e:.3e}", flush=True)
    if "f32" in got:
        for dt, o in got.items():
            if dt != "f32":
                e = float((o.double() - got["f32"].double()).norm() / got["f32"].double().norm())
                res["dtypes"][dt + "_vs_hip_f32"] = e
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
