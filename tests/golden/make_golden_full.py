"""Golden fixtures at the BASELINE sizes, produced by running the REFERENCE itself (build container only).

    python tests/golden/make_golden_full.py [--only name,name]

Complements make_golden.py (tiny shapes, per-block taps) with what the benchmark configurations need:

  full_<yaml stem>.npz   one score evaluation of the reference UNetModel built from the reference's OWN
                         configs/<yaml> (read from /root/reference, L / N overridden as BASELINE.md section 5
                         says) on this repo's synthetic weights, x and text context (all regenerated from
                         text2protein_amd.synth on the test side: only the score is stored, float32).
  param_tables.json      named_parameters() names and shapes of the reference model for the four YAMLs --
                         the order the EMA shadow list follows (models/ema.py:51-64).
  cfg1_run100.npz        BASELINE configs[0]: test_config.yml, B=2, L=64, N=100, the complete 100-step PC run
                         of the reference sampler (sampling.py:245-289) with torch.randn / torch.randn_like
                         replaced by a counter-based generator (synth.normal keyed by the draw index), so that
                         the HIP side can inject the identical noise; the final sample only.
  tiny_sampler_ss.npz    5-step run with the `ss` condition (sampling.py:268-270), C = 8.
  tiny_checkpoint.pth    a checkpoint written by the reference's own save_checkpoint
  + tiny_checkpoint_expected.npz   (score_sde_pytorch/utils.py:19-26) with DataParallel keys, an Adam state and
                         an ExponentialMovingAverage.state_dict() whose shadow parameters DIFFER from the live
                         ones, and the score the reference computes after restore_checkpoint + ema.copy_to
                         (sampling_6d.py:64-73).

  run1000_cond_length_inpainting.npz / run1000_test_config_L128.npz   (round 4) the same for cond_length_inpainting.yml
                         (C = 8, length 100 + inpainting "1:5,10:15" on synthetic coords_6d: a shard of BASELINE configs[4]) and for
                         test_config.yml at the benchmark's own L = 128 (one chain: configs[1]).
  run1000_cond_length.npz   the horizon the metric is quoted on: COMPLETE N = 1000 runs of the reference sampler on
  run1000_test_config.npz   counter-based noise, B = 2: cond_length.yml at L = 128 with the `length` condition
                         (100 residues: a shard of BASELINE configs[2]) and test_config.yml at L = 64 (no condition);
                         the final sample only.  `--only run1000_cond_length,run1000_test_config` (15 - 40 min each).

Only data is written; no reference source text goes into the repo.  The GPU box never runs this script.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

from helpers import FULL, CounterNoise, cfg_ckpt as ckpt_config, cfg_ss as ss_config, full_inputs   # noqa: E402
from text2protein_amd.config import finalize_config                 # noqa: E402
from text2protein_amd import synth                                  # noqa: E402
from text2protein_amd.arch import param_specs                       # noqa: E402
from oracle import t2p_oracle as O                                  # noqa: E402

from score_sde_pytorch.models import ncsnpp                         # noqa: E402  (reference)
from score_sde_pytorch.models.ema import ExponentialMovingAverage   # noqa: E402  (reference)
from score_sde_pytorch import sde_lib, sampling                     # noqa: E402  (reference)
from score_sde_pytorch import utils as ref_utils                    # noqa: E402  (reference)

def ref_config(fname, L, N):
    """The reference's own YAML, unchanged, through this repo's loader defaults (n_heads / context_dim are
    absent from cond_length*.yml although UNetModel reads them, ncsnpp.py:94-95)."""
    with open(os.path.join(REF, "configs", fname)) as f:
        raw = yaml.safe_load(f)
    cfg = finalize_config(raw, **{"data.max_res_num": L, "model.num_scales": N})
    cfg.device = "cpu"
    return cfg


def reference_model(cfg, seed):
    torch.manual_seed(0)
    model = ncsnpp.UNetModel(cfg)
    sd = synth.synth_state_dict(cfg, seed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert (list(missing) == ["sigmas"] or not missing) and not unexpected, (missing, unexpected)
    return model.eval(), sd


def full_fixture(stem, tables):
    fname, L, N, B, T, _ = FULL[stem]
    cfg = ref_config(fname, L, N)
    t0 = time.time()
    model, sd = reference_model(cfg, 0)
    names = [(n, list(p.shape)) for n, p in model.named_parameters()]
    tables[stem] = {"yaml": fname, "L": L, "n_params": int(sum(int(np.prod(s)) for _, s in names)), "named_parameters": names}
    assert [(s.name, list(s.shape)) for s in param_specs(cfg)] == names, "arch.param_specs order differs from the reference"
    x, labels, ctx = full_inputs(cfg, B, T)
    with torch.no_grad():
        score = model(x, labels, ctx)
    assert score.dtype == torch.float64 and torch.isfinite(score).all()
    out = {"score": score.float().numpy(), "labels": labels.numpy(), "B": np.int64(B), "T": np.int64(T), "L": np.int64(L),
           "N": np.int64(N), "score_rms": np.float64(score.pow(2).mean().sqrt())}
    # the oracle at the real size (it takes its topology from text2protein_amd.arch: this also pins that)
    with torch.no_grad():
        o = O.unet_forward(sd, cfg, x, labels, ctx)
    err = float((o - score).norm() / score.norm())
    print(f"[full_{stem}] {tables[stem]['n_params'] / 1e6:.1f} M params, oracle vs reference rel-L2 = {err:.3e}, "
          f"|score| rms = {float(out['score_rms']):.4g}, {time.time() - t0:.0f} s", flush=True)
    assert err < 1e-5
    out["oracle_rel_l2"] = np.float64(err)
    np.savez_compressed(os.path.join(HERE, f"full_{stem}.npz"), **out)


def run100_fixture():
    """configs[0]: B=2, L=64, N=100 -- the whole run of the reference sampler on counter-based noise."""
    B, L, N, T, seed = 2, 64, 100, 128, 2024
    cfg = ref_config("test_config.yml", L, N)
    model, sd = reference_model(cfg, 0)
    C = cfg.data.num_channels
    shape = (B, C, L, L)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 5)
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, shape, 1e-5)
    noise = CounterNoise(seed)
    real = torch.randn, torch.randn_like
    t0 = time.time()
    torch.randn, torch.randn_like = noise.randn, noise.randn_like
    try:
        ref, nfe = fn(model, condition={}, context=ctx)
    finally:
        torch.randn, torch.randn_like = real
    assert noise.k == 1 + 2 * N and nfe == 2 * N, (noise.k, nfe)
    print(f"[cfg1_run100] reference run {time.time() - t0:.0f} s, draws = {noise.k}, |sample| rms = {float(ref.pow(2).mean().sqrt()):.4g}",
          flush=True)
    # the oracle on the same draws
    on = CounterNoise(seed)
    t0 = time.time()
    got, _ = O.pc_sampler_ve(sd, cfg, shape, ctx, noise_fn=on.draw)
    err = float((got - ref).norm() / ref.norm())
    print(f"[cfg1_run100] oracle vs reference after {N} PC steps rel-L2 = {err:.3e} ({time.time() - t0:.0f} s)", flush=True)
    assert err < 1e-4
    np.savez_compressed(os.path.join(HERE, "cfg1_run100.npz"), sample=ref.numpy(), nfe=np.int64(nfe), noise_seed=np.int64(seed),
                        B=np.int64(B), L=np.int64(L), N=np.int64(N), T=np.int64(T), context_seed=np.int64(5),
                        oracle_rel_l2=np.float64(err))


RUN1000 = {   # stem -> (yaml, L, chains, text tokens, noise seed, context seed, length condition or None, inpainting ranges or None)
    "cond_length": ("cond_length.yml", 128, 2, 64, 31337, 21, 100, None),
    "test_config": ("test_config.yml", 64, 2, 64, 31338, 22, None, None),
    # round 4: the two BASELINE configurations the horizon was not yet pinned on
    "cond_length_inpainting": ("cond_length_inpainting.yml", 128, 2, 64, 31339, 23, 100, "1:5,10:15"),   # configs[4], C = 8
    "test_config_L128": ("test_config.yml", 128, 1, 64, 31340, 24, None, None),                            # configs[1]'s own L
}
INPAINT_COORDS_SEED = 3     # coords_6d ~ U(-1, 1) = synth.uniform_pm1(3, "coords_6d", B C L L), regenerated on the test side


def run1000_condition(B, C, L, length, mask_info):
    """The condition dict pc_sampler reads (sampling.py:260-275); tensors only."""
    cond = {}
    if length is not None:
        m = torch.zeros(B, L, L).bool()
        m[:, :length, :length] = True
        cond["length"] = m
    if mask_info is not None:
        coords = torch.from_numpy(synth.uniform_pm1(INPAINT_COORDS_SEED, "coords_6d", B * C * L * L).reshape(B, C, L, L))
        cond["inpainting"] = {"coords_6d": coords, "mask_inpaint": O.selected_mask(mask_info, B, L)}
    return cond


def run1000_fixture(stem):
    """N = 1000 (2000 score evaluations, 2001 draws): the reference's pc_sampler, sampling.py:245-289."""
    fname, L, B, T, seed, cseed, length, mask_info = RUN1000[stem]
    N = 1000
    cfg = ref_config(fname, L, N)
    model, _ = reference_model(cfg, 0)
    C = cfg.data.num_channels
    shape = (B, C, L, L)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, cseed)
    cond = run1000_condition(B, C, L, length, mask_info)
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, shape, 1e-5)
    noise = CounterNoise(seed)
    real = torch.randn, torch.randn_like
    t0 = time.time()
    torch.randn, torch.randn_like = noise.randn, noise.randn_like
    try:
        ref, nfe = fn(model, condition=cond, context=ctx)
    finally:
        torch.randn, torch.randn_like = real
    assert noise.k == 1 + 2 * N and nfe == 2 * N and torch.isfinite(ref).all(), (noise.k, nfe)
    print(f"[run1000_{stem}] reference run {time.time() - t0:.0f} s, draws = {noise.k}, |sample| rms = "
          f"{float(ref.pow(2).mean().sqrt()):.4g}", flush=True)
    np.savez_compressed(os.path.join(HERE, f"run1000_{stem}.npz"), sample=ref.numpy(), nfe=np.int64(nfe),
                        noise_seed=np.int64(seed), B=np.int64(B), L=np.int64(L), N=np.int64(N), T=np.int64(T),
                        context_seed=np.int64(cseed), length=np.int64(-1 if length is None else length),
                        mask_info=np.str_(mask_info or ""), coords_seed=np.int64(INPAINT_COORDS_SEED))


def ss_fixture():
    cfg = ss_config()
    seed, B, T = 2, 2, 3
    model, sd = reference_model(cfg, seed)
    C, L, N = cfg.data.num_channels, cfg.data.max_res_num, cfg.model.num_scales
    shape = (B, C, L, L)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, seed)
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, shape, 1e-5)

    def make_condition():
        m = torch.zeros(B, L, L).bool()
        m[:, :11, :11] = True
        ss = torch.from_numpy(synth.uniform_pm1(seed, "ss", B * 3 * L * L).reshape(B, 3, L, L))
        return {"length": m, "ss": ss}

    torch.manual_seed(99)
    ref, nfe = fn(model, condition=make_condition(), context=ctx)
    draws = []

    def noise_fn(shp):
        z = torch.randn(*shp)
        draws.append(z)
        return z

    torch.manual_seed(99)
    got, _ = O.pc_sampler_ve(sd, cfg, shape, ctx, condition=make_condition(), noise_fn=noise_fn)
    err = float((got - ref).norm() / ref.norm())
    print(f"[tiny_sampler_ss] oracle vs reference rel-L2 = {err:.3e}, nfe = {nfe}")
    assert err < 1e-5
    c = make_condition()
    np.savez_compressed(os.path.join(HERE, "tiny_sampler_ss.npz"), context=ctx.numpy(), sample=ref.numpy(), nfe=np.int64(nfe),
                        seed=np.int64(seed), noise=torch.stack(draws).numpy(), cond_length=c["length"].numpy(), cond_ss=c["ss"].numpy())


def checkpoint_fixture():
    cfg = ckpt_config()
    live_seed, ema_seed = 0, 7
    # training-side state as train.py builds it: DataParallel model, Adam, EMA of model.parameters()
    torch.manual_seed(0)
    model = ref_utils.get_model(cfg)                                      # DataParallel(UNetModel), utils.py:4-9
    model.module.load_state_dict(synth.synth_state_dict(cfg, ema_seed), strict=False)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)   # shadow = the seed-7 weights
    model.module.load_state_dict(synth.synth_state_dict(cfg, live_seed), strict=False)   # live weights differ from the EMA
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    path = os.path.join(HERE, "tiny_checkpoint.pth")
    ref_utils.save_checkpoint(path, dict(optimizer=opt, model=model, ema=ema, step=1234))
    # sampling-side restore, as sampling_6d.py:64-73 does it
    torch.manual_seed(1)
    m2 = ref_utils.get_model(cfg)
    ema2 = ExponentialMovingAverage(m2.parameters(), decay=cfg.model.ema_rate)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-4)
    state = ref_utils.restore_checkpoint(path, dict(optimizer=opt2, model=m2, ema=ema2, step=0), "cpu")
    state["ema"].copy_to(m2.parameters())
    m2.eval()
    B, T = 2, 3
    C, L = cfg.data.num_channels, cfg.data.max_res_num
    x = torch.from_numpy(synth.normal(11, "ckpt_x", B * C * L * L).reshape(B, C, L, L)) * 2.0
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 11)
    labels = torch.tensor([1, 8]).long()
    with torch.no_grad():
        score = m2.module(x, labels, ctx)
        # the same network with the seed-7 weights loaded directly, and with the live (seed-0) weights
        direct, _ = reference_model(cfg, ema_seed)
        s_direct = direct(x, labels, ctx)
        live, _ = reference_model(cfg, live_seed)
        s_live = live(x, labels, ctx)
    assert torch.equal(score, s_direct) and not torch.allclose(score, s_live)
    keys = list(torch.load(path, map_location="cpu", weights_only=False)["model"].keys())
    assert all(k.startswith("module.") for k in keys) and "module.sigmas" in keys
    print(f"[tiny_checkpoint] {os.path.getsize(path) / 1e6:.2f} MB, step = {state['step']}, {len(keys)} model keys, "
          f"restored == direct EMA weights, != live weights")
    np.savez_compressed(os.path.join(HERE, "tiny_checkpoint_expected.npz"), x=x.numpy(), labels=labels.numpy(), context=ctx.numpy(),
                        score=score.numpy(), score_live=s_live.numpy(), step=np.int64(state["step"]), ema_seed=np.int64(ema_seed),
                        live_seed=np.int64(live_seed))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    only = set(filter(None, a.only.split(",")))
    torch.set_num_threads(a.threads)
    want = lambda n: not only or n in only   # noqa: E731
    tpath = os.path.join(HERE, "param_tables.json")
    tables = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for stem in FULL:
        if want("full_" + stem):
            full_fixture(stem, tables)
            json.dump(tables, open(tpath, "w"), separators=(",", ":"), sort_keys=True)
    if want("ss"):
        ss_fixture()
    if want("checkpoint"):
        checkpoint_fixture()
    if want("run100"):
        run100_fixture()
    for stem in RUN1000:          # long: only on request
        if "run1000_" + stem in only:
            run1000_fixture(stem)


if __name__ == "__main__":
    main()
