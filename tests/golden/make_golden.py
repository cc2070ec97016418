"""Generate the golden fixtures by running the REFERENCE itself (build container only).

Usage (in the build container, where /root/reference exists):
    python tests/golden/make_golden.py

Imports the reference's hot-path modules from /root/reference (they
import on CPU with torch/numpy/einops/tqdm), loads this repo's synthetic weights into the
reference ``UNetModel`` through ``load_state_dict``, runs it, and stores inputs + outputs as
``tests/golden/*.npz``.  Only data is written: no reference source text goes into the repo.
The fixtures are what pins ``oracle/t2p_oracle.py`` (tests/test_oracle_golden.py) and, through
it, the HIP path.  The GPU box never runs this script (no /root/reference there).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from text2protein_amd.config import tiny_config            # noqa: E402
from text2protein_amd import synth                         # noqa: E402
from text2protein_amd.arch import build_arch               # noqa: E402
from oracle import t2p_oracle as O                         # noqa: E402

from score_sde_pytorch.models import ncsnpp                # noqa: E402  (reference)
from score_sde_pytorch import sde_lib, sampling            # noqa: E402  (reference)
from score_sde_pytorch.models import utils as mutils       # noqa: E402  (reference)


def build_reference_model(cfg, seed):
    torch.manual_seed(0)
    model = ncsnpp.UNetModel(cfg)
    sd = synth.synth_state_dict(cfg, seed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert list(missing) == ["sigmas"] or not missing, missing
    assert not unexpected, unexpected
    model.eval()
    return model, sd


def forward_fixture(name, cfg, seed, B, T):
    model, sd = build_reference_model(cfg, seed)
    C, L = cfg.data.num_channels, cfg.data.max_res_num
    x = torch.from_numpy(synth.normal(seed, "x0", B * C * L * L).reshape(B, C, L, L)) * 3.0
    ctx = synth.synth_context(B, T, cfg.model.context_dim, seed)
    labels = torch.tensor([1, cfg.model.num_scales - 2][:B] + [0] * max(0, B - 2)).long()
    taps = {}
    hooks = []
    arch = build_arch(cfg)
    mods = dict(model.named_modules())
    for l in arch.all_layers():
        hooks.append(mods[l.prefix].register_forward_hook(
            lambda m, i, o, k=l.prefix: taps.__setitem__(k, o.detach().clone())))
    with torch.no_grad():
        score = model(x, labels, ctx)
    for h in hooks:
        h.remove()
    assert score.dtype == torch.float64
    # cross-check the oracle right here
    otaps = {}
    with torch.no_grad():
        oscore = O.unet_forward(sd, cfg, x, labels, ctx, taps=otaps)
    for k, v in taps.items():
        err = (otaps[k] - v).norm() / v.norm()
        assert err < 1e-5, (k, float(err))
    err = float((oscore - score).norm() / score.norm())
    print(f"[{name}] oracle vs reference score rel-L2 = {err:.3e}; |score| rms = {float(score.pow(2).mean().sqrt()):.4g}")
    assert err < 1e-5
    out = {"x": x.numpy(), "labels": labels.numpy(), "context": ctx.numpy(), "score": score.numpy(),
           "seed": np.int64(seed)}
    for k, v in taps.items():
        out["tap:" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def sampler_fixture(name, cfg, seed, B, T, cond_kind):
    model, sd = build_reference_model(cfg, seed)
    C, L = cfg.data.num_channels, cfg.data.max_res_num
    N = cfg.model.num_scales
    shape = (B, C, L, L)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, seed)
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, shape, 1e-5)

    def make_condition():
        cond = {}
        if "length" in cond_kind:
            m = torch.zeros(B, L, L).bool()
            m[:, :12, :12] = True
            cond["length"] = m
        if "inpainting" in cond_kind:
            coords = torch.from_numpy(synth.uniform_pm1(seed, "coords_6d", B * C * L * L).reshape(shape))
            cond["inpainting"] = {"coords_6d": coords, "mask_inpaint": O.selected_mask("1:3,6:8", B, L)}
        return cond

    torch.manual_seed(1234 + seed)
    ref, nfe = fn(model, condition=make_condition(), context=ctx)
    # same noise stream through the oracle, recording the draws
    draws = []

    def noise_fn(shp):
        z = torch.randn(*shp)
        draws.append(z)
        return z

    trace = []
    torch.manual_seed(1234 + seed)
    got, nfe2 = O.pc_sampler_ve(sd, cfg, shape, ctx, condition=make_condition(), noise_fn=noise_fn, trace=trace)
    err = float((got - ref).norm() / ref.norm())
    print(f"[{name}] oracle vs reference sample rel-L2 = {err:.3e}, nfe={nfe}")
    assert err < 1e-5 and nfe == nfe2
    out = {"context": ctx.numpy(), "sample": ref.numpy(), "nfe": np.int64(nfe), "seed": np.int64(seed),
           "noise": torch.stack(draws).numpy()}
    for i, (x, xm) in enumerate(trace):
        out[f"x_step{i}"] = x.numpy()
        out[f"xmean_step{i}"] = xm.numpy().astype(np.float32)
    cond = make_condition()
    if "length" in cond:
        out["cond_length"] = cond["length"].numpy()
    if "inpainting" in cond:
        out["cond_coords_6d"] = cond["inpainting"]["coords_6d"].numpy()
        out["cond_mask_inpaint"] = cond["inpainting"]["mask_inpaint"].numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def sampler_fixture_vp(name, cfg, seed, B, T):
    """5-step VP run of the reference (training.sde == vpsde, eps = 1e-3 as in sampling_6d.py:80-82)."""
    model, sd = build_reference_model(cfg, seed)
    C, L = cfg.data.num_channels, cfg.data.max_res_num
    N = cfg.model.num_scales
    shape = (B, C, L, L)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, seed)
    sde = sde_lib.VPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, shape, 1e-3)
    torch.manual_seed(4321 + seed)
    ref, nfe = fn(model, condition={}, context=ctx)
    draws = []

    def noise_fn(shp):
        z = torch.randn(*shp)
        draws.append(z)
        return z

    torch.manual_seed(4321 + seed)
    got, _ = O.pc_sampler_vp(sd, cfg, shape, ctx, noise_fn=noise_fn)
    err = float((got - ref).norm() / ref.norm())
    print(f"[{name}] oracle vs reference VP sample rel-L2 = {err:.3e}, nfe={nfe}")
    assert err < 1e-5
    np.savez_compressed(os.path.join(HERE, name + ".npz"), context=ctx.numpy(), sample=ref.numpy(), nfe=np.int64(nfe),
                        seed=np.int64(seed), noise=torch.stack(draws).numpy())


def tables_fixture():
    """discrete_sigmas, G_i, labels for N in {100, 1000, 2000} from the reference VESDE and
    get_score_fn (sde_lib.py:199-245, models/utils.py:159-171), plus the model sigmas buffer."""
    out = {}

    class _Probe(torch.nn.Module):
        def forward(self, x, labels, ctx):
            self.labels = labels.clone()
            return x

    for N in (100, 1000, 2000):
        sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=100.0, N=N)
        ts = torch.linspace(sde.T, 1e-5, N)
        probe = _Probe()
        score_fn = mutils.get_score_fn(sde, probe, train=False)
        labels, G = [], []
        x = torch.zeros(1, 1, 1, 1)
        for i in range(N):
            vt = torch.ones(1) * ts[i]
            score_fn(x, vt.clone(), None)
            labels.append(int(probe.labels[0]))
            G.append(float(sde.discretize(x, vt)[1][0]))
        out[f"discrete_sigmas_{N}"] = sde.discrete_sigmas.numpy()
        out[f"labels_{N}"] = np.array(labels, dtype=np.int64)
        out[f"G_{N}"] = np.array(G, dtype=np.float32)
        cfg = tiny_config(**{"model.num_scales": N})
        out[f"model_sigmas_{N}"] = mutils.get_sigmas(cfg)
        # VP tables as well (lowest-priority row 21 of SURVEY 8(a))
        vp = sde_lib.VPSDE(beta_min=0.1, beta_max=20.0, N=N)
        out[f"vp_alphas_{N}"] = vp.alphas.numpy()
        out[f"vp_sqrt_1m_acp_{N}"] = vp.sqrt_1m_alphas_cumprod.numpy()
    emb = __import__("score_sde_pytorch.models.layers", fromlist=["x"]).get_timestep_embedding(
        torch.tensor([0, 1, 7, 999, 1999]), 32)
    out["temb_32"] = emb.numpy()
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)
    print("[tables] written")


def main():
    torch.set_num_threads(8)
    cfg = tiny_config()
    forward_fixture("tiny_forward", cfg, seed=0, B=2, T=3)
    # a second architecture: 3 levels, 2 res-blocks, 8 channels, attention at two resolutions
    cfg_b = tiny_config(**{"model.ch_mult": [1, 1, 2], "model.num_res_blocks": 2, "data.num_channels": 8,
                           "model.attn_resolutions": [4, 8], "model.n_heads": 2, "model.context_dim": 24,
                           "model.nf": 32})
    forward_fixture("tinyB_forward", cfg_b, seed=3, B=2, T=5)
    for kind in ("none", "length", "length+inpainting"):
        sampler_fixture("tiny_sampler_" + kind.replace("+", "_"), cfg, seed=0, B=2, T=3, cond_kind=kind)
    # VP needs beta_max / N < 1 (alpha = 1 - beta > 0): 40 steps instead of the 5 of the VE fixtures
    sampler_fixture_vp("tiny_sampler_vp", tiny_config(**{"model.num_scales": 40, "training.sde": "vpsde"}), seed=0, B=2, T=3)
    tables_fixture()


if __name__ == "__main__":
    main()
