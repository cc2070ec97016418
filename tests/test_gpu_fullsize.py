"""Size-independent properties at BASELINE's full model size (configs[1]: test_config.yml, L=128,
nf=256, 379.5 M parameters, 512 text tokens), where the CPU oracle is too slow to be the checker:
determinism, per-sample independence of the score network, f16 against exact-f32 MFMA, and the
condition invariants of the sampler.  A smaller chain count than the benchmark keeps it short."""
import os

import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full():
    from text2protein_amd import synth
    from text2protein_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "configs", "test_config.yml"), **{"data.max_res_num": 128, "model.num_scales": 1000})
    cfg.device = "cuda:0"
    sd = synth.synth_state_dict(cfg, 0)
    B, T = 4, 512
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 3).cuda()
    x = (torch.from_numpy(synth.normal(5, "x", B * 5 * 128 * 128).reshape(B, 5, 128, 128)) * 50.0).cuda()
    labels = torch.tensor([0, 10, 500, 999]).cuda()
    return cfg, sd, ctx, x, labels


def _model(cfg, sd, dtype):
    from text2protein_amd.model import HipScoreModel
    m = HipScoreModel(cfg, dtype=dtype)
    m.load_state_dict(sd)
    return m


def test_full_size_score_properties(full):
    cfg, sd, ctx, x, labels = full
    m32 = _model(cfg, sd, "f32")
    a = m32(x, labels, ctx)
    b = m32(x, labels, ctx)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)                                    # bitwise reproducible
    assert m32.pool_reclaimed() == 0                            # every block returned its temporaries (the lease scope found none)
    # chains are independent inside the network (only the Langevin step size couples them):
    # evaluating a sample alone gives the same score as inside a batch
    for i in (0, 3):
        one = m32(x[i:i + 1], labels[i:i + 1], ctx[i:i + 1])
        assert rel_l2(one.cpu(), a[i:i + 1].cpu()) < 1e-5
    m32.set_context(ctx)
    # scale_by_sigma: the same input at a smaller sigma label is the raw output over a smaller sigma
    m16 = _model(cfg, sd, "f16")
    c = m16(x, labels, ctx)
    d = m16(x, labels, ctx)
    torch.cuda.synchronize()
    assert torch.equal(c, d)
    assert m16.pool_reclaimed() == 0 and m16.device_bytes() == (m16(x, labels, ctx), m16.device_bytes())[1]   # no leak, no growth
    err = rel_l2(c.cpu(), a.cpu())
    print(f"full-size score: f16 vs exact-f32 rel-L2 = {err:.3e}")
    assert err < 2e-3
    del m32
    mb = _model(cfg, sd, "bf16")
    e = mb(x, labels, ctx)
    err_b = rel_l2(e.cpu(), a.cpu())
    print(f"full-size score: bf16 vs exact-f32 rel-L2 = {err_b:.3e}")
    assert err_b < 2e-2


def test_full_size_sampler_invariants(full):
    from text2protein_amd import sampling, sde_lib
    from text2protein_amd.conditions import synthetic_condition
    cfg, sd, ctx, _, _ = full
    m = _model(cfg, sd, "f16")
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    B = 4
    fn = sampling.get_sampling_fn(cfg, sde, (B, 5, 128, 128), 1e-5, seed=11)
    cond = synthetic_condition(cfg, B, "length+inpainting", "cuda:0", length=100)
    out, nfe = fn(m, condition=cond, context=ctx, n_iter=3, call_index=0)
    out2, _ = fn(m, condition=cond, context=ctx, n_iter=3, call_index=0)
    torch.cuda.synchronize()
    assert nfe == 2000 and torch.isfinite(out).all()
    assert torch.equal(out, out2)                               # counter-based device noise: reproducible
    free = (cond["length"] & cond["inpainting"]["mask_inpaint"]).unsqueeze(1).expand_as(out).clone()
    free[:, -1] = False
    assert torch.equal(out[~free], cond["inpainting"]["coords_6d"][~free])   # frozen region untouched (sampling.py:271-287)
    assert float(out[free].abs().max()) > 1.0                   # the free region evolved from the sigma_max prior


def test_cfg5_sampler_at_full_size():
    """BASELINE configs[4]: the C = 8 cond_length_inpainting model at its per-GPU batch (16 chains, L = 128) with the length +
    inpainting conditions, 3 PC steps of the fused sampler: what sampling.py:259-287 guarantees whatever the network does --
    the known region and the mask channel come back exactly, everything outside the length mask stays at its initial value,
    the free region moved -- plus bitwise reproducibility, and the class route (operator ABI) agreeing with the fused one."""
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.conditions import synthetic_condition
    from text2protein_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "configs", "cond_length_inpainting.yml"), **{"data.max_res_num": 128, "model.num_scales": 1000})
    cfg.device = "cuda:0"
    B, C, L, T = 16, 8, 128, 512
    assert cfg.data.num_channels == C and cfg.model.condition == ["length", "inpainting"]
    sd = synth.synth_state_dict(cfg, 0)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 31).cuda()
    m = _model(cfg, sd, "f16")
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    cond = synthetic_condition(cfg, B, "length+inpainting", "cuda:0", length=100)
    fn = sampling.get_sampling_fn(cfg, sde, (B, C, L, L), 1e-5, seed=5)
    out, nfe = fn(m, condition=cond, context=ctx, n_iter=3, call_index=0)
    again, _ = fn(m, condition=cond, context=ctx, n_iter=3, call_index=0)
    torch.cuda.synchronize()
    assert nfe == 2000 and torch.isfinite(out).all() and torch.equal(out, again)
    # mask_inpaint is True where a pixel is to be inpainted (pairs with a selected residue, "1:5,10:15" = 11 residues)
    length, selected, coords = cond["length"], cond["inpainting"]["mask_inpaint"], cond["inpainting"]["coords_6d"]
    free = (length & selected).unsqueeze(1).expand_as(out).clone()
    free[:, -1] = False
    assert torch.equal(out[~free], coords[~free])               # known residues, the outside of the length mask, the mask channel
    assert float(out[free].abs().max()) > 1.0 and float((out[free] - coords[free]).abs().mean()) > 1.0
    assert int(free[0].sum()) == 7 * (100 * 100 - 89 * 89)      # 7 data channels x pairs inside the length mask with a selected residue
    # the predictor / corrector classes on the operator ABI against the fused stepper, on identical injected noise
    from helpers import CounterNoise
    outs = {}
    for force in (False, True):
        fn_n = sampling.get_sampling_fn(cfg, sde, (B, C, L, L), 1e-5, seed=5, force_classes=force)
        outs[force], _ = fn_n(m, condition=cond, context=ctx, n_iter=2, noise_fn=CounterNoise(77).draw)
        torch.cuda.synchronize()
        assert torch.equal(outs[force][~free], coords[~free])
    e = rel_l2(outs[True][free], outs[False][free])
    print(f"cfg5 shape, 2 PC steps at 16 chains on injected noise: class route vs fused route rel-L2 = {e:.3e}")
    assert e < 1e-5


def test_midsize_splitk_plan_matches_unsplit(full):
    """24 chains put the 16x16 level at 48 tiles of 256x256: those convolutions then run as 256x256 tiles
    with the K loop split (GroupNorm statistics produced by the split-K second pass).  Both plans use the
    same f16 operands; they must be equally close to the exact-f32 engine and close to each other (the
    fp32 summation order differs, which flips a few f16 roundings downstream)."""
    from text2protein_amd import _lib, synth
    cfg, sd, _, _, _ = full
    lib = _lib.load()
    B = 24
    ctx = synth.synth_context(B, 64, cfg.model.context_dim, 5).cuda()
    x = (torch.from_numpy(synth.normal(6, "x", B * 5 * 128 * 128).reshape(B, 5, 128, 128)) * 30.0).cuda()
    labels = torch.arange(B).cuda() * 40
    ref = _model(cfg, sd, "f32")(x, labels, ctx).cpu()
    m = _model(cfg, sd, "f16")
    try:
        lib.t2p_debug_set(12, 0)
        a = m(x, labels, ctx).cpu()
        lib.t2p_debug_set(12, 1)
        b = m(x, labels, ctx).cpu()
        c = m(x, labels, ctx).cpu()
    finally:
        lib.t2p_debug_set(12, 0)
    assert torch.isfinite(b).all() and torch.equal(b, c)
    ea, eb, eab = rel_l2(a, ref), rel_l2(b, ref), rel_l2(b, a)
    print(f"vs exact-f32: 128x128 plan {ea:.3e}, 256x256 split-K plan {eb:.3e}; between plans {eab:.3e}")
    assert eab > 0.0                      # a different plan really ran
    assert ea < 2e-3 and eb < 2e-3 and abs(ea - eb) < 0.2 * ea and eab < ea


@pytest.mark.skipif(os.environ.get("T2P_LONG_TESTS") != "1", reason="superseded by the reference-pinned test_full_size_score_vs_reference[test_config_large] and test_cfg4_sampler_at_full_size; T2P_LONG_TESTS=1 runs it")
def test_large_config_score_properties():
    """BASELINE configs[3]: test_config_large.yml at L=256 (863.3 M parameters, 3 res-blocks per level, channel
    multiplier 4 at the lowest level: AttnBlockpp runs single-head attention with d up to 1024).  The oracle is
    far too slow here; checked: finite, bitwise reproducible, f16 against the exact-f32 engine, and independence
    of a sample from the rest of its batch."""
    from text2protein_amd import synth
    from text2protein_amd.arch import param_specs
    from text2protein_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "configs", "test_config_large.yml"), **{"data.max_res_num": 256, "model.num_scales": 1000})
    cfg.device = "cuda:0"
    assert abs(sum(int(torch.tensor(s.shape).prod()) for s in param_specs(cfg)) / 1e6 - 863.3) < 0.1
    sd = synth.synth_state_dict(cfg, 0)
    B, T = 2, 64
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 4).cuda()
    x = (torch.from_numpy(synth.normal(9, "x", B * 5 * 256 * 256).reshape(B, 5, 256, 256)) * 20.0).cuda()
    labels = torch.tensor([100, 900]).cuda()
    m32 = _model(cfg, sd, "f32")
    ref = m32(x, labels, ctx).cpu()
    one = m32(x[1:2], labels[1:2], ctx[1:2]).cpu()
    del m32
    assert torch.isfinite(ref).all() and rel_l2(one, ref[1:2]) < 1e-5
    m16 = _model(cfg, sd, "f16")
    a = m16(x, labels, ctx).cpu()
    b = m16(x, labels, ctx).cpu()
    assert torch.equal(a, b)
    err = rel_l2(a, ref)
    print(f"large config (L=256): f16 vs exact-f32 score rel-L2 = {err:.3e}")
    assert err < 2e-3


def test_cfg4_sampler_at_full_size():
    """BASELINE configs[3] as the SAMPLER runs it: test_config_large.yml, L = 256, its per-GPU batch of 16 chains, 512 text tokens,
    3 PC steps of the fused loop in the benchmarked precision (1 M-row maps at the top level, AttnBlockpp with d up to 1024):
    finite, bitwise reproducible from the same seed, different from another seed, every chain moved, the chains of a batch are
    distinct (per-chain Philox streams), and the denoised mean differs from the noisy state by the last predictor's G z only."""
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "configs", "test_config_large.yml"), **{"data.max_res_num": 256, "model.num_scales": 1000})
    cfg.device = "cuda:0"
    B, C, L, T = 16, 5, 256, 512
    sd = synth.synth_state_dict(cfg, 0)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 41).cuda()
    m = _model(cfg, sd, "f16")
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    fn = sampling.get_sampling_fn(cfg, sde, (B, C, L, L), 1e-5, seed=7)
    out, nfe = fn(m, context=ctx, n_iter=3, call_index=0)
    again, _ = fn(m, context=ctx, n_iter=3, call_index=0)
    other, _ = fn(m, context=ctx, n_iter=3, call_index=1)
    torch.cuda.synchronize()
    assert nfe == 2000 and out.shape == (B, C, L, L) and torch.isfinite(out).all()
    assert torch.equal(out, again) and not torch.equal(out, other)
    per_chain = out.reshape(B, -1)
    assert float(per_chain.std(dim=1).min()) > 10.0                      # every chain is still at the noise scale of step 3 (sigma ~ 97)
    d = torch.cdist(per_chain[:, ::97].double(), per_chain[:, ::97].double())
    assert float((d + torch.eye(B, device=d.device) * 1e9).min()) > 1.0  # no two chains coincide
    print(f"cfg4 shape, 3 PC steps at 16 chains: |x| rms {float(out.pow(2).mean().sqrt()):.1f}, device memory {m.device_bytes() / 2 ** 30:.1f} GiB")
