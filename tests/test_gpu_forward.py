"""Score network and PC sampler on the GPU against the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from helpers import load_golden, cfg_tiny, cfg_tinyB, cfg_smallC, rel_l2

pytestmark = pytest.mark.gpu

# rel-L2 of one score evaluation against the fp32 CPU reference
SCORE_TOL = {"f32": 2e-5, "f16": 4e-3, "bf16": 3e-2}


def make_model(cfg, seed, dtype):
    from text2protein_amd import synth
    from text2protein_amd.model import HipScoreModel
    m = HipScoreModel(cfg, dtype=dtype)
    m.load_state_dict(synth.synth_state_dict(cfg, seed))
    return m


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("name,cfgf", [("tiny_forward", cfg_tiny), ("tinyB_forward", cfg_tinyB)])
def test_score_matches_reference_golden(name, cfgf, dtype):
    g = load_golden(name)
    cfg = cfgf()
    m = make_model(cfg, int(g["seed"]), dtype)
    x = torch.from_numpy(g["x"]).cuda()
    out = m(x, torch.from_numpy(g["labels"]).cuda(), torch.from_numpy(g["context"]).cuda())
    torch.cuda.synchronize()
    err = rel_l2(out.cpu(), g["score"])
    print(f"{name} {dtype}: score rel-L2 vs reference = {err:.3e}")
    assert err < SCORE_TOL[dtype]


def test_engine_param_table_matches_arch():
    from text2protein_amd.arch import param_specs
    for cfg in (cfg_tiny(), cfg_tinyB()):
        from text2protein_amd.model import HipScoreModel
        m = HipScoreModel(cfg)
        assert m.engine_param_table() == [(s.name, tuple(s.shape)) for s in param_specs(cfg)]


def test_missing_weight_fails_loudly():
    from text2protein_amd import synth
    from text2protein_amd._lib import T2PError
    from text2protein_amd.model import HipScoreModel
    cfg = cfg_tiny()
    sd = synth.synth_state_dict(cfg, 0)
    sd.pop("mid_blocks.1.NIN_2.W")
    m = HipScoreModel(cfg)
    with pytest.raises(T2PError):
        m.load_state_dict(sd)
    with pytest.raises(T2PError):
        m(torch.zeros(1, 5, 16, 16).cuda(), torch.zeros(1).long().cuda(), torch.zeros(1, 3, 32).cuda())


def test_failed_evaluation_returns_its_buffers_to_the_pool():
    """An evaluation that fails half-way (here: a batch the cached text keys / values were not projected for, detected inside the
    first SpatialTransformer block, with the skip stack and the block's temporaries checked out) must not leave those buffers checked
    out for the lifetime of the engine: the lease scope of Engine::score takes them back (t2p_engine_pool_reclaimed > 0), the next
    evaluation reuses them (no growth of the pool) and is bit-identical to the one before the failure."""
    from text2protein_amd import synth
    from text2protein_amd._lib import T2PError, check, ptr, stream_ptr
    from text2protein_amd.model import HipScoreModel
    cfg = cfg_tiny()
    m = HipScoreModel(cfg, dtype="f16")
    m.load_state_dict(synth.synth_state_dict(cfg, 0))
    x = torch.from_numpy(synth.normal(3, "x", 2 * 5 * 16 * 16).reshape(2, 5, 16, 16)).cuda()
    labels = torch.tensor([3, 40]).cuda()
    ctx = synth.synth_context(2, 3, cfg.model.context_dim, 1).cuda()
    a = m(x, labels, ctx)
    torch.cuda.synchronize()
    assert m.pool_reclaimed() == 0
    held = m.device_bytes()
    out = torch.empty(1, 5, 16, 16, device="cuda")
    li = labels[:1].to(torch.int32).contiguous()
    with pytest.raises(T2PError):               # one chain against a context cached for two: fails inside the network, not at the door
        check(m.lib.t2p_engine_score_ex(m._h, ptr(x[:1].contiguous()), ptr(li), None, ptr(out), 1, stream_ptr()))
    assert m.pool_reclaimed() > 0
    held1 = m.device_bytes()                    # (the one-chain attempt added its own buffer sizes to the exact-size pool)
    b = m(x, labels, ctx)
    torch.cuda.synchronize()
    assert m.pool_reclaimed() == 0 and m.device_bytes() == held1 and held1 - held < (1 << 20) and torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_score_wide_config_matches_oracle(dtype):
    """64/128-channel maps at 32x32: the configuration where the LDS-DMA kernel carries the convolutions."""
    from oracle import t2p_oracle as O
    from text2protein_amd import synth
    cfg = cfg_smallC()
    sd = synth.synth_state_dict(cfg, 11)
    B, T = 3, 20
    x = torch.from_numpy(synth.normal(11, "x", B * 5 * 32 * 32).reshape(B, 5, 32, 32)) * 20.0
    ctx = synth.synth_context(B, T, cfg.model.context_dim, 11)
    labels = torch.tensor([0, 4, 9])
    with torch.no_grad():
        want = O.unet_forward(sd, cfg, x, labels, ctx)
    m = make_model(cfg, 11, dtype)
    got = m(x.cuda(), labels.cuda(), ctx.cuda())
    torch.cuda.synchronize()
    err = rel_l2(got.cpu(), want)
    print(f"wide config {dtype}: score rel-L2 vs oracle = {err:.3e}")
    assert err < SCORE_TOL[dtype]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_head_conv_thin_kernel_matches_gemm_path(dtype):
    """The network head (nf -> 5 channels, NCHW fp32 output x 1/sigma) on the thin-output MFMA kernel against the
    same layer on the generic GEMM kernel (development key 15): identical 16-bit operands, fp32 accumulation in
    a different order.  Ragged tiles: a 24 x 24 map is 3 x 2 tiles of 8 x 16 with half-empty columns."""
    from text2protein_amd import _lib, synth
    from text2protein_amd.config import tiny_config
    lib = _lib.load()
    cfg = tiny_config(**{"data.max_res_num": 24, "model.attn_resolutions": [12], "model.nf": 64, "model.ch_mult": [1, 2]})
    m = make_model(cfg, 3, dtype)
    B = 3
    x = (torch.from_numpy(synth.normal(8, "x", B * 5 * 24 * 24).reshape(B, 5, 24, 24)) * 10.0).cuda()
    ctx = synth.synth_context(B, 7, cfg.model.context_dim, 2).cuda()
    labels = torch.tensor([0, 2, 4]).cuda()
    try:
        lib.t2p_debug_set(15, 0)
        a = m(x, labels, ctx).cpu()
        lib.t2p_debug_set(15, 1)
        b = m(x, labels, ctx).cpu()
    finally:
        lib.t2p_debug_set(15, 1)
    assert torch.isfinite(b).all()
    err = rel_l2(b, a)
    print(f"thin head conv vs GEMM head conv ({dtype}): rel-L2 = {err:.3e}")
    assert err < 5e-6      # (bit-identical in practice: same K order, fp32 accumulation)


def test_gn_apply_16byte_kernel_is_bit_identical():
    """GroupNorm apply on 16-bit maps: the 16-byte / 4-pixels-in-flight kernel (plan switch 17) against the 8-byte one --
    the same arithmetic per element, so the whole score must be bit-identical (f16 engine, maps of 32^2 and 16^2,
    64 .. 192 channels incl. the two-source concat inputs of the up path)."""
    from text2protein_amd import _lib, synth
    from helpers import cfg_smallC
    lib = _lib.load()
    cfg = cfg_smallC()
    m = make_model(cfg, 5, "f16")
    B = 3
    L = cfg.data.max_res_num
    x = (torch.from_numpy(synth.normal(8, "x", B * 5 * L * L).reshape(B, 5, L, L)) * 10.0).cuda()
    ctx = synth.synth_context(B, 7, cfg.model.context_dim, 2).cuda()
    labels = torch.tensor([0, 4, 9]).cuda()
    try:
        lib.t2p_debug_set(17, 0)
        a = m(x, labels, ctx).cpu()
        lib.t2p_debug_set(17, 1)
        b = m(x, labels, ctx).cpu()
    finally:
        lib.t2p_debug_set(17, 1)
    assert torch.isfinite(b).all() and torch.equal(a, b)


def test_groupnorm_folded_statistics_in_the_apply_launch():
    """Maps of <= 4096 pixels whose GroupNorm statistics arrive as column sums: finalize and apply in one launch (plan switch
    27: a block folds the sums of its own 64-channel slab) against the separate finalize launch -- the same double-precision
    sums, only their order inside a group differs, so the scores agree to rounding of the statistics (f16 engine, 32^2 and
    16^2 maps, one- and two-source inputs)."""
    from text2protein_amd import _lib, synth
    from helpers import cfg_smallC
    lib = _lib.load()
    cfg = cfg_smallC()
    m = make_model(cfg, 5, "f16")
    B = 3
    L = cfg.data.max_res_num
    x = (torch.from_numpy(synth.normal(8, "x", B * 5 * L * L).reshape(B, 5, L, L)) * 10.0).cuda()
    ctx = synth.synth_context(B, 7, cfg.model.context_dim, 2).cuda()
    labels = torch.tensor([0, 4, 9]).cuda()
    try:
        lib.t2p_debug_set(27, 0)
        a = m(x, labels, ctx).cpu()
        lib.t2p_debug_set(27, 1)
        b = m(x, labels, ctx).cpu()
    finally:
        lib.t2p_debug_set(27, 1)
    d = rel_l2(b, a)
    print(f"GroupNorm statistics folded in the apply launch vs separate finalize: rel-L2 = {d:.3e}, bit-equal = {torch.equal(a, b)}")
    assert torch.isfinite(b).all() and d < 1e-4
