"""Every BASELINE configuration against outputs of the REFERENCE itself (tests/golden/make_golden_full.py):

  * one score evaluation per shipped YAML at its real size (configs[1..4]: test_config L=128, cond_length,
    cond_length_inpainting (C=8), test_config_large L=256): exact-f32 engine <= 1e-5, f16 engine within the
    north star's 1e-3; the same fixture samples embedded in a batch of the BENCHMARK's size (32 / 16 chains,
    up to M = 1 M rows: 32-bit offset arithmetic, tile scheduling) between unrelated filler chains; and under
    every forced LDS-DMA tile geometry, the 512 x 128 one of cfg3 / cfg5 included;
  * configs[0] (test_config.yml, B=2, L=64, N=100): the complete 100-step PC run on the reference's noise;
  * the 1000-step horizon the metric is quoted on: f16 against the exact-f32 engine on identical Philox noise at
    cfg2 and cfg3 shapes (the 100-step fixture shows f16-vs-f32-engine == f16-vs-reference).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import FULL, CounterNoise, full_inputs, load_golden, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LONG = os.environ.get("T2P_LONG_TESTS") == "1"      # variants that only repeat a pinned figure on a second size run under T2P_LONG_TESTS=1 (suite time on the GPU box)
long_only = pytest.mark.skipif(not LONG, reason="T2P_LONG_TESTS=1 runs it")
OUT = os.path.join(ROOT, "gpurun_out")

F32_TOL = 1e-5
F16_TOL = 1e-3          # BASELINE.json north_star: samples "matching CPU reference within 1e-3 rel-L2" -- asserted on
                        # the RUNS below (100 steps vs the reference, 1000 steps vs the exact-f32 engine)
F16_SCORE_TOL = 2e-3    # ONE score evaluation in f16: the rounding of 11-bit operands through ~100 layers is 0.5-1.0e-3
                        # of the score (largest at small sigma); a run averages it down (2.9e-4 after 100 steps)


def _cfg(stem, **over):
    from text2protein_amd.config import load_config
    fname, L, N, B, T, chains = FULL[stem]
    cfg = load_config(os.path.join(ROOT, "configs", fname), **{"data.max_res_num": L, "model.num_scales": N, **over})
    cfg.device = "cuda:0"
    return cfg, B, T, chains


def _model(cfg, sd, dtype):
    from text2protein_amd.model import HipScoreModel
    m = HipScoreModel(cfg, dtype=dtype)
    m.load_state_dict(sd)
    return m


def _record(name, value):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "r04_parity.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[name] = value
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.fixture(scope="module", params=list(FULL))
def full(request):
    from text2protein_amd import synth
    stem = request.param
    cfg, B, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    assert int(g["B"]) == B and int(g["T"]) == T
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B, T)
    return stem, cfg, sd, x.cuda(), labels.cuda(), ctx.cuda(), torch.from_numpy(g["score"]), chains


def test_param_table_matches_reference_named_parameters():
    """arch.param_specs and the engine's own table against the reference's named_parameters() (the order the EMA
    shadow list follows, models/ema.py:51-64) for the four shipped YAMLs."""
    from text2protein_amd.arch import param_specs
    tables = json.load(open(os.path.join(ROOT, "tests", "golden", "param_tables.json")))
    assert sorted(tables) == sorted(FULL)
    for stem, t in tables.items():
        cfg, _, _, _ = _cfg(stem)
        want = [(n, tuple(s)) for n, s in t["named_parameters"]]
        assert [(s.name, tuple(s.shape)) for s in param_specs(cfg)] == want
        from text2protein_amd.model import HipScoreModel
        m = HipScoreModel(cfg, dtype="f16")
        assert m.engine_param_table() == want
        assert sum(int(np.prod(s)) for _, s in want) == t["n_params"]


def test_full_size_score_vs_reference(full):
    stem, cfg, sd, x, labels, ctx, want, _ = full
    m32 = _model(cfg, sd, "f32")
    e32 = rel_l2(m32(x, labels, ctx).cpu(), want)
    del m32
    m16 = _model(cfg, sd, "f16")
    e16 = rel_l2(m16(x, labels, ctx).cpu(), want)
    print(f"{stem}: one score evaluation vs the reference: f32 {e32:.3e}, f16 {e16:.3e}")
    _record(f"score_{stem}", {"f32": e32, "f16": e16, "L": cfg.data.max_res_num, "batch": int(x.shape[0])})
    assert e32 < F32_TOL
    assert e16 < F16_SCORE_TOL


def test_benchmark_batch_holds_the_reference_samples(full):
    """The fixture's samples inside a batch of the benchmark's size (first and last slots), every other slot an
    unrelated chain with its own text and time label: each must still come out as the reference computed it."""
    from text2protein_amd import synth
    stem, cfg, sd, x, labels, ctx, want, chains = full
    B0, T = x.shape[0], ctx.shape[1]
    xs = torch.from_numpy(synth.normal(77, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 78).cuda()
    ls = (torch.arange(chains, device="cuda") * 37 + 11) % cfg.model.num_scales
    slots = [0, chains - 1][:B0] if B0 > 1 else [chains - 1]
    for i, s in enumerate(slots):
        xs[s], cs[s], ls[s] = x[i], ctx[i], labels[i]
    m16 = _model(cfg, sd, "f16")
    out = m16(xs, ls, cs)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    for i, s in enumerate(slots):
        e = rel_l2(out[s].cpu(), want[i])
        print(f"{stem}: sample {i} in slot {s} of {chains} chains: f16 vs reference {e:.3e}")
        _record(f"score_{stem}_batch{chains}_slot{s}", e)
        assert e < F16_SCORE_TOL


@pytest.mark.parametrize("geom", [1, 2, 3, 4])
def test_forced_tile_geometries_on_cfg3(geom):
    """cfg3 / cfg5 run their wide maps on the 512 x 128 geometry (plan value 4), which no small shape selects by
    itself: every forced geometry of the LDS-DMA GEMM must reproduce the reference's score."""
    from text2protein_amd import _lib, synth
    stem = "cond_length"
    cfg, B, T, _ = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B, T)
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    try:
        _lib.check(lib.t2p_debug_set(2, geom))
        lib.t2p_profile_begin()
        out = m16(x.cuda(), labels.cuda(), ctx.cuda()).cpu()
        o9 = (__import__("ctypes").c_double * 9)()
        lib.t2p_profile_end(o9)
        name = __import__("ctypes").create_string_buffer(256)
        dom = (__import__("ctypes").c_double * 4)()
        lib.t2p_profile_dominant(dom, name, 256)
    finally:
        lib.t2p_debug_set(2, 0)
    e = rel_l2(out, g["score"])
    kname = name.value.decode()
    print(f"geometry {geom}: dominant conv kernel {kname}: f16 vs reference {e:.3e}")
    want = {1: "256, 128", 2: "128, 128", 3: "256, 256", 4: "512, 128"}[geom]
    assert want in kname
    _record(f"score_cond_length_geom{geom}", {"kernel": kname, "f16": e})
    assert e < F16_SCORE_TOL


def test_cfg1_hundred_step_run_vs_reference():
    """BASELINE configs[0]: the reference sampler's complete run (B=2, L=64, N=100) on counter-based noise."""
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.config import load_config
    g = load_golden("cfg1_run100")
    B, L, N, T = int(g["B"]), int(g["L"]), int(g["N"]), int(g["T"])
    cfg = load_config(os.path.join(ROOT, "configs", "test_config.yml"), **{"data.max_res_num": L, "model.num_scales": N})
    cfg.device = "cuda:0"
    sd = synth.synth_state_dict(cfg, 0)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, int(g["context_seed"]))
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, cfg.data.num_channels, L, L), 1e-5)
    res = {}
    for dt, tol in (("f32", F32_TOL), ("f16", F16_TOL)):
        m = _model(cfg, sd, dt)
        noise = CounterNoise(int(g["noise_seed"]))
        out, nfe = fn(m, context=ctx, noise_fn=noise.draw)
        torch.cuda.synchronize()
        assert nfe == int(g["nfe"]) and noise.k == 1 + 2 * N
        res[dt] = rel_l2(out.cpu(), g["sample"])
        print(f"configs[0], 100 PC steps: {dt} final sample vs the reference run {res[dt]:.3e}")
        del m
    _record("cfg1_run100", res)
    assert res["f32"] < F32_TOL and res["f16"] < F16_TOL


F32_RUN1000_TOL = 1e-5   # 2000 chained fp32 evaluations against the reference's own fp32 (CPU) arithmetic (measured 1.9e-6)
                         # differences of ~1e-6 per evaluation compound over the run (measured values: gpurun_out/r03_parity.json)


RUN1000_YAML = {"cond_length": "cond_length.yml", "test_config": "test_config.yml", "cond_length_inpainting": "cond_length_inpainting.yml",
                "test_config_L128": "test_config.yml"}


@pytest.mark.parametrize("stem", ["cond_length", pytest.param("test_config", marks=long_only), "cond_length_inpainting", "test_config_L128"])
def test_thousand_step_run_vs_reference(stem):
    """The horizon the metric is quoted on, pinned by the REFERENCE: complete N = 1000 runs of the reference sampler
    (tests/golden/make_golden_full.py, run1000_<stem>.npz) on counter-based noise -- cond_length.yml at L = 128 with the length
    condition (B = 2: a shard of BASELINE configs[2]), test_config.yml at L = 64 (B = 2) and, since round 4, cond_length_inpainting.yml
    (C = 8, length 100 + inpainting "1:5,10:15" on synthetic coords_6d: a shard of configs[4]) and test_config.yml at the benchmark's own
    L = 128 (B = 1: configs[1]).  The f16 engine (the benchmarked precision) must end within the north star's 1e-3 of the reference's
    final samples."""
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.conditions import pair_mask, parse_mask_info
    from text2protein_amd.config import load_config
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", f"run1000_{stem}.npz")):
        pytest.skip(f"run1000_{stem}.npz has not been generated yet (tests/golden/make_golden_full.py --only run1000_{stem})")
    g = load_golden("run1000_" + stem)
    B, L, N, T, length = int(g["B"]), int(g["L"]), int(g["N"]), int(g["T"]), int(g["length"])
    mask_info = str(g["mask_info"]) if "mask_info" in g else ""
    assert N == 1000
    cfg = load_config(os.path.join(ROOT, "configs", RUN1000_YAML[stem]), **{"data.max_res_num": L, "model.num_scales": N})
    cfg.device = "cuda:0"
    C = cfg.data.num_channels
    sd = synth.synth_state_dict(cfg, 0)
    ctx = synth.synth_context(B, T, cfg.model.context_dim, int(g["context_seed"]))
    cond = {}
    if length > 0:
        m = torch.zeros(B, L, L).bool()
        m[:, :length, :length] = True
        cond["length"] = m.cuda()
    if mask_info:
        coords = torch.from_numpy(synth.uniform_pm1(int(g["coords_seed"]), "coords_6d", B * C * L * L).reshape(B, C, L, L))
        cond["inpainting"] = {"coords_6d": coords.cuda(), "mask_inpaint": pair_mask(parse_mask_info(mask_info, B, L)).cuda()}
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, C, L, L), 1e-5)
    res = {}
    # the exact-f32 engine's 1000-step figure is taken on the cond_length fixture; on the others (measured in round 4 with T2P_LONG_TESTS=1:
    # cond_length_inpainting 2.1e-6, test_config at L = 128 2.0e-6, at L = 64 2.0e-6; profiles/r04_parity.json) it costs 25 - 80 s each of
    # the GPU box's time limit and runs only with T2P_LONG_TESTS=1
    dts = ("f32", "f16") if stem == "cond_length" or LONG else ("f16",)
    for dt in dts:
        m = _model(cfg, sd, dt)
        noise = CounterNoise(int(g["noise_seed"]))
        out, nfe = fn(m, condition=cond, context=ctx, noise_fn=noise.draw)
        torch.cuda.synchronize()
        assert nfe == int(g["nfe"]) == 2 * N and noise.k == 1 + 2 * N
        res[dt] = rel_l2(out.cpu(), g["sample"])
        print(f"{stem}: 1000 PC steps, {dt} final sample vs the REFERENCE's run: rel-L2 = {res[dt]:.3e}")
        if mask_info:      # the known region (outside the inpainted rows / columns, inside the length mask) is coords_6d exactly
            known = (~cond["inpainting"]["mask_inpaint"]).unsqueeze(1).expand(B, C, L, L).clone()
            known[:, -1] = False
            assert torch.equal(out[known], cond["inpainting"]["coords_6d"][known])
        del m
    _record(f"run1000_vs_reference_{stem}", res)
    assert res.get("f32", 0.0) < F32_RUN1000_TOL and res["f16"] < F16_TOL


# (test_thousand_step_f16_within_tolerance -- the f16 engine against the exact-f32 engine on Philox noise, 4.9e-4 / 5.2e-4 in rounds 2 and 3
# -- was retired in round 4: every shipped YAML now has a 1000-step run of the REFERENCE itself above, and the 42 s go to those.)


def test_up_phase_convolution_matches_gather_form():
    """The first convolution of an up-sampling block runs as four 2x2 phase convolutions on the source map (taps that read
    the same source pixel summed on the host, plan switch 21) instead of a 3x3 gather from the half-resolution map: at the
    benchmark batch of cfg3 (the plan selects it from 200 tiles on) both forms must reproduce the reference's samples."""
    from text2protein_amd import _lib, synth
    stem = "cond_length"
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B0, T)
    xs = torch.from_numpy(synth.normal(79, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 80).cuda()
    ls = (torch.arange(chains, device="cuda") * 29 + 5) % cfg.model.num_scales
    for i, s in enumerate((3, chains - 2)):
        xs[s], cs[s], ls[s] = x[i].cuda(), ctx[i].cuda(), labels[i].cuda()
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(21, sw))
            outs[sw] = m16(xs, ls, cs).cpu()
    finally:
        lib.t2p_debug_set(21, 1)
    assert not torch.equal(outs[0], outs[1]), "the phase form did not run"
    d = rel_l2(outs[1], outs[0])
    print(f"phase form vs gather form of the up-sampling convolutions: rel-L2 = {d:.3e}")
    assert d < 1e-3
    for i, s in enumerate((3, chains - 2)):
        for sw in (0, 1):
            assert rel_l2(outs[sw][s], g["score"][i]) < F16_SCORE_TOL


@pytest.mark.parametrize("stem", ["test_config", "cond_length"])
def test_shortcut_in_the_convolution_matches_separate_shortcut(stem):
    """Blocks with a 1x1 shortcut at their own resolution run it as an extra K segment of their second convolution (plan
    switch 23) instead of a GEMM of its own whose fp32 result the convolution then re-reads: both forms against the
    reference's scores at full size (the 16x16 level of test_config takes the split-K plan, whose splits cross the segment)."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = (t.cuda() for t in full_inputs(cfg, B0, T))
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(23, sw))
            outs[sw] = m16(x, labels, ctx).cpu()
    finally:
        lib.t2p_debug_set(23, 1)
    assert not torch.equal(outs[0], outs[1]), "the fused form did not run"
    d = rel_l2(outs[1], outs[0])
    e0, e1 = rel_l2(outs[0], g["score"]), rel_l2(outs[1], g["score"])
    print(f"{stem}: shortcut in the convolution vs separate: rel-L2 = {d:.3e}; vs reference: separate {e0:.3e}, fused {e1:.3e}")
    _record(f"shortcut_fused_{stem}", {"fused_vs_separate": d, "separate_vs_reference": e0, "fused_vs_reference": e1})
    # two f16 evaluations with differently rounded intermediates are as far from each other as each is from the reference
    assert d < F16_SCORE_TOL and e0 < F16_SCORE_TOL and e1 < F16_SCORE_TOL and e1 < 1.05 * e0


@pytest.mark.parametrize("stem", ["cond_length", "test_config"])
def test_groupnorm_in_the_split_k_second_pass_matches_separate_launches(stem):
    """Plan switch 28: the second pass of the split-K convolutions (16x16 .. 4x4 levels) applies the GroupNorm that follows
    (GroupNorm_1 inside a block, GroupNorm_0 of the next block) instead of a launch of its own.  Both plans against the
    reference's full-size scores at the benchmark batch; the fused one normalises the fp32 values (one rounding fewer)."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B0, T)
    xs = torch.from_numpy(synth.normal(79, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 80).cuda()
    ls = (torch.arange(chains, device="cuda") * 29 + 5) % cfg.model.num_scales
    for i, s in enumerate((3, chains - 2)):
        xs[s], cs[s], ls[s] = x[i].cuda(), ctx[i].cuda(), labels[i].cuda()
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(28, sw))
            outs[sw] = m16(xs, ls, cs).cpu()
            assert torch.equal(outs[sw], m16(xs, ls, cs).cpu())
    finally:
        lib.t2p_debug_set(28, 1)
    assert not torch.equal(outs[0], outs[1]), "the fused second pass did not run"
    d = rel_l2(outs[1], outs[0])
    e = {sw: max(rel_l2(outs[sw][s], g["score"][i]) for i, s in enumerate((3, chains - 2))) for sw in (0, 1)}
    print(f"{stem}: GroupNorm in the split-K second pass vs separate launches: rel-L2 = {d:.3e}; vs reference: separate {e[0]:.3e}, fused {e[1]:.3e}")
    _record(f"post_gn_{stem}", {"fused_vs_separate": d, "separate_vs_reference": e[0], "fused_vs_reference": e[1]})
    assert d < F16_SCORE_TOL and e[1] < F16_SCORE_TOL and e[1] < 1.05 * e[0]


@pytest.mark.parametrize("stem", ["cond_length", "test_config", pytest.param("test_config_large", marks=long_only)])
def test_merged_projections_match_the_reference(stem):
    """Two products the engine forms once at load time in the 16-bit modes, both exact in real arithmetic:
    NIN_2 . NIN_3 of AttnBlockpp (the rows of its softmax sum to 1, layers.py:168-176; plan switch 32) and
    proj_out . ff.net.2 of SpatialTransformer (no nonlinearity between them, attention.py:213-215, 259-263; plan switch 33).
    Merged and unmerged engines against the reference's full-size scores."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = (t.cuda() for t in full_inputs(cfg, B0, T))
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(32, sw))
            _lib.check(lib.t2p_debug_set(33, sw))
            outs[sw] = m16(x, labels, ctx).cpu()
    finally:
        lib.t2p_debug_set(32, 1)
        lib.t2p_debug_set(33, 1)
    assert not torch.equal(outs[0], outs[1]), "the merged forms did not run"
    d = rel_l2(outs[1], outs[0])
    e0, e1 = rel_l2(outs[0], g["score"]), rel_l2(outs[1], g["score"])
    print(f"{stem}: merged projections vs separate: rel-L2 = {d:.3e}; vs reference: separate {e0:.3e}, merged {e1:.3e}")
    _record(f"merged_projections_{stem}", {"merged_vs_separate": d, "separate_vs_reference": e0, "merged_vs_reference": e1})
    assert d < F16_SCORE_TOL and e1 < F16_SCORE_TOL and e1 < 1.1 * e0


@pytest.mark.parametrize("stem", ["cond_length", "cond_length_inpainting", "test_config"])
def test_small_map_convolution_kernel_matches_the_split_k_path(stem):
    """Plan switch 36: the residual blocks of the 8x8 / 4x4 levels run each convolution with its shortcut segment, biases, residual and
    the GroupNorm that follows in one launch (small_conv_gn_kernel) instead of split-K convolution + second pass.  Both plans against
    the reference's full-size scores at the benchmark batch (the kernel is chosen where all its workgroups are resident at once)."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B0, T)
    xs = torch.from_numpy(synth.normal(79, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 80).cuda()
    ls = (torch.arange(chains, device="cuda") * 29 + 5) % cfg.model.num_scales
    for i, s in enumerate((3, chains - 2)):
        xs[s], cs[s], ls[s] = x[i].cuda(), ctx[i].cuda(), labels[i].cuda()
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(36, sw))
            outs[sw] = m16(xs, ls, cs).cpu()
            assert torch.equal(outs[sw], m16(xs, ls, cs).cpu())
    finally:
        lib.t2p_debug_set(36, 1)
    assert not torch.equal(outs[0], outs[1]), "the small-map kernel did not run"
    d = rel_l2(outs[1], outs[0])
    e = {sw: max(rel_l2(outs[sw][s], g["score"][i]) for i, s in enumerate((3, chains - 2))) for sw in (0, 1)}
    print(f"{stem}: small-map convolution kernel vs split-K path: rel-L2 = {d:.3e}; vs reference: split-K {e[0]:.3e}, one launch {e[1]:.3e}")
    _record(f"small_conv_{stem}", {"fused_vs_split_k": d, "split_k_vs_reference": e[0], "fused_vs_reference": e[1]})
    assert d < F16_SCORE_TOL and e[1] < F16_SCORE_TOL and e[1] < 1.05 * e[0]


@pytest.mark.parametrize("stem", ["cond_length", "cond_length_inpainting"])
def test_spatial_transformer_row_chains_match_separate_launches(stem):
    """Plan switch 39: the row-wise chains of every SpatialTransformer block of the C = 256 configurations as one launch each over
    32-row blocks (st_entry_kernel): GroupNorm -> proj_in -> LayerNorm_1 -> q | k | v, and to_out + residual -> LayerNorm_2 -> to_q;
    with switch 40 also to_out + residual -> LayerNorm_3 -> ff.net.0 (GEGLU), with switch 42 the merged ff.net.2 / proj_out product and
    the block's column sums behind it in the same launch (>= 4096 rows; without the third product >= 8192: cond_length's 32 chains,
    not cond_length_inpainting's 16).  All plans against the reference's full-size scores at the benchmark batch."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B0, T)
    xs = torch.from_numpy(synth.normal(79, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 80).cuda()
    ls = (torch.arange(chains, device="cuda") * 29 + 5) % cfg.model.num_scales
    for i, s in enumerate((3, chains - 2)):
        xs[s], cs[s], ls[s] = x[i].cuda(), ctx[i].cuda(), labels[i].cuda()
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for name, sw39, sw40, sw42 in (("separate", 0, 0, 0), ("chains", 1, 0, 0), ("chains+tail", 1, 1, 0), ("chains+tail3", 1, 1, 1)):
            _lib.check(lib.t2p_debug_set(39, sw39))
            _lib.check(lib.t2p_debug_set(40, sw40))
            _lib.check(lib.t2p_debug_set(42, sw42))
            outs[name] = m16(xs, ls, cs).cpu()
            assert torch.equal(outs[name], m16(xs, ls, cs).cpu())
    finally:
        lib.t2p_debug_set(39, 1)
        lib.t2p_debug_set(40, 1)
        lib.t2p_debug_set(42, 1)
    assert not torch.equal(outs["separate"], outs["chains"]), "the row-chain kernel did not run"
    rows = chains * cfg.data.max_res_num ** 2 // 64                  # rows of the 16x16 level
    assert torch.equal(outs["chains"], outs["chains+tail"]) == (rows < 8192), "GEGLU chain without the third product: rows >= 8192 only"
    assert not torch.equal(outs["chains+tail"], outs["chains+tail3"]) and rows >= 4096, "the third product did not run"
    e = {k: max(rel_l2(v[s], g["score"][i]) for i, s in enumerate((3, chains - 2))) for k, v in outs.items()}
    d = {k: rel_l2(outs[k], outs["separate"]) for k in ("chains", "chains+tail", "chains+tail3")}
    print(f"{stem}: SpatialTransformer row chains vs separate launches: rel-L2 = {d}; vs reference: {e}")
    _record(f"st_chains_{stem}", {"vs_separate": d, "vs_reference": e})
    for k in d:
        assert d[k] < F16_SCORE_TOL and e[k] < F16_SCORE_TOL and e[k] < 1.05 * e["separate"]


@pytest.mark.parametrize("stem", ["test_config", pytest.param("test_config_large", marks=long_only)])
def test_fragment_major_attention_operands_match_row_major(stem):
    """Plan switch 45: at the 32 x 32 level of the C = 512 configurations the q | k projection writes its k columns and the transposed
    value projection all of its output fragment-major, and the wide-head attention kernel streams them with whole cache lines per load.
    The values are the same ones in another place: bit-identical scores, and both against the reference's full-size scores."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B0, T)
    xs = torch.from_numpy(synth.normal(79, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 80).cuda()
    ls = (torch.arange(chains, device="cuda") * 29 + 5) % cfg.model.num_scales
    slots = (3, chains - 2)[:B0]
    for i, s in enumerate(slots):
        xs[s], cs[s], ls[s] = x[i].cuda(), ctx[i].cuda(), labels[i].cuda()
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(45, sw))
            lib.t2p_profile_begin()
            outs[sw] = m16(xs, ls, cs).cpu()
            o9 = (__import__("ctypes").c_double * 9)()
            lib.t2p_profile_end(o9)
    finally:
        lib.t2p_debug_set(45, 1)
    e = {sw: max(rel_l2(outs[sw][s], g["score"][i]) for i, s in enumerate(slots)) for sw in (0, 1)}
    print(f"{stem}: fragment-major attention operands: vs reference: row-major {e[0]:.3e}, fragment-major {e[1]:.3e}; identical: {torch.equal(outs[0], outs[1])}")
    _record(f"attn_frag_major_{stem}", {"row_major_vs_reference": e[0], "frag_major_vs_reference": e[1], "identical": bool(torch.equal(outs[0], outs[1]))})
    assert torch.equal(outs[0], outs[1])
    assert e[1] < F16_SCORE_TOL


@pytest.mark.parametrize("stem", ["cond_length", "cond_length_inpainting"])
def test_attention_block_projections_in_one_launch_match(stem):
    """Plan switch 46: GroupNorm apply, q | k and the transposed value projection of every AttnBlockpp of the C = 256 configurations as one
    launch over 32-row blocks (attn_proj_kernel).  Both plans against the reference's full-size scores at the benchmark batch."""
    from text2protein_amd import _lib, synth
    cfg, B0, T, chains = _cfg(stem)
    g = load_golden("full_" + stem)
    sd = synth.synth_state_dict(cfg, 0)
    x, labels, ctx = full_inputs(cfg, B0, T)
    xs = torch.from_numpy(synth.normal(79, "filler_x", chains * x[0].numel()).reshape(chains, *x.shape[1:])).cuda() * 20.0
    cs = synth.synth_context(chains, T, cfg.model.context_dim, 80).cuda()
    ls = (torch.arange(chains, device="cuda") * 29 + 5) % cfg.model.num_scales
    for i, s in enumerate((3, chains - 2)):
        xs[s], cs[s], ls[s] = x[i].cuda(), ctx[i].cuda(), labels[i].cuda()
    lib = _lib.load()
    m16 = _model(cfg, sd, "f16")
    outs = {}
    try:
        for sw in (0, 1):
            _lib.check(lib.t2p_debug_set(46, sw))
            outs[sw] = m16(xs, ls, cs).cpu()
            assert torch.equal(outs[sw], m16(xs, ls, cs).cpu())
    finally:
        lib.t2p_debug_set(46, 1)
    # (the same products accumulated in the same order: the scores come out bit-identical -- that the kernel runs is visible in the
    # dispatch count, 10 fewer per PC step: profiles/README.md)
    d = rel_l2(outs[1], outs[0])
    e = {sw: max(rel_l2(outs[sw][s], g["score"][i]) for i, s in enumerate((3, chains - 2))) for sw in (0, 1)}
    print(f"{stem}: AttnBlockpp projections in one launch vs separate: rel-L2 = {d:.3e}; vs reference: separate {e[0]:.3e}, one launch {e[1]:.3e}")
    _record(f"attn_proj_{stem}", {"fused_vs_separate": d, "separate_vs_reference": e[0], "fused_vs_reference": e[1]})
    assert d < F16_SCORE_TOL and e[1] < F16_SCORE_TOL and e[1] < 1.05 * e[0]
