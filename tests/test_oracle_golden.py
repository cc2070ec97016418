"""The oracle (oracle/t2p_oracle.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import t2p_oracle as O
from text2protein_amd import synth
from text2protein_amd.config import tiny_config
from helpers import load_golden, cfg_tiny, cfg_tinyB, rel_l2

TOL = 1e-5   # float32 restatement with the same torch CPU ops as the reference


@pytest.mark.parametrize("name,cfgf", [("tiny_forward", cfg_tiny), ("tinyB_forward", cfg_tinyB)])
def test_forward_matches_reference_taps(name, cfgf):
    g = load_golden(name)
    cfg = cfgf()
    P = synth.synth_state_dict(cfg, int(g["seed"]))
    taps = {}
    with torch.no_grad():
        score = O.unet_forward(P, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["labels"]),
                               torch.from_numpy(g["context"]), taps=taps)
    assert score.dtype == torch.float64          # reference divides by the float64 sigmas buffer
    n = 0
    for k, v in g.items():
        if k.startswith("tap:"):
            assert rel_l2(taps[k[4:]], v) < TOL, k
            n += 1
    assert n >= 10
    assert rel_l2(score, g["score"]) < TOL


@pytest.mark.parametrize("kind", ["none", "length", "length_inpainting"])
def test_pc_sampler_matches_reference(kind):
    g = load_golden("tiny_sampler_" + kind)
    cfg = cfg_tiny()
    P = synth.synth_state_dict(cfg, int(g["seed"]))
    cond = {}
    if "cond_length" in g:
        cond["length"] = torch.from_numpy(g["cond_length"])
    if "cond_mask_inpaint" in g:
        cond["inpainting"] = {"coords_6d": torch.from_numpy(g["cond_coords_6d"]),
                              "mask_inpaint": torch.from_numpy(g["cond_mask_inpaint"])}
    noise = [torch.from_numpy(z) for z in g["noise"]]
    it = iter(noise)
    trace = []
    B, C, L = 2, cfg.data.num_channels, cfg.data.max_res_num
    out, nfe = O.pc_sampler_ve(P, cfg, (B, C, L, L), torch.from_numpy(g["context"]), condition=cond,
                               noise_fn=lambda shp: next(it), trace=trace)
    assert nfe == int(g["nfe"]) == cfg.model.num_scales * 2
    for i, (x, xm) in enumerate(trace):
        assert rel_l2(x, g[f"x_step{i}"]) < TOL
        assert rel_l2(xm, g[f"xmean_step{i}"]) < TOL
    assert rel_l2(out, g["sample"]) < TOL
    # invariants of sampling.py:260-275
    if "cond_mask_inpaint" in g:      # frozen region (outside length, last channel, not inpainted) == coords_6d
        free = (torch.from_numpy(g["cond_length"]) & torch.from_numpy(g["cond_mask_inpaint"])).unsqueeze(1)
        free = free.expand_as(out).clone()
        free[:, -1] = False
        assert torch.equal(out[~free], torch.from_numpy(g["cond_coords_6d"])[~free])
    elif "cond_length" in g:          # last channel == length mask, zero outside it
        m = torch.from_numpy(g["cond_length"])
        assert torch.equal(out[:, -1], m.float())
        assert float((out[:, :-1] * (~m).unsqueeze(1)).abs().max()) == 0.0


@pytest.mark.parametrize("N", [100, 1000, 2000])
def test_schedule_tables(N):
    g = load_golden("tables")
    dsig = O.ve_discrete_sigmas(0.01, 100.0, N)
    assert np.array_equal(dsig.numpy(), g[f"discrete_sigmas_{N}"])
    ts = O.timesteps(N, 1e-5)
    labels = torch.stack([O.ve_label(ts[i:i + 1].clone(), N)[0] for i in range(N)])
    assert np.array_equal(labels.numpy(), g[f"labels_{N}"])
    assert np.array_equal(labels.numpy(), np.arange(N))              # label at step i is i
    G = torch.cat([O.ve_discretize_G(ts[i:i + 1], dsig, N) for i in range(N)])
    assert np.array_equal(G.numpy(), g[f"G_{N}"])
    cfg = tiny_config(**{"model.num_scales": N})
    assert np.array_equal(O.model_sigmas(cfg).numpy(), g[f"model_sigmas_{N}"])
    vp = O.vp_tables(0.1, 20.0, N)
    assert np.array_equal(vp["alphas"].numpy(), g[f"vp_alphas_{N}"])
    assert np.allclose(vp["sqrt_1m_alphas_cumprod"].numpy(), g[f"vp_sqrt_1m_acp_{N}"], rtol=0, atol=0)


def test_timestep_embedding():
    g = load_golden("tables")
    e = O.timestep_embedding(torch.tensor([0, 1, 7, 999, 1999]), 32)
    assert np.array_equal(e.numpy(), g["temb_32"])


def test_condition_builders():
    m = O.mask_all_lengths(4, 16, 3)
    assert m.shape == (13, 3, 16, 16)
    assert m[0, 0, :4, :4].all() and not m[0, 0, 4:, :].any() and m[-1].all()
    s = O.selected_mask("1:3,6", 2, 8)
    rows = torch.zeros(8, dtype=torch.bool)
    rows[[1, 2, 3, 6]] = True
    assert torch.equal(s[0], rows[:, None] | rows[None, :])


def test_vp_sampler_matches_reference():
    """Lowest-priority row of the path: VP SDE, 40 steps (VP needs beta_max / N < 1)."""
    g = load_golden("tiny_sampler_vp")
    cfg = tiny_config(**{"model.num_scales": 40, "training.sde": "vpsde"})
    P = synth.synth_state_dict(cfg, int(g["seed"]))
    it = iter([torch.from_numpy(z) for z in g["noise"]])
    out, nfe = O.pc_sampler_vp(P, cfg, (2, 5, 16, 16), torch.from_numpy(g["context"]), noise_fn=lambda shp: next(it))
    assert nfe == int(g["nfe"]) == 80
    assert rel_l2(out, g["sample"]) < TOL


# ---- fixtures of tests/golden/make_golden_full.py (BASELINE sizes, `ss`, reference-written checkpoint) ------------
def test_ss_sampler_matches_reference():
    from helpers import cfg_ss
    g = load_golden("tiny_sampler_ss")
    cfg = cfg_ss()
    P = synth.synth_state_dict(cfg, int(g["seed"]))
    it = iter([torch.from_numpy(z) for z in g["noise"]])
    cond = {"length": torch.from_numpy(g["cond_length"]), "ss": torch.from_numpy(g["cond_ss"])}
    out, nfe = O.pc_sampler_ve(P, cfg, (2, 8, 16, 16), torch.from_numpy(g["context"]), condition=cond, noise_fn=lambda s: next(it))
    assert nfe == int(g["nfe"]) and rel_l2(out, g["sample"]) < TOL
    assert torch.equal(out[:, 4:7], torch.from_numpy(g["cond_ss"]))      # sampling.py:268-270: given and frozen


def test_reference_written_checkpoint_layout_and_ema_order():
    """tests/golden/tiny_checkpoint.pth (written by the reference's save_checkpoint): DataParallel keys, float64 sigmas
    buffer, EMA shadow list in parameters() order; the oracle on the EMA tensors reproduces the score the reference
    computed after restore_checkpoint + ema.copy_to, the live tensors the other one."""
    import os
    from helpers import GOLDEN, cfg_ckpt
    from text2protein_amd.arch import param_specs
    from text2protein_amd.checkpoint import ema_state_dict, strip_module_prefix
    g = load_golden("tiny_checkpoint_expected")
    cfg = cfg_ckpt()
    st = torch.load(os.path.join(GOLDEN, "tiny_checkpoint.pth"), map_location="cpu", weights_only=False)
    assert sorted(st) == ["ema", "model", "optimizer", "step"] and st["step"] == 1234
    assert all(k.startswith("module.") for k in st["model"]) and st["model"]["module.sigmas"].dtype == torch.float64
    specs = param_specs(cfg)
    assert len(st["ema"]["shadow_params"]) == len(specs)
    x, labels, ctx = (torch.from_numpy(g[k]) for k in ("x", "labels", "context"))
    ema = ema_state_dict(cfg, st)
    want = synth.synth_state_dict(cfg, int(g["ema_seed"]))
    for s in specs:                                                    # the shadow list is the seed-7 weights, in order
        assert torch.equal(ema[s.name], want[s.name]), s.name
    assert rel_l2(O.unet_forward(ema, cfg, x, labels, ctx), g["score"]) < TOL
    live = {k: v for k, v in strip_module_prefix(st["model"]).items() if k != "sigmas"}
    assert rel_l2(O.unet_forward(live, cfg, x, labels, ctx), g["score_live"]) < TOL


def test_param_tables_match_reference_named_parameters():
    """arch.param_specs (what the checkpoint loader and the oracle build on) against the reference's
    named_parameters() for the four shipped YAMLs at the BASELINE sizes."""
    import json
    import os
    from helpers import FULL, GOLDEN
    from text2protein_amd.arch import param_specs
    from text2protein_amd.config import load_config
    root = GOLDEN.rsplit("/tests", 1)[0]
    tables = json.load(open(os.path.join(GOLDEN, "param_tables.json")))
    counts = {"test_config": 379.5, "cond_length": 75.0, "cond_length_inpainting": 75.0, "test_config_large": 863.3}
    for stem, (fname, L, N, _, _, _) in FULL.items():
        cfg = load_config(os.path.join(root, "configs", fname), **{"data.max_res_num": L, "model.num_scales": N})
        want = [(n, tuple(s)) for n, s in tables[stem]["named_parameters"]]
        assert [(s.name, tuple(s.shape)) for s in param_specs(cfg)] == want
        assert abs(tables[stem]["n_params"] / 1e6 - counts[stem]) < 0.06


def test_oracle_at_full_size_cond_length():
    """One evaluation at a BASELINE size on the CPU (cond_length.yml, L=128, 75 M parameters, a few seconds)."""
    import os
    from helpers import FULL, GOLDEN, full_inputs
    from text2protein_amd.config import load_config
    root = GOLDEN.rsplit("/tests", 1)[0]
    fname, L, N, B, T, _ = FULL["cond_length"]
    cfg = load_config(os.path.join(root, "configs", fname), **{"data.max_res_num": L, "model.num_scales": N})
    g = load_golden("full_cond_length")
    x, labels, ctx = full_inputs(cfg, B, T)
    with torch.no_grad():
        s = O.unet_forward(synth.synth_state_dict(cfg, 0), cfg, x, labels, ctx)
    assert rel_l2(s, g["score"]) < TOL


@pytest.mark.parametrize("name", ["train_tiny", "train_tinyB"])
def test_training_step_matches_reference_autograd(name):
    """the oracle's restated loss / optimize_fn / EMA update (losses.py:26-51,105-134,165-176; ema.py:32-49) against
    loss, gradients and post-step state produced by autograd through the REFERENCE UNetModel (tests/golden/make_golden_train.py):
    every tensor through its norm and a random projection, one tensor of each kind element by element."""
    from helpers import TRAIN_CASES, train_inputs, CounterDropout, projection
    g = load_golden(name)
    case = TRAIN_CASES[name]
    cfg = case["config"]()
    inp = train_inputs(cfg, case)
    sd = synth.synth_state_dict(cfg, case["seed"])
    names = [str(n) for n in g["names"]]
    assert names == list(sd)                                    # parameters() order (the EMA shadow list follows it)
    P = {n: w.clone().requires_grad_(True) for n, w in sd.items()}
    state = dict(step=case["step0"], adam_k=0, ema_updates=0, m={n: torch.zeros_like(w) for n, w in sd.items()},
                 v={n: torch.zeros_like(w) for n, w in sd.items()}, ema={n: w.clone() for n, w in sd.items()})
    drop = CounterDropout(case["seed"], cfg.model.dropout)
    loss, raw = O.train_step(P, state, cfg, dict(inp), inp["t"], inp["z"], cfg.model.condition,
                             dropout=drop.module if cfg.model.dropout > 0 else None)
    assert drop.k == int(g["n_dropout_calls"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    got = {"grads": raw, "post": {n: P[n].detach() for n in names}, "ema": state["ema"], "m": state["m"], "v": state["v"]}
    T = float(g["grad_total_norm"])     # exactly-zero gradients (key bias of an AttnBlockpp) hold rounding noise only: floor the scale
    floor = {"grads": 3e-5 * T, "m": 3e-6 * T, "v": 1e-12 * T * T, "post": 0.0, "ema": 0.0}
    for key, tol in (("grads", 1e-4), ("post", 1e-6), ("ema", 1e-6), ("m", 1e-4), ("v", 2e-4)):
        for i, n in enumerate(names):
            scale = max(float(g[key + "_norm"][i]), floor[key], 1e-30)
            assert abs(float(got[key][n].double().norm()) - float(g[key + "_norm"][i])) <= tol * scale, (key, n)
            assert abs(projection(n, got[key][n]) - float(g[key + "_proj"][i])) <= tol * scale * 10, (key, n)
    full = [k[5:] for k in g if k.startswith("grad:")]
    assert len(full) >= 15
    for n in full:
        assert rel_l2(raw[n], g["grad:" + n]) < 1e-4, n
        assert rel_l2(P[n].detach(), g["post:" + n]) < 1e-6, n
