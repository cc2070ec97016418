"""Training step (first slice) on the GPU, through the C ABI (t2p_train_*, t2p_op_*_backward, t2p_op_tgemm).

Oracle: tests/golden/train_*.npz -- loss, gradients and post-step state produced by autograd through the REFERENCE UNetModel
(tests/golden/make_golden_train.py) -- and, for inputs the fixtures do not hold, the CPU restatement oracle.t2p_oracle.train_step
(itself pinned by those fixtures, tests/test_oracle_golden.py).  Operator-level checks compare with torch autograd in float64.
"""
import ctypes as C

import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import TRAIN_CASES, CounterDropout, load_golden, projection, rel_l2, train_inputs

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-5      # |loss - reference| / |reference|
GRAD_TOL = 1e-4      # rel-L2 per tensor (fp32 sums in another order than torch's CPU kernels)
PARAM_TOL = 1e-5     # rel-L2 of the post-step parameters / EMA per tensor


@pytest.fixture(scope="module")
def lib():
    from text2protein_amd import _lib
    return _lib.load()


_KEEP = []


def dev(t):
    d = t.contiguous().to("cuda")
    _KEEP.append(d)
    return d


@pytest.fixture(autouse=True)
def _release_device_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def check(lib, rc):
    assert rc == 0, lib.t2p_last_error().decode()


# ---- the strided GEMM ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (70, 33, 19), (256, 300, 129), (5, 288, 1000), (1000, 5, 77), (384, 256, 512)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_tgemm_views(lib, M, N, K, ta, tb):
    """C = alpha A B + bias + beta C with A / B given as row- or column-major views (ta / tb = the operand is stored transposed)."""
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K + ta * 2 + tb)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    ref = 0.7 * (a.double() @ b.double()) + bias.double() + 0.5 * c0.double()
    da = dev(a.T.contiguous() if ta else a)
    db = dev(b.T.contiguous() if tb else b)
    out = dev(c0.clone())
    sAm, sAk = (1, M) if ta else (K, 1)
    sBk, sBn = (1, K) if tb else (N, 1)
    check(lib, lib.t2p_op_tgemm(P(da), sAm, sAk, P(db), sBk, sBn, P(out), N, M, N, K, 1, 0, 0, 0, 0.7, 0.5, P(dev(bias)), 1, 0, 0, 0, 0, None))
    assert rel_l2(out.cpu(), ref) < 2e-6


def test_tgemm_split_k_and_batches(lib):
    """Weight-gradient form: K = 20000 rows cut over workgroups with fp32 atomics into an initialised C; and a batch of heads
    addressed by strides (q k^T of 3 samples x 4 heads out of [B][n][heads d] tensors)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 96, 160, 20000
    dy, x = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    out = dev(c0.clone())
    check(lib, lib.t2p_op_tgemm(P(dev(dy)), 1, M, P(dev(x)), N, 1, P(out), N, M, N, K, 1, 0, 0, 0, 1.0, 1.0, None, 0, 0, 0, 0, 0, None))
    assert rel_l2(out.cpu(), c0.double() + dy.double().T @ x.double()) < 3e-6
    B, n, h, d = 3, 50, 4, 24
    q, k = torch.randn(B, n, h * d, generator=g), torch.randn(B, n, h * d, generator=g)
    S = dev(torch.zeros(B, h, n, n))
    dq, dk = dev(q), dev(k)
    for hh in range(h):        # the op-level entry batches over one axis; the engine uses two (sample, head)
        check(lib, lib.t2p_op_tgemm(C.c_void_p(dq.data_ptr() + 4 * hh * d), h * d, 1, C.c_void_p(dk.data_ptr() + 4 * hh * d), 1, h * d,
                                    C.c_void_p(S.data_ptr() + 4 * hh * n * n), n, n, n, d, B, n * h * d, n * h * d, h * n * n, 1.0, 0.0, None, 1,
                                    0, 0, 0, 0, None))
    ref = torch.einsum("bihd,bjhd->bhij", q.double().reshape(B, n, h, d), k.double().reshape(B, n, h, d))
    assert rel_l2(S.cpu(), ref) < 2e-6


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 8, 32, 64), (3, 16, 16, 8, 32), (1, 12, 20, 40, 5)])
def test_tgemm_convolution_weight_gradient(lib, B, H, W, Ci, Co):
    """dW[co][tap][ci] = sum_pixels dY[pixel][co] X[pixel + tap][ci] against torch's conv2d weight gradient (layers.py:89-95)."""
    g = torch.Generator().manual_seed(B + H + Ci)
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, Co, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, padding=1).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, 9 * Ci)                      # [co][tap][ci]
    xn = dev(x.detach().float().permute(0, 2, 3, 1).contiguous())            # NHWC
    dyn = dev(dy.float().permute(0, 2, 3, 1).contiguous())
    out = dev(torch.zeros(Co, 9 * Ci))
    check(lib, lib.t2p_op_tgemm(P(dyn), 1, Co, P(xn), 0, 0, P(out), 9 * Ci, Co, 9 * Ci, B * H * W, 1, 0, Ci, 0, 1.0, 1.0, None, 0, 1, H, W, Ci, None))
    assert rel_l2(out.cpu(), ref) < 3e-6


# ---- backward halves of the operators -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,HW,Cc,G,silu", [(2, 64, 32, 8, 1), (3, 256, 64, 16, 0), (2, 100, 128, 32, 1), (1, 16, 512, 32, 1), (2, 1024, 96, 24, 0)])
def test_groupnorm_backward(lib, B, HW, Cc, G, silu):
    g = torch.Generator().manual_seed(HW + Cc)
    x = (torch.randn(B, HW, Cc, generator=g) * 2 + 0.3).double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).double().requires_grad_(True)
    beta = (0.1 * torch.randn(Cc, generator=g)).double().requires_grad_(True)
    dy = torch.randn(B, HW, Cc, generator=g).double()
    y = F.group_norm(x.permute(0, 2, 1), G, gamma, beta, eps=1e-6)
    y = F.silu(y) if silu else y
    y.backward(dy.permute(0, 2, 1))
    dx0, dg0, db0 = torch.randn(B, HW, Cc, generator=g), torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)   # accumulated into
    dx, dg, db = dev(dx0.clone()), dev(dg0.clone()), dev(db0.clone())
    check(lib, lib.t2p_op_groupnorm_backward(P(dev(x.detach().float())), P(dev(dy.float())), P(dev(gamma.detach().float())),
                                            P(dev(beta.detach().float())), silu, B, HW, Cc, G, 1e-6, P(dx), P(dg), P(db), None))
    assert rel_l2(dx.cpu() - dx0, x.grad) < 2e-5
    assert rel_l2(dg.cpu() - dg0, gamma.grad) < 2e-5 and rel_l2(db.cpu() - db0, beta.grad) < 2e-5


@pytest.mark.parametrize("rows,Cc", [(70, 32), (512, 256), (33, 1024), (200, 96)])
def test_layernorm_backward(lib, rows, Cc):
    g = torch.Generator().manual_seed(rows + Cc)
    x = (torch.randn(rows, Cc, generator=g) * 1.5 - 0.2).double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).double().requires_grad_(True)
    beta = torch.zeros(Cc).double().requires_grad_(True)
    dy = torch.randn(rows, Cc, generator=g).double()
    F.layer_norm(x, (Cc,), gamma, beta, eps=1e-5).backward(dy)
    dx, dg, db = dev(torch.zeros(rows, Cc)), dev(torch.zeros(Cc)), dev(torch.zeros(Cc))
    check(lib, lib.t2p_op_layernorm_backward(P(dev(x.detach().float())), P(dev(dy.float())), P(dev(gamma.detach().float())), rows, Cc, 1e-5,
                                            P(dx), P(dg), P(db), None))
    assert rel_l2(dx.cpu(), x.grad) < 1e-5 and rel_l2(dg.cpu(), gamma.grad) < 1e-5 and rel_l2(db.cpu(), beta.grad) < 1e-5


def test_softmax_and_geglu_backward(lib):
    g = torch.Generator().manual_seed(8)
    rows, n, scale = 300, 77, 0.25
    s = (torch.randn(rows, n, generator=g) * 3).double().requires_grad_(True)
    dp = torch.randn(rows, n, generator=g).double()
    p = F.softmax(scale * s, dim=-1)
    p.backward(dp)
    buf = dev(dp.float())
    check(lib, lib.t2p_op_softmax_backward(P(dev(p.detach().float())), P(buf), rows, n, scale, None))
    assert rel_l2(buf.cpu(), s.grad) < 5e-6
    inner = 96
    u = torch.randn(rows, 2 * inner, generator=g).double().requires_grad_(True)
    dy = torch.randn(rows, inner, generator=g).double()
    a, gate = u.chunk(2, dim=-1)
    (a * F.gelu(gate)).backward(dy)
    du = dev(torch.zeros(rows, 2 * inner))
    check(lib, lib.t2p_op_geglu_backward(P(dev(u.detach().float())), P(dev(dy.float())), P(du), rows, inner, None))
    assert rel_l2(du.cpu(), u.grad) < 5e-6


# ---- the whole step against the reference ---------------------------------------------------------------------------------------
def _model_for(case, cfg, seed_offset=0):
    from text2protein_amd import synth
    from text2protein_amd.losses import HipTrainModel
    cfg.device = "cuda:0"
    m = HipTrainModel(cfg, device="cuda:0", seed=11)
    m.load_state_dict(synth.synth_state_dict(cfg, case["seed"] + seed_offset))
    return m


def _dropout_masks(case, cfg, model):
    """The counter-based keep-masks of the fixture, one per residual block in forward order, NHWC uint8."""
    if cfg.model.dropout <= 0:
        return []
    from oracle import t2p_oracle as O
    drop = CounterDropout(case["seed"], cfg.model.dropout)
    inputs, mid, outs = O.unet_plan(cfg)
    L, nf, B = cfg.data.max_res_num, cfg.model.nf, case["B"]
    table = dict(model.param_table())
    masks, k = [], 0
    side = L
    for stage in inputs + [mid] + outs:
        for kind, prefix, up, down in stage:
            if kind != "res":
                continue
            side = side * 2 if up else side // 2 if down else side
            co = table[prefix + ".Conv_1.weight"][0]
            masks.append(drop.mask(k, (B, co, side, side)).permute(0, 2, 3, 1).contiguous().to(torch.uint8))
            k += 1
    return masks


@pytest.mark.parametrize("name", ["train_tiny", "train_tinyB", "train_cond_length",
                                  pytest.param("train_test_config", marks=pytest.mark.skipif(os.environ.get("T2P_LONG_TESTS") != "1",
                                               reason="49 s (380 M parameters through host-side norms and projections): T2P_LONG_TESTS=1 runs it"))])
def test_training_step_vs_reference(name):
    """ONE training step (losses.py:165-176) against autograd through the reference UNetModel: the loss, every gradient (norm + random
    projection for all tensors, element by element for one tensor of each kind), the parameters, the EMA and both Adam moments after
    the update.  train_tinyB runs with Dropout_0 active (counter-based keep-masks injected), up / down blocks, C = 8 and all three
    conditions; the warm-up factor is step0 / 5000 resp. 1.  train_cond_length (round 4) is BASELINE configs[2]'s model at its REAL size
    (cond_length.yml, L = 128, 75.0 M parameters in 622 tensors, 42 dropout masks, one sample of 100 residues): every tensor through its
    norm and projection, the small ones element by element, the score on an 8-strided grid.  train_test_config is the other architecture
    family at its real WIDTH (test_config.yml: nf 256, 8 heads, AttnBlockpp + SpatialTransformer at three resolutions, 379.5 M parameters)
    on L = 64 maps, no condition (measured in round 4: loss 3.0e-7, gradients 9.7e-6, post-step parameters 5.9e-8; profiles/r04_parity.json)."""
    from text2protein_amd import losses, sde_lib
    g = load_golden(name)
    case = TRAIN_CASES[name]
    cfg = case["config"]()
    inp = train_inputs(cfg, case)
    model = _model_for(case, cfg)
    names = [str(n) for n in g["names"]]
    assert [n for n, _ in model.param_table()] == names
    model.set_dropout_masks(_dropout_masks(case, cfg, model))
    assert len(model._keep) == int(g["n_dropout_calls"])
    batch = {k: inp[k] for k in ("coords_6d", "mask_pair", "context", "mask_inpaint") if k in inp}
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=case["step0"])
    # the loss and the score it is computed from, no update yet
    loss0, score = model.loss(batch, t=inp["t"], z=inp["z"], backward=True, return_score=True)
    e_score = rel_l2(score.cpu()[:, :, ::8, ::8] if case.get("full_size") else score.cpu(), g["score"])
    e_loss = abs(loss0 - float(g["loss"])) / abs(float(g["loss"]))
    grads = model.read(losses.GRAD)
    worst = {}
    # tensors whose gradient is zero in exact arithmetic (the key bias of an AttnBlockpp: softmax rows are shift-invariant) hold rounding noise
    # on both sides: differences are held against max(|tensor|, 3e-5 of the whole gradient's norm)
    T = float(g["grad_total_norm"])
    floor = {"grads": 3e-5 * T, "m": 3e-6 * T, "v": 1e-12 * T * T, "post": 0.0, "ema": 0.0}
    pcache = {}
    for key, got, tol in (("grads", grads, GRAD_TOL),):
        for i, n in enumerate(names):
            scale = max(float(g[key + "_norm"][i]), floor[key], 1e-30)
            assert abs(float(got[n].double().norm()) - float(g[key + "_norm"][i])) <= tol * scale, (key, n)
            assert abs(projection(n, got[n], cache=pcache) - float(g[key + "_proj"][i])) <= 10 * tol * scale, (key, n)
    full = [k[5:] for k in g if k.startswith("grad:")]
    worst["grad"] = max(rel_l2(grads[n], g["grad:" + n]) for n in full)
    print(f"{name}: loss {loss0:.6f} (reference {float(g['loss']):.6f}, rel {e_loss:.1e}), score rel-L2 {e_score:.1e}, "
          f"worst stored gradient rel-L2 {worst['grad']:.1e} over {len(full)} tensors")
    assert e_loss < LOSS_TOL and e_score < 1e-5 and worst["grad"] < GRAD_TOL
    # the step itself
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    loss1 = step_fn(state, batch, condition=cfg.model.condition, t=inp["t"], z=inp["z"])
    assert abs(loss1 - loss0) <= 1e-6 * abs(loss0) and state["step"] == case["step0"] + 1
    assert model.get_step() == (case["step0"] + 1, 1, 1)
    post = {"post": model.read(losses.PARAM), "ema": model.read(losses.EMA), "m": model.read(losses.EXP_AVG), "v": model.read(losses.EXP_AVG_SQ)}
    # Adam's first update is lr g / (|g| + eps): sign-like, so an element whose gradient is within rounding of zero moves by up to lr on either
    # side; at the full learning rate (step0 >= warmup) that is 3e-5 of a small bias tensor's norm: the full-size case gets 1e-4 there
    ptol = 1e-4 if case.get("full_size") else PARAM_TOL
    for key, tol in (("post", ptol), ("ema", ptol), ("m", GRAD_TOL), ("v", 2 * GRAD_TOL)):
        for i, n in enumerate(names):
            scale = max(float(g[key + "_norm"][i]), floor[key], 1e-30)
            assert abs(float(post[key][n].double().norm()) - float(g[key + "_norm"][i])) <= tol * scale, (key, n)
            assert abs(projection(n, post[key][n], cache=pcache) - float(g[key + "_proj"][i])) <= 10 * tol * scale, (key, n)
    worst["post"] = max(rel_l2(post["post"][n], g["post:" + n]) for n in full)
    # the update itself, not only parameters that barely move: (p_after - p_before) against the reference's
    from text2protein_amd import synth
    sd = synth.synth_state_dict(cfg, case["seed"])
    worst["delta"] = max(rel_l2(post["post"][n] - sd[n], torch.from_numpy(g["post:" + n]) - sd[n]) for n in full)
    print(f"{name}: post-step parameters worst rel-L2 {worst['post']:.1e}, parameter UPDATE worst rel-L2 {worst['delta']:.1e}")
    assert worst["post"] < ptol and worst["delta"] < (2e-2 if case.get("full_size") else 5e-3)   # (sign-like first update, see above)
    from test_gpu_baseline import _record
    _record(f"train_step_{name}", {"loss_rel": e_loss, "score_rel_l2": e_score, "grad_rel_l2": worst["grad"], "post_rel_l2": worst["post"],
                                   "update_rel_l2": worst["delta"]})


def test_training_step_vs_oracle_at_a_wider_shape():
    """Shapes the fixtures do not hold (nf = 64, 32 x 32 maps, 64 / 128 channels, 4 heads): loss and every gradient against the
    oracle's autograd on the CPU, then three more steps on both sides (warm-up reached, clipping active)."""
    from oracle import t2p_oracle as O
    from text2protein_amd import losses, sde_lib, synth
    from helpers import cfg_smallC
    cfg = cfg_smallC()
    cfg.model.condition = ["length"]
    cfg.model.dropout = 0.0
    cfg.model.num_scales = 100
    case = dict(seed=6, B=2, T=4, lengths=[32, 20], step0=6000, mask_info=None)
    inp = train_inputs(cfg, case)
    model = _model_for(case, cfg)
    sd = synth.synth_state_dict(cfg, case["seed"])
    Pw = {n: w.clone().requires_grad_(True) for n, w in sd.items()}
    state_o = dict(step=case["step0"], adam_k=0, ema_updates=0, m={n: torch.zeros_like(w) for n, w in sd.items()},
                   v={n: torch.zeros_like(w) for n, w in sd.items()}, ema={n: w.clone() for n, w in sd.items()})
    batch = {k: inp[k] for k in ("coords_6d", "mask_pair", "context")}
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=case["step0"])
    for it in range(3):
        t = inp["t"] * (0.9 ** it)
        z = torch.roll(inp["z"], it, dims=0)
        loss_o, raw = O.train_step(Pw, state_o, cfg, batch, t, z, cfg.model.condition)
        if it == 0:
            l0 = model.loss(batch, t=t, z=z, backward=True)
            gr = model.read(losses.GRAD)
            T = float(torch.sqrt(sum((v.double() ** 2).sum() for v in raw.values())))
            # (the key bias of an AttnBlockpp has a zero gradient in exact arithmetic: rounding noise on both sides, hence the floor)
            eg = max(float((gr[n].double() - raw[n].double()).norm()) / max(float(raw[n].double().norm()), 3e-5 * T) for n in raw)
            print(f"wider shape: loss {l0:.6f} vs oracle {float(loss_o):.6f}, worst gradient rel-L2 over {len(raw)} tensors {eg:.1e}")
            assert abs(l0 - float(loss_o)) <= LOSS_TOL * abs(float(loss_o)) and eg < GRAD_TOL
        loss_h = step_fn(state, batch, condition=cfg.model.condition, t=t, z=z)
        assert abs(loss_h - float(loss_o)) <= 5e-5 * abs(float(loss_o)), (it, loss_h, float(loss_o))
    post, ema = model.read(losses.PARAM), model.read(losses.EMA)
    ep = max(rel_l2(post[n], Pw[n].detach()) for n in post)
    ee = max(rel_l2(ema[n], state_o["ema"][n]) for n in ema)
    print(f"wider shape: after 3 steps parameters {ep:.1e}, EMA {ee:.1e}")
    # three full-rate Adam updates: an element whose gradient is below Adam's eps (1e-8; e.g. the key bias of an AttnBlockpp, zero in exact
    # arithmetic) moves by lr g / (|g| + eps) -- rounding noise in, up to 1e-5 per step out -- so whole tensors agree to ~3e-5, not 1e-6
    assert ep < 1e-4 and ee < 1e-4 and model.get_step() == (case["step0"] + 3, 3, 3)


def test_loss_falls_on_a_fixed_batch_and_eval_uses_the_ema():
    """Properties that need no oracle: repeated steps on one batch with fixed (t, z) lower the loss; the evaluation loss is computed
    from the EMA weights (it lags the training loss); device-drawn t / z / dropout masks give a finite loss and differ call to call."""
    from text2protein_amd import losses, sde_lib
    case = dict(TRAIN_CASES["train_tinyB"], step0=5000)
    cfg = case["config"]()
    inp = train_inputs(cfg, case)
    model = _model_for(case, cfg)
    batch = {k: inp[k] for k in ("coords_6d", "mask_pair", "context", "mask_inpaint")}
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    eval_fn = losses.get_step_fn(sde, train=False)
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=5000)
    a, b = model.loss(batch), model.loss(batch)                       # device-drawn t, z and dropout masks
    assert np.isfinite(a) and np.isfinite(b) and a != b
    model.set_dropout_masks(_dropout_masks(case, cfg, model))          # fixed masks from here on
    e0 = eval_fn(state, batch, condition=cfg.model.condition, t=inp["t"], z=inp["z"])
    seq = [step_fn(state, batch, condition=cfg.model.condition, t=inp["t"], z=inp["z"]) for _ in range(12)]
    e1 = eval_fn(state, batch, condition=cfg.model.condition, t=inp["t"], z=inp["z"])
    print("loss over 12 steps on one batch:", " ".join(f"{v:.4f}" for v in seq), f"| EMA loss {e0:.4f} -> {e1:.4f}")
    assert seq[-1] < seq[0] and all(np.isfinite(seq))
    assert e1 < e0 and e1 > seq[-1]                                    # the EMA follows, behind the live weights
    sd = state["ema"].state_dict()
    assert sd["num_updates"] == 12 and len(sd["shadow_params"]) == len(model.param_table())


def test_trainer_refuses_what_it_does_not_cover():
    from text2protein_amd import losses
    from text2protein_amd._lib import T2PError
    case = TRAIN_CASES["train_tiny"]
    cfg = case["config"]()
    cfg.device = "cuda:0"
    with pytest.raises(T2PError):
        losses.HipTrainModel(cfg, device="cpu")
    m = losses.HipTrainModel(cfg, device="cuda:0")
    inp = train_inputs(cfg, case)
    with pytest.raises(T2PError):                                      # no weights yet
        m.loss({k: inp[k] for k in ("coords_6d", "mask_pair", "context")})
    with pytest.raises(NotImplementedError):
        cfg2 = case["config"]()
        cfg2.optim.optimizer = "SGD"
        losses.HipTrainModel(cfg2, device="cuda:0")
    with pytest.raises(ValueError):
        losses.condition_flags(["length", "shape"])


def test_training_checkpoint_round_trip_in_the_reference_layout(tmp_path):
    """save_checkpoint / restore_training_state (score_sde_pytorch/utils.py:11-26): three steps, save, two more steps; a fresh model
    restored from the file and stepped twice ends at the same parameters, EMA and moments.  The file is the reference's: torch's own
    Adam accepts its optimizer entry, the sampling-side loader (checkpoint.restore_checkpoint) reads its EMA weights, and a
    checkpoint WRITTEN BY THE REFERENCE (tests/golden/tiny_checkpoint.pth) restores into a training state."""
    from text2protein_amd import checkpoint, losses, sde_lib, synth
    from text2protein_amd.model import HipScoreModel
    case = dict(TRAIN_CASES["train_tiny"], step0=4000)
    cfg = case["config"]()
    inp = train_inputs(cfg, case)
    batch = {k: inp[k] for k in ("coords_6d", "mask_pair", "context")}
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))

    def new_state(seed_offset=0, step=4000):
        m = _model_for(case, cfg, seed_offset)
        return dict(model=m, optimizer=losses.get_optimizer(cfg, m.parameters()),
                    ema=losses.ExponentialMovingAverage(m.parameters(), decay=cfg.model.ema_rate), step=step)

    a = new_state()
    for i in range(3):
        step_fn(a, batch, condition=cfg.model.condition, t=inp["t"], z=torch.roll(inp["z"], i, 0))
    path = str(tmp_path / "train_state.pth")
    checkpoint.save_checkpoint(path, a)
    for i in range(3, 5):
        step_fn(a, batch, condition=cfg.model.condition, t=inp["t"], z=torch.roll(inp["z"], i, 0))
    b = new_state(seed_offset=9, step=0)                       # other weights, other counters: everything must come from the file
    checkpoint.restore_training_state(path, b)
    assert b["step"] == 4003 and b["model"].get_step() == (4003, 3, 3)
    for i in range(3, 5):
        step_fn(b, batch, condition=cfg.model.condition, t=inp["t"], z=torch.roll(inp["z"], i, 0))
    for which in (losses.PARAM, losses.EMA, losses.EXP_AVG, losses.EXP_AVG_SQ):
        ta, tb = a["model"].read(which), b["model"].read(which)
        # (fp32 atomics in the weight gradients: not bitwise; tensors whose gradient is zero in exact arithmetic hold noise: floor)
        total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ta.values())))
        worst = max(float((tb[n].double() - ta[n].double()).norm()) / max(float(ta[n].double().norm()), 3e-5 * total) for n in ta)
        assert worst < 1e-4, (which, worst)
    # the file itself
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"optimizer", "model", "ema", "step"} and ck["step"] == 4003
    assert all(k.startswith("module.") for k in ck["model"]) and ck["model"]["module.sigmas"].dtype == torch.float64
    shapes = [tuple(s) for _, s in a["model"].param_table()]
    dummy = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    opt = torch.optim.Adam(dummy, lr=1e-4)
    opt.load_state_dict(ck["optimizer"])                         # torch's own validation of the layout
    assert int(opt.state[dummy[0]]["step"]) == 3 and opt.state[dummy[5]]["exp_avg"].shape == shapes[5]
    assert ck["ema"]["num_updates"] == 3 and [tuple(t.shape) for t in ck["ema"]["shadow_params"]] == shapes
    cfg.device = "cuda:0"
    sm = HipScoreModel(cfg, dtype="f32")
    assert checkpoint.restore_checkpoint(path, sm, cfg) == 4003   # the sampling driver's loader takes the EMA weights from it
    # a checkpoint written by the reference's own save_checkpoint (DataParallel keys, empty Adam state, EMA shadow != live weights)
    from helpers import cfg_ckpt
    import os
    cfg_r = cfg_ckpt()
    cfg_r.model.dropout = 0.0
    cfg_r.device = "cuda:0"
    mr = losses.HipTrainModel(cfg_r, device="cuda:0")
    st = dict(model=mr, optimizer=losses.get_optimizer(cfg_r, mr.parameters()),
              ema=losses.ExponentialMovingAverage(mr.parameters(), decay=cfg_r.model.ema_rate), step=0)
    checkpoint.restore_training_state(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_checkpoint.pth"), st)
    assert st["step"] == 1234 and mr.get_step() == (1234, 0, 0)
    live, ema = mr.read(losses.PARAM), mr.read(losses.EMA)
    sd_live, sd_ema = synth.synth_state_dict(cfg_r, 0), synth.synth_state_dict(cfg_r, 7)
    assert all(torch.equal(live[n], sd_live[n]) and torch.equal(ema[n], sd_ema[n]) for n in live)


def test_data_parallel_training_two_ranks_equal_one_process(tmp_path):
    """The reference trains under DataParallel (score_sde_pytorch/utils.py:8): one gradient over the whole batch.  Two processes
    with half the batch each -- loss + backward, ONE all-reduce of the flat gradient buffer, the same update on both -- must end where
    a single process holding all four samples ends, and both ranks must hold identical parameters.  (Two ranks on one card over gloo;
    on a multi-GPU node the same code runs over RCCL.)"""
    import os
    from text2protein_amd import distributed as D
    from text2protein_amd import losses, sde_lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = D.launch_local(2, [os.path.join(root, "tests", "dist_worker.py"), "gpu_train_dp", str(tmp_path)],
                        env_extra={"T2P_FORCE_DEVICE": "0", "T2P_DIST_BACKEND": "gloo"}, timeout=600)
    assert rc == 0
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert r0["step"] == r1["step"] == (6002, 2, 2) and r0["losses"] == r1["losses"]
    assert all(torch.equal(r0["param"][n], r1["param"][n]) for n in r0["param"])          # the ranks stay in lock step, bit for bit
    case = dict(TRAIN_CASES["train_tiny"], B=4, lengths=[12, 9, 16, 7], step0=6000)
    cfg = case["config"]()
    inp = train_inputs(cfg, case)
    model = _model_for(case, cfg)
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=case["step0"])
    batch = {k: inp[k] for k in ("coords_6d", "mask_pair", "context")}
    want = [step_fn(state, batch, condition=cfg.model.condition, t=inp["t"] * (0.8 ** it), z=inp["z"]) for it in range(2)]
    assert all(abs(a - b) <= 2e-6 * abs(b) for a, b in zip(r0["losses"], want)), (r0["losses"], want)
    for key, which in (("param", losses.PARAM), ("ema", losses.EMA)):
        ref = model.read(which)
        worst = max(rel_l2(r0[key][n], ref[n]) for n in ref)
        print(f"2 ranks x 2 samples (gradient all-reduce) vs 1 process x 4 samples, after 2 steps: {key} worst rel-L2 {worst:.1e}")
        assert worst < 1e-4           # (sign-like Adam updates on near-zero gradient elements, see test_training_step_vs_oracle_at_a_wider_shape)
