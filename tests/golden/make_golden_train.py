"""Golden fixtures of the TRAINING STEP, produced by autograd through the REFERENCE UNetModel
(build container only).

    python tests/golden/make_golden_train.py

score_sde_pytorch/losses.py does not import in the build container (module-level `import biotite`), so the ~15 lines of
its loss body (losses.py:105-134) and the 6 lines of optimize_fn (:41-49) are evaluated here as written there, on the
reference's own objects: the reference `UNetModel` in train mode through the reference's `get_score_fn(sde, model,
train=True)` (models/utils.py:126-176), `VESDE.marginal_prob` (sde_lib.py:225-228), `torch.optim.Adam` with the arguments
`get_optimizer` passes (losses.py:26-36), `torch.nn.utils.clip_grad_norm_`, and the reference's
`ExponentialMovingAverage` (models/ema.py).  t, z and the Dropout_0 keep-masks come from the counter-based generator
(text2protein_amd.synth), so the test side regenerates every input; stored are the loss, the gradients (whole tensors for
one parameter of each kind, norm + a fixed random projection for ALL of them) and the parameters / EMA / Adam moments after
ONE step (same form).  Only data is written; no reference source text goes into the repo.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

from helpers import TRAIN_CASES, train_inputs, CounterDropout, projection          # noqa: E402
from text2protein_amd import synth                         # noqa: E402
from oracle import t2p_oracle as O                         # noqa: E402

from score_sde_pytorch.models import ncsnpp                # noqa: E402  (reference)
from score_sde_pytorch.models import utils as mutils       # noqa: E402  (reference)
from score_sde_pytorch.models.ema import ExponentialMovingAverage   # noqa: E402  (reference)
from score_sde_pytorch import sde_lib                      # noqa: E402  (reference)


def reference_step(cfg, case, inp):
    """One training step on the reference model: step_fn (losses.py:165-176) with loss_fn (:105-134) and optimize_fn (:41-49)."""
    torch.manual_seed(0)
    model = ncsnpp.UNetModel(cfg)
    sd = synth.synth_state_dict(cfg, case["seed"])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert (list(missing) == ["sigmas"] or not missing) and not unexpected
    o = cfg.optim
    optimizer = torch.optim.Adam(model.parameters(), lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps, weight_decay=o.weight_decay)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=case["step0"])
    coords_6d, mask_pair, t, z = inp["coords_6d"], inp["mask_pair"], inp["t"].clone(), inp["z"]
    drop = CounterDropout(case["seed"], cfg.model.dropout)
    real = torch.nn.functional.dropout
    torch.nn.functional.dropout = drop.functional
    try:
        optimizer.zero_grad()
        # ---- loss_fn ------------------------------------------------------------------------------
        score_fn = mutils.get_score_fn(sde, model, train=True)
        mean, std = sde.marginal_prob(coords_6d, t)
        perturbed_data = mean + std[:, None, None, None] * z
        conditional_mask = torch.ones_like(coords_6d).bool()
        for c in cfg.model.condition:
            if c == "length":
                conditional_mask[:, -1] = False
            elif c == "ss":
                conditional_mask[:, 4:7] = False
            elif c == "inpainting":
                conditional_mask = conditional_mask * inp["mask_inpaint"].unsqueeze(1)
        mask = mask_pair.unsqueeze(1) * conditional_mask
        num_elem = mask.reshape(mask.shape[0], -1).sum(dim=-1)
        perturbed_data = torch.where(mask, perturbed_data, coords_6d)
        score = score_fn(perturbed_data, t, inp["context"])
        losses = torch.square(score * std[:, None, None, None] + z) * mask
        losses = torch.sum(losses.reshape(losses.shape[0], -1), dim=-1)
        losses = losses / (num_elem + 1e-8)
        loss = torch.mean(losses)
        loss.backward()
    finally:
        torch.nn.functional.dropout = real
    n_drop = drop.k
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    # ---- optimize_fn ------------------------------------------------------------------------------
    if o.warmup > 0:
        for g in optimizer.param_groups:
            g["lr"] = o.lr * np.minimum(state["step"] / o.warmup, 1.0)
    if o.grad_clip >= 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=o.grad_clip)
    optimizer.step()
    state["step"] += 1
    state["ema"].update(model.parameters())
    names = [n for n, _ in model.named_parameters()]
    post = {n: p.detach().clone() for n, p in model.named_parameters()}
    shadow = dict(zip(names, [s.detach().clone() for s in ema.shadow_params]))
    st = optimizer.state
    m = {n: st[p]["exp_avg"].clone() for n, p in model.named_parameters()}
    v = {n: st[p]["exp_avg_sq"].clone() for n, p in model.named_parameters()}
    return dict(loss=loss.detach(), score=score.detach(), grads=grads, post=post, ema=shadow, m=m, v=v, names=names, sd=sd,
                n_drop=n_drop)


def oracle_step(cfg, case, inp, sd):
    P = {n: w.clone().requires_grad_(True) for n, w in sd.items()}
    state = dict(step=case["step0"], adam_k=0, ema_updates=0, m={n: torch.zeros_like(w) for n, w in sd.items()},
                 v={n: torch.zeros_like(w) for n, w in sd.items()}, ema={n: w.clone() for n, w in sd.items()})
    drop = CounterDropout(case["seed"], cfg.model.dropout)
    batch = dict(coords_6d=inp["coords_6d"], mask_pair=inp["mask_pair"], context=inp["context"], mask_inpaint=inp.get("mask_inpaint"))
    loss, raw = O.train_step(P, state, cfg, batch, inp["t"], inp["z"], cfg.model.condition,
                             dropout=drop.module if cfg.model.dropout > 0 else None)
    return loss, raw, P, state


FULL_TENSORS = [   # one parameter of each kind whose gradient / post-step value is stored whole
    "pre_blocks.0.weight", "pre_blocks.1.bias", "pre_conv.weight", "pre_conv.bias",
    "input_blocks.0.0.GroupNorm_0.weight", "input_blocks.0.0.Conv_0.weight", "input_blocks.0.0.Dense_0.weight",
    "input_blocks.0.0.Conv_1.bias", "mid_blocks.1.NIN_0.W", "mid_blocks.1.NIN_3.b", "mid_blocks.2.norm.bias",
    "mid_blocks.2.proj_in.weight", "mid_blocks.2.transformer_blocks.0.attn1.to_q.weight",
    "mid_blocks.2.transformer_blocks.0.attn2.to_k.weight", "mid_blocks.2.transformer_blocks.0.attn2.to_out.0.bias",
    "mid_blocks.2.transformer_blocks.0.ff.net.0.proj.weight", "mid_blocks.2.transformer_blocks.0.norm2.weight",
    "mid_blocks.2.proj_out.weight", "out.0.bias", "out.2.weight",
]


def fixture(name):
    case = TRAIN_CASES[name]
    cfg = case["config"]()
    if case.get("full_size"):        # the reference's OWN YAML for the reference model (the repo's configs/ hold views of it with the same values)
        import yaml
        from text2protein_amd.config import finalize_config
        with open(os.path.join("/root/reference", "configs", case.get("yaml", "cond_length.yml"))) as f:
            cfg_ref = finalize_config(yaml.safe_load(f), **{"data.max_res_num": cfg.data.max_res_num, "model.num_scales": cfg.model.num_scales})
        cfg_ref.device = "cpu"
        for k in ("nf", "ch_mult", "num_res_blocks", "attn_resolutions", "dropout", "ema_rate", "condition", "n_heads", "context_dim"):
            assert cfg_ref.model[k] == cfg.model[k], k
        assert dict(cfg_ref.optim) == dict(cfg.optim)
        cfg = cfg_ref
    inp = train_inputs(cfg, case)
    r = reference_step(cfg, case, inp)
    names = r["names"]
    full = [n for n in FULL_TENSORS + case.get("extra_full", []) if n in r["grads"]]
    if case.get("full_size"):
        full = [n for n in full if r["grads"][n].numel() <= 20000]
    # the oracle's restatement through its own functional forward
    loss_o, raw_o, P_o, st_o = oracle_step(cfg, case, inp, r["sd"])
    e_loss = abs(float(loss_o) - float(r["loss"])) / abs(float(r["loss"]))
    # (a gradient that is zero in exact arithmetic -- the key bias of an AttnBlockpp -- is rounding noise on both sides, and with several
    # threads not the same noise twice: differences are held against max(|tensor|, 3e-5 of the whole gradient's norm), as in the tests)
    gtot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in r["grads"].values())))
    e_g = max(float((raw_o[n] - r["grads"][n]).norm() / r["grads"][n].norm().clamp_min(3e-5 * gtot)) for n in names)
    e_p = max(float((P_o[n].detach() - r["post"][n]).norm() / r["post"][n].norm()) for n in names)
    e_e = max(float((st_o["ema"][n] - r["ema"][n]).norm() / r["ema"][n].norm()) for n in names)
    moved = max(float((r["post"][n] - r["sd"][n]).abs().max()) for n in names)
    print(f"[{name}] loss {float(r['loss']):.6g} ({r['n_drop']} dropout calls); oracle vs reference: loss {e_loss:.2e}, worst gradient {e_g:.2e}, "
          f"worst post-step parameter {e_p:.2e}, worst EMA {e_e:.2e}; largest parameter move {moved:.3e}", flush=True)
    assert e_loss < 1e-6 and e_g < 1e-4 and e_p < 1e-6 and e_e < 1e-6 and moved > 0
    out = {"loss": np.float64(r["loss"]), "score": (r["score"][:, :, ::8, ::8] if case.get("full_size") else r["score"]).float().numpy(), "names": np.array(names), "n_dropout_calls": np.int64(r["n_drop"])}
    for key in ("grads", "post", "ema", "m", "v"):
        out[key + "_norm"] = np.array([float(r[key][n].double().norm()) for n in names])
        out[key + "_proj"] = np.array([projection(n, r[key][n]) for n in names])
    for n in full:
        out["grad:" + n] = r["grads"][n].numpy()
        out["post:" + n] = r["post"][n].numpy()
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in r["grads"].values())))
    out["grad_total_norm"] = np.float64(total)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("T2P_GOLDEN_THREADS", "2")))
    for nm in (sys.argv[1:] or list(TRAIN_CASES)):
        fixture(nm)
