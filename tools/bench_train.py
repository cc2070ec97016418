"""Time the training step (t2p_train_step: loss + backward + Adam + EMA, fp32) at a BASELINE model size.

    python tools/bench_train.py --config cond_length.yml --batch 8 --steps 5 [--tokens 64] [--dropout 0.1]

Prints one JSON line: ms per step, samples/s, the loss sequence, device memory, and the achieved fp32 matrix rate against the
157.3 TFLOP/s f32 MFMA peak, counting a step as 3 x the forward pass AS EXECUTED (the text K / V projections are inside a training
step: the context changes with every batch).  Measurement tool, not part of the product path or of bench.py's headline."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FWD_GFLOP = {"test_config.yml": 714.0, "cond_length.yml": 151.2, "cond_length_inpainting.yml": 151.4, "test_config_large.yml": 3099.6}   # SURVEY 8(d), L = 128 / 256, T = 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cond_length.yml")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--dropout", type=float, default=-1.0)
    a = ap.parse_args()
    from text2protein_amd import losses, sde_lib, synth
    from text2protein_amd.config import load_config
    over = {"data.max_res_num": a.L or (256 if "large" in a.config else 128)}
    if a.dropout >= 0:
        over["model.dropout"] = a.dropout
    cfg = load_config(os.path.join(ROOT, "configs", a.config), **over)
    cfg.device = "cuda:0"
    if "optim" not in cfg:
        cfg.optim = dict(optimizer="Adam", lr=1e-4, beta1=0.9, eps=1e-8, weight_decay=0, warmup=5000, grad_clip=1.0)
    cfg.model.setdefault("ema_rate", 0.999)
    cfg.model.setdefault("dropout", 0.1)
    model = losses.HipTrainModel(cfg, device="cuda:0", seed=1)
    model.load_state_dict(synth.synth_state_dict(cfg, 0))
    B, C, L = a.batch, cfg.data.num_channels, cfg.data.max_res_num
    x = torch.from_numpy(synth.uniform_pm1(1, "bench_train_x", B * C * L * L).reshape(B, C, L, L))
    mp = torch.zeros(B, L, L).bool()
    mp[:, :100, :100] = True
    x = x * mp.unsqueeze(1)
    x[:, -1] = mp.float()
    batch = dict(coords_6d=x.cuda(), mask_pair=mp.cuda(), context=synth.synth_context(B, a.tokens, cfg.model.context_dim, 3).cuda())
    if "inpainting" in (cfg.model.condition or []):
        batch["mask_inpaint"] = mp.cuda()
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=5000)
    seq = []
    for _ in range(a.warmup):
        seq.append(step_fn(state, batch, condition=cfg.model.condition))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        seq.append(step_fn(state, batch, condition=cfg.model.condition))
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    gf = FWD_GFLOP.get(a.config, 0.0) * 3 * B
    print(json.dumps({"metric": "training step (fp32)", "config": a.config, "batch": B, "L": L, "tokens": a.tokens, "ms_per_step": dt * 1e3,
                      "samples_per_s": B / dt, "losses": [round(v, 5) for v in seq], "device_GiB": model.device_bytes() / 2 ** 30,
                      "tflops_f32": gf / dt / 1e3, "frac_of_f32_peak": gf / dt / 1e3 / 157.3, "dropout": float(cfg.model.dropout)}))


if __name__ == "__main__":
    main()
