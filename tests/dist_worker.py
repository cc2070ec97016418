"""Rank body of the multi-process tests (started by text2protein_amd.distributed.launch_local, one process per rank).

  cpu_control_flow <outdir>   the N > 1 control flow of bench.py / sampling_6d.py on CPU (gloo): process group from
                              the environment, per-rank seeds and chain ids, a timed region between two barriers with
                              the max over ranks, ONE all_gather of the per-rank samples; rank 0 prints a JSON line
  gpu_global_batch <outdir>   (GPU box) fused sampler with the norm all-reduce hook: this rank's single chain of a
                              two-chain global batch, noise injected; saves its final sample
  gpu_train_dp <outdir>       (GPU box) data-parallel training: this rank's half of a 4-sample batch, two steps through
                              losses.get_step_fn(..., dist=...) (gradient all-reduce between backward and the update); saves its
                              parameters, EMA and the losses
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from text2protein_amd import distributed as D   # noqa: E402


def cpu_control_flow(outdir):
    rank, world, _ = D.env_rank_world()
    dist = D.init_process_group("cpu")
    B, C, L = 2, 5, 8
    g = torch.Generator().manual_seed(D.rank_seed(3, rank))
    x = torch.randn(B, C, L, L, generator=g)
    D.barrier(dist, "cpu")
    t0 = time.perf_counter()
    for _ in range(3):                                  # stand-in for the per-rank PC steps: no collective inside
        x = x * 0.5 + 1.0
    if rank == 1:
        time.sleep(0.2)                                 # a slow rank: the job's time is the slowest rank's
    full = D.gather_samples(x, dist)
    D.barrier(dist, "cpu")
    dt = D.max_over_ranks(time.perf_counter() - t0, dist, "cpu")
    torch.save({"x": x, "ids": D.chain_ids(B, rank)}, os.path.join(outdir, f"rank{rank}.pt"))
    if rank == 0:
        torch.save(full, os.path.join(outdir, "gathered.pt"))
        print(json.dumps({"n_gpus": world, "chains": int(full.shape[0]), "seconds": dt}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def gpu_global_batch(outdir):
    from helpers import cfg_tiny
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.model import HipScoreModel
    rank, world, local = D.env_rank_world()
    dev = torch.device("cuda", int(os.environ.get("T2P_FORCE_DEVICE", local)))
    torch.cuda.set_device(dev)
    dist = D.init_process_group(dev)
    cfg = cfg_tiny()
    cfg.device = str(dev)
    model = HipScoreModel(cfg, dtype="f32", device=str(dev))
    model.load_state_dict(synth.synth_state_dict(cfg, 0))
    ctx = synth.synth_context(world, 3, cfg.model.context_dim, 0)[rank:rank + 1]
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    g = torch.Generator().manual_seed(31)
    draws = [torch.randn(world, 5, 16, 16, generator=g) for _ in range(1 + 2 * sde.N)]
    fn = sampling.get_sampling_fn(cfg, sde, (1, 5, 16, 16), 1e-5, global_batch=world,
                                  all_reduce=lambda sums: D.allreduce_norm_sums(sums, dist))
    it = iter(draws)
    out, _ = fn(model, context=ctx, noise_fn=lambda s: next(it)[rank:rank + 1])
    torch.cuda.synchronize()
    torch.save(out.cpu(), os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def gpu_train_dp(outdir):
    from helpers import TRAIN_CASES, train_inputs
    from text2protein_amd import losses, sde_lib, synth
    rank, world, local = D.env_rank_world()
    dev = torch.device("cuda", int(os.environ.get("T2P_FORCE_DEVICE", local)))
    torch.cuda.set_device(dev)
    dist = D.init_process_group(dev)
    case = dict(TRAIN_CASES["train_tiny"], B=4, lengths=[12, 9, 16, 7], step0=6000)
    cfg = case["config"]()
    cfg.device = str(dev)
    inp = train_inputs(cfg, case)
    per = case["B"] // world
    sl = slice(rank * per, (rank + 1) * per)
    model = losses.HipTrainModel(cfg, device=str(dev), seed=1)
    model.load_state_dict(synth.synth_state_dict(cfg, case["seed"]))
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), dist=dist)
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=case["step0"])
    batch = {k: inp[k][sl] for k in ("coords_6d", "mask_pair", "context")}
    out = []
    for it in range(2):
        out.append(step_fn(state, batch, condition=cfg.model.condition, t=inp["t"][sl] * (0.8 ** it), z=inp["z"][sl]))
    torch.cuda.synchronize()
    torch.save({"losses": out, "param": model.read(losses.PARAM), "ema": model.read(losses.EMA), "step": model.get_step()},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    {"cpu_control_flow": cpu_control_flow, "gpu_global_batch": gpu_global_batch, "gpu_train_dp": gpu_train_dp}[sys.argv[1]](sys.argv[2])
