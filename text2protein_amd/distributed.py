"""Multi-GPU plumbing: one process per GPU, chains sharded by rank, one gather at the end.

The reference's only multi-device construct is ``nn.DataParallel`` around the model
(score_sde_pytorch/utils.py:8): scatter + gather inside every forward.  Here each rank runs whole
chains on its own GPU (SURVEY.md 8(e) option A: equivalent to launching the reference once per
GPU with ``--batch_size B/G``) and the only data-path collective of a run is the final
``all_gather`` of the ``(B_local, C, L, L)`` samples (RCCL over xGMI on GPUs, gloo in CPU tests).
``allreduce_norm_sums`` gives option B (global-batch Langevin step size) when wanted.
"""
from __future__ import annotations

import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(device=None, backend=None):
    """torch.distributed over RCCL ("nccl" backend on ROCm) or gloo; rendezvous on 127.0.0.1."""
    import torch.distributed as dist
    rank, world, _ = env_rank_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
    kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def chain_ids(chains_per_rank: int, rank: int):
    """Global ids of the chains a rank owns (rank-major, the order ``gather_samples`` returns)."""
    return list(range(rank * chains_per_rank, (rank + 1) * chains_per_rank))


def rank_seed(seed: int, rank: int) -> int:
    """Distinct noise streams per rank: chains must not repeat across GPUs."""
    return int(seed) * 1000003 + int(rank)


def gather_samples(x: torch.Tensor, dist=None) -> torch.Tensor:
    """all_gather of the per-rank samples, concatenated in rank order (every rank gets the result)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    x = x.contiguous()
    parts = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, x)
    return torch.cat(parts, 0)


def allreduce_norm_sums(sums: torch.Tensor, dist=None) -> torch.Tensor:
    """In-place sum over ranks of ``[sum_b ||grad_b||, sum_b ||noise_b||]`` (reference
    sampling.py:193-195 takes the mean over the whole batch; with the batch sharded over ranks the
    per-rank sums are added and divided by the global chain count)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums
