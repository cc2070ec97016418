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
import socket
import subprocess
import sys

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
        # T2P_DIST_BACKEND=gloo: rehearsals with several ranks on one card (RCCL wants one GPU per rank)
        backend = os.environ.get("T2P_DIST_BACKEND") or (
            "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo")
    kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_local(n_ranks: int, argv, env_extra=None, timeout=None, poll_s=0.2) -> int:
    """Start ``n_ranks`` fresh child interpreters running ``argv`` (a script path + its arguments), one rank per
    GPU of this node, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way
    ``torch.distributed.run`` sets them; returns the exit code of the job: 0 when every rank succeeded, otherwise the
    code of the lowest-numbered rank found failed at that poll (signals as positive numbers).

    The ranks are watched together: as soon as one exits non-zero the others are terminated (they would otherwise sit in
    ``init_process_group`` / a barrier / the gather until the collective's own 10-30 minute timeout, holding every GPU).
    ``timeout`` is a deadline in seconds for the WHOLE job (``subprocess.TimeoutExpired`` after the ranks are killed).

    This is what ``python bench.py --gpus N`` does when it is started directly (no WORLD_SIZE in the
    environment).  It must run BEFORE the calling process touches the GPU: the children are new processes (never
    a re-exec of a process that holds a GPU context), and the parent only waits for them."""
    import time
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # ROCm shares device buffers between the ranks of a node (RCCL's intra-node transport, CUDA-tensor IPC) through IPC
        # handles; hosts whose driver offers only the dmabuf flavour (this pool's: without the variable RCCL start-up fails
        # with "hipIpcGetMemHandle: invalid argument") need the legacy mode switched off.  The launcher passes the caller's
        # value through and sets 0 only when the variable is absent; HSA_ENABLE_IPC_MODE_LEGACY=1 in the caller's
        # environment keeps the legacy mode on hosts that want it.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, *argv], env=env))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    deadline = None if timeout is None else time.monotonic() + float(timeout)
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed:
                stop_all()
                return abs(failed[0])
            if all(c == 0 for c in codes):
                return 0
            if deadline is not None and time.monotonic() > deadline:
                stop_all()
                raise subprocess.TimeoutExpired([sys.executable, *argv], timeout)
            time.sleep(poll_s)
    except BaseException:
        stop_all()                           # interrupted: do not leave ranks behind
        raise


def barrier(dist=None, device=None):
    """Process-group barrier followed by a device synchronisation: both sides of a timed region."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, dist=None, device="cpu") -> float:
    """The slowest rank's figure (a timed region ends when the last rank is done)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, dist=None, device="cpu"):
    """Every rank's own figure, in rank order (one all_gather of a scalar): lets a scaling record separate the code from a slow device."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def allreduce_mean_(t: torch.Tensor, dist=None) -> torch.Tensor:
    """In-place mean over the ranks (data-parallel training: the flat gradient buffer, one collective)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())
    return t


def mean_over_ranks(value: float, dist=None, device="cpu") -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()


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
