"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): chains shard by rank with no
per-step collective, one all_gather at the end; the optional global-batch Langevin step size is
an all-reduce of two floats.  The oracle stands in for the per-rank compute."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import t2p_oracle as O
    from text2protein_amd import distributed as D
    dist = D.init_process_group(device="cpu")
    B, C, L = 2, 5, 8
    ids = D.chain_ids(B, rank)
    # per-rank chains: a toy score function keeps this fast; noise stream keyed by rank
    g = torch.Generator().manual_seed(D.rank_seed(3, rank))
    x = torch.randn(B, C, L, L, generator=g) * 10
    grad = -x.double() / 50.0
    noise = torch.randn(B, C, L, L, generator=g)
    # option A: per-rank batch mean (reference semantics of a B-sized batch)
    xa, _ = O.langevin_update(x, grad, noise, 0.17)
    # option B: global-batch mean through an all-reduce of the two norm sums
    sums = torch.stack([grad.reshape(B, -1).norm(dim=-1).sum(), noise.reshape(B, -1).double().norm(dim=-1).sum()])
    D.allreduce_norm_sums(sums, dist)
    step = (0.17 * (sums[1] / (B * world)) / (sums[0] / (B * world))) ** 2 * 2
    xb = x + step * grad + torch.sqrt(step * 2) * noise
    full_a = D.gather_samples(xa.float(), dist)
    full_b = D.gather_samples(xb.float(), dist)
    gsum = torch.full((5,), float(rank + 1))
    D.allreduce_mean_(gsum, dist)                                   # data-parallel training: mean of the flat gradient over the ranks
    lmean = D.mean_over_ranks(10.0 * (rank + 1), dist, "cpu")
    assert torch.equal(gsum, torch.full((5,), (1 + world) / 2.0)) and abs(lmean - 5.0 * (1 + world)) < 1e-12
    all_x = D.gather_samples(x, dist)
    all_grad = D.gather_samples(grad, dist)
    all_noise = D.gather_samples(noise, dist)
    if rank == 0:
        q.put((ids, full_a, full_b, all_x, all_grad, all_noise))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharding_and_gather():
    from oracle import t2p_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ids, full_a, full_b, all_x, all_grad, all_noise = q.get(timeout=100)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    B = 2
    assert ids == [0, 1] and full_a.shape[0] == world * B
    # rank-major order, distinct chains per rank
    assert not torch.equal(all_x[:B], all_x[B:])
    # option A == running the reference once per shard
    for r in range(world):
        sl = slice(r * B, (r + 1) * B)
        xa, _ = O.langevin_update(all_x[sl], all_grad[sl], all_noise[sl], 0.17)
        assert torch.allclose(full_a[sl], xa.float(), rtol=0, atol=0)
    # option B == one reference batch of world * B chains
    xb, _ = O.langevin_update(all_x, all_grad, all_noise, 0.17)
    assert torch.allclose(full_b, xb.float(), rtol=1e-6, atol=1e-5)


@pytest.mark.timeout(180)
def test_launch_local_control_flow(tmp_path, capfd):
    """What `python bench.py --gpus N` does when started directly: N fresh rank processes from the launcher in
    text2protein_amd/distributed.py, a timed region whose figure is the slowest rank's, one all_gather in rank
    order, one JSON line from rank 0."""
    import json
    import sys
    from text2protein_amd import distributed as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = D.launch_local(2, [os.path.join(root, "tests", "dist_worker.py"), "cpu_control_flow", str(tmp_path)], timeout=150)
    assert rc == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["chains"] == 4 and rec["seconds"] >= 0.2      # the slow rank's time
    full = torch.load(tmp_path / "gathered.pt")
    parts = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    assert torch.equal(full, torch.cat([p["x"] for p in parts], 0))                 # rank-major order
    assert [i for p in parts for i in p["ids"]] == [0, 1, 2, 3]
    assert not torch.equal(parts[0]["x"], parts[1]["x"])     # per-rank noise streams differ
    # a failing rank is reported through the exit code
    assert D.launch_local(2, ["-c", "import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)"]) == 3


def test_bench_and_cli_use_the_distributed_module():
    """Both entry points route their multi-rank control flow through text2protein_amd/distributed.py."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("bench.py", "sampling_6d.py"):
        src = open(os.path.join(root, f)).read()
        for name in ("D.launch_local(", "D.init_process_group(", "D.gather_samples(", "D.env_rank_world("):
            assert name in src, (f, name)
        assert "dist.init_process_group(" not in src and "dist.all_gather(" not in src


@pytest.mark.timeout(60)
def test_launch_local_stops_the_other_ranks_when_one_fails(tmp_path):
    """One rank exits 3 while the other sits in a barrier that can never complete: the launcher must return that code within
    seconds and leave no rank behind (it used to wait for the blocked rank's own collective timeout)."""
    import time
    from text2protein_amd import distributed as D
    pidfile = tmp_path / "blocked.pid"
    body = ("import os, sys, time\n"
            "if os.environ['RANK'] == '1':\n"
            "    time.sleep(0.5); sys.exit(3)\n"
            f"open({str(pidfile)!r}, 'w').write(str(os.getpid()))\n"
            "import torch.distributed as dist\n"
            "dist.init_process_group('gloo', rank=0, world_size=2)\n"      # the peer never joins: blocks here
            "dist.barrier()\n")
    t0 = time.monotonic()
    rc = D.launch_local(2, ["-c", body])
    assert rc == 3 and time.monotonic() - t0 < 30
    pid = int(pidfile.read_text())
    with pytest.raises(ProcessLookupError):
        os.kill(pid, 0)                                     # the blocked rank was reaped


@pytest.mark.timeout(60)
def test_launch_local_deadline_covers_the_whole_job():
    import subprocess
    import time
    from text2protein_amd import distributed as D
    t0 = time.monotonic()
    with pytest.raises(subprocess.TimeoutExpired):
        D.launch_local(2, ["-c", "import time; time.sleep(60)"], timeout=1.5)
    assert time.monotonic() - t0 < 15


def _trampoline_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from text2protein_amd import distributed as D
    from text2protein_amd import sampling
    from text2protein_amd._lib import T2PError
    dist = D.init_process_group(device="cpu")
    sums = torch.tensor([1.0 + rank, 10.0 * (1 + rank)])
    calls = []

    def all_reduce(t):
        calls.append(len(calls))
        if len(calls) == 2 and rank == 1:
            raise ConnectionError("xGMI link 3 went away (rank 1, corrector step 2)")
        D.allreduce_norm_sums(t, dist)

    cb, pending = sampling.allreduce_trampoline(all_reduce, sums)
    # what t2p_sampler_step does with the hook (engine.cpp, Sampler::step): call it through the C function pointer; non-zero = fail the step
    rc1 = cb(C.c_void_p(sums.data_ptr()), None, None)
    first = sums.clone()
    rc2 = cb(C.c_void_p(sums.data_ptr()), None, None) if rank == 1 else 0     # rank 0 is not in a collective here: rank 1 fails BEFORE it
    text = None
    try:
        if rc2 != 0:
            sampling.raise_allreduce_error(pending, T2PError("libt2p_hip call failed (status 1): all-reduce callback"))
    except T2PError as e:
        text = (str(e), type(e.__cause__).__name__, len(pending))
    q.put((rank, rc1, first.tolist(), rc2, text))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduce_trampoline_surfaces_the_callables_own_exception():
    """Option B's hook (sampling.allreduce_trampoline, what PCStepper installs with t2p_sampler_set_norm_allreduce) on two gloo ranks:
    the first call sums the two norm sums over the ranks; then rank 1's callable raises -- the C-side sees status 1, and the caller
    gets a T2PError carrying the ORIGINAL exception's text with the exception itself as __cause__ (the advisor fix of round 3)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trampoline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=100)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        rc1, first, _, _ = got[rank]
        assert rc1 == 0 and first == [3.0, 30.0]                     # summed over both ranks, in place
    assert got[0][2] == 0 and got[0][3] is None
    rc2, text = got[1][2], got[1][3]
    assert rc2 == 1 and text is not None
    assert "xGMI link 3 went away (rank 1, corrector step 2)" in text[0] and "norm all-reduce failed" in text[0]
    assert text[1] == "ConnectionError" and text[2] == 0             # the cause is attached, nothing stays parked
