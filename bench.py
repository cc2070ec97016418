#!/usr/bin/env python3
"""Benchmark of the reverse-diffusion sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ...` the ranks come from
the environment; started directly, this script first starts N fresh child processes of itself (before it touches
the GPU, text2protein_amd/distributed.py:launch_local) and relays rank 0's JSON line.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "cfg2"): configs/test_config.yml with
data.max_res_num = 128 and model.num_scales = 1000, 32 chains per GPU, 512 text tokens of width
4096, synthetic non-degenerate weights (text2protein_amd/synth.py), unconditional sampling.

A "step" is ONE predictor-corrector step over the batch: 2 score-network evaluations + Langevin
corrector update + reverse-diffusion predictor update (reference sampling.py:279-285).  Every
PC step has identical shapes and cost, and one sample is exactly `num_scales` = 1000 of them, so
    samples/s = chains / (1000 * seconds_per_step).
`--steps 1000` times a complete run.  Inputs (state, text keys/values, weights) are resident in
HBM when the timed region starts; the per-run text K/V projection is loop-invariant and untimed.
Chains shard over ranks with no per-step collective (per-GPU batches, SURVEY 8(e) option A); one
RCCL all_gather of the final samples closes the timed region.

The JSON line (rank 0), beyond the contract's keys:
  dispatches_per_step       device dispatches one PC step enqueues (graph nodes of a captured, unexecuted step)
  roofline                  the dominant 3x3-convolution instantiation of the benchmarked precision against the dense MFMA peak:
                            every GEMM launch of one PC step timed with HIP events on the launch stream (t2p_profile_*)
  traffic_measured_in_run   false: roofline.traffic is the HBM byte count of the committed rocprofv3 --pmc passes of this
                            command (profiles/), not a figure of this run
  f32                       N = 1: the SAME workload on the exact-f32 engine (v_mfma_f32_32x32x2_f32, the reference's own
                            arithmetic type): --f32-steps (10) steps after a warm-up step, with its own kernel roofline
                            against the 157.3 TFLOP/s f32 matrix peak
  train_step                N = 1: the fp32 training step (loss + backward + Adam + EMA) on cond_length.yml at full size, 16 samples: an
                            extra, not the headline
  cpu_baseline              N = 1: the oracle on the host cores, 2 chains x 3 PC steps after one warm-up step, scaled by N
  cfg3, cfg5                N > 1: BASELINE configs[2] (cond_length.yml, 32 chains per GPU, length condition: the workload the north
                            star's 1000 samples/min is about) and configs[4] (cond_length_inpainting.yml, 16 chains per GPU, length +
                            inpainting) measured by the same ranks with the same bracketing
  per_rank_ms_per_step      every rank's OWN device time per PC step (HIP events around its steps, all-gathered after the timed
                            region), in rank order, next to the max-over-ranks wall time `ms_per_step` is computed from
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config file, overrides, chains per GPU, text tokens, condition)
    "cfg2": ("test_config.yml", {"data.max_res_num": 128, "model.num_scales": 1000}, 32, 512, None),
    "cfg3": ("cond_length.yml", {"data.max_res_num": 128, "model.num_scales": 1000}, 32, 512, "length"),
    "cfg4": ("test_config_large.yml", {"data.max_res_num": 256, "model.num_scales": 1000}, 16, 512, None),
    "cfg5": ("cond_length_inpainting.yml", {"data.max_res_num": 128, "model.num_scales": 1000}, 16, 512, "length+inpainting"),
}
# algorithmic GFLOP per score evaluation per sample, K/V projections excluded
ALG_GFLOP = {"cfg2": 645.3, "cfg3": 138.3, "cfg4": 2970.8, "cfg5": 138.5}
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}   # dense, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=os.environ.get("T2P_BENCH_DTYPE", "f16"), choices=["f32", "bf16", "f16"])
    ap.add_argument("--batch", type=int, default=None, help="chains per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", type=int, default=0, help="measurement only (tools/r04_graph_gaps.sh): replay each PC step from a captured "
                    "hipGraph; slower than eager launches at every workload (DESIGN.md section 8), never the benchmark's mode")
    ap.add_argument("--plan", default="", help="A/B measurements: 'key=value,...' plan switches of t2p_debug_set "
                    "(include/t2p.h: tile geometry, split-K, individual fusions; all produce correct results)")
    ap.add_argument("--lib", default="", help="A/B measurements: load this libt2p_hip.so (built from another revision)")
    ap.add_argument("--shapes", default="", help="write the profiled steps' GEMM / convolution launches by operand shape (CSV) to this file")
    ap.add_argument("--layers", default="", help="write the per-block times of one PC step (HIP events at block boundaries, CSV) to this file")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-f32 engine's line (rank 0, N = 1)")
    ap.add_argument("--f32-steps", type=int, default=10)
    ap.add_argument("--no-train", action="store_true", help="skip the training-step line (rank 0, N = 1)")
    ap.add_argument("--no-cfg3", action="store_true", help="with --gpus N > 1: skip the additional cfg3 / cfg5 (BASELINE configs[2] / configs[4]) measurements")
    return ap.parse_args()


def cpu_baseline(cfg, sd, ctx_cpu, n_scales, k_steps=3):
    """The oracle (CPU restatement of the reference, checked against the reference's own runs) timed on this box's host
    cores on a bounded sample of the same workload: 2 chains, one warm-up PC step (thread pools,
    allocator, caches), then k = 3 PC steps (6 score evaluations) timed (2 or 1 when the warm-up step shows a host on which
    three would take more than about half a minute); scaled by N PC steps per sample -- every step has the same shapes and cost."""
    from oracle import t2p_oracle as O
    B = 2
    C_, L = cfg.data.num_channels, cfg.data.max_res_num
    g = torch.Generator().manual_seed(0)
    draw = lambda s: torch.randn(*s, generator=g)   # noqa: E731
    with torch.no_grad():
        t0 = time.perf_counter()
        O.pc_sampler_ve(sd, cfg, (B, C_, L, L), ctx_cpu[:B], noise_fn=draw, n_steps_limit=1)          # warm-up step
        warm = time.perf_counter() - t0
        if warm > 25.0:            # a slow host: keep the leg bounded (about 30 s of CPU work), and say so in `sample`
            k_steps = 1
        elif warm > 12.0:
            k_steps = 2
        t0 = time.perf_counter()
        O.pc_sampler_ve(sd, cfg, (B, C_, L, L), ctx_cpu[:B], noise_fn=draw, n_steps_limit=k_steps)
        dt = time.perf_counter() - t0
    return {"value": B / (dt / k_steps * n_scales), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{B} chains x {k_steps} PC steps ({2 * k_steps} score evaluations) of the same workload after one warm-up "
                      f"step, on the host CPU with {torch.get_num_threads()} torch intra-op threads ({os.cpu_count()} logical cores "
                      f"visible), scaled by {n_scales} PC steps per sample; {dt:.2f} s measured"}


class Job:
    """One workload on this rank's GPU: model, text K/V, fused stepper, state."""

    def __init__(self, name, dtype, dev, rank, batch=None):
        from text2protein_amd import distributed as D
        from text2protein_amd import sampling, sde_lib, synth
        from text2protein_amd.config import load_config
        from text2protein_amd.model import HipScoreModel
        fname, overrides, chains, T, cond_kind = WORKLOADS[name]
        self.name, self.fname, self.dtype, self.dev, self.T, self.cond_kind = name, fname, dtype, dev, T, cond_kind
        self.B = B = batch or chains
        self.cfg = cfg = load_config(os.path.join(ROOT, "configs", fname), **overrides)
        cfg.device = str(dev)
        self.N = cfg.model.num_scales
        self.C, self.L = cfg.data.num_channels, cfg.data.max_res_num
        t0 = time.perf_counter()
        self.sd = synth.synth_state_dict(cfg, seed=0)                 # same weights on every rank (replicated model)
        self.model = HipScoreModel(cfg, dtype=dtype, device=str(dev))
        self.model.load_state_dict(self.sd)
        self.ctx_cpu = synth.synth_context(B, T, cfg.model.context_dim, seed=1000 + rank)
        self.ctx = self.ctx_cpu.to(dev)
        self.model.set_context(self.ctx)                              # one-off K/V projection of the frozen text
        self.sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=self.N)
        self.stepper = self.new_stepper(self.model, rank)
        self.x = sampling._device_randn_like(torch.empty(B, self.C, self.L, self.L, device=dev), D.rank_seed(12345, rank), 0) * self.sde.prior_scale()
        self.cond = None
        if cond_kind:
            from text2protein_amd.conditions import synthetic_condition
            x, mask = sampling.apply_conditions(self.x, synthetic_condition(cfg, B, cond_kind, dev))
            self.x = x.float().contiguous()
            self.cond = (mask.to(torch.uint8).contiguous(), self.x.clone())
            self.stepper.set_condition(*self.cond)
        self.x_mean = torch.empty_like(self.x)
        self.stepper.reset(0)
        torch.cuda.synchronize()
        self.setup_s = time.perf_counter() - t0

    def new_stepper(self, model, rank):
        from text2protein_amd import distributed as D
        from text2protein_amd import sampling
        c = self.cfg.sampling
        return sampling.PCStepper(model, self.sde, self.B, c.snr, c.n_steps_each, c.probability_flow, c.noise_removal, 1e-5,
                                  seed=D.rank_seed(0, rank))

    def timed(self, steps, warmup, dist, graph=0):
        """W untimed steps, then exactly `steps` PC steps between barrier + device synchronisation on both sides; the
        slowest rank's time; the single all_gather of a run is inside the region."""
        from text2protein_amd import distributed as D
        st, x, xm, dev = self.stepper, self.x, self.x_mean, self.dev
        side = torch.cuda.Stream(device=dev) if graph else None           # stream capture needs a real stream
        run_step = st.step
        if graph:
            def run_step(a, b):
                with torch.cuda.stream(side):
                    st.step_graph(a, b)
            torch.cuda.synchronize()
            for _ in range(2):                                            # eager step + capture
                run_step(x, xm)
            torch.cuda.synchronize()

        def rewind():
            # the device-side step counter is written on the current stream: nothing of the side stream may be in flight
            if side is not None:
                side.synchronize()
            st.reset(0)

        for _ in range(warmup):
            run_step(x, xm)
        if dist is not None:
            D.gather_samples(xm, dist)                       # RCCL communicator set-up is not part of a run
        # the schedule tables hold N steps: the timed region starts at step 0 of a run and rewinds every N steps
        rewind()
        D.barrier(dist, dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            if i and i % self.N == 0:
                rewind()
            run_step(x, xm)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        ev1.record()                                         # this rank's own steps end here (device time, no host wait added)
        gathered = D.gather_samples(xm, dist)                # the single collective of a run (no-op at N = 1)
        D.barrier(dist, dev)
        dt = D.max_over_ranks(time.perf_counter() - t0, dist, dev)
        # every rank's OWN device time per step, gathered after the region: a short scaling curve can then be read as "the code"
        # (all ranks slow) or "one slow device" (the timed region ends at the slowest rank)
        self.per_rank_ms = D.all_ranks(ev0.elapsed_time(ev1) / steps, dist, dev)
        world = dist.get_world_size() if dist is not None else 1
        finite = bool(torch.isfinite(gathered).all().item()) and gathered.shape[0] == self.B * world
        return dt, finite

    def describe(self):
        return (f"{self.name}: {self.fname} L={self.L} N={self.N} chains/GPU={self.B} C={self.C} text_tokens={self.T} "
                f"condition={self.cond_kind or 'none'}; step = 1 PC step (2 score evals + SDE updates); sample = {self.N} PC steps")


def kernel_roofline(job, stepper, lib, peak, shapes="", layers=""):
    """Every GEMM / convolution launch of ONE PC step timed with HIP events on the launch stream (t2p_profile_*): the dominant
    3x3-convolution instantiation against the dense MFMA peak of the dtype, plus the other kernel classes."""
    from text2protein_amd._lib import check
    x, x_mean = job.x, job.x_mean
    stepper.reset(0)
    check(lib.t2p_profile_begin())
    stepper.step(x, x_mean)
    o = (C.c_double * 9)()
    check(lib.t2p_profile_end(o))
    conv_ms, conv_fl, conv_n, g_ms, g_fl, g_n, c1_ms, c1_fl, c1_n = list(o)
    dom, dom_name = (C.c_double * 4)(), C.create_string_buffer(256)
    check(lib.t2p_profile_dominant(dom, dom_name, 256))
    dom_ms, dom_fl, dom_n, dom_bytes = list(dom)
    att = (C.c_double * 3)()
    check(lib.t2p_profile_attention(att))
    att_ms, att_fl, att_n = list(att)
    if shapes:           # the GEMM / convolution launches of the profiled step by operand shape (tools/shapes_table.py)
        buf = C.create_string_buffer(1 << 20)
        check(lib.t2p_profile_shapes(buf, len(buf)))
        os.makedirs(os.path.dirname(os.path.abspath(shapes)), exist_ok=True)
        with open(shapes, "w") as f:
            f.write(f"# {job.name} {job.dtype} chains={job.B}: 1 profiled PC step(s)\n" + buf.value.decode())
    if layers:           # per-block times of one PC step (tools/layer_table.py)
        check(lib.t2p_profile_layers_begin())
        stepper.step(x, x_mean)
        buf = C.create_string_buffer(1 << 20)
        check(lib.t2p_profile_layers_end(buf, len(buf)))
        os.makedirs(os.path.dirname(os.path.abspath(layers)), exist_ok=True)
        with open(layers, "w") as f:
            f.write(f"# {job.name} {job.dtype} chains={job.B}: one PC step (2 score evaluations)\n" + buf.value.decode())
    tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0   # noqa: E731
    if conv_n == 0:            # fp32 mode: every convolution runs on the register-staged exact-f32 kernel
        conv_ms, conv_fl, conv_n, c1_ms, c1_fl, c1_n = c1_ms, c1_fl, c1_n, 0.0, 0.0, 0.0
        kname = "gemm_kernel<float, ...> (register-staged implicit-GEMM 3x3 convolution, v_mfma_f32_32x32x2_f32)"
    else:
        kname = "gemm_dma_kernel (LDS-DMA implicit-GEMM 3x3 convolution)"
    all_conv = {"achieved": tf(conv_fl, conv_ms), "launches_per_step": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
                "share_of_step_ms": conv_ms}
    if dom_n > 0:
        # the dominant instantiation by name: its average launch duration is the figure to hold against the rocprofv3
        # --stats average of the same kernel in profiles/<round>_kernel_stats.csv
        kname = dom_name.value.decode()
        conv_ms, conv_fl, conv_n = dom_ms, dom_fl, dom_n
    ach = tf(conv_fl, conv_ms)
    return {
        "bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
        "launches_per_step": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
        "algorithmic_gflop_per_launch": conv_fl / max(conv_n, 1) / 1e9,
        "algorithmic_bytes_per_launch": (dom_bytes / dom_n) if dom_n > 0 else None,
        "share_of_step_ms": conv_ms, "all_conv3x3_launches": all_conv,
        "other_gemm": {"achieved": tf(g_fl, g_ms), "launches_per_step": g_n, "share_of_step_ms": g_ms, "frac": tf(g_fl, g_ms) / peak},
        "conv_on_v1_kernel": {"launches_per_step": c1_n, "share_of_step_ms": c1_ms},
        # fused self / text cross-attention (attn_flash_kernel): the "attention roofline" of the north star
        "attention": {"achieved": tf(att_fl, att_ms), "launches_per_step": att_n, "share_of_step_ms": att_ms, "frac": tf(att_fl, att_ms) / peak},
    }


def train_step_line(dev, fname="cond_length.yml", batch=16, steps=3):
    """The training step (reference losses.py:165-176: loss, backward, warm-up + clip + Adam, EMA), fp32, on BASELINE
    configs[2]'s model at its real size: `batch` samples of 100 residues at L = 128, 512 text tokens, dropout as shipped, t / z / masks
    drawn on the device; one warm-up step, then `steps` timed.  An extra of the line, not the headline: the reference publishes no
    training figure.  FLOPs: 3 x the forward pass as executed (151.2 GFLOP per sample, text K / V projections included)."""
    from text2protein_amd import losses, sde_lib, synth
    from text2protein_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "configs", fname), **{"data.max_res_num": 128, "model.num_scales": 1000})
    cfg.device = str(dev)
    model = losses.HipTrainModel(cfg, device=str(dev), seed=1)
    model.load_state_dict(synth.synth_state_dict(cfg, 0))
    B, C_, L = batch, cfg.data.num_channels, cfg.data.max_res_num
    x = torch.from_numpy(synth.uniform_pm1(1, "bench_train_x", B * C_ * L * L).reshape(B, C_, L, L))
    mp = torch.zeros(B, L, L).bool()
    mp[:, :100, :100] = True
    x = x * mp.unsqueeze(1)
    x[:, -1] = mp.float()
    b = dict(coords_6d=x.to(dev), mask_pair=mp.to(dev), context=synth.synth_context(B, 512, cfg.model.context_dim, 3).to(dev))
    sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=losses.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=5000)
    seq = [step_fn(state, b, condition=cfg.model.condition)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        seq.append(step_fn(state, b, condition=cfg.model.condition))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = 151.2e9 * 3 * B / dt / 1e12
    return {"value": B / dt, "unit": "training samples/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": 1, "dtype": "f32",
            "workload": f"{fname} L=128 batch={B} text_tokens=512 dropout={cfg.model.dropout}: loss + backward + Adam + EMA",
            "losses": [round(v, 5) for v in seq], "finite": all(v == v and abs(v) < 1e9 for v in seq),
            "tflops_f32": tf, "frac_of_f32_mfma_peak": tf / MFMA_PEAK_TFLOPS["f32"], "device_gib": model.device_bytes() / 2 ** 30}


def main():
    args = parse()
    from text2protein_amd import distributed as D            # imports torch; does not touch the GPU
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started directly with --gpus N: N fresh children, one rank per GPU; this process never initialises HIP
        sys.exit(D.launch_local(args.gpus, [os.path.abspath(__file__), *sys.argv[1:]]))
    rank, world, local_rank = D.env_rank_world()
    args.gpus = world
    # rehearsal on a one-GPU box: T2P_FORCE_DEVICE=0 T2P_DIST_BACKEND=gloo put every rank on one card
    dev_index = int(os.environ.get("T2P_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = D.init_process_group(dev)                         # "nccl" = RCCL over xGMI on ROCm; None at N = 1

    from text2protein_amd._lib import load, set_plan_switches
    from text2protein_amd.model import HipScoreModel
    if args.lib:
        from text2protein_amd import _lib as _L
        _L.load_path(args.lib)
    set_plan_switches(args.plan)
    lib = load()

    job = Job(args.workload, args.dtype, dev, rank, args.batch)
    B, N = job.B, job.N
    dt, finite = job.timed(args.steps, args.warmup, dist, args.graph)
    ms_per_step = dt / args.steps * 1e3
    value = B * world / (N * dt / args.steps)
    alg = ALG_GFLOP[args.workload] * 1e9
    out = {
        "metric": "6D backbone samples/sec (128-res, 1000-step)" if job.L == 128 else f"6D backbone samples/sec ({job.L}-res, {N}-step)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic (hash-generated non-degenerate weights, N(0,1) text embeddings, on-device Philox noise)",
        "config": {"workload": job.describe(),
                   "samples_per_min": value * 60.0, "setup_s": job.setup_s, "finite": finite,
                   "mfma_frac_end_to_end": value * 2 * N * alg / world / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12),
                   "device_gib": job.model.device_bytes() / 2 ** 30},
        # roofline.traffic is NOT measured by this run: it is the HBM byte count of the committed rocprofv3 --pmc passes of this
        # same command (profiles/), see roofline.traffic_from_profile
        "traffic_measured_in_run": False,
        "per_rank_ms_per_step": job.per_rank_ms,
    }
    if rank == 0 and not args.graph:
        job.stepper.reset(0)
        out["dispatches_per_step"] = job.stepper.count_dispatches(job.x, job.x_mean)     # graph nodes of one captured PC step

    if rank == 0 and not args.no_roofline:
        r = kernel_roofline(job, job.stepper, lib, MFMA_PEAK_TFLOPS[args.dtype], args.shapes, args.layers)
        tfile = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        r["traffic_from_profile"] = None
        if tfile and args.workload == "cfg2" and args.dtype == "f16" and B == 32:
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
            # (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction; profiles/README.md)
            t = json.load(open(tfile[-1]))
            if r["kernel"] in t:
                r["traffic"] = t[r["kernel"]]["hbm_bytes_per_launch"]
                r["traffic_from_profile"] = {"bytes_per_launch": r["traffic"], "file": os.path.relpath(tfile[-1], ROOT)}
        out["roofline"] = r
    if world > 1 and args.workload == "cfg2" and not args.no_cfg3:
        # BASELINE.json configs[2] (cond_length.yml, 32 chains per GPU, length condition: the workload the north star's
        # ">= 1000 samples/min on 8 GPUs" is about) and configs[4] (cond_length_inpainting.yml, 16 chains per GPU, length + inpainting:
        # the other 8-GPU configuration), measured by the same ranks with the same bracketing; cfg2 stays the headline
        for name in ("cfg3", "cfg5"):
            jobx = Job(name, args.dtype, dev, rank)
            dtx, finx = jobx.timed(args.steps, args.warmup, dist, 0)
            vx = jobx.B * world / (jobx.N * dtx / args.steps)
            out[name] = {"value": vx, "unit": "samples/s", "samples_per_min": vx * 60.0, "ms_per_step": dtx / args.steps * 1e3,
                         "per_rank_ms_per_step": jobx.per_rank_ms, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                         "finite": finx, "workload": jobx.describe()}
            del jobx
    if rank == 0 and world == 1 and not args.no_f32 and args.dtype != "f32":
        # the same workload on the exact-f32 engine (v_mfma_f32_32x32x2_f32: the reference's own arithmetic type), timed over
        # the same kind of region, with its own kernel roofline against the 157.3 TFLOP/s f32 matrix peak
        job.stepper = None
        m32 = HipScoreModel(job.cfg, dtype="f32", device=str(dev))
        m32.load_state_dict(job.sd)
        m32.set_context(job.ctx)
        st32 = job.new_stepper(m32, rank)
        if job.cond:
            st32.set_condition(*job.cond)
        st32.reset(0)
        st32.step(job.x, job.x_mean)                          # warm-up (fills the activation pool)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.f32_steps):
            st32.step(job.x, job.x_mean)
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t0) / args.f32_steps
        v32 = B / (N * d32)
        out["f32"] = {"value": v32, "unit": "samples/s", "ms_per_step": d32 * 1e3, "steps": args.f32_steps, "warmup": 1, "dtype": "f32",
                      "mfma_frac_end_to_end": v32 * 2 * N * alg / (MFMA_PEAK_TFLOPS["f32"] * 1e12), "peak": MFMA_PEAK_TFLOPS["f32"],
                      "finite": bool(torch.isfinite(job.x_mean).all().item()),
                      "dispatches_per_step": (st32.reset(0), st32.count_dispatches(job.x, job.x_mean))[1]}
        if not args.no_roofline:
            out["f32"]["roofline"] = kernel_roofline(job, st32, lib, MFMA_PEAK_TFLOPS["f32"])
        del st32, m32
    if rank == 0 and world == 1 and not args.no_train and not args.no_f32:      # (the measurement tools pass --no-f32: no extras)
        out["train_step"] = train_step_line(dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(job.cfg, job.sd, job.ctx_cpu, N)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
