#!/usr/bin/env python3
"""6D backbone sampling on MI355X: the reference's ``sampling_6d.py`` command line on the HIP engine.

    python sampling_6d.py <config.yml> <checkpoint.pth> [--batch_size 32] [--tag test] [--device cuda]
                          [--select_length True --length_index 61] [--mask_info 1:5,10:15] [--n_iter 1]

Same positional arguments and flags as the reference (sampling_6d.py:41-53) and the same output
contract (one pickle per sample, tensor (1, C, L, L), under
sampling/coords_6d/<config stem>/<run>/<tag>/sampled_<id>.pkl, sampling_6d.py:160-162).
Differences, all at the edges of the hot path:
  * text context: the reference embeds captions with a LLaMA embedding table fetched by name
    (sampling_6d.py:121-137).  Here the same producer runs from local files: ``--captions <file>`` (one
    caption per line, optionally ``id<TAB>caption``) with ``--tokenizer_path <dir>`` and ``--embed_table
    <file or HF checkpoint dir>`` (only ``embed_tokens`` is read; the lookup is a HIP gather).  Without
    captions the context comes from ``--context <file.pt>`` (a (B, T, context_dim) float tensor) or is
    synthetic (``--context synthetic``).
  * ``--decode`` additionally writes ``decoded_<id>.npz`` per sample: the reference's first folding stage
    (sampling_rosetta.py:69-96: mask rounding, crop, clip, inverse scaling) done on the device.
  * ``--pdb`` conditions need biotite and are broken in the reference: refused.  Their tensor half is
    reachable through ``--inpaint_coords <file.pt|synthetic>`` (the featurised chain the reference would have computed from
    the PDB file): the conditions named in ``config.model.condition`` (length / ss / inpainting with ``--mask_info``) are
    built from it exactly as ``get_condition_from_batch`` does (utils.py:84-106).  ``--mask_info`` alone is an error.
  * ``checkpoint`` may be the word ``synthetic`` (hash-generated weights, no file needed).
  * extra flags: --dtype (f32|f16|bf16), --seed, --ids, --num_scales / --max_res_num overrides.
Under ``python -m torch.distributed.run --nproc-per-node N`` (or with ``--gpus N``, which starts the N ranks
itself) every rank samples ``--batch_size`` chains on its own GPU and rank 0 gathers them with one RCCL
all_gather before writing (text2protein_amd/distributed.py).  ``--global_batch_norm`` reproduces the reference's
multi-GPU (DataParallel) Langevin step size: batch means over all ranks' chains, one 2-float all-reduce per step.
"""
import argparse
import os
import pickle as pkl
from pathlib import Path

import torch


def str2bool_like_reference(v):
    # the reference declares type=bool, so any non-empty string is True (sampling_6d.py:51)
    return bool(v)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("config", type=str)
    parser.add_argument("checkpoint", type=str)
    parser.add_argument("--pdb", type=str, default=None)
    parser.add_argument("--chain", type=str, default="A")
    parser.add_argument("--mask_info", type=str, default=None,
                        help="inpainting residue ranges (reference default '1:5,10:15'); needs --inpaint_coords")
    parser.add_argument("--tag", type=str, default="test")
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--batch_size", type=int, default=32)
    parser.add_argument("--n_iter", type=int, default=1)
    parser.add_argument("--select_length", type=str2bool_like_reference, default=False)
    parser.add_argument("--length_index", type=int, default=1)  # Index starts at 1
    # additions
    parser.add_argument("--dtype", type=str, default="f16", choices=["f32", "f16", "bf16"])
    parser.add_argument("--context", type=str, default="synthetic")
    parser.add_argument("--context_tokens", type=int, default=512)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--ids", type=str, default=None, help="comma separated sample ids (default 0..B-1)")
    parser.add_argument("--num_scales", type=int, default=None)
    parser.add_argument("--max_res_num", type=int, default=None)
    parser.add_argument("--outdir", type=str, default=None)
    parser.add_argument("--captions", type=str, default=None, help="text file: one caption (or id<TAB>caption) per line")
    parser.add_argument("--tokenizer_path", type=str, default=None)
    parser.add_argument("--embed_table", type=str, default=None)
    parser.add_argument("--inpaint_coords", type=str, default=None,
                        help="source of the known 6D maps for the conditions of config.model.condition (what the reference "
                             "builds from --pdb, utils.py:84-137): a torch file holding {'coords_6d': (B or 1, C, L, L) in "
                             "[-1, 1], 'lengths': ints or 'aa_str': strings}, or the word 'synthetic' (U(-1, 1) maps, 100 residues)")
    parser.add_argument("--decode", action="store_true", help="also write decoded_<id>.npz (sampling_rosetta.py:69-96)")
    parser.add_argument("--gpus", type=int, default=1, help="start this many ranks (one per GPU) when not under torch.distributed.run")
    parser.add_argument("--global_batch_norm", action="store_true",
                        help="Langevin batch means over every rank's chains (the reference's DataParallel semantics)")
    args = parser.parse_args()

    assert not (args.pdb is not None and args.select_length)
    if args.pdb is not None:
        raise SystemExit("--pdb conditions are outside the sampling hot path (need biotite; see SURVEY.md section 2, row 10); "
                         "pass the featurised maps with --inpaint_coords instead")
    if args.mask_info is not None and args.inpaint_coords is None:
        raise SystemExit("--mask_info selects residues of KNOWN 6D maps: give their source with --inpaint_coords <file.pt|synthetic>")
    assert not (args.inpaint_coords is not None and args.select_length)

    from text2protein_amd import distributed as D
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:      # before anything touches the GPU
        import sys
        raise SystemExit(D.launch_local(args.gpus, [os.path.abspath(__file__), *sys.argv[1:]]))
    from text2protein_amd import sampling, sde_lib, synth
    from text2protein_amd.checkpoint import restore_checkpoint
    from text2protein_amd.conditions import get_condition_from_batch, get_mask_all_lengths
    from text2protein_amd.config import load_config
    from text2protein_amd.model import HipScoreModel

    rank, world, local_rank = D.env_rank_world()
    overrides = {}
    if args.num_scales:
        overrides["model.num_scales"] = args.num_scales
    if args.max_res_num:
        overrides["data.max_res_num"] = args.max_res_num
    config = load_config(args.config, **overrides)
    device = f"cuda:{local_rank}" if args.device == "cuda" else args.device
    config.device = device
    torch.cuda.set_device(torch.device(device))
    dist = D.init_process_group(device)

    run = Path(args.checkpoint).parent.parent.stem if args.checkpoint != "synthetic" else "synthetic"
    workdir = Path(args.outdir) if args.outdir else Path("sampling", "coords_6d", Path(args.config).stem, run, args.tag)
    if rank == 0:
        workdir.mkdir(parents=True, exist_ok=True)

    # Initialize model (get_model + restore_checkpoint + ema.copy_to, sampling_6d.py:64-73)
    score_model = HipScoreModel(config, dtype=args.dtype, device=device)
    if args.checkpoint == "synthetic":
        score_model.load_state_dict(synth.synth_state_dict(config, args.seed))
    else:
        restore_checkpoint(args.checkpoint, score_model, config)

    # Load SDE (sampling_6d.py:76-82)
    if config.training.sde == "vesde":
        sde = sde_lib.VESDE(sigma_min=config.model.sigma_min, sigma_max=config.model.sigma_max, N=config.model.num_scales)
        sampling_eps = 1e-5
    elif config.training.sde == "vpsde":
        sde = sde_lib.VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
        sampling_eps = 1e-3
    else:
        raise SystemExit(f"unknown training.sde {config.training.sde}")

    B = args.batch_size
    sampling_shape = (B, config.data.num_channels, config.data.max_res_num, config.data.max_res_num)
    kw = {}
    if args.global_batch_norm and dist is not None:
        kw = {"global_batch": B * world, "all_reduce": lambda sums: D.allreduce_norm_sums(sums, dist)}
    # every rank its own noise stream; every iteration of the --n_iter loop a fresh one (call index, sampling.py)
    sampling_fn = sampling.get_sampling_fn(config, sde, sampling_shape, sampling_eps, seed=D.rank_seed(args.seed, rank), **kw)

    caption_ids = None
    if args.captions:
        if not (args.tokenizer_path and args.embed_table):
            raise SystemExit("--captions needs --tokenizer_path and --embed_table (local paths)")
        from text2protein_amd.text_context import TextContextProducer
        rows = [ln.rstrip("\n") for ln in open(args.captions) if ln.strip()]
        rows = rows[rank * B:(rank + 1) * B]
        if len(rows) != B:
            raise SystemExit(f"--captions must hold batch_size x world_size = {B * world} captions")
        if all("\t" in r for r in rows):
            caption_ids = [r.split("\t", 1)[0] for r in rows]
            rows = [r.split("\t", 1)[1] for r in rows]
        producer = TextContextProducer.from_local(args.tokenizer_path, args.embed_table, device=device)
        context = producer(rows)                           # sampling_6d.py:134-137
        if context.shape[-1] != config.model.context_dim:
            raise SystemExit(f"embedding width {context.shape[-1]} != model.context_dim {config.model.context_dim}")
        del producer
    elif args.context == "synthetic":
        context = synth.synth_context(B, args.context_tokens, config.model.context_dim, seed=D.rank_seed(args.seed, rank))
    else:
        context = torch.load(args.context, map_location="cpu")
        if context.shape[0] != B:
            raise SystemExit(f"context batch {context.shape[0]} != --batch_size {B}")
    ids = args.ids.split(",") if args.ids else (caption_ids or [str(rank * B + i) for i in range(B)])
    if len(ids) != B:
        raise SystemExit("--ids must list batch_size ids")

    known = None
    if args.inpaint_coords is not None:
        # the tensor half of get_conditions_from_pdb (utils.py:119-137): one featurised chain repeated over the batch
        C_, L_ = config.data.num_channels, config.data.max_res_num
        if args.inpaint_coords == "synthetic":
            coords = torch.from_numpy(synth.uniform_pm1(args.seed, "coords_6d", C_ * L_ * L_).reshape(1, C_, L_, L_))
            known = {"coords_6d": coords, "lengths": [min(100, L_)]}
        else:
            known = torch.load(args.inpaint_coords, map_location="cpu", weights_only=True)
            if "coords_6d" not in known or ("lengths" not in known and "aa_str" not in known):
                raise SystemExit("--inpaint_coords file must hold 'coords_6d' and 'lengths' (or 'aa_str')")
        coords = known["coords_6d"].float()
        if tuple(coords.shape[1:]) != (C_, L_, L_) or coords.shape[0] not in (1, B):
            raise SystemExit(f"coords_6d must be (1 or {B}, {C_}, {L_}, {L_}), got {tuple(coords.shape)}")
        rep = B // coords.shape[0]
        known = {"coords_6d": coords.repeat(rep, 1, 1, 1),
                 **({"lengths": list(known["lengths"]) * rep} if "lengths" in known else {"aa_str": list(known["aa_str"]) * rep})}

    def to_device(c):
        return {k: to_device(v) if isinstance(v, dict) else v.to(device) for k, v in c.items()}

    for it in range(args.n_iter):
        if args.select_length:
            mask = get_mask_all_lengths(config, batch_size=B)[args.length_index - 1]
            condition = {"length": mask.to(device)}
        elif known is not None:
            # sampling_6d.py:146-147 -> get_condition_from_batch (utils.py:84-106): every condition the model was trained with
            condition = to_device(get_condition_from_batch(config, known, mask_info=args.mask_info or "1:5,10:15"))
        else:
            condition = {}
        sample, n = sampling_fn(score_model, condition=condition, context=context)
        if dist is not None:
            sample = D.gather_samples(sample, dist)          # the single data-path collective of a run
            all_ids = [None] * world
            dist.all_gather_object(all_ids, ids)
            out_ids = [i for sub in all_ids for i in sub]
        else:
            out_ids = ids
        generated = sample.cpu()
        if rank == 0:
            print("show generated samples shape: ", generated.shape)
            for i, sid in enumerate(out_ids):
                suffix = f"_{it}" if args.n_iter > 1 else ""
                with open(workdir.joinpath(f"sampled_{sid}{suffix}.pkl"), "wb") as f:
                    pkl.dump(generated[i].unsqueeze(0), f)
            if args.decode:
                import numpy as np
                from text2protein_amd.decode import NAMES, decode_6d_batch
                lengths, clipped, absval = decode_6d_batch(sample)
                for i, sid in enumerate(out_ids):
                    Lb = int(lengths[i])
                    if Lb < 0:          # sampling_rosetta.py:72-73 raises for such a sample; here it is reported and skipped
                        print(f"sample {sid}: improper masking channel, not decoded")
                        continue
                    suffix = f"_{it}" if args.n_iter > 1 else ""
                    arrs = {}
                    for c, nm in enumerate(NAMES):
                        arrs[nm] = clipped[i, c, :Lb * Lb].reshape(Lb, Lb).cpu().numpy()
                        arrs[nm + "_abs"] = absval[i, c, :Lb * Lb].reshape(Lb, Lb).cpu().numpy()
                    np.savez(workdir.joinpath(f"decoded_{sid}{suffix}.npz"), **arrs)
            print(f"[{it + 1} / {args.n_iter}] save samples to {workdir} ({n} score evaluations per chain)")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
