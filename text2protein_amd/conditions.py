"""Condition builders of the sampling path: the pure-tensor parts of reference ``utils.py``.

  get_mask_all_lengths     utils.py:139-148   (n_lengths, B, L, L) bool; the driver indexes it with
                                              ``length_index - 1`` (sampling_6d.py:145)
  selected_mask_batch      utils.py:62-81     "1:5,10:15" -> (B, L, L) bool, inclusive 0-based ranges
  get_condition_from_batch utils.py:84-106    {"length" | "ss" | "inpainting"} from a batch of 6D maps
The PDB-driven builders (utils.py:108-137) need biotite and call ``ProteinDataset`` with a
signature it does not have; they are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch


def get_mask_all_lengths(config, batch_size=16):
    """(n_lengths, B, L, L) bool: entry k masks the top-left square of ``min_res_num + k`` residues (utils.py:139-148)."""
    L = config.data.max_res_num
    lengths = torch.arange(config.data.min_res_num, L + 1)
    inside = torch.arange(L)[None, :] < lengths[:, None]                      # (n_lengths, L)
    square = inside[:, :, None] & inside[:, None, :]                          # (n_lengths, L, L)
    return square[:, None].expand(-1, batch_size, -1, -1).clone()


def parse_mask_info(mask_info: str, batch: int, n: int) -> torch.Tensor:
    """(B, N) 0/1 residue mask from ``"a:b,c"``: comma separated 0-based entries, ``a:b`` inclusive (utils.py:69-76).
    Indices follow tensor slicing semantics (negative values count from the end), as in the reference."""
    row = torch.zeros(n)
    for item in mask_info.split(","):
        lo, _, hi = item.partition(":")
        if hi:
            row[int(lo):int(hi) + 1] = 1
        else:
            row[int(lo)] = 1
    return row.expand(batch, n).clone()


def pair_mask(residue_mask: torch.Tensor) -> torch.Tensor:
    """(B, N) -> (B, N, N) bool: a pixel (i, j) is selected when residue i OR residue j is (utils.py:78)."""
    m = residue_mask.bool()
    return m[:, :, None] | m[:, None, :]


def selected_mask_batch(batch, mask_info, config):
    """utils.py:62-81 on a ``{"coords_6d": (B,C,N,N)}`` batch; adds ``mask_inpaint`` (B,N,N) bool."""
    if "inpainting" not in config.model.condition:
        batch["mask_inpaint"] = None
        return batch
    B, _, N, _ = batch["coords_6d"].shape
    batch["mask_inpaint"] = pair_mask(parse_mask_info(mask_info, B, N))
    return batch


def get_condition_from_batch(config, batch, mask_info=None):
    """utils.py:84-106 for a tensor batch: one entry per name in ``config.model.condition``.

    ``batch`` holds ``coords_6d`` (B, C, L, L) and the residue counts either as ``lengths`` (B ints) or as the
    reference's ``aa_str`` (padded with "_").  ``inpainting`` needs ``mask_info`` (the reference's alternative, a random
    mask drawn by ``random_mask_batch``, belongs to training)."""
    coords = batch["coords_6d"]
    B, L = coords.shape[0], config.data.max_res_num
    out = {}
    for name in config.model.condition:
        if name == "length":
            if "lengths" in batch:
                lengths = torch.as_tensor(batch["lengths"]).long()
            else:
                lengths = torch.tensor([sum(ch != "_" for ch in s) for s in batch["aa_str"]])
            inside = torch.arange(L)[None, :] < lengths[:, None]
            out[name] = inside[:, :, None] & inside[:, None, :]
        elif name == "ss":
            out[name] = coords[:, 4:7]
        elif name == "inpainting":
            if mask_info is None:
                raise ValueError("the inpainting condition needs mask_info (residue ranges such as '1:5,10:15')")
            out[name] = {"coords_6d": coords, "mask_inpaint": pair_mask(parse_mask_info(mask_info, B, L))}
        else:
            raise ValueError(f"unknown condition {name!r}")
    return out


def synthetic_condition(config, batch, kind, device, length=100, mask_info="1:5,10:15", seed=0):
    """Benchmark conditions: a length mask for `length` residues and, for
    inpainting, the reference's default ``--mask_info`` with coords_6d ~ U(-1, 1)."""
    from . import synth
    L, C = config.data.max_res_num, config.data.num_channels
    cond = {}
    if "length" in kind:
        idx = length - config.data.min_res_num            # == length_index - 1 of the driver
        cond["length"] = get_mask_all_lengths(config, batch)[idx].to(device)
    if "inpainting" in kind:
        coords = torch.from_numpy(synth.uniform_pm1(seed, "coords_6d", batch * C * L * L).reshape(batch, C, L, L))
        cond["inpainting"] = {"coords_6d": coords.to(device),
                              "mask_inpaint": pair_mask(parse_mask_info(mask_info, batch, L)).to(device)}
    return cond
