"""Condition builders of the sampling path: the pure-tensor parts of reference ``utils.py``.

  get_mask_all_lengths     utils.py:139-148   (n_lengths, B, L, L) bool; the driver indexes it with
                                              ``length_index - 1`` (sampling_6d.py:145)
  selected_mask_batch      utils.py:62-81     "1:5,10:15" -> (B, L, L) bool, inclusive 0-based ranges
The PDB-driven builders (utils.py:108-137) need biotite and call ``ProteinDataset`` with a
signature it does not have; they are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch


def get_mask_all_lengths(config, batch_size=16):
    all_lengths = np.arange(config.data.min_res_num, config.data.max_res_num + 1)
    mask = torch.zeros(len(all_lengths), batch_size, config.data.max_res_num, config.data.max_res_num).bool()
    for idx, l in enumerate(all_lengths):
        mask[idx, :, :l, :l] = True
    return mask


def parse_mask_info(mask_info: str, batch: int, n: int) -> torch.Tensor:
    """(B, N) residue mask from "a:b,c" (ranges inclusive, 0-based), utils.py:69-76."""
    mask = torch.zeros(batch, n)
    for r in mask_info.split(","):
        if ":" in r:
            start_idx, end_idx = r.split(":")
            mask[:, int(start_idx):int(end_idx) + 1] = 1
        else:
            mask[:, int(r)] = 1
    return mask


def selected_mask_batch(batch, mask_info, config):
    """utils.py:62-81 on a ``{"coords_6d": (B,C,N,N)}`` batch; adds ``mask_inpaint`` (B,N,N) bool."""
    if "inpainting" not in config.model.condition:
        batch["mask_inpaint"] = None
        return batch
    B, _, N, _ = batch["coords_6d"].shape
    mask = parse_mask_info(mask_info, B, N)
    mask = torch.logical_or(mask.unsqueeze(-1), mask.unsqueeze(1))
    batch["mask_inpaint"] = mask.to(dtype=torch.bool)
    return batch


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
        m = parse_mask_info(mask_info, batch, L)
        cond["inpainting"] = {"coords_6d": coords.to(device),
                              "mask_inpaint": torch.logical_or(m.unsqueeze(-1), m.unsqueeze(1)).bool().to(device)}
    return cond
