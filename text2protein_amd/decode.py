"""6D decode of finished samples on the device -- the first stage of the reference's folding script.

Mirrors ``sampling_rosetta.py:59-96``: ``msk = round(coords[-1])``; ``L = sqrt(#(msk == 1))`` must be
an integer (``ValueError`` otherwise); the four geometry channels are masked, reshaped to ``(L, L)``,
clipped to [-1, 1] and scaled back (``dist_abs = (dist + 1) * 10``, ``omega_abs = omega * pi``,
``theta_abs = theta * pi``, ``phi_abs = (phi + 1) * pi / 2``).  One kernel launch decodes a whole
batch straight from the sampler's output tensor (``t2p_op_decode_6d``); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import T2PError, check, ptr, stream_ptr

NAMES = ("dist", "omega", "theta", "phi")


def decode_6d_batch(samples: torch.Tensor):
    """``samples``: ``(B, C, L, L)`` float32 (C >= 5, mask last).  Returns ``(lengths, clipped, absval)``:
    ``lengths`` int32 ``(B,)`` on the host (-1 = improper mask), the other two ``(B, 4, L*L)`` device
    tensors whose first ``lengths[b]**2`` entries per channel are the row-major ``(L_b, L_b)`` maps."""
    if samples.dim() != 4 or samples.shape[2] != samples.shape[3]:
        raise ValueError("samples must be (B, C, L, L)")
    if samples.device.type != "cuda":
        raise T2PError("decode_6d needs the samples on a GPU device (there is no CPU fallback)")
    lib = _lib.load()
    x = samples.to(torch.float32).contiguous()
    B, Cc, L, _ = x.shape
    clipped = torch.empty(B, 4, L * L, device=x.device, dtype=torch.float32)
    absval = torch.empty_like(clipped)
    lengths = torch.empty(B, device=x.device, dtype=torch.int32)
    with torch.cuda.device(x.device):
        check(lib.t2p_op_decode_6d(ptr(x), B, Cc, L, ptr(clipped), ptr(absval), ptr(lengths), stream_ptr()))
    return lengths.cpu(), clipped, absval


def decode_6d(samples: torch.Tensor):
    """Per sample the reference's ``npz`` dict (numpy ``(L, L)`` float32 arrays under ``dist``, ``omega``,
    ``theta``, ``phi`` and ``*_abs``, plus ``L``).  An improper mask raises ``ValueError`` with the
    reference's message (sampling_rosetta.py:72-73)."""
    if samples.dim() == 3:
        samples = samples.unsqueeze(0)
    lengths, clipped, absval = decode_6d_batch(samples)
    out = []
    for b, Lb in enumerate(lengths.tolist()):
        if Lb < 0:
            raise ValueError("Terminated due to improper masking channel...")
        d = {"L": Lb}
        cl, ab = clipped[b, :, :Lb * Lb].cpu().numpy(), absval[b, :, :Lb * Lb].cpu().numpy()
        for i, nm in enumerate(NAMES):
            d[nm] = cl[i].reshape(Lb, Lb)
            d[nm + "_abs"] = ab[i].reshape(Lb, Lb)
        out.append(d)
    return out
