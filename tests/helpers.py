"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from text2protein_amd.config import tiny_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def cfg_tiny():
    return tiny_config()


def cfg_tinyB():
    return tiny_config(**{"model.ch_mult": [1, 1, 2], "model.num_res_blocks": 2, "data.num_channels": 8,
                          "model.attn_resolutions": [4, 8], "model.n_heads": 2, "model.context_dim": 24,
                          "model.nf": 32})


def rel_l2(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cfg_smallC():
    """Wide enough (64/128 channels, 32x32 maps) for the LDS-DMA GEMM kernel to be selected."""
    return tiny_config(**{"model.nf": 64, "model.ch_mult": [1, 2], "model.num_res_blocks": 1, "data.max_res_num": 32,
                          "model.attn_resolutions": [16], "model.n_heads": 4, "model.context_dim": 64,
                          "model.num_scales": 10})
