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


# ---- BASELINE-size fixtures (tests/golden/make_golden_full.py writes them, the GPU tests read them) ----------------
# stem -> (YAML, L, N, batch of the fixture, text tokens, chains per GPU of the benchmark), BASELINE.md section 5
FULL = {
    "test_config": ("test_config.yml", 128, 1000, 2, 512, 32),
    "cond_length": ("cond_length.yml", 128, 1000, 2, 512, 32),
    "cond_length_inpainting": ("cond_length_inpainting.yml", 128, 1000, 2, 512, 16),
    "test_config_large": ("test_config_large.yml", 256, 1000, 1, 512, 16),
}
FULL_LABELS = [3, 700]     # time labels of the fixture's samples (sigma ~ 97 and ~ 0.16)


def full_inputs(cfg, B, T, seed=0):
    """x, labels and text context of the full-size fixtures: regenerated on both sides from the counter-hash
    generator, so only the score is stored."""
    from text2protein_amd import synth
    C, L = cfg.data.num_channels, cfg.data.max_res_num
    x = torch.from_numpy(synth.normal(seed, "full_x", B * C * L * L).reshape(B, C, L, L))
    labels = torch.tensor(FULL_LABELS[:B]).long()
    sig = torch.from_numpy(np.exp(np.linspace(np.log(cfg.model.sigma_max), np.log(cfg.model.sigma_min), cfg.model.num_scales)))
    x = x * sig[labels].float()[:, None, None, None]          # a state of the right magnitude for its noise level
    ctx = synth.synth_context(B, T, cfg.model.context_dim, seed + 17)
    return x, labels, ctx


class CounterNoise:
    """Stand-in for torch.randn / torch.randn_like: draw k = synth.normal(seed, "draw<k>", n).  Patched over the
    reference's draws when the 100-step fixture is made, injected through ``noise_fn`` on the HIP side."""

    def __init__(self, seed):
        self.seed, self.k = int(seed), 0

    def draw(self, shape):
        from text2protein_amd import synth
        n = int(np.prod(shape))
        z = torch.from_numpy(synth.normal(self.seed, f"draw{self.k}", n).reshape(tuple(shape)))
        self.k += 1
        return z

    def randn(self, *shape, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        return self.draw(shape)

    def randn_like(self, x, **kw):
        return self.draw(tuple(x.shape)).to(x.device)


def cfg_ss():
    return tiny_config(**{"data.num_channels": 8, "model.condition": ["length", "ss"]})


def cfg_ckpt():
    """Small enough for a committed .pth (about 1 MB with the EMA copy)."""
    return tiny_config(**{"model.nf": 8, "model.ch_mult": [4], "model.attn_resolutions": [], "data.max_res_num": 8,
                          "model.n_heads": 2, "model.context_dim": 16, "model.num_scales": 10})
