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


# ---- training-step fixtures (tests/golden/make_golden_train.py writes them) ------------------------------------------
def cfg_train_tiny():
    """The tiny configuration in training trim: `length` condition, no dropout (the gradient check proper)."""
    return tiny_config(**{"model.condition": ["length"], "model.dropout": 0.0, "model.num_scales": 50})


def cfg_train_tinyB():
    """Up / down blocks, two attention levels, C = 8, all three conditions, Dropout_0 active with counter-based keep-masks."""
    return tiny_config(**{"model.ch_mult": [1, 1, 2], "model.num_res_blocks": 2, "data.num_channels": 8,
                          "model.attn_resolutions": [4, 8], "model.n_heads": 2, "model.context_dim": 24, "model.nf": 32,
                          "model.condition": ["length", "ss", "inpainting"], "model.dropout": 0.1, "model.num_scales": 50})


def cfg_train_cond_length():
    """BASELINE configs[2]'s model at its real size: configs/cond_length.yml, L = 128 (75.0 M parameters), dropout as shipped (0.1)."""
    from text2protein_amd.config import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return load_config(os.path.join(root, "configs", "cond_length.yml"), **{"data.max_res_num": 128, "model.num_scales": 1000})


def cfg_train_test_config():
    """BASELINE configs[1]'s model at its real WIDTH (configs/test_config.yml: nf 256, channel multipliers up to 2, 8 heads, AttnBlockpp and
    SpatialTransformer at three resolutions, 379.5 M parameters) on L = 64 maps (six levels down to 2 x 2), dropout as shipped (0.1), no condition."""
    from text2protein_amd.config import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return load_config(os.path.join(root, "configs", "test_config.yml"), **{"data.max_res_num": 64, "model.num_scales": 1000})


TRAIN_CASES = {   # name -> what the fixture script and the tests share; step0 = state['step'] before the update (warm-up factor step0 / 5000)
    "train_tiny": dict(config=cfg_train_tiny, seed=3, B=2, T=3, lengths=[12, 9], step0=2000, mask_info=None),
    "train_tinyB": dict(config=cfg_train_tinyB, seed=4, B=3, T=5, lengths=[16, 11, 6], step0=7000, mask_info="1:3,6:8"),
    # full size (round 4): norms + projections of all 622 tensors, whole tensors for the small ones only (the file stays ~1 MB)
    "train_cond_length": dict(config=cfg_train_cond_length, seed=5, B=1, T=16, lengths=[100], step0=9000, mask_info=None, full_size=True),
    # the other architecture family at its real width (C = 256 / 512, attention at three resolutions)
    "train_test_config": dict(config=cfg_train_test_config, seed=6, B=1, T=8, lengths=[50], step0=3000, mask_info=None, full_size=True,
                              yaml="test_config.yml"),
}


def train_inputs(cfg, case):
    """coords_6d ~ U(-1, 1) with the padding channel = the pair mask, pair masks of the given lengths, t ~ U(eps, 1), z ~ N(0, 1),
    text context: all from the counter-hash generator, regenerated identically on both sides."""
    from text2protein_amd import synth
    B, T, seed = case["B"], case["T"], case["seed"]
    C, L = cfg.data.num_channels, cfg.data.max_res_num
    x = torch.from_numpy(synth.uniform_pm1(seed, "train_coords", B * C * L * L).reshape(B, C, L, L))
    mp = torch.zeros(B, L, L).bool()
    for b, n in enumerate(case["lengths"]):
        mp[b, :n, :n] = True
    x = x * mp.unsqueeze(1)
    x[:, -1] = mp.float()
    u = torch.from_numpy((synth.uniform_pm1(seed, "train_t", B).astype(np.float64) + 1.0) / 2.0).float()
    t = u * (1.0 - 1e-5) + 1e-5                               # losses.py:106 with sde.T = 1, eps = 1e-5
    z = torch.from_numpy(synth.normal(seed, "train_z", B * C * L * L).reshape(B, C, L, L))
    out = dict(coords_6d=x, mask_pair=mp, t=t, z=z, context=synth.synth_context(B, T, cfg.model.context_dim, seed + 31))
    if case.get("mask_info"):
        m = torch.zeros(B, L)
        for r in case["mask_info"].split(","):
            a, b = r.split(":")
            m[:, int(a):int(b) + 1] = 1
        out["mask_inpaint"] = torch.logical_or(m.unsqueeze(-1), m.unsqueeze(1)).bool()
    return out


class CounterDropout:
    """Dropout_0 keep-masks from the counter-hash generator: call k keeps element i iff uniform_pm1(seed, "drop<k>", n)[i] >= 2p - 1
    (row-major over the NCHW tensor).  ``functional`` is patched over torch.nn.functional.dropout when the fixture is made
    (calls with p == 0 or in eval mode pass through uncounted: FeedForward / CrossAttention hold Dropout(0.)); ``module`` is the
    oracle's hook; ``mask(k, shape)`` gives the test the same mask to upload."""

    def __init__(self, seed, p):
        self.seed, self.p, self.k = int(seed), float(p), 0

    def mask(self, k, shape):
        from text2protein_amd import synth
        n = int(np.prod(shape))
        u = torch.from_numpy(synth.uniform_pm1(self.seed, f"drop{k}", n).reshape(tuple(shape)))
        return u >= (2.0 * self.p - 1.0)

    def module(self, h):
        keep = self.mask(self.k, h.shape)
        self.k += 1
        return h * keep / (1.0 - self.p)

    def functional(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        assert abs(p - self.p) < 1e-12
        return self.module(input)


def projection(name, tensor, seed=99, cache=None):
    """<tensor, r> with r ~ U(-1, 1) keyed by the tensor's name: pins every element of a tensor with one stored number.
    ``cache`` (a dict) keeps r between calls: the full-size fixture projects five buffers of 75 M elements on the same vectors."""
    from text2protein_amd import synth
    v = torch.as_tensor(tensor).detach().double().reshape(-1)
    r = cache.get(name) if cache is not None else None
    if r is None:
        r = torch.from_numpy(synth.uniform_pm1(seed, name + ":proj", v.numel()))
        if cache is not None:
            cache[name] = r
    return float((v * r.double()).sum())
