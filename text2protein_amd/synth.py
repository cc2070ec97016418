"""Deterministic synthetic weights, text embeddings and noise for benchmarks and parity tests.

The reference's construction-time init is degenerate for sampling: 75 tensors are zero or
1e-10-scaled (layers.py:73-80 ``default_init(scale=0)``, attention.py:66-72 ``zero_module``), so
the network output is ~1e-6 and the Langevin step size explodes.  The
benchmark therefore uses this generator: every tensor -- including the ones the reference
zero-initialises -- gets fan-in-scaled uniform values from a counter-based hash keyed by
``(seed, tensor name, element index)``.  It does not touch torch's RNG stream, so the oracle,
the reference (through ``load_state_dict``) and the HIP engine all see identical weights.
"""
from __future__ import annotations

import hashlib

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + _GOLD).astype(np.uint64)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def name_key(seed: int, name: str) -> np.uint64:
    h = hashlib.blake2b(f"{seed}:{name}".encode(), digest_size=8).digest()
    return np.uint64(int.from_bytes(h, "little"))


def uniform_pm1(seed: int, name: str, n: int, chunk: int = 1 << 24) -> np.ndarray:
    """n float32 values uniform in [-1, 1), element i = f(seed, name, i)."""
    key = name_key(seed, name)
    out = np.empty(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            idx = np.arange(s, e, dtype=np.uint64)
            bits = _splitmix64(idx * _GOLD + key) >> np.uint64(40)          # 24 random bits
            out[s:e] = bits.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return out


def normal(seed: int, name: str, n: int) -> np.ndarray:
    """n float32 standard-normal values (Box-Muller on two hashed uniforms)."""
    key = name_key(seed, name)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h1 = _splitmix64(idx * _GOLD + key)
        h2 = _splitmix64(h1 ^ _M2)
    u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = (h2 >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def synth_tensor(spec, seed: int) -> np.ndarray:
    """Value of one parameter (see arch.ParamSpec.kind)."""
    n = int(np.prod(spec.shape))
    u = uniform_pm1(seed, spec.name, n)
    if spec.kind == "weight":
        # variance 1/fan_in, uniform: keeps activations O(1) through every stage
        u *= np.float32(np.sqrt(3.0 / max(spec.fan_in, 1)))
    elif spec.kind == "bias":
        u *= np.float32(0.05)
    elif spec.kind == "norm_scale":
        u = np.float32(1.0) + np.float32(0.1) * u
    elif spec.kind == "norm_shift":
        u *= np.float32(0.05)
    else:
        raise ValueError(spec.kind)
    return u.reshape(spec.shape)


def synth_state_dict(config, seed: int = 0, as_torch: bool = True):
    """name -> tensor for every learnable tensor of the score network (arch.param_specs order)."""
    from .arch import param_specs
    out = {}
    for spec in param_specs(config):
        t = synth_tensor(spec, seed)
        if as_torch:
            import torch
            t = torch.from_numpy(t)
        out[spec.name] = t
    return out


def synth_context(batch: int, tokens: int, dim: int, seed: int = 0, as_torch: bool = True):
    """Stand-in for ``llm.model.embed_tokens(tokens)`` (reference sampling_6d.py:134-137):
    ``(B, T, context_dim)`` float32 ~ N(0, 1)."""
    t = normal(seed, "context", batch * tokens * dim).reshape(batch, tokens, dim)
    if as_torch:
        import torch
        t = torch.from_numpy(t)
    return t
