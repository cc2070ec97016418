"""YAML -> attribute-dict config, the surface `sampling_6d.py` reads.

The reference loads its YAML into an ``EasyDict`` (reference sampling_6d.py:57-60) and every
layer below reads ``config.model.nf``-style attributes.  ``easydict`` is not a dependency here;
``AttrDict`` gives the same attribute/indexing behaviour for the keys the sampling path reads
("Config / flags").
"""
from __future__ import annotations

import copy

import yaml


class AttrDict(dict):
    """dict with attribute access, recursive over nested dicts (EasyDict work-alike)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


# The reference's cond_length*.yml lack these two keys although UNetModel reads them
# unconditionally (ncsnpp.py:94-95); the values are those of configs/test_config.yml:54-55.
_MODEL_DEFAULTS = {"n_heads": 8, "context_dim": 4096}


def load_config(path, **overrides):
    """Read a YAML config; ``overrides`` use dotted keys, e.g. ``**{"data.max_res_num": 128}``."""
    with open(path, "r") as f:
        cfg = AttrDict(yaml.safe_load(f))
    return finalize_config(cfg, **overrides)


def finalize_config(cfg, **overrides):
    cfg = cfg if isinstance(cfg, AttrDict) else AttrDict(cfg)
    for k, v in _MODEL_DEFAULTS.items():
        cfg.model.setdefault(k, v)
    if cfg.model.get("condition") is None:  # reference no_cond.yml has an empty `condition:`
        cfg.model["condition"] = []
    for dotted, v in overrides.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def tiny_config(**overrides):
    """The small configuration used by golden fixtures and fast parity tests."""
    cfg = AttrDict(
        device="cpu",
        seed=0,
        training=dict(sde="vesde", batch_size=2),
        sampling=dict(method="pc", predictor="reverse_diffusion", corrector="langevin", snr=0.17,
                      n_steps_each=1, probability_flow=False, noise_removal=True),
        data=dict(num_channels=5, min_res_num=4, max_res_num=16),
        model=dict(name="ncsnpp", condition=[], sigma_min=0.01, sigma_max=100.0, num_scales=5,
                   beta_min=0.1, beta_max=20.0, nf=32, ch_mult=[1, 2], num_res_blocks=1,
                   attn_resolutions=[8], resblock_type="biggan", resamp_with_conv=True,
                   skip_rescale=True, scale_by_sigma=True, embedding_type="positional",
                   nonlinearity="swish", dropout=0.1, init_scale=0.0, ema_rate=0.999,
                   n_heads=4, context_dim=32),
        optim=dict(optimizer="Adam", lr=1e-4, beta1=0.9, eps=1e-8, weight_decay=0, warmup=5000,
                   grad_clip=1.0),
    )
    return finalize_config(cfg, **overrides)
