"""Checkpoint side of the sampling driver: reference ``score_sde_pytorch/utils.py:11-26`` and
``score_sde_pytorch/models/ema.py:51-93``: the load side of the sampling driver, and (round 4) ``save_checkpoint`` /
``restore_training_state`` for the training state of ``text2protein_amd.losses`` in the SAME file layout, so that a run trained
here can be sampled -- or continued -- by the reference and the other way round.

A reference checkpoint is ``torch.save({'optimizer', 'model', 'ema', 'step'})`` where ``model`` is a
DataParallel state dict (every key prefixed ``module.``, plus the float64 buffer
``module.sigmas``) and ``ema`` is ``{'decay', 'num_updates', 'shadow_params'}`` with
``shadow_params`` a list of tensors in ``model.parameters()`` order.  The driver restores the
model and then overwrites the live parameters with the EMA ones (sampling_6d.py:71-73); the HIP
engine is immutable once finalized, so ``restore_checkpoint`` loads the EMA weights directly.
"""
from __future__ import annotations

import torch

from .arch import param_specs
from ._lib import T2PError


def strip_module_prefix(state_dict):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def ema_state_dict(config, loaded_state):
    """What ``ema.copy_to(model.parameters())`` leaves in the model, as a name -> tensor dict."""
    specs = param_specs(config)
    shadow = loaded_state["ema"]["shadow_params"]
    if len(shadow) != len(specs):
        raise T2PError(f"checkpoint EMA has {len(shadow)} tensors, the model for this config has {len(specs)}")
    out = {}
    for s, p in zip(specs, shadow):
        if tuple(p.shape) != tuple(s.shape):
            raise T2PError(f"EMA tensor for {s.name} has shape {tuple(p.shape)}, expected {tuple(s.shape)}")
        out[s.name] = p
    return out


def restore_checkpoint(ckpt_path, model, config, device="cpu", use_ema=True):
    """Load a reference ``.pth`` into a ``HipScoreModel``; returns the checkpoint's ``step``."""
    # a reference checkpoint holds tensors, lists, dicts and numbers only: no arbitrary unpickling of user-supplied files
    loaded = torch.load(ckpt_path, map_location=device, weights_only=True)
    if use_ema and "ema" in loaded and loaded["ema"].get("shadow_params"):
        sd = ema_state_dict(config, loaded)
    else:
        sd = strip_module_prefix(loaded["model"])
    model.load_state_dict(sd)
    return loaded.get("step", 0)


def save_synthetic_checkpoint(path, config, seed=0):
    """Write a checkpoint in the reference's layout from the synthetic weights (tests, demos)."""
    from . import synth
    sd = synth.synth_state_dict(config, seed)
    specs = param_specs(config)
    model_sd = {"module." + k: v for k, v in sd.items()}
    model_sd["module.sigmas"] = torch.tensor(__import__("text2protein_amd.model", fromlist=["x"]).get_sigmas(config))
    state = {"optimizer": {}, "model": model_sd,
             "ema": {"decay": config.model.ema_rate, "num_updates": 0, "shadow_params": [sd[s.name] for s in specs]},
             "step": 0}
    torch.save(state, path)
    return path


def save_checkpoint(ckpt_dir, state):
    """score_sde_pytorch/utils.py:19-26 for a training ``state`` of text2protein_amd.losses (``optimizer`` = AdamView, ``model`` =
    HipTrainModel, ``ema`` = ExponentialMovingAverage view, ``step``): the reference's file layout, DataParallel key prefix and the
    float64 ``sigmas`` buffer included, tensors on the CPU."""
    from .model import get_sigmas
    model = state["model"]
    model_sd = {"module.sigmas": torch.tensor(get_sigmas(model.config))}
    model_sd.update({"module." + k: v for k, v in model.state_dict().items()})
    torch.save({"optimizer": state["optimizer"].state_dict(), "model": model_sd, "ema": state["ema"].state_dict(),
                "step": int(state["step"])}, ckpt_dir)


def restore_training_state(ckpt_dir, state, device="cpu"):
    """score_sde_pytorch/utils.py:11-17: optimizer, model, EMA and step of a reference-layout checkpoint into ``state``."""
    loaded = torch.load(ckpt_dir, map_location=device, weights_only=True)
    state["model"].load_state_dict(strip_module_prefix(loaded["model"]), strict=False)
    state["ema"].load_state_dict(loaded["ema"])
    state["optimizer"].load_state_dict(loaded["optimizer"])
    state["step"] = int(loaded["step"])
    state["model"].set_step(state["step"])
    return state
