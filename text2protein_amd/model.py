"""HIP-backed score network: the drop-in for the reference ``UNetModel`` / ``get_model``.

Mirrors, for the sampling path only:
  * ``score_sde_pytorch/models/ncsnpp.py:71-263``  ``UNetModel(config)``; ``model(x, labels, context)``
  * ``score_sde_pytorch/utils.py:4-17``            ``get_model`` / ``restore_checkpoint``
  * ``score_sde_pytorch/models/ema.py:51-93``      EMA ``shadow_params`` -> parameters

All compute happens in libt2p_hip.so (hand-written HIP kernels); torch only owns the device
buffers and the stream.  No CPU path exists: constructing the model without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .arch import param_specs
from ._lib import ModelConfig, T2PError, check, ptr, stream_ptr


def _model_config(config, dtype) -> ModelConfig:
    m = config.model
    mc = ModelConfig()
    mc.num_channels = config.data.num_channels
    mc.max_res_num = config.data.max_res_num
    mc.nf = m.nf
    mc.num_res_blocks = m.num_res_blocks
    ch_mult = list(m.ch_mult)
    attn = list(m.attn_resolutions)
    if len(ch_mult) > 8 or len(attn) > 8:
        raise ValueError("ch_mult / attn_resolutions longer than 8")
    mc.n_ch_mult = len(ch_mult)
    for i, v in enumerate(ch_mult):
        mc.ch_mult[i] = int(v)
    mc.n_attn_resolutions = len(attn)
    for i, v in enumerate(attn):
        mc.attn_resolutions[i] = int(v)
    mc.n_heads = m.n_heads
    mc.context_dim = m.context_dim
    mc.num_scales = m.num_scales
    mc.sigma_min = float(m.sigma_min)
    mc.sigma_max = float(m.sigma_max)
    mc.skip_rescale = int(bool(m.skip_rescale))
    mc.scale_by_sigma = int(bool(m.scale_by_sigma))
    mc.compute_dtype = _lib.DTYPE_NAMES[dtype] if isinstance(dtype, str) else int(dtype)
    if m.resblock_type.lower() != "biggan" or m.embedding_type.lower() != "positional":
        raise ValueError("only resblock_type=biggan / embedding_type=positional are on the sampling path")
    if m.nonlinearity.lower() != "swish":
        raise ValueError("only nonlinearity=swish (SiLU) is implemented (every shipped config uses it)")
    return mc


class HipScoreModel:
    """``model(x, labels, context) -> score`` computed by the HIP engine.

    Differences from the reference module that a caller can observe: the result is float32 (the
    reference returns float64 because it divides by its float64 ``sigmas`` buffer,
    ncsnpp.py:259-261; the sampler casts back with ``.float()`` at sampling.py:283); a failing block
    raises instead of being skipped (the reference swallows exceptions at ncsnpp.py:235-243).
    """

    def __init__(self, config, dtype="f32", device="cuda:0"):
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise T2PError("HipScoreModel needs a GPU device (there is no CPU fallback)")
        self.lib = _lib.load()
        torch.cuda.set_device(self.device)
        self._mc = _model_config(config, dtype)
        self.dtype = dtype
        h = C.c_void_p()
        check(self.lib.t2p_engine_create(C.byref(self._mc), C.byref(h)))
        self._h = h
        self._finalized = False
        self._ctx_key = None
        self._specs = param_specs(config)
        # the engine derives the same table on its own; both must agree (tests check it too)
        n = self.lib.t2p_engine_num_params(self._h)
        if n != len(self._specs):
            raise T2PError(f"parameter table mismatch: engine {n} vs arch {len(self._specs)}")
        self.training = False

    # -- nn.Module-like surface used by the sampling path --------------------------------------
    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise T2PError("the HIP engine is inference-only")
        return self

    def to(self, *_a, **_k):
        return self

    def parameters(self):
        return iter(())

    def engine_param_table(self):
        out = []
        name = C.c_char_p()
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self.lib.t2p_engine_num_params(self._h)):
            check(self.lib.t2p_engine_param_info(self._h, i, C.byref(name), shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's state dict (with or without the DataParallel ``module.`` prefix)."""
        if self._finalized:
            raise T2PError("weights already loaded")
        seen = set()
        for k, v in state_dict.items():
            name = k[7:] if k.startswith("module.") else k
            if name == "sigmas":
                continue
            t = torch.as_tensor(v).detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            check(self.lib.t2p_engine_load_param(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            seen.add(name)
        missing = [s.name for s in self._specs if s.name not in seen]
        if missing:
            if strict:
                raise T2PError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
            return missing
        check(self.lib.t2p_engine_finalize(self._h))
        self._finalized = True
        return []

    def load_ema_shadow(self, shadow_params):
        """``ema.copy_to(model.parameters())`` (ema.py:59-70): a list in ``parameters()`` order."""
        if len(shadow_params) != len(self._specs):
            raise T2PError(f"EMA list has {len(shadow_params)} tensors, model has {len(self._specs)}")
        return self.load_state_dict(OrderedDict((s.name, p) for s, p in zip(self._specs, shadow_params)))

    def set_context(self, context):
        """Project the frozen text embedding through every to_k / to_v once (loop-invariant)."""
        context = context.to(self.device, torch.float32).contiguous()
        B, T, D = context.shape
        if D != self.config.model.context_dim:
            raise T2PError(f"context dim {D} != model.context_dim {self.config.model.context_dim}")
        check(self.lib.t2p_engine_set_context(self._h, ptr(context), B, T, stream_ptr()))
        self._ctx_key = (context.data_ptr(), context._version, tuple(context.shape))
        self._ctx_ref = context

    def __call__(self, x, labels, context=None):
        if not self._finalized:
            raise T2PError("load weights before calling the model")
        if context is not None:
            key = (context.data_ptr(), context._version, tuple(context.shape))
            if key != self._ctx_key or context.device != self.device or context.dtype != torch.float32:
                self.set_context(context)
                if context.device == self.device and context.dtype == torch.float32 and context.is_contiguous():
                    self._ctx_key = key
        elif self._ctx_key is None:
            raise T2PError("context is required (context_dim differs from the block width, SURVEY 8(a))")
        x = x.to(self.device, torch.float32).contiguous()
        labels = labels.to(self.device)
        # float labels (VP branch): embed the float, index sigmas with .long() like ncsnpp.py:223-224
        labels_f = labels.to(torch.float32).contiguous() if labels.is_floating_point() else None
        labels_i = labels.long().to(torch.int32).contiguous()
        out = torch.empty_like(x)
        check(self.lib.t2p_engine_score_ex(self._h, ptr(x), ptr(labels_i), ptr(labels_f), ptr(out), x.shape[0], stream_ptr()))
        return out

    forward = __call__

    def device_bytes(self):
        return int(self.lib.t2p_engine_device_bytes(self._h))

    def pool_reclaimed(self):
        """Activation buffers the last evaluation left checked out and the engine took back (0 after every complete evaluation)."""
        return int(self.lib.t2p_engine_pool_reclaimed(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.t2p_engine_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass


def get_model(config, dtype="f32"):
    """score_sde_pytorch/utils.py:4-9 -- no DataParallel wrapper: one process drives one GPU."""
    return HipScoreModel(config, dtype=dtype, device=config.device if str(config.device) != "cuda" else "cuda:0")


def get_sigmas(config):
    """models/utils.py:50-60."""
    m = config.model
    return np.exp(np.linspace(np.log(m.sigma_max), np.log(m.sigma_min), m.num_scales))
