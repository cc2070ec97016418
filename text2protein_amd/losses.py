"""Training step on MI355X: Python mirror of the reference's ``score_sde_pytorch/losses.py`` over the C ABI
(``t2p_train_*``, include/t2p.h).  First slice of SURVEY.md 8(f)4: fp32 arithmetic, VE SDE.

The reference keeps four objects in ``state`` -- ``model`` (DataParallel(UNetModel)), ``optimizer`` (torch Adam),
``ema`` (ExponentialMovingAverage) and ``step`` (train.py:118-124) -- and ``step_fn`` (losses.py:165-176) drives them:
zero_grad, loss_fn, backward, optimize_fn (warm-up, clip, Adam), ``step += 1``, ``ema.update``.  Here the parameters, their
gradients, both Adam moments and the EMA shadow live in ONE native object (flat device buffers in ``parameters()`` order) and
one call runs the whole step on the GPU; the classes below are views of that object with the reference's names and call
signatures, so a training script reads the same:

    model = get_train_model(config)                  # utils.get_model(config)
    optimizer = get_optimizer(config, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_rate)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=0)
    step_fn = get_step_fn(sde, train=True, optimize_fn=optimization_manager(config))
    loss = step_fn(state, batch, condition=config.model.condition)

There is no CPU fallback: without a HIP device ``HipTrainModel`` raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import T2PError, TrainBatch, TrainConfig, check, ptr, stream_ptr
from .arch import param_specs
from .model import _model_config
from .sde_lib import VESDE

_COND_FLAGS = {"length": 1, "ss": 2, "inpainting": 4}
PARAM, GRAD, EMA, EXP_AVG, EXP_AVG_SQ = 0, 1, 2, 3, 4


def condition_flags(condition) -> int:
    flags = 0
    for c in (condition or []):
        if c not in _COND_FLAGS:
            raise ValueError(f"unknown condition {c!r}")       # the reference ignores unknown names silently (losses.py:115-123)
        flags |= _COND_FLAGS[c]
    return flags


class HipTrainModel:
    """The score network in training form: ``UNetModel`` + its optimizer state + its EMA, resident on one GPU."""

    def __init__(self, config, device="cuda:0", seed=0):
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise T2PError("HipTrainModel needs a GPU device (there is no CPU fallback)")
        self.lib = _lib.load()
        torch.cuda.set_device(self.device)
        self._mc = _model_config(config, "f32")
        o, m = config.optim, config.model
        if o.optimizer != "Adam":
            raise NotImplementedError(f"Optimizer {o.optimizer} not supported yet!")          # losses.py:32-34
        tc = TrainConfig()
        tc.lr, tc.beta1, tc.eps, tc.weight_decay = float(o.lr), float(o.beta1), float(o.eps), float(o.weight_decay)
        tc.warmup, tc.grad_clip = float(o.warmup), float(o.grad_clip)
        tc.ema_rate, tc.dropout, tc.t_eps = float(m.ema_rate), float(m.dropout), 1e-5
        tc.cond_flags = condition_flags(m.condition)
        tc.seed = int(seed)
        self._tc = tc
        h = C.c_void_p()
        check(self.lib.t2p_train_create(C.byref(self._mc), C.byref(tc), C.byref(h)))
        self._h = h
        self._specs = param_specs(config)
        if self.lib.t2p_train_num_params(self._h) != len(self._specs):
            raise T2PError("parameter table mismatch between the trainer and arch.param_specs")
        self._loaded = False
        self.training = True
        self._keep = []

    # -- nn.Module-like surface ----------------------------------------------------------------------------------------
    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, *_a, **_k):
        return self

    def parameters(self):
        """Handle the optimizer / EMA views are built from (the tensors themselves stay on the device)."""
        return _ParamHandle(self)

    def param_table(self):
        out = []
        name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        for i in range(self.lib.t2p_train_num_params(self._h)):
            check(self.lib.t2p_train_param_info(self._h, i, C.byref(name), shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def load_state_dict(self, state_dict, strict=True):
        seen = set()
        for k, v in state_dict.items():
            name = k[7:] if k.startswith("module.") else k
            if name == "sigmas":
                continue
            t = torch.as_tensor(v).detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            check(self.lib.t2p_train_load_param(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            seen.add(name)
        missing = [s.name for s in self._specs if s.name not in seen]
        if missing and strict:
            raise T2PError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self._loaded = not missing
        return missing

    def read(self, which, name=None):
        """One tensor (``name``) or all of them (OrderedDict in ``parameters()`` order) of buffer ``which``
        (PARAM / GRAD / EMA / EXP_AVG / EXP_AVG_SQ) as CPU float32 tensors."""
        specs = [s for s in self._specs if name is None or s.name == name]
        out = OrderedDict()
        for s in specs:
            t = torch.empty(tuple(s.shape), dtype=torch.float32)
            check(self.lib.t2p_train_read(self._h, which, s.name.encode(), C.c_void_p(t.data_ptr())))
            out[s.name] = t
        return out[name] if name is not None else out

    def write(self, which, tensors):
        for k, v in tensors.items():
            t = torch.as_tensor(v).detach().to("cpu", torch.float32).contiguous()
            check(self.lib.t2p_train_write(self._h, which, k.encode(), C.c_void_p(t.data_ptr())))

    def state_dict(self):
        return self.read(PARAM)

    def set_step(self, step, adam_updates=None, ema_updates=None):
        cur = self.get_step()
        check(self.lib.t2p_train_set_step(self._h, int(step), int(cur[1] if adam_updates is None else adam_updates),
                                          int(cur[2] if ema_updates is None else ema_updates)))

    def get_step(self):
        out = (C.c_int64 * 3)()
        check(self.lib.t2p_train_get_step(self._h, out))
        return tuple(int(v) for v in out)

    def set_dropout_masks(self, masks):
        """Parity runs: NHWC uint8 keep-masks (device tensors), one per residual block in forward order; None / [] = Philox."""
        masks = [m.to(self.device, torch.uint8).contiguous() for m in (masks or [])]
        self._keep = masks
        arr = (C.c_void_p * max(len(masks), 1))(*[m.data_ptr() for m in masks])
        check(self.lib.t2p_train_set_dropout_masks(self._h, arr, len(masks)))

    def _batch(self, batch, t=None, z=None):
        dev = self.device
        hold = [batch["coords_6d"].to(dev, torch.float32).contiguous(), batch["mask_pair"].to(dev, torch.uint8).contiguous(),
                batch["context"].to(dev, torch.float32).contiguous()]
        tb = TrainBatch()
        tb.coords_6d, tb.mask_pair, tb.context = hold[0].data_ptr(), hold[1].data_ptr(), hold[2].data_ptr()
        tb.batch, tb.tokens = hold[0].shape[0], hold[2].shape[1]
        if hold[2].shape[2] != self.config.model.context_dim:
            raise T2PError("context width differs from model.context_dim")
        for key, val, dt in (("mask_inpaint", batch.get("mask_inpaint"), torch.uint8), ("t", t, torch.float32), ("z", z, torch.float32)):
            if val is not None:
                v = val.to(dev, dt).contiguous()
                hold.append(v)
                setattr(tb, key, v.data_ptr())
        return tb, hold

    def loss(self, batch, t=None, z=None, backward=False, return_score=False):
        """``loss_fn`` (losses.py:105-134); ``t`` / ``z`` = the draws of :106-107 (None: drawn on the device)."""
        if not self._loaded:
            raise T2PError("load weights before training")
        tb, hold = self._batch(batch, t, z)
        out = C.c_float()
        score = torch.empty_like(hold[0]) if return_score else None
        check(self.lib.t2p_train_loss(self._h, C.byref(tb), int(backward), C.byref(out), ptr(score), stream_ptr()))
        return (float(out.value), score) if return_score else float(out.value)

    def step(self, batch, t=None, z=None):
        """``step_fn`` with train=True (losses.py:165-176) in one call."""
        if not self._loaded:
            raise T2PError("load weights before training")
        tb, hold = self._batch(batch, t, z)
        out = C.c_float()
        check(self.lib.t2p_train_step(self._h, C.byref(tb), C.byref(out), stream_ptr()))
        return float(out.value)

    def apply(self):
        """optimize_fn + ``step += 1`` + ``ema.update`` (losses.py:41-49, 174-176; ema.py:32-49) on the gradient buffer as it stands."""
        check(self.lib.t2p_train_apply(self._h, stream_ptr()))

    def grad_view(self):
        """The flat fp32 gradient buffer (every tensor in ``parameters()`` order) as a torch tensor sharing the trainer's device memory:
        what data-parallel training all-reduces between ``loss(..., backward=True)`` and ``apply()``."""
        p, n = C.c_void_p(), C.c_int64()
        check(self.lib.t2p_train_grad_buffer(self._h, C.byref(p), C.byref(n)))

        class _Buf:          # CUDA array interface: torch wraps the memory without copying
            __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f4", "data": (int(p.value), False), "version": 2}

        keep = _Buf()
        t = torch.as_tensor(keep, device=self.device)
        t._t2p_owner = (self, keep)
        return t

    def eval_loss(self, batch, t=None, z=None):
        """``step_fn`` with train=False (losses.py:177-183): the loss under the EMA weights."""
        tb, hold = self._batch(batch, t, z)
        out = C.c_float()
        check(self.lib.t2p_train_eval_loss(self._h, C.byref(tb), C.byref(out), stream_ptr()))
        return float(out.value)

    def device_bytes(self):
        return int(self.lib.t2p_train_device_bytes(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.t2p_train_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass


class _ParamHandle:
    """What ``model.parameters()`` returns: the optimizer and the EMA are views of the model's native state."""

    def __init__(self, model):
        self.model = model


def get_train_model(config, seed=0):
    """``utils.get_model(config)`` (score_sde_pytorch/utils.py:4-9) for training: no DataParallel, one process per GPU."""
    return HipTrainModel(config, device=config.device if str(config.device) != "cuda" else "cuda:0", seed=seed)


class AdamView:
    """``get_optimizer``'s return value: hyper-parameters are those of the config the model was created from."""

    def __init__(self, model, lr, betas, eps, weight_decay):
        self.model = model
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]

    def zero_grad(self):
        pass            # the native step zeroes the gradient buffer itself

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what the reference's save_checkpoint stores, score_sde_pytorch/utils.py:19-26):
        ``state`` = {parameter index: {step, exp_avg, exp_avg_sq}} in ``parameters()`` order (empty before the first update),
        ``param_groups`` = one group listing every index."""
        k = self.model.get_step()[1]
        names = [s.name for s in self.model._specs]
        state = {}
        if k > 0:
            m, v = self.model.read(EXP_AVG), self.model.read(EXP_AVG_SQ)
            state = {i: {"step": torch.tensor(float(k)), "exp_avg": m[n], "exp_avg_sq": v[n]} for i, n in enumerate(names)}
        g = dict(self.param_groups[0])
        group = dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"], amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False,
                     params=list(range(len(names))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        names = [s.name for s in self.model._specs]
        st = sd.get("state", {})
        cur = self.model.get_step()
        if not st:
            self.model.set_step(cur[0], adam_updates=0)
            return
        if len(st) != len(names):
            raise T2PError(f"optimizer state holds {len(st)} tensors, the model has {len(names)}")
        self.model.write(EXP_AVG, {n: st[i]["exp_avg"] for i, n in enumerate(names)})
        self.model.write(EXP_AVG_SQ, {n: st[i]["exp_avg_sq"] for i, n in enumerate(names)})
        steps = {int(torch.as_tensor(st[i]["step"]).item()) for i in range(len(names))}
        if len(steps) != 1:
            raise T2PError("per-parameter Adam step counts differ: not a state this trainer can continue from")
        self.model.set_step(cur[0], adam_updates=steps.pop())


def get_optimizer(config, params):
    """losses.py:26-36."""
    if config.optim.optimizer != "Adam":
        raise NotImplementedError(f"Optimizer {config.optim.optimizer} not supported yet!")
    if not isinstance(params, _ParamHandle):
        raise TypeError("get_optimizer takes model.parameters() of a HipTrainModel")
    o = config.optim
    return AdamView(params.model, o.lr, (o.beta1, 0.999), o.eps, o.weight_decay)


def optimization_manager(config):
    """losses.py:38-51.  The returned function only checks that its arguments are the ones the native step was built with
    (the warm-up, the clipping and the Adam update themselves run inside ``t2p_train_step``)."""
    o = config.optim

    def optimize_fn(optimizer, params, step, lr=o.lr, warmup=o.warmup, grad_clip=o.grad_clip):
        tc = optimizer.model._tc
        if (float(lr), float(warmup), float(grad_clip)) != (tc.lr, tc.warmup, tc.grad_clip):
            raise T2PError("optimize_fn arguments differ from the configuration the model was created with")

    return optimize_fn


class ExponentialMovingAverage:
    """models/ema.py:8-93 as a view of the model's EMA buffer."""

    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")                                # ema.py:24-25
        if not isinstance(parameters, _ParamHandle):
            raise TypeError("ExponentialMovingAverage takes model.parameters() of a HipTrainModel")
        self.model = parameters.model
        if abs(decay - self.model._tc.ema_rate) > 1e-12 or not use_num_updates:
            raise T2PError("the EMA of a HipTrainModel runs with model.ema_rate and use_num_updates=True")
        self.decay = decay

    @property
    def num_updates(self):
        return self.model.get_step()[2]

    @property
    def shadow_params(self):
        return list(self.model.read(EMA).values())

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)      # ema.py:86-88

    def load_state_dict(self, state_dict):
        names = [s.name for s in self.model._specs]
        self.model.write(EMA, dict(zip(names, state_dict["shadow_params"])))
        cur = self.model.get_step()
        self.model.set_step(cur[0], ema_updates=int(state_dict["num_updates"]))


def get_sde_loss_fn(sde, train, eps=1e-5):
    """losses.py:66-136.  ``batch["context"]`` holds the caption embedding (B, T, context_dim); with raw captions pass
    ``llm_components`` = a callable ``captions -> embedding`` (text2protein_amd.text_context.TextContextProducer)."""
    if not isinstance(sde, VESDE):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def loss_fn(model, batch, condition=None, llm_components=None, t=None, z=None):
        if condition_flags(condition) != model._tc.cond_flags:
            raise T2PError("`condition` differs from the model.condition the model was created with")
        if "context" not in batch:
            if llm_components is None:
                raise T2PError("the batch holds no `context`; pass llm_components=TextContextProducer(...)")
            batch = dict(batch, context=llm_components(batch["caption"]))
        if train:
            return model.loss(batch, t=t, z=z, backward=False)
        return model.eval_loss(batch, t=t, z=z)

    return loss_fn


def get_step_fn(sde, train, optimize_fn=None, dist=None):
    """losses.py:140-186.  ``dist`` (an initialised ``torch.distributed`` module, text2protein_amd.distributed.init_process_group):
    data-parallel training, one process per GPU -- each rank takes its shard of the batch, the flat gradient buffer is averaged over
    the ranks with ONE all-reduce (RCCL over xGMI) and every rank applies the same update; the returned loss is the mean over the
    ranks.  With equal shards this is the reference's DataParallel step on the concatenated batch (the loss is a mean of per-sample
    terms, losses.py:128-131)."""
    if not isinstance(sde, VESDE):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1

    def step_fn(state, batch, condition=None, t=None, z=None):
        model = state["model"]
        if condition_flags(condition) != model._tc.cond_flags:
            raise T2PError("`condition` differs from the model.condition the model was created with")
        if "context" not in batch:
            producer = state.get("llm")
            if producer is None:
                raise T2PError("the batch holds no `context`; put a TextContextProducer under state['llm']")
            batch = dict(batch, context=producer(batch["caption"]))
        if train:
            if optimize_fn is not None:
                optimize_fn(state["optimizer"], model.parameters(), step=state["step"])
            model.set_step(state["step"])
            if world == 1:
                loss = model.step(batch, t=t, z=z)                # zero_grad, loss, backward, optimize_fn, ema.update
            else:
                from . import distributed as D
                loss = model.loss(batch, t=t, z=z, backward=True)
                D.allreduce_mean_(model.grad_view(), dist)        # one collective over the whole gradient
                loss = D.mean_over_ranks(loss, dist, model.device)
                model.apply()
            state["step"] += 1
            return loss
        return model.eval_loss(batch, t=t, z=z)                  # ema.store / copy_to / loss / restore (losses.py:177-183)

    return step_fn
