"""Predictor-corrector sampling on the HIP engine: mirror of ``score_sde_pytorch/sampling.py``.

Same names, arguments, registries and error behaviour as the reference:

  register_predictor / register_corrector / get_predictor / get_corrector   sampling.py:32-75
      duplicate name -> ValueError, unknown name -> KeyError
  get_sampling_fn(config, sde, shape, eps)                                  sampling.py:78-104
  Predictor / Corrector ABCs, ReverseDiffusionPredictor, LangevinCorrector  sampling.py:107-199
  get_pc_sampler(...) -> pc_sampler(model, condition=None, context=None)     sampling.py:213-291
  get_score_fn(sde, model, train=False, continuous=False)                   models/utils.py:126-176

The state update of every step (score network, noise, norms, Euler-Maruyama / Langevin update,
conditional masking) runs in libt2p_hip.so.  torch is used for the once-per-run condition set-up
and for (B,)-sized schedule scalars.  Two execution routes produce the same numbers:

  * fused   -- VE SDE + 'reverse_diffusion' + 'langevin' + a ``HipScoreModel`` (every shipped
               config): the whole step is enqueued by one C call (t2p_sampler_step), no host sync.
  * classes -- anything else registered by the user, or the VP SDE: the loop below calls
               ``update_fn`` of the predictor / corrector objects, which call the operator ABI.
"""
from __future__ import annotations

import abc
import ctypes as C
import functools

import torch

from . import _lib, sde_lib
from ._lib import SamplerConfig, T2PError, check, ptr, stream_ptr
from .model import HipScoreModel

class _Registry(dict):
    """name -> class table behind ``register_predictor`` / ``register_corrector`` (reference sampling.py:28-75):
    registering a taken name raises ``ValueError``, looking up an unknown one raises ``KeyError``."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def decorator(self, cls=None, *, name=None):
        """``@reg`` or ``@reg(name=...)``; the class is returned unchanged."""
        def add(c):
            key = name or c.__name__
            if key in self:
                raise ValueError(f"Already registered model with name: {key}")
            self[key] = c
            return c
        return add(cls) if cls is not None else add


_PREDICTORS, _CORRECTORS = _Registry("predictor"), _Registry("corrector")
register_predictor, register_corrector = _PREDICTORS.decorator, _CORRECTORS.decorator
get_predictor, get_corrector = _PREDICTORS.__getitem__, _CORRECTORS.__getitem__


# ------------------------------------------------------------------------------------------------
# score function adapter
# ------------------------------------------------------------------------------------------------
def get_score_fn(sde, model, train=False, continuous=False):
    """models/utils.py:126-176: ``score_fn(x, t, context=None)`` for ``model(x, labels, context)`` (a ``HipScoreModel`` or
    any callable with the reference signature).  VE: integer labels ``round((T - t)(N - 1))`` (noisiest level first) or,
    ``continuous``, the marginal std; the network output is the score.  VP / subVP: fractional labels ``t (N - 1)``
    (``t * 999`` when continuous or subVP) and the output is divided by minus the marginal std."""
    if train:
        raise T2PError("the HIP engine is inference-only (train=True is not on the sampling path)")
    ve = isinstance(sde, sde_lib.VESDE)
    sub = isinstance(sde, sde_lib.subVPSDE)
    if not (ve or sub or isinstance(sde, sde_lib.VPSDE)):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def evaluate(x, labels, context):
        model.eval()
        return model(x, labels, context)

    def score_ve(x, t, context=None):
        labels = sde.marginal_prob_std(t) if continuous else torch.round((sde.T - t) * (sde.N - 1)).long()
        return evaluate(x, labels, context)

    def score_vp(x, t, context=None):
        if continuous or sub:
            out = evaluate(x, t * 999, context)
            std = sde.marginal_prob_std(t)
        else:
            labels = t * (sde.N - 1)
            out = evaluate(x, labels, context)
            std = sde.sqrt_1m_alphas_cumprod.to(labels.device)[labels.long()]
        return -out / std[:, None, None, None].to(out.device)

    return score_ve if ve else score_vp


# ------------------------------------------------------------------------------------------------
# predictors / correctors
# ------------------------------------------------------------------------------------------------
class _UpdateRule(abc.ABC):
    """Common part of ``Predictor`` and ``Corrector`` (sampling.py:107-150): holds the SDE and the score function;
    ``update_fn(x, t, context=None) -> (x, x_mean)`` is the one method a registered class supplies."""

    def __init__(self, sde, score_fn):
        self.sde, self.score_fn = sde, score_fn

    @abc.abstractmethod
    def update_fn(self, x, t, context=None):
        ...


class Predictor(_UpdateRule):
    """Predictor algorithm: ``Predictor(sde, score_fn, probability_flow=False)``."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__(sde, score_fn)
        self.probability_flow = probability_flow


class Corrector(_UpdateRule):
    """Corrector algorithm: ``Corrector(sde, score_fn, snr, n_steps)``."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn)
        self.snr, self.n_steps = snr, n_steps


def _device_randn_like(x, seed, stream_id):
    out = torch.empty_like(x)
    check(_lib.load().t2p_op_philox_normal(ptr(out), out.numel(), seed, stream_id, stream_ptr()))
    return out


class _NoiseSource:
    """Standard-normal draws for the loop.  ``device``: Philox kernel keyed by (seed, draw index);
    a callable ``noise_fn(shape) -> CPU tensor`` reproduces a host stream (parity runs)."""

    def __init__(self, seed=0, noise_fn=None):
        self.seed = int(seed)
        self.noise_fn = noise_fn
        self.count = 0

    def like(self, x):
        self.count += 1
        if self.noise_fn is not None:
            return self.noise_fn(tuple(x.shape)).to(x.device, torch.float32).contiguous()
        return _device_randn_like(x, self.seed, self.count)


_default_noise = _NoiseSource()

_M64 = (1 << 64) - 1


def call_seed(seed: int, call: int) -> int:
    """Seed of the ``call``-th invocation of a sampler built with ``seed``.  The reference draws fresh
    ``torch.randn`` noise on every call of ``pc_sampler`` (sampling.py:253,164,192); the counter-based
    generator here is keyed by (seed, stream, step) alone, so the call index is folded into the seed
    (call 0 keeps ``seed`` itself; later calls go through one splitmix64 round)."""
    seed, call = int(seed) & _M64, int(call)
    if call == 0:
        return seed
    z = (seed + 0x9E3779B97F4A7C15 * call) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


@register_predictor(name="reverse_diffusion")
class ReverseDiffusionPredictor(Predictor):
    """sampling.py:157-167 with RSDE.discretize (sde_lib.py:96-101)."""

    noise = _default_noise

    def update_fn(self, x, t, context=None):
        lib = _lib.load()
        a, G = self.sde.discretize_coeffs(t)
        score = self.score_fn(x, t, context).float().contiguous()
        z = self.noise.like(x)
        x_in = x
        if float(a[0]) != 0.0:      # VP: f = (sqrt(alpha) - 1) x  ->  x - f = (2 - sqrt(alpha)) x
            x_in = (x * (1.0 - a.to(x.device))[:, None, None, None]).contiguous()
        x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
        check(lib.t2p_op_predictor(ptr(x_in), ptr(score), ptr(z), None, None, ptr(x_new), ptr(x_mean), x.numel(),
                                   float(G[0]), int(bool(self.probability_flow)), stream_ptr()))
        return x_new, x_mean


@register_corrector(name="langevin")
class LangevinCorrector(Corrector):
    """sampling.py:170-199.  ``all_reduce`` (optional callable on the 2-float norm-sum tensor)
    and ``global_batch`` give the global-batch step size of SURVEY 8(e) option B."""

    noise = _default_noise
    all_reduce = None
    global_batch = None

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t, context=None):
        lib = _lib.load()
        sde = self.sde
        if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
            timestep = (t.detach().cpu() * (sde.N - 1) / sde.T).long()
            alpha = float(sde.alphas[timestep][0])
        else:
            alpha = 1.0
        B = x.shape[0]
        x_mean = x
        ws = torch.empty(B * 128, device=x.device)
        sums = torch.empty(2, device=x.device)
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t, context).float().contiguous()
            noise = self.noise.like(x)
            check(lib.t2p_op_langevin_norms(ptr(grad), ptr(noise), B, x[0].numel(), ptr(ws), ptr(sums), stream_ptr()))
            total = B
            if self.all_reduce is not None:
                self.all_reduce(sums)
                total = self.global_batch or B
            x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
            check(lib.t2p_op_langevin_update(ptr(x), ptr(grad), ptr(noise), None, None, ptr(x_new), ptr(x_mean),
                                             x.numel(), ptr(sums), float(total), float(self.snr), alpha, stream_ptr()))
            x = x_new
        return x, x_mean


def shared_predictor_update_fn(x, t, context, sde, model, predictor, probability_flow):
    """A wrapper that configures and returns the update function of predictors (sampling.py:201-205)."""
    score_fn = get_score_fn(sde, model, train=False)
    return predictor(sde, score_fn, probability_flow).update_fn(x, t, context)


def shared_corrector_update_fn(x, t, context, sde, model, corrector, snr, n_steps):
    """sampling.py:207-211."""
    score_fn = get_score_fn(sde, model, train=False)
    return corrector(sde, score_fn, snr, n_steps).update_fn(x, t, context)


# ------------------------------------------------------------------------------------------------
# the sampler
# ------------------------------------------------------------------------------------------------
def allreduce_trampoline(all_reduce, sums):
    """The C-callable (``t2p_allreduce_fn``) the fused sampler calls once per corrector step, wrapping ``all_reduce(sums)``.
    A Python exception must not unwind through the C frame: it is parked in the returned list and the callback returns 1, which
    makes ``t2p_sampler_step`` fail; ``raise_allreduce_error`` then surfaces the ORIGINAL exception to the caller.  The closure
    holds the list and the tensor, never the stepper (no reference cycle)."""
    err = []

    def _cb(_ptr, _stream, _user, _t=sums, _fn=all_reduce, _err=err):
        try:
            _fn(_t)                                # sums the caller-owned buffer the sampler just filled
            return 0
        except Exception as e:  # noqa: BLE001
            _err.append(e)
            return 1

    return _lib.ALLREDUCE_FN(_cb), err


def raise_allreduce_error(pending, fallback):
    """Re-raise what the all-reduce callable raised (with its text and as ``__cause__``), or ``fallback`` when it did not."""
    if pending:
        cause = pending.pop()
        pending.clear()
        raise T2PError(f"the norm all-reduce failed: {cause!r}") from cause
    raise fallback


class PCStepper:
    """Handle on the fused C++ sampler (t2p_sampler_*): one ``step`` = one iteration of the loop
    body of the reference ``pc_sampler`` (sampling.py:279-285) enqueued on the current stream."""

    def __init__(self, model, sde, batch, snr, n_steps=1, probability_flow=False, denoise=True, eps=1e-5, seed=0,
                 global_batch=None, all_reduce=None):
        """``global_batch`` / ``all_reduce``: the Langevin batch means run over ``global_batch`` chains spread
        over several processes (the reference's DataParallel semantics, sampling.py:193-195): ``all_reduce`` is
        called once per corrector step with a 2-float device tensor it must sum over the processes in place,
        in stream order (``text2protein_amd.distributed.allreduce_norm_sums``)."""
        vp = isinstance(sde, sde_lib.VPSDE)
        if not (vp or isinstance(sde, sde_lib.VESDE)):
            raise T2PError("the fused stepper covers the VE and VP SDEs; others run through the predictor/corrector classes")
        if not isinstance(model, HipScoreModel):
            raise T2PError("the fused stepper needs a HipScoreModel")
        self.model, self.lib = model, model.lib
        sc = SamplerConfig()
        sc.sde = _lib.SDE_VP if vp else _lib.SDE_VE
        sc.N = sde.N
        if vp:
            sc.sigma_min, sc.sigma_max = 0.01, 1.0
            sc.beta_min, sc.beta_max = float(sde.beta_0), float(sde.beta_1)
        else:
            sc.sigma_min, sc.sigma_max = float(sde.sigma_min), float(sde.sigma_max)
            sc.beta_min, sc.beta_max = 0.1, 20.0
        sc.snr = float(snr)
        sc.n_steps_each = int(n_steps)
        sc.probability_flow = int(bool(probability_flow))
        sc.denoise = int(bool(denoise))
        sc.eps = float(eps)
        sc.batch = int(batch)
        sc.global_batch = int(global_batch or batch)
        if (all_reduce is None) != (sc.global_batch == sc.batch):
            raise T2PError("global_batch > batch and the all_reduce callable go together")
        sc.seed = int(seed) & _M64
        g = sde.g_table(eps)                       # the reference's own float32 arithmetic
        labels = sde.label_table(eps)
        vp_tables = sde.vp_tables(eps) if vp else None
        h = C.c_void_p()
        check(self.lib.t2p_sampler_create(model._h, C.byref(sc), C.c_void_p(g.data_ptr()), C.c_void_p(labels.data_ptr()),
                                          C.byref(h)))
        self._h = h
        if vp:
            check(self.lib.t2p_sampler_set_vp_tables(h, *[C.c_void_p(t.data_ptr()) for t in vp_tables]))
        self._keep = ()
        self.N = int(sde.N)
        if all_reduce is not None:
            self._sums = torch.zeros(2, device=model.device, dtype=torch.float32)
            self._cb, self._cb_error = allreduce_trampoline(all_reduce, self._sums)   # kept alive as long as the sampler
            check(self.lib.t2p_sampler_set_norm_allreduce(self._h, ptr(self._sums), self._cb, None))

    def set_seed(self, seed):
        check(self.lib.t2p_sampler_set_seed(self._h, int(seed) & _M64))

    def set_condition(self, mask_u8=None, x_initial=None):
        self._keep = (mask_u8, x_initial)          # the C side borrows these pointers
        check(self.lib.t2p_sampler_set_condition(self._h, ptr(mask_u8), ptr(x_initial)))

    def reset(self, step=0):
        check(self.lib.t2p_sampler_reset(self._h, int(step), stream_ptr()))

    def step(self, x, x_mean, noise_corrector=None, noise_predictor=None):
        try:
            check(self.lib.t2p_sampler_step(self._h, ptr(x), ptr(x_mean), ptr(noise_corrector), ptr(noise_predictor),
                                            stream_ptr()))
        except T2PError as e:
            raise_allreduce_error(getattr(self, "_cb_error", None), e)

    def step_graph(self, x, x_mean):
        """One PC step replayed from a captured hipGraph (device noise; needs a non-default stream)."""
        check(self.lib.t2p_sampler_step_graph(self._h, ptr(x), ptr(x_mean), stream_ptr()))

    def count_dispatches(self, x, x_mean):
        """Device dispatches one PC step enqueues (captured on a side stream and discarded; measurement only)."""
        n = C.c_int(0)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            check(self.lib.t2p_sampler_count_dispatches(self._h, ptr(x), ptr(x_mean), stream_ptr(), C.byref(n)))
        torch.cuda.current_stream().wait_stream(side)
        return int(n.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.t2p_sampler_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass


def get_sampling_fn(config, sde, shape, eps, **kw):
    """sampling.py:78-104."""
    predictor = get_predictor(config.sampling.predictor.lower())
    corrector = get_corrector(config.sampling.corrector.lower())
    return get_pc_sampler(sde=sde, shape=shape, predictor=predictor, corrector=corrector,
                          snr=config.sampling.snr, n_steps=config.sampling.n_steps_each,
                          probability_flow=config.sampling.probability_flow,
                          denoise=config.sampling.noise_removal, eps=eps, device=config.device, **kw)


def apply_conditions(x, condition):
    """sampling.py:259-275: ``condition`` -> (x, conditional_mask).  One-off torch glue."""
    mask = torch.ones_like(x).bool()
    if condition is not None:
        for k, v in condition.items():
            if k == "length":
                v = v.to(x.device)
                x = x * v.unsqueeze(1)
                mask = mask * v.unsqueeze(1)
                x[:, -1] = v
                mask[:, -1] = False
            elif k == "ss":
                x[:, 4:7] = v.to(x.device)
                mask[:, 4:7] = False
            elif k == "inpainting":
                mask = mask * v["mask_inpaint"].to(x.device).unsqueeze(1)
                x = torch.where(mask, x, v["coords_6d"].to(x.device))
    return x, mask


def get_pc_sampler(sde, shape, predictor, corrector, snr, n_steps=1, probability_flow=False, denoise=True,
                   eps=1e-3, device="cuda", seed=0, force_classes=False, global_batch=None, all_reduce=None):
    """Create a Predictor-Corrector (PC) sampler (sampling.py:213-291).

    Extra keyword arguments (not in the reference): ``seed`` of the on-device noise generator;
    ``force_classes`` runs the predictor / corrector objects even where the fused route applies;
    ``global_batch`` + ``all_reduce`` give the Langevin step size the reference computes when its batch is
    spread over devices (mean norms over all ``global_batch`` chains, SURVEY.md 8(e) option B).
    The returned ``pc_sampler(model, condition=None, context=None, noise_fn=None, n_iter=None, call_index=None)``
    accepts ``noise_fn(shape) -> CPU tensor`` to inject the standard-normal draws in the
    reference's order (prior, then corrector and predictor of each step), and ``n_iter`` to stop
    after that many PC steps (benchmarks / tests).  Like the reference, every call draws fresh noise:
    call number k of this sampler uses ``call_seed(seed, k)``; ``call_index`` pins k (reproducing a call).
    """
    device = torch.device("cuda:0" if str(device) == "cuda" else device)
    if device.type != "cuda":
        raise T2PError("the HIP sampler needs a GPU device (no CPU fallback)")
    fused_ok = (isinstance(sde, (sde_lib.VESDE, sde_lib.VPSDE)) and predictor is ReverseDiffusionPredictor
                and corrector is LangevinCorrector and not force_classes)
    state = {"sampler": None, "model": None, "calls": 0}

    def _fused_sampler(model):
        if state["sampler"] is None or state["model"] is not model:
            state["sampler"] = PCStepper(model, sde, shape[0], snr, n_steps, probability_flow, denoise, eps, seed,
                                         global_batch=global_batch, all_reduce=all_reduce)
            state["model"] = model
        return state["sampler"]

    def pc_sampler(model, condition=None, context=None, noise_fn=None, n_iter=None, call_index=None):
        """The PC sampler function -> (samples, number of function evaluations)."""
        lib = _lib.load()
        torch.cuda.set_device(device)
        n_iter = sde.N if n_iter is None else int(n_iter)
        if n_iter > sde.N:
            raise ValueError(f"n_iter {n_iter} exceeds sde.N {sde.N}")
        if call_index is None:
            call_index = state["calls"]
            state["calls"] += 1
        run_seed = call_seed(seed, call_index)
        noise = _NoiseSource(run_seed, noise_fn)
        with torch.no_grad():
            # Initial sample (sde_lib.py:229-230): the reference draws on the CPU and moves it
            if noise_fn is not None:
                x = (noise_fn(tuple(shape)) * sde.prior_scale()).to(device, torch.float32)
            else:
                x = _device_randn_like(torch.empty(*shape, device=device), run_seed, 0) * sde.prior_scale()
            x, conditional_mask = apply_conditions(x, condition)
            x = x.float().contiguous()
            x_initial = x.detach().clone()
            conditioned = bool(condition)
            mask_u8 = conditional_mask.to(torch.uint8).contiguous() if conditioned else None
            if context is not None:
                context = context.to(device, torch.float32).contiguous()
                if isinstance(model, HipScoreModel):
                    model.set_context(context)
                    context = model._ctx_ref

            if fused_ok and isinstance(model, HipScoreModel):
                stepper = _fused_sampler(model)
                stepper.set_condition(mask_u8, x_initial if conditioned else None)
                stepper.set_seed(run_seed)
                stepper.reset(0)
                x_mean = torch.empty_like(x)
                for _ in range(n_iter):
                    nc = npred = None
                    if noise_fn is not None:
                        nc, npred = noise.like(x), noise.like(x)
                    stepper.step(x, x_mean, nc, npred)
            else:
                ReverseDiffusionPredictor.noise = LangevinCorrector.noise = noise
                LangevinCorrector.all_reduce = staticmethod(all_reduce) if all_reduce is not None else None
                LangevinCorrector.global_batch = global_batch
                timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
                predictor_update_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                                        probability_flow=probability_flow)
                corrector_update_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector,
                                                        snr=snr, n_steps=n_steps)
                x_mean = x
                n_el = x.numel()
                for i in range(n_iter):
                    vec_t = torch.ones(shape[0], device=device) * timesteps[i]
                    x, x_mean = corrector_update_fn(x, vec_t, model=model, context=context)
                    if conditioned:
                        check(lib.t2p_op_apply_mask(ptr(x), ptr(mask_u8), ptr(x_initial), n_el, stream_ptr()))
                    x, x_mean = predictor_update_fn(x, vec_t, model=model, context=context)
                    if conditioned:
                        check(lib.t2p_op_apply_mask(ptr(x), ptr(mask_u8), ptr(x_initial), n_el, stream_ptr()))
            if conditioned:
                x_mean = x_mean.float().contiguous()
                check(lib.t2p_op_apply_mask(ptr(x_mean), ptr(mask_u8), ptr(x_initial), x_mean.numel(), stream_ptr()))
            return (x_mean if denoise else x), sde.N * (n_steps + 1)

    return pc_sampler
