"""ctypes binding of libt2p_hip.so (C ABI: include/t2p.h).

There is no fallback: if the library is missing or a call fails, a ``T2PError`` is raised.
"""
from __future__ import annotations

import ctypes as C

from . import build as _build

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
DTYPE_NAMES = {"f32": DT_F32, "fp32": DT_F32, "float32": DT_F32, "bf16": DT_BF16, "bfloat16": DT_BF16,
               "f16": DT_F16, "fp16": DT_F16, "float16": DT_F16}
SDE_VE, SDE_VP = 0, 1


class T2PError(RuntimeError):
    pass


class ModelConfig(C.Structure):
    _fields_ = [("num_channels", C.c_int32), ("max_res_num", C.c_int32), ("nf", C.c_int32),
                ("num_res_blocks", C.c_int32), ("n_ch_mult", C.c_int32), ("ch_mult", C.c_int32 * 8),
                ("n_attn_resolutions", C.c_int32), ("attn_resolutions", C.c_int32 * 8),
                ("n_heads", C.c_int32), ("context_dim", C.c_int32), ("num_scales", C.c_int32),
                ("sigma_min", C.c_double), ("sigma_max", C.c_double),
                ("skip_rescale", C.c_int32), ("scale_by_sigma", C.c_int32), ("compute_dtype", C.c_int32)]


class SamplerConfig(C.Structure):
    _fields_ = [("sde", C.c_int32), ("N", C.c_int32),
                ("sigma_min", C.c_double), ("sigma_max", C.c_double), ("beta_min", C.c_double), ("beta_max", C.c_double),
                ("snr", C.c_double), ("n_steps_each", C.c_int32), ("probability_flow", C.c_int32),
                ("denoise", C.c_int32), ("eps", C.c_double), ("batch", C.c_int32), ("global_batch", C.c_int32),
                ("seed", C.c_uint64)]


class TrainConfig(C.Structure):          # t2p_train_config
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("warmup", C.c_double), ("grad_clip", C.c_double), ("ema_rate", C.c_double), ("dropout", C.c_double),
                ("t_eps", C.c_double), ("cond_flags", C.c_int32), ("seed", C.c_uint64)]


class TrainBatch(C.Structure):           # t2p_train_batch
    _fields_ = [("coords_6d", C.c_void_p), ("mask_pair", C.c_void_p), ("mask_inpaint", C.c_void_p), ("context", C.c_void_p),
                ("batch", C.c_int32), ("tokens", C.c_int32), ("t", C.c_void_p), ("z", C.c_void_p)]


_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); every symbol declared in include/t2p.h
SIGNATURES = {
    "t2p_last_error": (C.c_char_p, []),
    "t2p_device_count": (_i, []),
    "t2p_engine_create": (_i, [C.POINTER(ModelConfig), C.POINTER(_vp)]),
    "t2p_engine_destroy": (None, [_vp]),
    "t2p_engine_num_params": (_i, [_vp]),
    "t2p_engine_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i)]),
    "t2p_engine_load_param": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "t2p_engine_finalize": (_i, [_vp]),
    "t2p_engine_set_context": (_i, [_vp, _vp, _i, _i, _vp]),
    "t2p_engine_score": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "t2p_engine_score_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "t2p_engine_device_bytes": (_i64, [_vp]),
    "t2p_engine_pool_reclaimed": (_i, [_vp]),
    "t2p_sampler_create": (_i, [_vp, C.POINTER(SamplerConfig), _vp, _vp, C.POINTER(_vp)]),
    "t2p_sampler_set_seed": (_i, [_vp, _u64]),
    "t2p_sampler_set_norm_allreduce": (_i, [_vp, _vp, _vp, _vp]),
    "t2p_built_with_ablation": (_i, []),
    "t2p_sampler_destroy": (None, [_vp]),
    "t2p_sampler_set_condition": (_i, [_vp, _vp, _vp]),
    "t2p_sampler_reset": (_i, [_vp, _i, _vp]),
    "t2p_sampler_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "t2p_sampler_step_graph": (_i, [_vp, _vp, _vp, _vp]),
    "t2p_sampler_run": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "t2p_sampler_set_vp_tables": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "t2p_sampler_count_dispatches": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    "t2p_op_gemm": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _vp, _vp, _f, _vp]),
    "t2p_op_conv3x3_shortcut": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2p_op_conv3x3_groupnorm": (_i, [_i, _vp, _vp, _vp, _vp, _vp, C.c_float, _i, _i, _vp, _vp, C.c_float, _i, _vp, _i, _vp, _vp,
                                      _i, _i, _i, _i, _i, _vp]),
    "t2p_op_attention_wide": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _vp, _vp, _i, _f, _vp, _i, _i, _i, _f, _vp]),
    "t2p_op_attention_wide_fm": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _i, _i, _i, _f, _vp]),
    "t2p_op_gemm_frag_major": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2p_op_small_conv_groupnorm": (_i, [_i, _vp, _vp, _i64, _vp, _i, _vp, _i, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _vp, _vp, _f, _i,
                                         _i, _i, _i, _i, _i, _vp]),
    "t2p_op_attn_proj": (_i, [_i, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "t2p_op_st_entry": (_i, [_i, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                             _i, _i, _i, _vp]),
    "t2p_op_input_conv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "t2p_op_gemm_r16": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _vp, _vp, _f, _vp]),
    "t2p_op_conv3x3": (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "t2p_op_groupnorm": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _i, _vp, _i, _vp]),
    "t2p_op_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp]),
    "t2p_op_layernorm16": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp]),
    "t2p_op_softmax": (_i, [_vp, _i64, _vp, _i64, _i, _i64, _i, _f, _vp]),
    "t2p_op_geglu": (_i, [_vp, _vp, _i, _i64, _i, _vp]),
    "t2p_op_attention_ws": (_i64, [_i, _i, _i, _i, _i]),
    "t2p_op_attention_qkv": (_i, [_i, _vp, _i64, _vp, _i, _i, _i, _i, _f, _vp]),
    "t2p_op_attention": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "t2p_op_langevin": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _f, _f, _vp, _vp]),
    "t2p_op_langevin_norms": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "t2p_op_langevin_update": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _vp]),
    "t2p_op_apply_mask": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "t2p_op_predictor": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f, _i, _vp]),
    "t2p_op_philox_normal": (_i, [_vp, _i64, _u64, _u64, _vp]),
    "t2p_profile_begin": (_i, []),
    "t2p_debug_set": (_i, [_i, _i]),
    "t2p_op_decode_6d": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "t2p_op_embedding_gather": (_i, [_vp, _i, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "t2p_profile_end": (_i, [C.POINTER(C.c_double)]),
    "t2p_profile_dominant": (_i, [C.POINTER(C.c_double), C.c_char_p, C.c_int]),
    "t2p_profile_attention": (_i, [C.POINTER(C.c_double)]),
    "t2p_profile_shapes": (_i, [C.c_char_p, C.c_int]),
    "t2p_debug_tap": (_i, [_i, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "t2p_profile_layers_begin": (_i, []),
    "t2p_profile_layers_end": (_i, [C.c_char_p, C.c_int]),
    "t2p_op_convert": (_i, [_vp, _vp, _i, _i64, _vp]),
    # training step
    "t2p_train_create": (_i, [C.POINTER(ModelConfig), C.POINTER(TrainConfig), C.POINTER(_vp)]),
    "t2p_train_destroy": (None, [_vp]),
    "t2p_train_num_params": (_i, [_vp]),
    "t2p_train_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i)]),
    "t2p_train_load_param": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "t2p_train_read": (_i, [_vp, _i, C.c_char_p, _vp]),
    "t2p_train_write": (_i, [_vp, _i, C.c_char_p, _vp]),
    "t2p_train_set_step": (_i, [_vp, _i64, _i64, _i64]),
    "t2p_train_get_step": (_i, [_vp, C.POINTER(_i64)]),
    "t2p_train_set_dropout_masks": (_i, [_vp, C.POINTER(_vp), _i]),
    "t2p_train_loss": (_i, [_vp, C.POINTER(TrainBatch), _i, C.POINTER(_f), _vp, _vp]),
    "t2p_train_step": (_i, [_vp, C.POINTER(TrainBatch), C.POINTER(_f), _vp]),
    "t2p_train_eval_loss": (_i, [_vp, C.POINTER(TrainBatch), C.POINTER(_f), _vp]),
    "t2p_train_apply": (_i, [_vp, _vp]),
    "t2p_train_grad_buffer": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "t2p_train_device_bytes": (_i64, [_vp]),
    "t2p_op_tgemm": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i, _i, _i, _i, _i64, _i64, _i64, _f, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "t2p_op_groupnorm_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "t2p_op_layernorm_backward": (_i, [_vp, _vp, _vp, _i64, _i, _f, _vp, _vp, _vp, _vp]),
    "t2p_op_softmax_backward": (_i, [_vp, _vp, _i64, _i, _f, _vp]),
    "t2p_op_geglu_backward": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
}

_lib = None


def lib_path() -> str:
    return _build.LIB


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)   # t2p_allreduce_fn


def load_path(path):
    """Measurement tools only (bench.py --lib): dlopen a library built from another revision, for A/B runs inside one call."""
    global _lib
    if _lib is not None:
        raise T2PError("load_path() must come before any other use of the library")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:          # an older revision: entry points added since are simply not there
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_ablation():
    """Measurement tools only: the -DT2P_ABLATION build (a separate file; results can be WRONG by design)."""
    global _lib
    if _lib is not None:
        raise T2PError("load_ablation() must come before any other use of the library")
    lib = C.CDLL(_build.build(verbose=False, ablation=True))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    assert lib.t2p_built_with_ablation() == 1
    _lib = lib
    return lib


def load(build_if_missing: bool = True):
    """dlopen libt2p_hip.so, building it first when it is missing or was built from other sources than the
    ones in the tree (content hash, text2protein_amd/build.py).  A stale library is never loaded: if the
    rebuild fails, so does this call."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if _build.needs_build():
        if not build_if_missing:
            raise T2PError(f"{path} is missing or stale (built from other sources): run `python -m text2protein_amd.build`")
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise T2PError(f"libt2p_hip.so is missing or stale and could not be rebuilt: {e}") from e
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_plan_switches(spec: str):
    """Measurement tools only (bench.py --plan, tools/): "key=value,key=value" -> t2p_debug_set (include/t2p.h).
    The product path never calls this and reads no environment variable."""
    lib = load()
    for kv in filter(None, (spec or "").split(",")):
        k, v = kv.split("=")
        if lib.t2p_debug_set(int(k), int(v)) != 0:
            msg = lib.t2p_last_error()
            raise T2PError(f"bad plan switch {kv}: {msg.decode() if msg else '?'}")


def check(rc: int):
    if rc != 0:
        msg = load().t2p_last_error()
        raise T2PError(f"libt2p_hip call failed (status {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
