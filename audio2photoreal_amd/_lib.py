"""ctypes binding of liba2p_hip.so (include/a2p_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call
fails the product raises -- a CPU/eager path would silently void parity
claims (the oracle lives in oracle/ and is test-only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("A2P_LIB") or os.path.join(_HERE, "liba2p_hip.so")   # A2P_LIB: A/B builds of the kernels
# The same sources built with -DA2P_HALF: the 16-bit operand type of the throughput mode is IEEE half instead of bfloat16
# (precision="fp16"; csrc/a2p_common.h).  Same C ABI, same exports.
LIB_PATH_F16 = os.environ.get("A2P_LIB_F16") or os.path.join(_HERE, "liba2p_hip_f16.so")

FACE, POSE = 0, 1
PREC_F32, PREC_BF16 = 0, 1
PASS_COND, PASS_UNCOND, PASS_CFG = 0, 1, 2
SAMPLER_DDIM, SAMPLER_DDPM = 0, 1
KERNEL_GEMM, KERNEL_ATTN_SELF, KERNEL_ATTN_CROSS, KERNEL_LNROPE, KERNEL_CHAIN = 0, 1, 2, 3, 4
KERNEL_CHAIN_PRE, KERNEL_CHAIN_MID, KERNEL_CHAIN_POST, KERNEL_CHAIN_MIDPOST, KERNEL_POSE_TAIL = 5, 6, 7, 8, 9   # finer classes (a2p_hip.h)
# a2p_table_id
TABLE_NAMES = (
    "posterior_mean_coef1", "posterior_mean_coef2", "posterior_variance", "posterior_log_variance_clipped",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "alphas_cumprod", "alphas_cumprod_prev",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "alphas_cumprod_next",
)
PLMS_PREDICT, PLMS_AB1, PLMS_AB2, PLMS_AB3, PLMS_AB4, PLMS_EULER = range(6)

EXPORTS = (
    "a2p_ctx_create", "a2p_ctx_destroy", "a2p_last_error", "a2p_version", "a2p_set_weight", "a2p_finalize_weights",
    "a2p_prepare_cond", "a2p_denoise_forward", "a2p_sample_step", "a2p_p_mean_variance", "a2p_ddim_update",
    "a2p_p_sample_update", "a2p_q_sample", "a2p_eps_from_xstart", "a2p_plms_update", "a2p_ddim_reverse_update", "a2p_decoder_layer_forward", "a2p_gemm", "a2p_attention",
    "a2p_kernel_timing", "a2p_kernel_time_ms", "a2p_debug_read", "a2p_reload_env", "a2p_set_batch_hint", "a2p_check_finite", "a2p_attention_logit_max", "a2p_precision_verdict",
    "a2p_guide_create", "a2p_guide_destroy", "a2p_guide_set_weight", "a2p_guide_finalize", "a2p_guide_prepare",
    "a2p_guide_forward", "a2p_guide_generate", "a2p_guide_debug_read", "a2p_vq_decode",
    "a2p_frontend_create", "a2p_frontend_destroy", "a2p_frontend_set_weight", "a2p_frontend_finalize",
    "a2p_frontend_encode_audio", "a2p_frontend_encode_lip",
)


class A2PConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "data_format", "nfeats", "latent_dim", "ff_size", "num_layers", "num_heads", "cond_feature_dim",
        "max_frames", "emb_len", "keyframe_dim", "keyframe_step", "precision", "max_batch", "reserved")]


class A2PGuideConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "tokens", "dim", "num_layers", "num_heads", "ff_size", "cond_feature_dim", "emb_len", "num_audio_layers",
        "max_batch", "max_positions")] + [("reserved", C.c_int32 * 2)]


class A2PFrontendConfig(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in (
        "conv_dim", "resample", "lip", "d_model", "num_heads", "ff_size", "enc_layers", "dec_layers", "lip_out", "lip_pad",
        "chunk_frames", "samples_per_frame", "max_batch", "max_frames", "conv_16bit",
        "a_group_norm", "l_group_norm", "a_activation", "l_activation", "a_log_compression", "l_log_compression", "a_skip", "l_skip")]
        + [("a_residual_scale", C.c_float), ("l_residual_scale", C.c_float), ("l_layers", C.c_int32), ("agg_layers", C.c_int32),
           ("agg_skip", C.c_int32), ("agg_residual_scale", C.c_float), ("agg_conv_bias", C.c_int32), ("agg_zero_pad", C.c_int32),
           ("agg_activation", C.c_int32), ("reserved", C.c_int32 * 2)])


class A2PError(RuntimeError):
    pass


class A2PPrecisionWarning(UserWarning):
    """The checkpoint / inputs leave the range the 16-bit throughput modes were validated on (FiLMTransformer.check_finite)."""


# Row maximum of the scaled attention scores beyond which the 16-bit modes are outside what the parity tests cover
# (profiles/r04_trained_like_budget.json, IEEE half, error of the ddim loop's return value: q/k rows x2 -> maximum 13.5, 5.1e-4;
# every weight x2 -> 13.1, 9.3e-4; q/k rows x3 -> 29.3, 2.8e-3; the xavier fixtures reach 17.7 late in the ddim10 loop at 3.7e-4)
LOGIT_ENVELOPE_FP16 = 20.0
ERR_NONFINITE = -5     # include/a2p_hip.h A2P_ERR_NONFINITE (a2p_check_finite)


_libs = {}


def load(half: bool = False) -> C.CDLL:
    """Load the shared library (`half`: the IEEE-half build) or raise (never falls back)."""
    if half in _libs:
        return _libs[half]
    # torch first: its bundled libamdhip64.so.7 must be the one HIP runtime of the process (the
    # library shares torch's device pointers and streams; loading /opt/rocm's copy first breaks both)
    import torch  # noqa: F401
    path = LIB_PATH_F16 if half else LIB_PATH
    if not os.path.exists(path):
        raise A2PError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.a2p_last_error.restype = C.c_char_p
    lib.a2p_version.restype = C.c_char_p
    sig = {
        "a2p_ctx_create": [C.POINTER(A2PConfig), C.POINTER(vp)],
        "a2p_ctx_destroy": [vp],
        "a2p_set_weight": [vp, C.c_char_p, vp, i64, vp],
        "a2p_finalize_weights": [vp, vp],
        "a2p_prepare_cond": [vp, vp, i32, i32, vp, vp, i32, i32, vp],
        "a2p_denoise_forward": [vp, vp, vp, vp, i32, vp, vp],
        "a2p_sample_step": [vp, i32, vp, vp, vp, vp, i32, vp, vp, f32, i32, vp, vp, vp],
        "a2p_p_mean_variance": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp],
        "a2p_ddim_update": [vp, vp, vp, vp, i32, vp, f32, i32, i64, vp, vp],
        "a2p_p_sample_update": [vp, vp, vp, i32, vp, i32, i64, vp, vp],
        "a2p_q_sample": [vp, vp, vp, i32, vp, i32, i64, vp, vp],
        "a2p_eps_from_xstart": [vp, vp, vp, vp, i32, i32, i64, vp, vp],
        "a2p_plms_update": [vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i64, vp, vp],
        "a2p_ddim_reverse_update": [vp, vp, vp, vp, i32, i32, i64, vp, vp],
        "a2p_decoder_layer_forward": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "a2p_gemm": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "a2p_attention": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "a2p_kernel_timing": [vp, i32, i32],
        "a2p_kernel_time_ms": [vp, C.POINTER(C.c_double), C.POINTER(i64)],
        "a2p_debug_read": [vp, C.c_char_p, vp, i64],
        "a2p_reload_env": [vp],
        "a2p_set_batch_hint": [vp, i32],
        "a2p_check_finite": [vp, vp],
        "a2p_attention_logit_max": [vp, C.POINTER(C.c_float), vp],
        "a2p_precision_verdict": [vp, C.POINTER(C.c_float), C.POINTER(i32), vp],
        "a2p_guide_create": [C.POINTER(A2PGuideConfig), C.POINTER(vp)],
        "a2p_guide_destroy": [vp],
        "a2p_guide_set_weight": [vp, C.c_char_p, vp, i64, vp],
        "a2p_guide_finalize": [vp, vp],
        "a2p_guide_prepare": [vp, vp, i32, i32, i32, vp],
        "a2p_guide_forward": [vp, vp, i32, i32, vp, vp],
        "a2p_guide_generate": [vp, i32, i32, f32, vp, vp, vp, vp],
        "a2p_guide_debug_read": [vp, C.c_char_p, vp, i64],
        "a2p_vq_decode": [vp, i32, i32, i32, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp],
        "a2p_frontend_create": [C.POINTER(A2PFrontendConfig), C.POINTER(vp)],
        "a2p_frontend_destroy": [vp],
        "a2p_frontend_set_weight": [vp, C.c_char_p, vp, i64, vp],
        "a2p_frontend_finalize": [vp, vp],
        "a2p_frontend_encode_audio": [vp, vp, i32, i64, vp, i32, vp],
        "a2p_frontend_encode_lip": [vp, vp, i32, i64, vp, i32, i32, vp, vp],
    }
    def note_failure(result, func, args, lib=lib):   # ctypes errcheck hook: remember WHICH build returned the error
        if result < 0:
            _failed.append(lib)
        return result
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
        fn.errcheck = note_failure
    _libs[half] = lib
    return lib


_failed = []   # libraries whose last call returned an error status, newest last


def check(rc: int, what: str) -> int:
    """Raise A2PError with the message of the library build that made the failing call (both builds keep their own
    `a2p_last_error` string, which is never cleared: joining them would report a stale message of the other build)."""
    if rc < 0:
        lib = _failed.pop() if _failed else None
        del _failed[:]
        msg = lib.a2p_last_error() if lib is not None else b"; ".join(l.a2p_last_error() for l in _libs.values() if l.a2p_last_error())
        raise A2PError(f"{what} failed ({rc}): {(msg or b'').decode()}")
    return rc


# the environment switches a context caches (csrc/a2p_lib.hip A2POpts); changing one after context creation takes a2p_reload_env
ENV_SWITCHES = ("A2P_KV_CACHED", "A2P_NO_CHAIN", "A2P_CHAIN_NW", "A2P_CHAIN_MT", "A2P_CHAIN_TUNE", "A2P_CHAIN_NO_MIX", "A2P_TUNE_VERBOSE",
                "A2P_SIDE_JOIN", "A2P_CHAIN_X_ROWMAJOR", "A2P_NO_SIDE_STREAM", "A2P_SIDE_EARLY_JOIN", "A2P_NO_SHARED_HALF", "A2P_NO_SMALL", "A2P_CHAIN_ROWS",
                "A2P_NO_KSPLIT", "A2P_ATTN_KSPLIT", "A2P_NO_FUSED_KF", "A2P_NO_FUSED_FINAL", "A2P_NO_FUSED_IN", "A2P_TIME_TABLE", "A2P_KSPLIT_NW", "A2P_KSPLIT_QT", "A2P_GRAPH", "A2P_CHAIN_V", "A2P_ATTN2", "A2P_ATTN3")


def env_signature():
    env = os.environ
    return tuple(env.get(k) for k in ENV_SWITCHES)


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None passes NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None) -> int:
    """HIP stream handle torch is enqueuing on for `device` (a tensor's device; default: the current device)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def on_device_of(t):
    """Context manager: make `t.device` the current HIP device for the library call (allocations and launches of the
    C ABI go to the *current* device; a tensor on cuda:1 with cuda:0 current would otherwise launch on the wrong GPU)."""
    import torch
    return torch.cuda.device(t.device)


def content_key(*tensors):
    """Cache key for "these tensors have not changed since I last looked".  Address + version + geometry per tensor.
    The caller MUST also keep strong references to the keyed tensors for as long as it keeps the key: while they are
    alive the caching allocator cannot hand their address to another tensor, so an equal key then means the same bytes
    (a freed-and-reallocated buffer of the same shape at the same address with `_version` 0 would otherwise be a false
    hit -- the stale-conditioning hazard of round 1)."""
    return tuple(None if t is None else (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), str(t.dtype), str(t.device))
                 for t in tensors)


def require_gpu_tensor(t, name: str):
    if not t.is_cuda:
        raise A2PError(f"{name} must live on the MI355X (got device {t.device}); the hot path has no CPU implementation")
