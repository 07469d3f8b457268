"""ctypes binding of libast_hip.so (include/ast_hip.h).

The product path has NO fallback: importing an op without the built library, or
calling one without a GPU, raises.  Build with `python __graft_entry__.py` (or
`make -C audio-style-transfer_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AST_HIP_LIB") or os.path.join(_HERE, "libast_hip.so")   # AST_HIP_LIB: same-box A/B of two builds (tools/ab.sh)

F32, BF16 = 0, 1
MAX_TAPS = 9

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class Gather(C.Structure):
    """ast_gather_t"""
    _fields_ = [(n, i32) for n in ("N", "Hs", "Ws", "Cs", "Hm", "Wm", "sh", "sw", "oh", "ow", "Hd", "Wd", "Cd",
                                   "dsh", "dsw", "doh", "dow", "ntaps", "wtaps")] + [("tap", i32 * MAX_TAPS)]


class WeightDesc(C.Structure):
    """ast_weight_desc_t"""
    _fields_ = [("w", vp), ("u", vp), ("v", vp), ("sigma", vp), ("scratch", vp), ("wf", vp), ("wb", vp)] + \
               [(n, i32) for n in ("Co", "Ci", "KK", "s_co", "s_ci", "Cop", "Cip", "power_iter")] + \
               [("dwp", vp), ("grad", vp), ("inner", vp), ("dwp_from_wb", i32), ("dwp_replicas", i32)]


class LinWg(C.Structure):
    """record of ast_linear_wgrad_batched"""
    _fields_ = [("dy", vp), ("x", vp), ("dW", vp), ("db", vp)] + [(n, i32) for n in ("M", "N", "K", "lddy", "ldw", "p0", "p1", "p2")]


_SIGS = {
    "ast_version": ([], i32),
    "ast_igemm": ([vp, vp, vp, vp, C.POINTER(Gather), i32, i32, vp, C.c_long, vp], i32),
    "ast_igemm_plan": ([C.POINTER(Gather), i32, C.POINTER(i32 * 5)], i32),
    "ast_igemm_ws_floats": ([C.POINTER(Gather), i32], C.c_long),
    "ast_wgrad": ([vp, vp, vp, C.POINTER(Gather), i32, vp], i32),
    "ast_skinny_gemm": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_skinny_gemm_ex": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, f32, C.c_uint64, vp, vp], i32),
    "ast_bigk_gemm": ([vp, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "ast_bign_dgrad": ([vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "ast_linear_wgrad": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "ast_linear_wgrad_batched": ([vp, i32, i32, vp], i32),
    "ast_linear_wgrad_batched_host": ([vp, i32, i32, vp], i32),
    "ast_nchw_to_nhwc": ([vp, vp, i32, i32, i32, i32, i64, i64, i64, i32, i32, vp], i32),
    "ast_nhwc_to_nchw": ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_cast": ([vp, i32, vp, i32, i64, vp], i32),
    "ast_weights_prepare_t": ([vp, vp, i32, i32, i32, vp, i32, vp], i32),
    "ast_sn_scratch_floats": ([i32, i32], C.c_long),
    "ast_weight_grads_flush_t": ([vp, vp, i32, vp], i32),
    "ast_dropout_fwd": ([vp, vp, vp, i64, f32, C.c_uint64, vp, vp], i32),
    "ast_weight_grad_unpack": ([vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp], i32),
    "ast_chan_stats": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "ast_norm_finalize": ([vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, f32, vp, vp, vp, vp, C.c_long, vp], i32),
    "ast_affine_act": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "ast_bn_apply_fwd": ([vp, vp, vp, vp, i32, C.c_long, vp, vp, vp, vp, vp, vp, f32, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_bn_apply_bwd": ([vp, vp, vp, vp, vp, vp, i32, C.c_long, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_norm_bwd_sums": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_norm_bwd_sums_pre": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp], i32),
    "ast_norm_bwd_apply_pre": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp], i32),
    "ast_norm_bwd_finalize": ([vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    "ast_norm_bwd_finalize_n": ([vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_long, vp], i32),
    "ast_igemm_bn": ([vp, vp, vp, vp, C.POINTER(Gather), i32, i32, vp, C.c_long, vp, vp, vp, vp], i32),
    "ast_norm_bwd_apply": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "ast_layernorm_fwd": ([vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp], i32),
    "ast_layernorm_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "ast_add_drop_ln_fwd": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, C.c_uint64, vp, vp], i32),
    "ast_add_drop_ln_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp], i32),
    "ast_adaptive_pool_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_adaptive_pool_bwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_bilinear_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_bilinear_bwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_attn_fwd": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp], i32),
    "ast_attn_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp], i32),
    "ast_rowmix": ([vp, vp, vp, i32, i32, i32, vp], i32),
    "ast_attn_fwd_p": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, f32, C.c_uint64, vp, vp], i32),
    "ast_attn_bwd_p": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, f32, C.c_uint64, vp, vp], i32),
    "ast_add": ([vp, vp, vp, i64, i32, vp], i32),
    "ast_relu_bwd": ([vp, vp, vp, i64, i32, vp], i32),
    "ast_dropout_mask": ([vp, i64, f32, C.c_uint64, vp, vp], i32),
    "ast_mul": ([vp, vp, vp, i64, i32, vp], i32),
    "ast_recon_loss": ([vp, vp, i64, i32, i32, i32, i32, f32, f32, f32, f32, f32, vp, vp, vp], i32),
    "ast_recon_loss_total": ([vp, vp, i64, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp, vp, vp], i32),
    "ast_infonce": ([vp, vp, i32, i32, f32, vp, vp, vp, vp], i32),
    "ast_margin": ([vp, i32, i32, f32, vp, vp, vp], i32),
    "ast_hsic": ([vp, vp, i32, i32, vp, vp, vp, vp, vp], i32),
    "ast_crosscov": ([vp, vp, i32, i32, vp, vp, vp, vp, vp], i32),
    "ast_istft": ([vp, i32, i32, vp, vp, vp], i32),
    "ast_bin_stats_acc": ([vp, vp, vp, i32, i32, i32, vp], i32),
    "ast_zscore": ([vp, vp, vp, vp, i32, i32, i32, f32, vp], i32),
    "ast_sections_overlap_avg": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "ast_cross_entropy": ([vp, vp, i32, i32, vp, vp, vp], i32),
    "ast_softmax_entropy": ([vp, i32, i32, vp, vp, vp], i32),
    "ast_scale": ([vp, vp, f32, vp, i64, i32, vp], i32),
    "ast_colsum_acc": ([vp, i64, i32, i32, vp, i32, vp], i32),
    "ast_sumsq": ([vp, i64, vp, vp], i32),
    "ast_adam": ([vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp, f32, vp], i32),
    "ast_adam_dev": ([vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp, vp, vp], i32),
    "ast_set_values": ([vp, C.POINTER(f32), i32, vp], i32),
    "ast_weighted_sum": ([C.POINTER(vp), C.POINTER(i32), i32, vp, vp, vp], i32),
    "ast_weighted_sum_bwd": ([vp, C.POINTER(i32), i32, vp, vp, vp], i32),
    "ast_counter_incr": ([vp, vp], i32),
    "ast_stft_sections": ([vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "ast_cqt_octaves": ([vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, vp], i32),
    "ast_cqt_sections": ([vp, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "ast_resample_poly": ([vp, i32, i32, vp, i32, i32, i32, i32, vp, i32, f32, vp], i32),
    "ast_wgrad_rep": ([vp, vp, vp, C.POINTER(Gather), i32, i32, vp], i32),
    "ast_wgrad_slab": ([vp, vp, vp, C.POINTER(Gather), i32, i32, C.POINTER(i32), vp], i32),
    "ast_slab_sum": ([C.POINTER(vp), C.POINTER(i64), C.POINTER(i32), i32, vp], i32),
    "ast_tok_max_ops": ([], i32),
    "ast_tok_program": ([vp, i32, i32, i32, vp, vp, vp, vp], i32),
}

EXPORTS = tuple(_SIGS) + ("ast_last_error",)

_lib = None


# ---- optional per-call profiling (bench.py's roofline.kernels[]): when PROFILE_CALLS is a list, every C-ABI call of the
# functions named in CALL_BYTES is bracketed by device events on the current stream and recorded as
# (name, algorithmic bytes, start event, end event).  Off (None) in normal operation: lib() returns the raw CDLL.
PROFILE_CALLS = None
NEXT_BYTES = None          # set by a caller whose byte count is not derivable from the arguments (weight pack / flush)


def _es(dtype_code):
    return 2 if dtype_code == BF16 else 4


CALL_BYTES = {
    # BatchNorm/InstanceNorm apply (+ReLU, + ResBlock add): read x (+ shortcut), write y
    "ast_affine_act": lambda a: a[7] * a[8] * a[9] * _es(a[11]) * (2 + (1 if a[3] else 0)),
    # norm backward apply: read dy, x (+ shortcut input), write dx (+ shortcut gradient)
    "ast_norm_bwd_apply_pre": lambda a: a[8] * a[9] * a[10] * _es(a[12]) * (3 + (2 if a[3] else 0)),
    "ast_norm_bwd_sums_pre": lambda a: a[5] * a[6] * a[7] * _es(a[9]) * (2 + (1 if a[3] else 0)),
    # the same two passes with the statistics finalize folded in (ast_bn_apply_fwd / _bwd)
    "ast_bn_apply_fwd": lambda a: a[18] * a[19] * a[20] * _es(a[23]) * (2 + (1 if a[1] else 0)),
    "ast_bn_apply_bwd": lambda a: a[22] * a[23] * a[24] * _es(a[27]) * (3 + (2 if a[2] else 0)),
    "ast_chan_stats": lambda a: a[2] * a[3] * a[4] * _es(a[5]),
    # compute_comprehensive_loss: read output and target, write the gradient (f32)
    "ast_recon_loss": lambda a: 3 * 4 * a[3] * a[4] * 2 * a[5] * a[6],
    "ast_recon_loss_total": lambda a: 3 * 4 * a[3] * a[4] * 2 * a[5] * a[6],
    "ast_adam": lambda a: 7 * 4 * a[4],                    # read p, g, m, v; write p, m, v
    "ast_sumsq": lambda a: 4 * a[1],
    "ast_weights_prepare_t": None,                         # bytes from NEXT_BYTES (WeightBank)
    "ast_weight_grads_flush_t": None,
    "ast_nchw_to_nhwc": lambda a: a[2] * a[4] * a[5] * (a[3] * 4 + a[9] * _es(a[10])),
    "ast_bilinear_fwd": lambda a: a[0] and a[2] * (a[5] * a[6] * a[4] * _es(a[9]) + a[3] * a[7] * a[8] * 4),
}


class _ProfiledLib:
    def __init__(self, raw):
        self._raw = raw

    def __getattr__(self, name):
        fn = getattr(self._raw, name)
        if name not in CALL_BYTES:
            return fn

        def wrapped(*args):
            global NEXT_BYTES
            f = CALL_BYTES[name]
            nbytes = NEXT_BYTES if f is None else f(args)
            NEXT_BYTES = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            PROFILE_CALLS.append((name, float(nbytes or 0), e0, e1))
            return rc
        return wrapped


def lib():
    """The loaded library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None and PROFILE_CALLS is not None:
        return _ProfiledLib(_lib)
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP library has not been built (run `python __graft_entry__.py` "
                "or `make -C audio-style-transfer_amd/csrc`). There is no CPU/PyTorch fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes, fn.restype = args, res
        L.ast_last_error.argtypes, L.ast_last_error.restype = [], C.c_char_p
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"libast_hip {what} failed ({rc}): {lib().ast_last_error().decode()}")


def dcode(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {dtype}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the
    kernels would fault on them."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libast_hip needs device tensors (got a CPU tensor); there is no CPU path")
    return t.data_ptr()
