"""ctypes binding of ``liblnn_hip.so`` (the C-ABI declared in ``include/lnn_hip.h``).

There is NO fallback: if the library is missing this module raises at import of :func:`lib`, and
every op in the package goes through it.  Status codes are mapped to ``RuntimeError`` with the
library's thread-local message, mirroring the reference's assert/exception behaviour
(e.g. multihead/nnUNetTrainerMultiHead.py:116,140).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LNN_LIB_PATH: another build of the same library (A/B measurements of two kernel versions on one box); never a fallback
LIB_PATH = os.environ.get("LNN_LIB_PATH") or os.path.join(_HERE, "csrc", "liblnn_hip.so")

_p = C.c_void_p
_i = C.c_int
_l = C.c_long
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list EVERY symbol include/lnn_hip.h declares (checked by tests)
SIGNATURES = {
    "lnn_last_error": (C.c_char_p, []),
    "lnn_version": (_i, []),
    "lnn_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "lnn_pack_weights": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l]),
    "lnn_packed_weight_elems": (_sz, [_i, _i, _i]),
    "lnn_conv3d_fwd": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_conv3d_dgrad": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_conv3d_wgrad": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_conv3d_fwd_cat": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_conv3d_fwd_in_stats": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _l]),
    "lnn_conv3d_dgrad_ws": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l]),
    "lnn_conv3d_dgrad_cat": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_conv3d_dgrad_cat_ws": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l]),
    "lnn_conv3d_wgrad_cat": (_i, [_p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    "lnn_convT3d_k2s2_fwd": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_convT3d_k2s2_dgrad": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_convT3d_k2s2_wgrad": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    "lnn_conv3d_wgrad_det": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _l]),
    "lnn_conv3d_wgrad_cat_det": (_i, [_p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _l]),
    "lnn_convT3d_k2s2_wgrad_det": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _l]),
    "lnn_unpack_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _f, _i]),
    "lnn_pack_weights_batched": (_i, [_p, _p, _p, _p, _i, _l]),
    "lnn_unpack_wgrad_batched": (_i, [_p, _p, _p, _p, _i, _l, _f, _i]),
    "lnn_wgrad_panel_elems": (_sz, [_i, _i, _i]),
    "lnn_instnorm_stats": (_i, [_p, _p, _i, _l, _i, _f, _p, _p, _p]),
    "lnn_instnorm_lrelu_fwd": (_i, [_p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f]),
    "lnn_instnorm_lrelu_seg_fwd": (_i, [_p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f, _p, _p, _i]),
    "lnn_instnorm_lrelu_bwd": (_i, [_p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f, _p, _p, _p, _f, _p]),
    "lnn_instnorm_ws_doubles": (_sz, [_i, _i]),
    "lnn_instnorm_lrelu_seg_bwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f, _p, _p, _f, _p]),
    "lnn_instnorm_lrelu_seg_bwd_ws_doubles": (_sz, [_i, _i]),
    "lnn_instnorm_lrelu_bwd_sums": (_i, [_p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f, _p, _p, _f, _p]),
    "lnn_conv3d_dgrad_in_bwd_sums": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _f, _p, _p, _l]),
    "lnn_instnorm_lrelu_bwd_apply": (_i, [_p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f, _p]),
    "lnn_instnorm_small_volume": (_i, []),
    "lnn_conv3d_fwd_in_lrelu": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _f, _p, _i, _p, _p, _l]),
    "lnn_conv3d_dgrad_in_bwd": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _f, _p, _p, _l]),
    "lnn_conv3d_wgrad_c1_in_bwd": (_i, [_p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _f, _p, _p, _l]),
    "lnn_seg1x1_fwd": (_i, [_p, _p, _i, _p, _p, _i, _l, _i, _i]),
    "lnn_seg1x1_bwd": (_i, [_p, _p, _i, _p, _p, _p, _i, _p, _i, _l, _i, _i, _i, _f, _p]),
    "lnn_seg1x1_bwd_ws_floats": (_sz, [_i, _i]),
    "lnn_dice_ce_fwd": (_i, [_p, _p, _p, _i, _i, _l, _i, _f, _p, _p]),
    "lnn_dice_ce_fwd_ds": (_i, [_p, _p, _p, _i, _i, _l, _i, _f, _p, _p, _f, _p, _i]),
    "lnn_dice_ce_bwd": (_i, [_p, _p, _p, _i, _i, _l, _i, _f, _p, _f, _p, _f, _p]),
    "lnn_dice_ce_loss_from_totals": (_i, [_p, _p, _i, _i, _l, _i, _f, _p]),
    "lnn_dice_ce_ws_doubles": (_sz, [_i, _i]),
    "lnn_online_dice_counts": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "lnn_kl_logits": (_i, [_p, _p, _p, _i, _i, _l, _f, _p, _p]),
    "lnn_kl_logits_ws_doubles": (_sz, [_i]),
    "lnn_plop_pseudo_labels": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "lnn_local_pod": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _i, _f, _i, _p, _p, _p]),
    "lnn_ewc_penalty_fwd": (_i, [_p, _p, _p, _p, _l, _f, _p, _p]),
    "lnn_ewc_penalty_bwd": (_i, [_p, _p, _p, _p, _l, _f, _f, _p, _p]),
    "lnn_fisher_square": (_i, [_p, _p, _p, _l, _f]),
    "lnn_fisher_accumulate": (_i, [_p, _p, _p, _l, _f, _f]),
    "lnn_fisher_ema": (_i, [_p, _p, _p, _l, _f, _f]),
    "lnn_rw_update": (_i, [_p, _p, _p, _p, _p, _p, _l, _f, _f, _p, _f, _f, _i]),
    "lnn_target_ce_fwd": (_i, [_p, _p, _p, _i, _i, _i, _l, _f, _i, _f, _p, _p]),
    "lnn_target_ce_bwd": (_i, [_p, _p, _p, _i, _i, _i, _l, _f, _i, _f, _p, _f, _p, _p]),
    "lnn_softmax_accumulate": (_i, [_p, _p, _p, _p, _p] + [_i] * 11 + [_f, _i]),
    "lnn_softmax_finalize": (_i, [_p, _p, _p, _i, _l, _p]),
    "lnn_gradnorm_sumsq": (_i, [_p, _p, _l, _f, _p, _i]),
    "lnn_flat_reduce_ws_doubles": (_sz, []),
    "lnn_sgd_nesterov_step_clipped": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p]),
    "lnn_sgd_nesterov_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _i]),
    "lnn_cast_f32_to_h": (_i, [_p, _p, _p, _l]),
    "lnn_f32_conv3d_fwd": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_f32_conv3d_dgrad": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_f32_conv3d_wgrad": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_f32_convT3d_k2s2_fwd": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_f32_convT3d_k2s2_dgrad": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i]),
    "lnn_f32_convT3d_k2s2_wgrad": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    "lnn_f32_conv3d_fwd_g": (_i, [_p, _p, _i, _p, _p, _p, _i] + [_i] * 12),
    "lnn_f32_conv3d_dgrad_g": (_i, [_p, _p, _i, _p, _p, _i] + [_i] * 13),
    "lnn_f32_conv3d_wgrad_g": (_i, [_p, _p, _i, _p, _i, _p] + [_i] * 12),
    "lnn_f32_convT3d_fwd_g": (_i, [_p, _p, _i, _p, _p, _i] + [_i] * 9),
    "lnn_f32_convT3d_dgrad_g": (_i, [_p, _p, _i, _p, _p, _i] + [_i] * 10),
    "lnn_f32_convT3d_wgrad_g": (_i, [_p, _p, _i, _p, _i, _p] + [_i] * 9),
    "lnn_image_to_cl_h": (_i, [_p, _p, _p, _i, _i, _l, _i]),
    "lnn_f32_image_to_cl": (_i, [_p, _p, _p, _i, _i, _l, _i]),
    "lnn_f32_instnorm_lrelu_fwd": (_i, [_p, _p, _i, _p, _i, _i, _l, _i, _f, _p, _p, _p, _p, _f]),
    "lnn_f32_instnorm_lrelu_bwd": (_i, [_p, _p, _i, _p, _i, _i, _l, _i, _p, _p, _p, _p, _f, _p, _p, _p]),
    "lnn_f32_seg1x1_fwd": (_i, [_p, _p, _i, _p, _p, _i, _l, _i, _i]),
    "lnn_f32_seg1x1_bwd": (_i, [_p, _p, _i, _p, _p, _p, _i, _p, _i, _l, _i, _i, _i]),
    "lnn_debug_tr16_probe": (_i, [_p, _p]),
    "lnn_debug_set_phase_buffer": (_i, [_p]),
    "lnn_debug_force_conv_kernel": (_i, [_i]),
    "lnn_debug_force_down2_kernel": (_i, [_i]),
    "lnn_debug_set_v9_zseg": (_i, [_i]),
    "lnn_debug_set_k133_v9": (_i, [_i]),
    "lnn_debug_last_k133_on_v9": (_i, []),
    "lnn_debug_set_gen_mode": (_i, [_i]),
    "lnn_debug_last_dgrad_reduce_fused": (_i, []),
    "lnn_debug_set_cu_budget": (_i, [_i]),
    # generic geometry (kernel extent 1 / 3 and stride 1 / 2 per axis; transposed convolutions with kernel == stride)
    "lnn_conv3d_fwd_g": (_i, [_p, _p, _i, _p, _p, _p, _i] + [_i] * 12 + [_p, _l]),
    "lnn_conv3d_dgrad_g": (_i, [_p, _p, _i, _p, _p, _i] + [_i] * 13 + [_p, _l]),
    "lnn_conv3d_wgrad_g": (_i, [_p, _p, _i, _p, _i, _p] + [_i] * 12 + [_p, _l]),
    "lnn_convT3d_fwd_g": (_i, [_p, _p, _i, _p, _p, _i] + [_i] * 9 + [_p, _l]),
    "lnn_convT3d_dgrad_g": (_i, [_p, _p, _i, _p, _p, _i] + [_i] * 10 + [_p, _l]),
    "lnn_convT3d_wgrad_g": (_i, [_p, _p, _i, _p, _i, _p] + [_i] * 9 + [_p, _l]),
    "lnn_convT3d_k2s2_fwd_ws": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _l]),
    "lnn_convT3d_k2s2_dgrad_ws": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l]),
}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises NativeLibraryMissing when not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(make -C lifelong-nnunet_amd/csrc).  There is no CPU / PyTorch fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def _ptr(t):
    if t is None:
        return None
    return t.data_ptr() if hasattr(t, "data_ptr") else int(t)


def stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream


_fn_cache = {}


def call(name, *args):
    """Invoke ``lnn_<name>`` on torch's current stream; tensors (anything with ``data_ptr``) are passed as device pointers.
    Host cost matters: a training step is ~270 calls and must stay ahead of the GPU."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(lib(), name)
    conv = []
    for a in args:
        c = a.__class__
        if c is int or c is float or a is None:
            conv.append(a)
        elif hasattr(a, "data_ptr"):
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    rc = fn(stream_handle(), *conv)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().lnn_last_error().decode()}")


def call_plain(name, *args):
    """Invoke an entry point that takes no stream (launch configuration)."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().lnn_last_error().decode()}")


def query(name, *args):
    return getattr(lib(), name)(*args)


def device_info():
    cu, khz = C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(128)
    rc = lib().lnn_device_info(C.byref(cu), C.byref(khz), buf, 128)
    if rc != 0:
        raise RuntimeError(lib().lnn_last_error().decode())
    return {"cu_count": cu.value, "clock_khz": khz.value, "arch": buf.value.decode()}
