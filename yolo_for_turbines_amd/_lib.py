"""ctypes binding of ``libyolo_mi355x.so`` (declared in ``include/yolo_mi355x.h``).

There is no CPU fallback: if the HIP library has not been built (``__graft_entry__.build()``
or ``make -C yolo_for_turbines_amd/csrc``) every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YOLO_MI355X_LIB") or os.path.join(_HERE, "libyolo_mi355x.so")

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_LEAKY, ACT_MISH = 0, 1, 2
OUT_NHWC, OUT_UPSAMPLE2X, OUT_HEAD = 0, 1, 2
FLAG_RESIDUAL, FLAG_NANCHECK = 1, 2


class PackItem(C.Structure):
    """`yolo_pack_item` (include/yolo_mi355x.h)."""
    _fields_ = [("w_oihw", C.c_void_p), ("w_packed", C.c_void_p), ("cout", C.c_int), ("cin", C.c_int), ("ksize", C.c_int),
                ("reserved", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "h", "w", "cin", "cout", "ksize", "stride", "x_ld", "x_off", "y_ld", "y_off", "r_ld", "r_off",
        "act", "out_mode", "dtype", "flags", "tile")]


class ConvOp(C.Structure):
    _fields_ = [("d", ConvDesc)] + [(n, C.c_uint64) for n in ("x", "w_packed", "scale", "shift", "residual", "y",
                                                              "workspace", "workspace_bytes")]


CALL_MAX_ARGS = 24


class Call(C.Structure):
    """`yolo_call` (include/yolo_mi355x.h): one recorded launch of a train-step table."""
    _fields_ = [("fn", C.c_int32), ("reserved", C.c_int32), ("a", C.c_uint64 * CALL_MAX_ARGS)]


class Reloc(C.Structure):
    """`yolo_reloc`: a[arg] of calls[call] = slots[slot] + offset at run time."""
    _fields_ = [("call", C.c_int32), ("arg", C.c_int32), ("slot", C.c_int32), ("reserved", C.c_int32), ("offset", C.c_int64)]


# YOLO_FN_* of the header, by exported name
FN_IDS = {name: i + 1 for i, name in enumerate((
    "yolo_fill_zero", "yolo_copy_d2d", "yolo_nchw_to_nhwc", "yolo_stem_fwd", "yolo_conv_fwd", "yolo_bn_stats", "yolo_bn_act_fwd",
    "yolo_bn_act_bwd", "yolo_upsample2x_bwd", "yolo_conv_wgrad", "yolo_pack_weights_dgrad", "yolo_pack_weights_batch",
    "yolo_conv_dgrad_s2", "yolo_head_grad_to_nhwc", "yolo_conv_fwd_stats", "yolo_bn_stats_from_partials",
    "yolo_conv_dgrad_bstats", "yolo_bn_act_bwd_rows", "yolo_conv_fwd_ws"))}


class YoloLibError(RuntimeError):
    pass


_SIGS = {
    "yolo_last_error": (C.c_char_p, []),
    "yolo_version": (C.c_int, []),
    "yolo_packed_weight_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "yolo_packed_weight_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "yolo_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yolo_unpack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yolo_bn_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_void_p]),
    "yolo_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    "yolo_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "yolo_fill_zero": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_copy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_run_calls": (C.c_int, [C.POINTER(Call), C.c_int, C.POINTER(Reloc), C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]),
    "yolo_train_fwd_batch": (C.c_int, [C.POINTER(Call), C.c_int, C.POINTER(Reloc), C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]),
    "yolo_train_bwd_batch": (C.c_int, [C.POINTER(Call), C.c_int, C.POINTER(Reloc), C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]),
    "yolo_sgd_chunk_elems": (C.c_int, []),
    "yolo_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "yolo_sgd_step_hp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "yolo_stem_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "yolo_stem_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_stem_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "yolo_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "yolo_conv_fwd_batch": (C.c_int, [C.POINTER(ConvOp), C.c_int, C.c_void_p, C.c_void_p]),
    "yolo_conv_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "yolo_conv_fwd_ws": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "yolo_conv_pick_tile": (C.c_int, [C.POINTER(ConvDesc)]),
    "yolo_conv_num_tiles": (C.c_int, []),
    "yolo_bn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "yolo_bn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                C.c_size_t, C.c_void_p]),
    "yolo_conv_stats_rows": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
    "yolo_conv_fwd_stats": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_bn_stats_from_partials": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "yolo_bn_act_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "yolo_bn_act_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_conv_bstats_rows": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
    "yolo_conv_dgrad_bstats": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_bn_act_bwd_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "yolo_upsample2x_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "yolo_wgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 8),
    "yolo_conv_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_debug_tr_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_packed_dgrad_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "yolo_pack_weights_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yolo_pack_weights_dgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yolo_conv_dgrad_s2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yolo_head_grad_to_nhwc": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]),
    "yolo_letterbox": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    "yolo_build_targets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "yolo_map_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "yolo_accuracy_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                       C.c_void_p]),
    "yolo_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "yolo_loss_fwd": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_loss_bwd": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "yolo_decode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "yolo_decode3": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                               C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_decode3_ex": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                  C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_nms_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "yolo_sort_u64_workspace_bytes": (C.c_size_t, [C.c_int]),
    "yolo_sort_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_size_t, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """The loaded library; raises YoloLibError when it is missing (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YoloLibError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C yolo_for_turbines_amd/csrc`. This package has no CPU fallback.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:                                  # pragma: no cover
            raise YoloLibError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().yolo_last_error()
        raise YoloLibError(f"{what or 'libyolo_mi355x'} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
