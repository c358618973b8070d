"""ctypes binding of libagan_hip.so (C ABI declared in include/agan.h).

The product path has no CPU fallback: if the shared library is missing or the tensors are not on an
MI355X the calls raise.  Only raw device pointers, sizes and the current HIP stream cross the boundary.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGAN_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libagan_hip.so")   # AGAN_LIB: kernel A/B builds

# enums of include/agan.h
PREC_F32, PREC_BF16, PREC_BF16X3, PREC_F16, PREC_BF16X6, PREC_F16X3 = 0, 1, 2, 3, 4, 5
PRECISIONS = {"f32": PREC_F32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "f16": PREC_F16, "bf16x6": PREC_BF16X6, "f16x3": PREC_F16X3}
PACK_FWD, PACK_DGRAD_S1, PACK_DGRAD_4x4S2, PACK_UP_FWD, PACK_UP_DGRAD = 0, 1, 2, 3, 4
ACT_NONE, ACT_GLU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
COMM_ID_BYTES = 128    # include/agan.h: AGAN_COMM_ID_BYTES
DT_F32, DT_BF16, DT_F16 = 0, 1, 2          # include/agan.h: AGAN_DT_* (storage type of an activation tensor)
AMAX_SLOT = 256        # floats per amax slot (include/agan.h: AGAN_AMAX_SLOT)
ABI_VERSION = 101      # include/agan.h: AGAN_VERSION this binding was written against (argument lists changed between versions)


class ConvGeom(Structure):
    """struct agan_conv_geom."""
    _fields_ = [("B", c_int32), ("Cin", c_int32), ("IH", c_int32), ("IW", c_int32),
                ("Cout", c_int32), ("OH", c_int32), ("OW", c_int32), ("R", c_int32), ("S", c_int32),
                ("OS", c_int32), ("SY", c_int32), ("DY", c_int32), ("OY", c_int32 * 2)]


_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "agan_version": (c_int, []),
    "agan_last_error": (c_char_p, []),
    "agan_packed_weight_bytes": (c_size_t, [c_int] * 6),
    "agan_timer_create": (c_int, [_P]),
    "agan_timer_destroy": (c_int, [_P]),
    "agan_timer_arm": (c_int, [_P, _P]),
    "agan_timer_elapsed_ms": (c_int, [_P, _P, _P]),
    "agan_timer_last_kernel": (c_int, [_P, c_size_t]),
    "agan_comm_unique_id": (c_int, [_P]),
    "agan_comm_init": (c_int, [_P, c_int, c_int, _P]),
    "agan_comm_destroy": (c_int, [_P]),
    "agan_allreduce_bucket": (c_int, [_P, _P, c_size_t, _P]),
    "agan_allreduce_chunk_elems": (c_size_t, [c_size_t, c_int]),
    "agan_allreduce_scratch_bytes": (c_size_t, [c_size_t, c_int, c_int]),
    "agan_allreduce_bucket_dt": (c_int, [_P, _P, c_size_t, c_int, _P, c_size_t, _P]),
    "agan_exchange_wire_elems": (c_size_t, [c_size_t, c_int]),
    "agan_exchange_pack_bf16": (c_int, [_P, _P, c_size_t, c_size_t, _P]),
    "agan_exchange_sum_bf16": (c_int, [_P, c_int, c_size_t, _P, _P]),
    "agan_exchange_unpack_bf16": (c_int, [_P, _P, c_size_t, _P]),
    "agan_pack_job_blocks": (c_int, [c_int] * 5),
    "agan_pack_job_blocks_prec": (c_int, [c_int] * 6),
    "agan_pack_weights": (c_int, [_P, c_int, c_int, c_int, _P]),
    "agan_pack_weight": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "agan_conv_effective_prec": (c_int, [POINTER(ConvGeom), c_int]),
    "agan_conv_wgrad_effective_prec": (c_int, [POINTER(ConvGeom), c_int, c_int]),
    "agan_conv_executed_fraction": (c_double, [POINTER(ConvGeom), c_int, c_int, c_int]),
    "agan_conv_gather_ws_bytes": (c_size_t, [POINTER(ConvGeom), c_int]),
    "agan_conv_ktable_elems": (c_size_t, [POINTER(ConvGeom)]),
    "agan_conv_ktable": (c_int, [POINTER(ConvGeom), _P, _P]),
    "agan_conv_gather": (c_int, [_P, _P, _P, _P, POINTER(ConvGeom), _P, c_int, c_int, _P, _P, c_size_t, _P, _P, _P]),
    "agan_conv_gather_dt": (c_int, [_P, _P, _P, _P, POINTER(ConvGeom), _P, c_int, c_int, _P, _P, c_size_t, _P, _P, _P, c_int, c_int]),
    "agan_conv_gather_dt_supported": (c_int, [POINTER(ConvGeom), c_int, c_int, c_int]),
    "agan_conv_wgrad_dt_supported": (c_int, [POINTER(ConvGeom), c_int, c_int, c_int, c_int]),
    "agan_conv_wgrad_dt": (c_int, [_P, _P, _P, POINTER(ConvGeom), _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P, c_int, c_int]),
    "agan_bn_stats_dt": (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_float, _P, c_size_t, _P, c_int]),
    "agan_bn_act_fwd_dt": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int]),
    "agan_bn_train_fwd_dt": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, c_int, _P, c_size_t, _P, _P, c_int, c_int]),
    "agan_bn_act_bwd_dt": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P, c_int, c_int]),
    "agan_absmax": (c_int, [_P, c_size_t, _P, _P]),
    "agan_conv_wgrad_ws_bytes": (c_size_t, [POINTER(ConvGeom)]),
    "agan_conv_wgrad": (c_int, [_P, _P, _P, POINTER(ConvGeom), _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "agan_bias_grad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "agan_bn_stats_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "agan_bn_stats": (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_float, _P, c_size_t, _P]),
    "agan_bn_train_fwd_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "agan_bn_train_fwd": (c_int, [_P] * 10 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int, _P, c_size_t, _P, _P]),
    "agan_bn_act_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "agan_bn_act_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "agan_bn_act_bwd": (c_int, [_P] * 9 + [c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "agan_act_fwd": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "agan_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "agan_glu_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "agan_glu_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "agan_attn_fwd": (c_int, [_P, _P, _P, _P, c_float, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "agan_attn_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "agan_attn_bwd": (c_int, [_P] * 7 + [c_float, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "agan_attn_fwd_dt": (c_int, [_P, _P, _P, _P, c_float, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int]),
    "agan_attn_bwd_dt": (c_int, [_P] * 7 + [c_float, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, c_int]),
    "agan_func_attention_fwd": (c_int, [_P, _P, c_float, c_float, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "agan_func_attention_bwd": (c_int, [_P, _P, _P, _P, c_float, c_float, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "agan_words_loss_save_elems": (c_size_t, [c_int, c_int, c_int, c_int]),
    "agan_words_loss_fwd": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "agan_words_loss_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "agan_words_loss_bwd": (c_int, [_P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "agan_sent_loss_fwd": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, _P, _P, c_int, c_int, _P]),
    "agan_sent_loss_bwd": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, _P, _P, c_int, c_int, _P]),
    "agan_disc_loss": (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    "agan_gen_loss": (c_int, [_P, _P, _P, c_int, _P]),
    "agan_kl_loss": (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    "agan_reparam_fwd": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "agan_reparam_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    "agan_adam_step": (c_int, [_P, _P, _P, _P, c_size_t, _P, c_double, c_double, c_double, c_double, c_float, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class AganError(RuntimeError):
    pass


def load():
    """dlopen libagan_hip.so and attach prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AganError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(or `make -C attention-gan_amd/csrc`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.agan_version() != ABI_VERSION:
        raise AganError(f"{LIB_PATH} reports C-ABI version {lib.agan_version()}, this binding is written against {ABI_VERSION} "
                        f"(include/agan.h: argument lists differ between versions) -- rebuild with `make -C attention-gan_amd/csrc`")
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().agan_last_error()
        raise AganError(f"{what or 'agan'} failed ({code}): {msg.decode() if msg else ''}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
