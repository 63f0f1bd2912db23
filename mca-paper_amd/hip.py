"""ctypes binding of libmca_hip.so (C ABI declared in include/mca_hip.h).

The library holds every compute kernel of the step; there is NO fallback: if it is missing or a call
returns an error code, this module raises.  Tensors are passed as raw device pointers
(``tensor.data_ptr()``) plus the current HIP stream handle — PyTorch only owns the memory.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCA_HIP_LIB") or os.path.join(_HERE, "libmca_hip.so")          # MCA_HIP_LIB: A/B against another build of the same ABI

_ERR = {-1: "bad argument", -2: "misaligned pointer / leading dimension", -3: "unsupported size", -4: "launch failed"}


class MCAHipError(RuntimeError):
    pass


class LossTerm(C.Structure):
    _fields_ = [("slot_a", C.c_int32), ("slot_b", C.c_int32), ("and_bits", C.c_uint32), ("or_bits", C.c_uint32)]


class CastDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("lds", C.c_int64), ("rows", C.c_int64), ("cols", C.c_int64),
                ("ldd", C.c_int64), ("rows_pad", C.c_int64), ("cols_pad", C.c_int64), ("transpose", C.c_int32), ("scale", C.c_float)]


MAX_MODALITIES = 16


class TnDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("N", C.c_int64), ("K", C.c_int64)]


class MaskDesc(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("rowmask", C.c_void_p), ("elem_bytes", C.c_int32), ("n", C.c_int32), ("offset", C.c_int32),
                ("pad_", C.c_int32)]


class FiniteArgs(C.Structure):
    _fields_ = [("p", C.c_void_p * MAX_MODALITIES), ("n", C.c_int64 * MAX_MODALITIES), ("count", C.c_int32), ("pad_", C.c_int32)]


class PackMasksArgs(C.Structure):
    _fields_ = [("m", MaskDesc * MAX_MODALITIES), ("n_mod", C.c_int32), ("batch", C.c_int32), ("n_tokens", C.c_int32),
                ("n_fusion", C.c_int32)]


class AttnFwdArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_bstride", C.c_int64), ("q_ld", C.c_int64),
        ("k", C.c_void_p), ("v", C.c_void_p), ("kv_bstride", C.c_int64), ("kv_ld", C.c_int64),
        ("o", C.c_void_p), ("o_bstride", C.c_int64), ("o_ld", C.c_int64),
        ("lse", C.c_void_p), ("qmask", C.c_void_p), ("keyinfo", C.c_void_p), ("ktile_flags", C.c_void_p),
        ("q_ptr", C.c_void_p), ("q_kt", C.c_void_p), ("q_order", C.c_void_p),
        ("vmean", C.c_void_p),
        ("batch", C.c_int), ("heads", C.c_int), ("nq", C.c_int), ("nk", C.c_int), ("nk_pad", C.c_int),
        ("n_qtiles", C.c_int), ("n_ktiles", C.c_int), ("scale", C.c_float), ("flags", C.c_int),
        ("khot", C.c_void_p),
    ]


ATTN_Q_PRESCALED = 1
ATTN_LAZY_REFERENCE = 2          # mca_attn_fwd: lazy softmax reference (include/mca_hip.h)


class AttnFp8Operands(C.Structure):
    _fields_ = [("q8", C.c_void_p), ("qs", C.c_void_p), ("k8", C.c_void_p), ("ks", C.c_void_p), ("v8t", C.c_void_p), ("vs", C.c_void_p),
                ("n_ktiles", C.c_int)]


class AttnFp8BwdOperands(C.Structure):
    _fields_ = [("q8", C.c_void_p), ("qs", C.c_void_p), ("k8", C.c_void_p), ("ks", C.c_void_p), ("v8", C.c_void_p), ("vs", C.c_void_p),
                ("do8", C.c_void_p), ("dos", C.c_void_p), ("n_ktiles", C.c_int)]


class AttnBwd2Args(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_bstride", C.c_int64), ("q_ld", C.c_int64),
        ("k", C.c_void_p), ("v", C.c_void_p), ("kv_bstride", C.c_int64), ("kv_ld", C.c_int64),
        ("d_o", C.c_void_p), ("o_bstride", C.c_int64), ("o_ld", C.c_int64),
        ("lse", C.c_void_p), ("delta", C.c_void_p), ("dvmean", C.c_void_p),
        ("dq", C.c_void_p), ("dq_bstride", C.c_int64), ("dq_ld", C.c_int64), ("dq_f32", C.c_int),
        ("dk", C.c_void_p), ("dv", C.c_void_p), ("dkv_bstride", C.c_int64), ("dkv_ld", C.c_int64),
        ("qmask", C.c_void_p), ("keyinfo", C.c_void_p), ("ktile_flags", C.c_void_p),
        ("q_ptr", C.c_void_p), ("q_kt", C.c_void_p), ("q_order", C.c_void_p), ("n_qtiles128", C.c_int), ("n_ktiles64", C.c_int),
        ("k_wg", C.c_void_p), ("k_qt", C.c_void_p), ("n_qtiles64", C.c_int), ("n_kblocks256", C.c_int),
        ("batch", C.c_int), ("heads", C.c_int), ("nq", C.c_int), ("nk", C.c_int), ("nk_pad", C.c_int),
        ("scale", C.c_float), ("flags", C.c_int),
        ("khot", C.c_void_p), ("qblk", C.c_void_p), ("kblock_keys", C.c_int),
    ]


class AttnBwd1Args(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_bstride", C.c_int64), ("q_ld", C.c_int64),
        ("k", C.c_void_p), ("v", C.c_void_p), ("kv_bstride", C.c_int64), ("kv_ld", C.c_int64),
        ("d_o", C.c_void_p), ("o_bstride", C.c_int64), ("o_ld", C.c_int64),
        ("q_hstride", C.c_int64), ("o_hstride", C.c_int64),
        ("rowc", C.c_void_p), ("dvmean", C.c_void_p),
        ("dq", C.c_void_p), ("dq_bstride", C.c_int64), ("dq_ld", C.c_int64),
        ("dk", C.c_void_p), ("dv", C.c_void_p), ("dkv_bstride", C.c_int64), ("dkv_ld", C.c_int64),
        ("dq_acc", C.c_void_p),
        ("keyinfo", C.c_void_p), ("ktile_flags", C.c_void_p), ("khot", C.c_void_p), ("qblk", C.c_void_p),
        ("qt_desc", C.c_void_p), ("kb_desc", C.c_void_p), ("kb_qt", C.c_void_p), ("visit", C.c_void_p),
        ("n_qtiles", C.c_int), ("n_kblocks", C.c_int), ("max_list", C.c_int), ("n_entries", C.c_int),
        ("batch", C.c_int), ("heads", C.c_int), ("n", C.c_int), ("nk_pad", C.c_int), ("n_ktiles64", C.c_int),
        ("scale", C.c_float), ("flags", C.c_int), ("split", C.c_int),
    ]


_P, _I64, _I, _F = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes).  Must list EVERY symbol include/mca_hip.h declares (tests check this).
SIGNATURES = {
    "mca_version": (C.c_char_p, []),
    "mca_flag_to_host": (_I, [_P, _P, _P]),
    "mca_build_keyhot": (_I, [_P, _P, _I, _I, _P]),
    "mca_rows_copy_add": (_I, [_P, _I64, _P, _I64, _I64, _I, _I, _I, _P]),
    "mca_segment_mean_fwd": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "mca_segment_mean_bwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "mca_gemm_nt": (_I, [_P, _I64, _P, _I64, _P, _I64, _I, _P, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "mca_gemm_nt_geglu_bwd": (_I, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "mca_gemm_nt_lnres": (_I, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _P]),
    "mca_gemm_nt_geglu_fwd": (_I, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P]),
    "mca_gemm_tn_acc": (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P]),
    "mca_gemm_tn_acc_group": (_I, [_P, _I, _I64, _P]),
    "mca_layernorm_fwd": (_I, [_P, _I64, _P, _P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _I, _P, _P, _I64, _I, _F, _P]),
    "mca_layernorm_bwd": (_I, [_P, _I64, _I64, _I64, _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _I64, _I, _P]),
    "mca_geglu_fwd": (_I, [_P, _P, _I64, _I, _P]),
    "mca_geglu_bwd": (_I, [_P, _P, _P, _I64, _I, _P]),
    "mca_cast_pad_bf16": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I, _P]),
    "mca_cast_pad_bf16_multi": (_I, [_P, _I, _P]),
    "mca_f32_to_bf16": (_I, [_P, _I64, _P, _I64, _I64, _I64, _F, _P]),
    "mca_bcast_rows": (_I, [_P, _I64, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "mca_reduce_rows": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I, _P]),
    "mca_embedding_renorm": (_I, [_P, _I64, _I, _F, _P]),
    "mca_tab_value_fwd": (_I, [_P, _P, _P, _P, _P, _I64, _I, _F, _F, _P]),
    "mca_tab_value_bwd": (_I, [_P, _I64, _P, _P, _P, _P, _I64, _I, _F, _P]),
    "mca_pack_masks": (_I, [_P, _P, _P, _P]),
    "mca_build_keyinfo": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "mca_attn_vmean": (_I, [_P, _I64, _I64, _P, _I, _I, _I, _P]),
    "mca_attn_vmean_if_needed": (_I, [_P, _I64, _I64, _P, _I, _I, _I, _P, _I, _P]),
    "mca_attn_fwd": (_I, [C.POINTER(AttnFwdArgs), _P]),
    "mca_attn_quant_mxfp8": (_I, [_P, _I64, _I64, _P, _P, _I64, _I64, C.POINTER(AttnFp8Operands), _I, _I, _I, _P]),
    "mca_attn_fwd_fp8": (_I, [C.POINTER(AttnFwdArgs), C.POINTER(AttnFp8Operands), _P]),
    "mca_attn_bwd_prep": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mca_attn_bwd_onepass": (_I, [C.POINTER(AttnBwd1Args), _P]),
    "mca_attn_bwd_prep_onepass": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I64, _I64, _P, _P, _P]),
    "mca_attn_bwd_dq": (_I, [C.POINTER(AttnBwd2Args), _P]),
    "mca_attn_bwd_dkv": (_I, [C.POINTER(AttnBwd2Args), _P]),
    "mca_attn_quant_bwd_mxfp8": (_I, [_P, _I64, _I64, _P, _P, _I64, _I64, _P, _I64, _I64, C.POINTER(AttnFp8BwdOperands), _I, _I, _I, _I, _P]),
    "mca_attn_bwd_dq_fp8": (_I, [C.POINTER(AttnBwd2Args), C.POINTER(AttnFp8BwdOperands), _P]),
    "mca_attn_bwd_dkv_fp8": (_I, [C.POINTER(AttnBwd2Args), C.POINTER(AttnFp8BwdOperands), _P]),
    "mca_contrastive_workspace_bytes": (_I64, [_I, _I]),
    "mca_contrastive_fwd_bwd": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mca_grad_sqnorm": (_I, [_P, _I64, _P, _P]),
    "mca_adamw_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "mca_adamw_hyper": (_I, [_P, _F, _F, _F, _P]),
    "mca_nonfinite_flag": (_I, [C.POINTER(FiniteArgs), _P, _I, _P]),
}
# measurement hooks (include/mca_hip_debug.h): exported by the library, not part of the drop-in ABI
DEBUG_SIGNATURES = {
    "mca_debug_set": (_I, [_I, _I]),
    "mca_debug_reset": (_I, []),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the shared library (built by ``__graft_entry__.build()`` / ``mca-paper_amd/build.py``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MCAHipError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py build`). "
                "There is no CPU or eager fallback for the MCA step.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(l, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_STREAM_CACHE = None     # set by the engine for the duration of a step (the stream does not change inside it)


def stream_ptr() -> int:
    if _STREAM_CACHE is not None:
        return _STREAM_CACHE
    return torch.cuda.current_stream().cuda_stream


class use_stream:
    """context manager: launch through an explicit stream handle (side-stream weight gradients)."""

    def __init__(self, handle: int):
        self.handle = handle

    def __enter__(self):
        global _STREAM_CACHE
        self.prev = _STREAM_CACHE
        _STREAM_CACHE = self.handle

    def __exit__(self, *exc):
        global _STREAM_CACHE
        _STREAM_CACHE = self.prev


class cached_stream:
    """context manager: query torch's current stream once and reuse the handle for every launch inside."""

    def __enter__(self):
        global _STREAM_CACHE
        self.prev = _STREAM_CACHE
        _STREAM_CACHE = torch.cuda.current_stream().cuda_stream

    def __exit__(self, *exc):
        global _STREAM_CACHE
        _STREAM_CACHE = self.prev


class knobs:
    """context manager for tests / tools: set measurement knobs (include/mca_hip_debug.h) and ALWAYS reset all of them on
    exit, so a failing test cannot leave the conservative kernels switched on for the rest of the process."""

    def __init__(self, **kv):
        self.kv = {int(k[1:]) if isinstance(k, str) else int(k): int(v) for k, v in kv.items()}

    def __enter__(self):
        for k, v in self.kv.items():
            lib().mca_debug_set(k, v)
        return self

    def __exit__(self, *exc):
        lib().mca_debug_reset()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def check(rc: int, what: str):
    if rc != 0:
        raise MCAHipError(f"{what} failed: {_ERR.get(rc, rc)}")


# ---- optional live kernel timing (bench.py): HIP events recorded on the launch stream around selected entry points
PROFILE = None          # None or {"names": set, "records": {key: [(start, end, flops)]}}
_PROFILE_ON = True      # sampling switch: bench.py records only every n-th step to keep the event overhead < 1 %
_TAG = ""


def profile_enable(flag: bool):
    global _PROFILE_ON
    _PROFILE_ON = bool(flag)


def set_tag(tag: str):
    global _TAG
    _TAG = tag


def profile_start(names):
    global PROFILE
    PROFILE = {"names": set(names), "records": {}}


def profile_collect():
    """Resolve the events recorded so far into (launches, ms, flops) totals and release them (synchronises).  Called after
    every sampled step so that at most one step's worth of timing events is ever outstanding on the queue."""
    torch.cuda.synchronize()
    tot = PROFILE.setdefault("totals", {})
    for key, recs in PROFILE["records"].items():
        n, ms, fl = tot.get(key, (0, 0.0, 0.0))
        tot[key] = (n + len(recs), ms + sum(s.elapsed_time(e) for s, e, _ in recs), fl + sum(f for _, _, f in recs))
    PROFILE["records"] = {}


def profile_stop():
    """-> {key: (launches, total_ms, total_flops)} (synchronises)."""
    global PROFILE
    profile_collect()
    prof, PROFILE = PROFILE, None
    return prof["totals"]


def call(name: str, *args, flops: float = 0.0):
    if PROFILE is not None and _PROFILE_ON and name in PROFILE["names"]:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(getattr(lib(), name)(*args), name)
        e.record()
        PROFILE["records"].setdefault(name + ("/" + _TAG if _TAG else ""), []).append((s, e, flops))
        return
    check(getattr(lib(), name)(*args), name)
