"""ctypes binding of the C ABI in include/texocr.h (libtexocr_hip.so, built in-tree by
``__graft_entry__.build()`` / ``texocr_amd/build.py``).

There is NO CPU fallback: if the HIP library is missing this module raises on load, and every
compute entry point requires device pointers."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TXO_LIB_PATH: load ANOTHER build of the library (development only: probes/ab_libs.sh compares builds on one GPU box without
# overwriting the in-tree product library)
LIB_PATH = os.environ.get("TXO_LIB_PATH") or os.path.join(_HERE, "libtexocr_hip.so")

TXO_F32, TXO_BF16 = 0, 1
TXO_E_INVALID, TXO_E_STATE, TXO_E_HIP = -1, -2, -3


class TxoConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "canvas_h", "canvas_w", "embed", "in_channels", "embed_dim", "enc_heads", "enc_layers", "dec_heads", "dec_layers",
        "enc_exp", "dec_exp", "vocab", "max_len", "bos", "eos", "pad", "dtype", "max_batch", "max_tokens")]


# every symbol include/texocr.h declares: (restype, argtypes)
_P, _I, _I64P, _FP = C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p
SYMBOLS = {
    "txo_engine_create": (C.c_int, [C.POINTER(TxoConfig), C.POINTER(_P)]),
    "txo_engine_destroy": (None, [_P]),
    "txo_engine_set_weight": (C.c_int, [_P, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), _I]),
    "txo_engine_finalize_weights": (C.c_int, [_P]),
    "txo_encode": (C.c_int, [_P, _FP, _I, _I, _I, _I, _FP, _P]),
    "txo_decode_begin": (C.c_int, [_P, _FP, _I, _I, _P]),
    "txo_decode_step": (C.c_int, [_P, _I64P, _I, _FP, _I64P, _P]),
    "txo_decode_prefill": (C.c_int, [_P, _I64P, _I, _FP, _P]),
    "txo_decode_set_key_mask": (C.c_int, [_P, C.c_void_p, _I, _P]),
    "txo_generate": (C.c_int, [_P, _FP, _I, _I, _I, _I, _I, _I, _I64P, C.POINTER(C.c_int32), _FP, _P]),
    "txo_generate_from_enc": (C.c_int, [_P, _FP, _I, _I, _I, _I, _I64P, C.POINTER(C.c_int32), _FP, _P]),
    "txo_generate_beam": (C.c_int, [_P, _FP, _I, _I, _I, _I, _I, _I, _I, _I64P, _FP, _I64P, C.POINTER(C.c_int32), _P]),
    "txo_set_sampling": (C.c_int, [_P, _I, _I, C.c_float, C.c_uint64]),
    "txo_set_stop_mode": (C.c_int, [_P, _I]),
    "txo_profile_enable": (C.c_int, [_P, _I]),
    "txo_profile_read": (C.c_int, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "txo_engine_query": (C.c_int, [_P, _I, C.POINTER(C.c_int64)]),
    "txo_last_error": (C.c_char_p, []),
    "txo_version": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library.  torch is imported first so that its bundled HIP runtime
    (libamdhip64.so.7) is the one and only runtime in the process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `python -m texocr_amd.build`). There is no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64 from torch/lib before ours resolves it)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = load().txo_last_error().decode(errors="replace")
    if rc == TXO_E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(f"texocr engine error {rc}: {msg}")
