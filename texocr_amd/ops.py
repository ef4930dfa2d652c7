"""PyTorch custom operators over the C ABI (include/texocr.h): ``torch.ops.texocr.*``.

north_star: "driven from Python via PyTorch-ROCm custom ops through a thin C-ABI".  Each operator is a thin
shim: it checks its tensors, takes the current HIP stream from torch and calls ONE ``txo_*`` entry point of
``libtexocr_hip.so`` through ctypes.  Fake (meta) implementations give the output shapes, so the operators can be
traced (``torch.compile`` / ``FakeTensorMode``) without a GPU.  The reference callables they stand for
(file:line in the reference tree):

  texocr::encode            VisionEncoder.forward                     model/encoder.py:128-152
  texocr::decode_begin      the ``enc=`` hand-over of decoder.generate model/decoder.py:56,103 (+ attention.py:125-126 once)
  texocr::decode_step       Transformer.forward, one position         model/decoder.py:41-67
  texocr::generate          OCRModel.generate                         model/ocr_model.py:46-66
  texocr::generate_from_enc AutoRegressiveDecoder.generate            model/decoder.py:77-122
  texocr::generate_beam     (build extension, BASELINE config 5)

An engine is named by an integer id (operators take tensors and scalars only); ``register_engine`` hands one out.
There is no CPU implementation: calling an operator on CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch.library import custom_op

from . import _lib

_ENGINES: Dict[int, "weakref.ReferenceType"] = {}
_NEXT = [1]


def register_engine(obj) -> int:
    """obj: anything with ``.dims`` (fake implementations) and, for real calls, ``.handle`` / ``.lib`` / ``.device``
    (texocr_amd.model.HipEngine)."""
    i = _NEXT[0]
    _NEXT[0] += 1
    _ENGINES[i] = weakref.ref(obj)
    return i


def unregister_engine(i: int) -> None:
    _ENGINES.pop(i, None)


def _eng(i: int, ready: bool = False):
    r = _ENGINES.get(int(i))
    e = r() if r is not None else None
    if e is None:
        raise RuntimeError(f"texocr: no live engine with id {i}")
    if ready and hasattr(e, "_ensure"):
        e._ensure()                      # upload the owning module's parameters if they changed since the last call
    return e


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32_dev(t: torch.Tensor, name: str, eng) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA/HIP tensor (this engine has no CPU path)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    if t.device.index != eng.device:
        raise ValueError(f"{name} lives on cuda:{t.device.index} but the engine was created on cuda:{eng.device}")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------------------------
@custom_op("texocr::encode", mutates_args=())
def encode(img: torch.Tensor, engine: int) -> torch.Tensor:
    e = _eng(engine, ready=True)
    if img.ndim != 4:
        raise ValueError("expected an image batch of shape (B, C, H, W)")
    img = _f32_dev(img, "src", e)
    B, Cc, H, W = img.shape
    e.dims.check_image(Cc, H, W)
    out = torch.empty((B, e.dims.n_tokens(H, W), e.dims.embed_dim), device=img.device, dtype=torch.float32)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_encode(e.handle, img.data_ptr(), B, Cc, H, W, out.data_ptr(), _stream()))
    return out


@encode.register_fake
def _(img, engine):
    d = _eng(engine).dims
    B, _, H, W = img.shape
    return img.new_empty((B, d.n_tokens(H, W), d.embed_dim), dtype=torch.float32)


@custom_op("texocr::decode_begin", mutates_args=())
def decode_begin(enc: torch.Tensor, engine: int) -> None:
    e = _eng(engine, ready=True)
    enc = _f32_dev(enc, "enc", e)
    if enc.ndim != 3 or enc.shape[2] != e.dims.embed_dim:
        raise ValueError(f"enc must be (B, N, {e.dims.embed_dim})")
    e._enc_keepalive = enc
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_decode_begin(e.handle, enc.data_ptr(), enc.shape[0], enc.shape[1], _stream()))
    e._B = enc.shape[0]


@decode_begin.register_fake
def _(enc, engine):
    return None


@custom_op("texocr::decode_set_key_mask", mutates_args=())
def decode_set_key_mask(mask: Optional[torch.Tensor], engine: int) -> None:
    """The `mask` argument of decoder.generate / decoder.net (decoder.py:95-101; attention.py:130-155) for the decode_step calls of
    the current session: (B, cols) bool, False = padding (never attended by later queries); None clears it."""
    e = _eng(engine)
    B = getattr(e, "_B", None)
    with torch.cuda.device(e.device):
        if mask is None:
            _lib.check(e.lib.txo_decode_set_key_mask(e.handle, None, 0, _stream()))
            return
        if mask.ndim != 2 or mask.shape[0] != B or not mask.is_cuda:
            raise ValueError("mask must be a GPU tensor of shape (B, cols) matching the session started by texocr::decode_begin")
        m8 = mask.to(torch.uint8).contiguous()
        e._mask_keepalive = m8
        _lib.check(e.lib.txo_decode_set_key_mask(e.handle, m8.data_ptr(), int(m8.shape[1]), _stream()))


@decode_set_key_mask.register_fake
def _(mask, engine):
    return None


@custom_op("texocr::decode_step", mutates_args=())
def decode_step(tok_in: Optional[torch.Tensor], engine: int, t: int, batch: int, want_logits: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """One position: returns (logits (B, V) -- (0, V) when want_logits is false --, argmax token (B,))."""
    e = _eng(engine)
    if batch != getattr(e, "_B", None):
        raise ValueError("batch does not match the decode session started by texocr::decode_begin")
    dev = torch.device("cuda", e.device)
    if tok_in is not None:
        if tok_in.dtype != torch.int64 or not tok_in.is_cuda or tuple(tok_in.shape) != (batch,):
            raise ValueError("tok_in must be an int64 GPU tensor of shape (B,)")
        if tok_in.device.index != e.device:
            raise ValueError(f"tok_in lives on cuda:{tok_in.device.index} but the engine was created on cuda:{e.device}")
        tok_in = tok_in.contiguous()
    logits = torch.empty((batch if want_logits else 0, e.dims.vocab), device=dev, dtype=torch.float32)
    nxt = torch.empty((batch,), device=dev, dtype=torch.int64)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_decode_step(e.handle, _ptr(tok_in), int(t), logits.data_ptr() if want_logits else None,
                                         nxt.data_ptr(), _stream()))
    return logits, nxt


@decode_step.register_fake
def _(tok_in, engine, t, batch, want_logits):
    d = _eng(engine).dims
    ref = tok_in if tok_in is not None else torch.empty(0)
    return (ref.new_empty((batch if want_logits else 0, d.vocab), dtype=torch.float32),
            ref.new_empty((batch,), dtype=torch.int64))


@custom_op("texocr::decode_prefill", mutates_args=())
def decode_prefill(tokens: torch.Tensor, engine: int, want_logits: bool) -> torch.Tensor:
    """Transformer.forward over a whole prefix in one pass (decoder.py:41-67): tokens (B, t) int64 at positions 0..t-1 ->
    logits (B, t, V) (or (0, t, V)); fills the self-attention K/V cache rows 0..t-1 of the session opened by decode_begin."""
    e = _eng(engine)
    B = getattr(e, "_B", None)
    if tokens.ndim != 2 or tokens.dtype != torch.int64 or not tokens.is_cuda or tokens.shape[0] != B:
        raise ValueError("tokens must be an int64 GPU tensor of shape (B, t) matching the session started by texocr::decode_begin")
    if tokens.device.index != e.device:
        raise ValueError(f"tokens live on cuda:{tokens.device.index} but the engine was created on cuda:{e.device}")
    tokens = tokens.contiguous()
    t = tokens.shape[1]
    logits = torch.empty((B if want_logits else 0, t, e.dims.vocab), device=tokens.device, dtype=torch.float32)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_decode_prefill(e.handle, tokens.data_ptr(), int(t), logits.data_ptr() if want_logits else None, _stream()))
    return logits


@decode_prefill.register_fake
def _(tokens, engine, want_logits):
    d = _eng(engine).dims
    return tokens.new_empty((tokens.shape[0] if want_logits else 0, tokens.shape[1], d.vocab), dtype=torch.float32)


def _gen_outputs(src, e, max_len, want_logits):
    B = src.shape[0]
    toks = torch.empty((B, max_len), device=src.device, dtype=torch.int64)
    if os.environ.get("TXO_DEBUG_POISON"):      # tests: a column the engine returns as valid but never wrote shows as -7
        toks.fill_(-7)
    logits = torch.empty((B if want_logits else 0, max_len, e.dims.vocab), device=src.device, dtype=torch.float32)
    return toks, logits


@custom_op("texocr::generate", mutates_args=())
def generate(img: torch.Tensor, engine: int, max_len: int, eos: int, want_logits: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Greedy OCRModel.generate: (tokens (B, max_len) of which the first n are valid, n as an int64 CPU tensor of shape (1,),
    logits (B, max_len, V) or (0, max_len, V)).  eos < 0: no eos test (eos_tok=None)."""
    e = _eng(engine, ready=True)
    if img.ndim != 4:
        raise ValueError("expected an image batch of shape (B, C, H, W)")
    img = _f32_dev(img, "src", e)
    B, Cc, H, W = img.shape
    e.dims.check_image(Cc, H, W)
    toks, logits = _gen_outputs(img, e, max_len, want_logits)
    n = C.c_int32(0)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_generate(e.handle, img.data_ptr(), B, Cc, H, W, int(max_len), int(eos), toks.data_ptr(), C.byref(n),
                                      logits.data_ptr() if want_logits else None, _stream()))
    e._B, e._enc_keepalive = B, img
    return toks, torch.tensor([n.value], dtype=torch.int64), logits


@generate.register_fake
def _(img, engine, max_len, eos, want_logits):
    d = _eng(engine).dims
    B = img.shape[0]
    return (img.new_empty((B, max_len), dtype=torch.int64), torch.empty((1,), dtype=torch.int64, device="cpu"),
            img.new_empty((B if want_logits else 0, max_len, d.vocab), dtype=torch.float32))


@custom_op("texocr::generate_from_enc", mutates_args=())
def generate_from_enc(enc: torch.Tensor, engine: int, max_len: int, eos: int, want_logits: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    e = _eng(engine, ready=True)
    enc = _f32_dev(enc, "enc", e)
    if enc.ndim != 3 or enc.shape[2] != e.dims.embed_dim:
        raise ValueError(f"enc must be (B, N, {e.dims.embed_dim})")
    toks, logits = _gen_outputs(enc, e, max_len, want_logits)
    n = C.c_int32(0)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_generate_from_enc(e.handle, enc.data_ptr(), enc.shape[0], enc.shape[1], int(max_len), int(eos),
                                               toks.data_ptr(), C.byref(n), logits.data_ptr() if want_logits else None, _stream()))
    e._B, e._enc_keepalive = enc.shape[0], enc
    return toks, torch.tensor([n.value], dtype=torch.int64), logits


@generate_from_enc.register_fake
def _(enc, engine, max_len, eos, want_logits):
    d = _eng(engine).dims
    B = enc.shape[0]
    return (enc.new_empty((B, max_len), dtype=torch.int64), torch.empty((1,), dtype=torch.int64, device="cpu"),
            enc.new_empty((B if want_logits else 0, max_len, d.vocab), dtype=torch.float32))


@custom_op("texocr::generate_beam", mutates_args=())
def generate_beam(img: torch.Tensor, engine: int, beams: int, max_len: int, eos: int, want_all: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Beam search (build extension): (best tokens (B, max_len), scores (B, beams), all beams (B*beams, max_len) or (0, max_len),
    n as an int64 CPU tensor)."""
    e = _eng(engine, ready=True)
    img = _f32_dev(img, "src", e)
    B, Cc, H, W = img.shape
    e.dims.check_image(Cc, H, W)
    toks = torch.empty((B, max_len), device=img.device, dtype=torch.int64)
    scores = torch.empty((B, beams), device=img.device, dtype=torch.float32)
    allt = torch.empty((B * beams if want_all else 0, max_len), device=img.device, dtype=torch.int64)
    n = C.c_int32(0)
    with torch.cuda.device(e.device):
        _lib.check(e.lib.txo_generate_beam(e.handle, img.data_ptr(), B, Cc, H, W, int(beams), int(max_len), int(eos), toks.data_ptr(),
                                           scores.data_ptr(), allt.data_ptr() if want_all else None, C.byref(n), _stream()))
    return toks, scores, allt, torch.tensor([n.value], dtype=torch.int64)


@generate_beam.register_fake
def _(img, engine, beams, max_len, eos, want_all):
    B = img.shape[0]
    return (img.new_empty((B, max_len), dtype=torch.int64), img.new_empty((B, beams), dtype=torch.float32),
            img.new_empty((B * beams if want_all else 0, max_len), dtype=torch.int64), torch.empty((1,), dtype=torch.int64, device="cpu"))
