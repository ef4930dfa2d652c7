"""Drop-in call surface of the reference's OCRModel / VisionEncoder / AutoRegressiveDecoder for the
``generate()`` hot path, backed by the HIP engine behind the C ABI (include/texocr.h).

Mirrors (reference file:line):
  OCRModel(encoder, decoder, bos_token, eos_token, trg_pad_idx, device)   model/ocr_model.py:16-32
  OCRModel.generate(src, max_len, temp=0.3)                               model/ocr_model.py:46-66
  model.encoder(src) -> (B, N, D)                                         model/encoder.py:128-152
  model.decoder.generate(start_tokens, eos_tok, max_len, temp, enc=)      model/decoder.py:77-122
  model.decoder.net(x, mask=, enc=) -> (B, t, V)                          model/decoder.py:41-67
  create_model(config)                                                    model/ocr_model.py:113-130
  model.load_state_dict(reference_state_dict)                             key layout: SURVEY.md 8a

Differences that are deliberate and documented:
  * decoding is greedy by default (``decode='greedy'``: argmax; the reference samples with top-k /
    temperature / multinomial -- available as ``decode='sample'``);
  * the decoder is KV-cached, so ``max_len`` must not exceed ``decoder.max_len`` (the reference would slide
    its window, decoder.py:99-100) -> ValueError instead;
  * errors are ValueError / RuntimeError, never ``assert``.
There is no CPU path: tensors must live on the GPU and the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C
import re
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import Dims
from .synth import state_dict_layout

_DTYPES = {"fp32": _lib.TXO_F32, "f32": _lib.TXO_F32, "float32": _lib.TXO_F32,
           "bf16": _lib.TXO_BF16, "bfloat16": _lib.TXO_BF16}


_LN_ALIAS = re.compile(r"\.layers\.(\d+)\.0\.(weight|bias)$")


def _is_ln_alias(key: str) -> bool:
    m = _LN_ALIAS.search(key)
    return bool(m) and int(m.group(1)) > 0


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t: torch.Tensor, name: str, device: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA/HIP tensor (this engine has no CPU path)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    if device is not None and t.device.index != device:
        raise ValueError(f"{name} lives on cuda:{t.device.index} but the engine was created on cuda:{device}")
    return t.contiguous()


class HipEngine:
    """Owns one txo_engine handle (weights copy, KV caches, workspace) on the current device."""

    def __init__(self, dims: Dims, dtype: str = "fp32", max_batch: int = 64, max_tokens: int = 0):
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}")
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the texocr_amd engine runs only on an MI355X (no CPU fallback)")
        self.dims, self.dtype = dims, dtype
        # the engine's memory and streams belong to the device that is current NOW; every later call checks its
        # tensors against it and runs with that device current
        self.device = torch.cuda.current_device()
        self.max_batch, self.max_tokens = max_batch, max_tokens or dims.n_pos
        ch, cw = dims.canvas_hw
        cfg = _lib.TxoConfig(
            canvas_h=ch, canvas_w=cw, embed=1 if dims.embed == "hybrid" else 0,
            in_channels=dims.in_channels, embed_dim=dims.embed_dim,
            enc_heads=dims.enc_heads, enc_layers=dims.enc_layers, dec_heads=dims.dec_heads,
            dec_layers=dims.dec_layers, enc_exp=dims.enc_exp, dec_exp=dims.dec_exp, vocab=dims.vocab,
            max_len=dims.max_len, bos=dims.bos, eos=dims.eos, pad=dims.pad, dtype=_DTYPES[dtype],
            max_batch=max_batch, max_tokens=self.max_tokens)
        h = C.c_void_p()
        _lib.check(self.lib.txo_engine_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        self.loaded = False

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.txo_engine_destroy(h)
            self.handle = None

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, "torch.Tensor | np.ndarray"], strict: bool = True) -> None:
        """Accepts the reference OCRModel.state_dict() layout (aliased shared-LN keys included)."""
        if self.loaded:
            raise RuntimeError("weights already loaded into this engine; create a new model to load others")
        want = {k: tuple(s) for k, s, _ in state_dict_layout(self.dims)}
        # aliased LN keys for s > 0 may be omitted (same tensor as layers.0.0.*)
        missing = [k for k in want if k not in sd and not _is_ln_alias(k)]
        unexpected = [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {sorted(set(missing))[:6]}, unexpected keys {unexpected[:6]}")
        for k, v in sd.items():
            if k not in want:
                continue
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if tuple(a.shape) != want[k]:
                raise RuntimeError(f"load_state_dict: size mismatch for {k}: got {tuple(a.shape)}, expected {want[k]}")
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self.lib.txo_engine_set_weight(self.handle, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        _lib.check(self.lib.txo_engine_finalize_weights(self.handle))
        self.loaded = True

    # ---- path ------------------------------------------------------------------------------------
    def encode(self, img: torch.Tensor) -> torch.Tensor:
        if img.ndim != 4:
            raise ValueError("expected an image batch of shape (B, C, H, W)")
        img = _dev_f32(img, "src", self.device)
        B, Cc, H, W = img.shape
        self.dims.check_image(Cc, H, W)
        out = torch.empty((B, self.dims.n_tokens(H, W), self.dims.embed_dim), device=img.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.txo_encode(self.handle, img.data_ptr(), B, Cc, H, W, out.data_ptr(), _stream()))
        return out

    def decode_begin(self, enc: torch.Tensor) -> None:
        enc = _dev_f32(enc, "enc", self.device)
        if enc.ndim != 3 or enc.shape[2] != self.dims.embed_dim:
            raise ValueError(f"enc must be (B, N, {self.dims.embed_dim})")
        self._enc_keepalive = enc
        with torch.cuda.device(self.device):
            _lib.check(self.lib.txo_decode_begin(self.handle, enc.data_ptr(), enc.shape[0], enc.shape[1], _stream()))
        self._B = enc.shape[0]

    def decode_step(self, t: int, tok_in: Optional[torch.Tensor] = None, want_logits: bool = True):
        B = self._B
        dev = self._enc_keepalive.device
        logits = torch.empty((B, self.dims.vocab), device=dev, dtype=torch.float32) if want_logits else None
        nxt = torch.empty((B,), device=dev, dtype=torch.int64)
        if tok_in is not None:
            if tok_in.dtype != torch.int64 or not tok_in.is_cuda or tok_in.shape != (B,):
                raise ValueError("tok_in must be an int64 GPU tensor of shape (B,)")
            if tok_in.device.index != self.device:
                raise ValueError(f"tok_in lives on cuda:{tok_in.device.index} but the engine was created on cuda:{self.device}")
            tok_in = tok_in.contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.txo_decode_step(self.handle, tok_in.data_ptr() if tok_in is not None else None, int(t),
                                                logits.data_ptr() if want_logits else None, nxt.data_ptr(), _stream()))
        return logits, nxt

    def generate(self, img: Optional[torch.Tensor], max_len: int, eos: Optional[int], enc: Optional[torch.Tensor] = None,
                 return_logits: bool = False):
        if (img is None) == (enc is None):
            raise ValueError("pass exactly one of img / enc")
        src = _dev_f32(img if img is not None else enc, "src" if img is not None else "enc", self.device)
        B = src.shape[0]
        toks = torch.empty((B, max_len), device=src.device, dtype=torch.int64)
        logits = torch.empty((B, max_len, self.dims.vocab), device=src.device, dtype=torch.float32) if return_logits else None
        n = C.c_int32(0)
        e = -1 if eos is None else int(eos)
        lp = logits.data_ptr() if return_logits else None
        if img is not None:
            _, Cc, H, W = src.shape
            self.dims.check_image(Cc, H, W)
            rc = self.lib.txo_generate(self.handle, src.data_ptr(), B, Cc, H, W, int(max_len), e, toks.data_ptr(),
                                       C.byref(n), lp, _stream())
        else:
            rc = self.lib.txo_generate_from_enc(self.handle, src.data_ptr(), B, src.shape[1], int(max_len), e,
                                                toks.data_ptr(), C.byref(n), lp, _stream())
        _lib.check(rc)
        self._B = B
        self._enc_keepalive = src
        toks = toks[:, :n.value]
        return (toks, logits[:, :n.value]) if return_logits else toks

    def generate_beam(self, img: torch.Tensor, beams: int, max_len: int, eos: Optional[int], return_beams: bool = False):
        """Beam search (build extension; the reference has none).  Returns the best beam's tokens (B, n), or with
        return_beams=True (tokens (B, beams, n), scores (B, beams)) sorted best first."""
        src = _dev_f32(img, "src", self.device)
        B, Cc, H, W = src.shape
        self.dims.check_image(Cc, H, W)
        toks = torch.empty((B, max_len), device=src.device, dtype=torch.int64)
        scores = torch.empty((B, beams), device=src.device, dtype=torch.float32)
        allt = torch.empty((B * beams, max_len), device=src.device, dtype=torch.int64) if return_beams else None
        n = C.c_int32(0)
        _lib.check(self.lib.txo_generate_beam(self.handle, src.data_ptr(), B, Cc, H, W, int(beams), int(max_len),
                                              -1 if eos is None else int(eos), toks.data_ptr(), scores.data_ptr(),
                                              allt.data_ptr() if return_beams else None, C.byref(n), _stream()))
        if return_beams:
            return allt[:, :n.value].reshape(B, beams, n.value), scores
        return toks[:, :n.value]

    def set_sampling(self, on: bool, temp: float = 1.0, seed: int = 0, threshold: float = 0.9) -> None:
        """on=True: the reference sampler (top-k with k = int((1 - threshold) * vocab), utils.py:85-91 -- 99 for
        vocab 1000 because of float rounding -- then softmax(/temp) and one multinomial draw, decoder.py:104-108)."""
        k = int((1 - threshold) * self.dims.vocab)
        _lib.check(self.lib.txo_set_sampling(self.handle, 1 if on else 0, max(k, 1), float(temp), int(seed) & (2**64 - 1)))

    def query(self, what: int) -> int:
        """txo_engine_query: 0 = the last generate() ran as one persistent launch, 1 = persistent launches that fell back."""
        out = C.c_int64(0)
        _lib.check(self.lib.txo_engine_query(self.handle, int(what), C.byref(out)))
        return out.value

    # profiling hooks used by bench.py
    def profile(self, on) -> None:
        """0/False off, 1/True full (markers around every encode and step), 2 cross-attention dispatch events only."""
        _lib.check(self.lib.txo_profile_enable(self.handle, int(on)))

    def profile_read(self, kind: int):
        ms, n = C.c_double(0), C.c_int64(0)
        _lib.check(self.lib.txo_profile_read(self.handle, kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value


# --------------------------------------------------------------------------------------------------
# reference-shaped facades
# --------------------------------------------------------------------------------------------------
class VisionEncoder:
    """model.encoder: callable (B,C,H,W) -> (B, N, D); CLS token at index 0 (encoder.py:128-152)."""

    def __init__(self, engine: HipEngine):
        self._engine = engine
        d = engine.dims
        self.height, self.width = d.canvas_hw
        self.patch_size = d.patch

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self._engine.encode(x)

    forward = __call__


class Transformer:
    """model.decoder.net: (B,t) int64 tokens -> (B,t,V) logits over the whole prefix (decoder.py:41-67).
    Implemented as t KV-cached steps, which equals the reference's full-prefix causal forward."""

    def __init__(self, engine: HipEngine):
        self._engine = engine
        self.max_len = engine.dims.max_len

    def __call__(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, enc: Optional[torch.Tensor] = None, **kw):
        if kw:
            raise ValueError(f"unsupported arguments for the inference path: {sorted(kw)}")
        if enc is None:
            raise ValueError("Must provide enc (cross-attending decoder)")       # attention.py:232-233
        if mask is not None and not bool(mask.all()):
            raise NotImplementedError("padding masks belong to the training forward (ocr_model.py:38-44), which is "
                                      "outside the generate() path this engine accelerates")
        if x.ndim != 2 or x.dtype != torch.int64 or not x.is_cuda:
            raise ValueError("x must be an int64 GPU tensor of shape (B, t)")
        if x.shape[1] > self.max_len:
            raise ValueError("prefix longer than decoder.max_len")
        eng = self._engine
        eng.decode_begin(enc)
        out = [eng.decode_step(t, x[:, t].contiguous())[0] for t in range(x.shape[1])]
        return torch.stack(out, dim=1)

    forward = __call__


class AutoRegressiveDecoder:
    """model.decoder (decoder.py:70-122)."""

    def __init__(self, engine: HipEngine):
        self._engine = engine
        self.net = Transformer(engine)
        self.max_len = self.net.max_len

    @torch.no_grad()
    def generate(self, start_tokens: torch.Tensor, eos_tok: Optional[int], max_len: int, temp: float = 1.0,
                 decode: str = "greedy", generator: Optional[torch.Generator] = None, seed: Optional[int] = None,
                 **kwargs) -> torch.Tensor:
        enc = kwargs.pop("enc", None)
        mask = kwargs.pop("mask", None)
        if kwargs:
            raise ValueError(f"unsupported arguments: {sorted(kwargs)}")
        if enc is None:
            raise ValueError("Must provide enc (cross-attending decoder)")
        if mask is not None and not bool(mask.all()):
            raise NotImplementedError("a padding mask over the start tokens is not part of the generate() path")
        if decode not in ("greedy", "sample"):
            raise ValueError("decode must be 'greedy' or 'sample'")
        squeeze = start_tokens.ndim == 1
        st = start_tokens[None, :] if squeeze else start_tokens                  # decoder.py:88
        B, T0 = st.shape
        if T0 + max_len - 1 > self.max_len:
            raise ValueError(f"start length {T0} + max_len {max_len} exceeds decoder.max_len {self.max_len}: the "
                             "reference would slide its window (decoder.py:99-100); the KV-cached engine refuses")
        eng = self._engine
        if decode == "sample":
            if seed is None:
                seed = generator.initial_seed() if generator is not None else int(torch.randint(0, 2**62, (1,)).item())
            eng.set_sampling(True, temp=temp, seed=seed)
        try:
            if T0 == 1 and bool((st == eng.dims.bos).all()):
                out = eng.generate(None, max_len, eos_tok, enc=enc)
            else:
                out = self._generate_stepwise(st, eos_tok, max_len, enc)
        finally:
            if decode == "sample":
                eng.set_sampling(False)
        return out.squeeze(0) if squeeze else out

    def _generate_stepwise(self, st, eos_tok, max_len, enc):
        """General form (arbitrary start prefix): one engine step per position with the reference's per-step
        host-side eos check (decoder.py:115-116); the engine picks the token (argmax or its sampler)."""
        eng = self._engine
        st = st.to(enc.device)
        B, T0 = st.shape
        eng.decode_begin(enc)
        for t in range(T0 - 1):
            eng.decode_step(t, st[:, t].contiguous(), want_logits=False)
        tok = st[:, T0 - 1].contiguous()
        output = st
        for i in range(max_len):
            _, tok = eng.decode_step(T0 - 1 + i, tok, want_logits=False)
            output = torch.cat((output, tok[:, None]), dim=-1)
            if eos_tok is not None and bool((output == eos_tok).any(dim=1).all()):
                break
        return output[:, T0:]


class OCRModel:
    """TeXOCR model for image-to-LaTeX conversion -- inference surface (ocr_model.py:14-66)."""

    def __init__(self, encoder: VisionEncoder, decoder: AutoRegressiveDecoder, bos_token: int, eos_token: int,
                 trg_pad_idx: int, device: torch.device):
        if encoder._engine is not decoder._engine:
            raise ValueError("encoder and decoder must share one engine")
        self.encoder, self.decoder = encoder, decoder
        self.bos_token, self.eos_token, self.trg_pad_idx = bos_token, eos_token, trg_pad_idx
        self.device = device
        self._engine = encoder._engine
        self.training = False

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise ValueError("this model only runs on the GPU")
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        self._engine.load_state_dict(state_dict, strict=strict)
        return self

    @torch.no_grad()
    def generate(self, src: torch.Tensor, max_len: int, temp: float = 0.3, *, decode: str = "greedy",
                 generator: Optional[torch.Generator] = None, seed: Optional[int] = None, return_logits: bool = False,
                 beam: int = 0, return_beams: bool = False):
        if beam:                                           # build extension (BASELINE config 5); engine max_batch >= B * beam
            return self._engine.generate_beam(src, beam, max_len, self.eos_token, return_beams=return_beams)
        if decode == "greedy" and self.bos_token == self._engine.dims.bos:
            if max_len > self.decoder.max_len:
                raise ValueError(f"max_len {max_len} exceeds decoder.max_len {self.decoder.max_len}: the reference "
                                 "would slide its window (decoder.py:99-100); the KV-cached engine refuses")
            return self._engine.generate(src, max_len, self.eos_token, return_logits=return_logits)
        enc = self.encoder(src)
        start = torch.full((src.shape[0], 1), self.bos_token, dtype=torch.int64, device=src.device)   # ocr_model.py:57
        return self.decoder.generate(start_tokens=start, eos_tok=self.eos_token, max_len=max_len, temp=temp,
                                     decode=decode, generator=generator, seed=seed, enc=enc)

    def forward(self, *a, **k):
        raise NotImplementedError("OCRModel.forward is the training loss (ocr_model.py:38-44); this engine "
                                  "implements the generate() inference path only")

    __call__ = forward


def create_model(config: dict, dtype: str = "fp32", max_batch: int = 64, max_tokens: int = 0) -> OCRModel:
    """create_model(config) (ocr_model.py:113-130) with the PatchEmbedding front end."""
    dims = Dims.from_config(config)
    eng = HipEngine(dims, dtype=dtype, max_batch=max_batch, max_tokens=max_tokens)
    device = torch.device(config.get("device", "cuda"))
    if device.type != "cuda":
        device = torch.device("cuda")
    return OCRModel(VisionEncoder(eng), AutoRegressiveDecoder(eng), dims.bos, dims.eos, dims.pad, device)


def model_from_dims(dims: Dims, dtype: str = "fp32", max_batch: int = 64, max_tokens: int = 0) -> OCRModel:
    eng = HipEngine(dims, dtype=dtype, max_batch=max_batch, max_tokens=max_tokens)
    return OCRModel(VisionEncoder(eng), AutoRegressiveDecoder(eng), dims.bos, dims.eos, dims.pad, torch.device("cuda"))
