"""Drop-in call surface of the reference's OCRModel / VisionEncoder / AutoRegressiveDecoder for the
``generate()`` hot path, backed by the HIP engine behind the C ABI (include/texocr.h) through the
``torch.ops.texocr.*`` custom operators (texocr_amd/ops.py).

Mirrors (reference file:line):
  OCRModel(encoder, decoder, bos_token, eos_token, trg_pad_idx, device)   model/ocr_model.py:16-32   (an nn.Module)
  OCRModel.generate(src, max_len, temp=0.3)                               model/ocr_model.py:46-66
  model.encoder(src) -> (B, N, D)                                         model/encoder.py:128-152
  model.decoder.generate(start_tokens, eos_tok, max_len, temp, enc=)      model/decoder.py:77-122
  model.decoder.net(x, mask=, enc=) -> (B, t, V)                          model/decoder.py:41-67
  create_model(config)                                                    model/ocr_model.py:113-130
  model.state_dict() / model.load_state_dict(reference_state_dict)        key layout: SURVEY.md 8a

The classes are nn.Modules whose parameters carry the reference's names -- including the aliases: ONE LayerNorm
parameter pair is registered under ``layers.{s}.0`` for every sub-layer s of a stack (attention.py:200,221), so
``state_dict()`` has the reference's keys (144 for the default dims) while ``parameters()`` counts each tensor once.
The parameters are the host-visible master copy; the engine keeps its own re-laid-out copy (bf16 or fp32), refreshed
lazily after ``load_state_dict`` (or ``model.sync_weights()`` after editing parameters in place).

Differences that are deliberate and documented:
  * decoding is greedy by default (``decode='greedy'``: argmax; the reference samples with top-k /
    temperature / multinomial -- available as ``decode='sample'``);
  * the decoder is KV-cached; once the output outgrows ``decoder.max_len`` the reference slides its window
    (decoder.py:99-100) and so does this build, by re-running the window through the cached path for every further
    token (exact, slow: window-length engine steps per token);
  * errors are ValueError / RuntimeError, never ``assert``.
There is no CPU path: tensors must live on the GPU and the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .config import Dims
from .synth import state_dict_layout

_DTYPES = {"fp32": _lib.TXO_F32, "f32": _lib.TXO_F32, "float32": _lib.TXO_F32,
           "bf16": _lib.TXO_BF16, "bfloat16": _lib.TXO_BF16}

_LN_ALIAS = re.compile(r"\.layers\.(\d+)\.0\.(weight|bias)$")


def _is_ln_alias(key: str) -> bool:
    m = _LN_ALIAS.search(key)
    return bool(m) and int(m.group(1)) > 0


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
        self.handle = None
        self.loaded = False
        self._provider = None            # callable -> reference-layout state dict (set by OCRModel)
        self._stale = False              # the provider's weights are newer than the engine's copy
        self._create()
        self.id = ops.register_engine(self)

    def _create(self) -> None:
        d = self.dims
        ch, cw = d.canvas_hw
        cfg = _lib.TxoConfig(
            canvas_h=ch, canvas_w=cw, embed=1 if d.embed == "hybrid" else 0,
            in_channels=d.in_channels, embed_dim=d.embed_dim,
            enc_heads=d.enc_heads, enc_layers=d.enc_layers, dec_heads=d.dec_heads,
            dec_layers=d.dec_layers, enc_exp=d.enc_exp, dec_exp=d.dec_exp, vocab=d.vocab,
            max_len=d.max_len, bos=d.bos, eos=d.eos, pad=d.pad, dtype=_DTYPES[self.dtype],
            max_batch=self.max_batch, max_tokens=self.max_tokens)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.txo_engine_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        self.loaded = False
        self._env_seen = self._txo_env()             # what the engine saw when it read its knobs

    @staticmethod
    def _txo_env():
        return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("TXO_")))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.txo_engine_destroy(h)
            self.handle = None
        i = getattr(self, "id", None)
        if i is not None:
            ops.unregister_engine(i)

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, "torch.Tensor | np.ndarray"], strict: bool = True) -> None:
        """Upload a reference-layout state dict (aliased shared-LN keys included; aliases for s > 0 may be omitted).
        Loading again replaces the engine (a handle's weights are immutable once finalized)."""
        want = {k: tuple(s) for k, s, _ in state_dict_layout(self.dims)}
        missing = [k for k in want if k not in sd and not _is_ln_alias(k)]
        unexpected = [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {sorted(set(missing))[:6]}, unexpected keys {unexpected[:6]}")
        arrays = {}
        for k, v in sd.items():
            if k not in want:
                continue
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if tuple(a.shape) != want[k]:
                raise RuntimeError(f"load_state_dict: size mismatch for {k}: got {tuple(a.shape)}, expected {want[k]}")
            arrays[k] = a
        if self.loaded:
            self.lib.txo_engine_destroy(self.handle)
            self.handle = None
            self._create()
        with torch.cuda.device(self.device):
            for k, a in arrays.items():
                shape = (C.c_int64 * a.ndim)(*a.shape)
                _lib.check(self.lib.txo_engine_set_weight(self.handle, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
            _lib.check(self.lib.txo_engine_finalize_weights(self.handle))
        self.loaded = True
        self._stale = False

    def _ensure(self) -> None:
        if (self._stale or not self.loaded) and self._provider is not None:
            self.load_state_dict(self._provider())
        # the engine reads its TXO_* development knobs once, at creation; if the environment changed since the last call (tests
        # flip TXO_PERSIST / TXO_LANES on a live engine) it is told to read them again
        env = self._txo_env()
        if env != self._env_seen and self.handle:
            self.query(4)                             # TXO_Q_RELOAD_KNOBS
            self._env_seen = env

    # ---- path (every call goes through a torch.ops.texocr operator) -------------------------------------------------
    def encode(self, img: torch.Tensor) -> torch.Tensor:
        self._ensure()
        return torch.ops.texocr.encode(img, self.id)

    def decode_begin(self, enc: torch.Tensor) -> None:
        self._ensure()
        torch.ops.texocr.decode_begin(enc, self.id)

    def set_key_mask(self, mask: Optional[torch.Tensor]) -> None:
        """Padding mask (B, cols) bool over the positions of the current decode_step session (False = padding); None clears."""
        torch.ops.texocr.decode_set_key_mask(mask, self.id)

    def decode_prefill(self, tokens: torch.Tensor, want_logits: bool = True) -> Optional[torch.Tensor]:
        """All positions of a prefix in one pass (txo_decode_prefill); the K/V cache then holds rows 0..t-1."""
        logits = torch.ops.texocr.decode_prefill(tokens, self.id, bool(want_logits))
        return logits if want_logits else None

    def decode_step(self, t: int, tok_in: Optional[torch.Tensor] = None, want_logits: bool = True):
        logits, nxt = torch.ops.texocr.decode_step(tok_in, self.id, int(t), int(self._B), bool(want_logits))
        return (logits if want_logits else None), nxt

    def generate(self, img: Optional[torch.Tensor], max_len: int, eos: Optional[int], enc: Optional[torch.Tensor] = None,
                 return_logits: bool = False):
        if (img is None) == (enc is None):
            raise ValueError("pass exactly one of img / enc")
        self._ensure()
        e = -1 if eos is None else int(eos)
        if img is not None:
            toks, n, logits = torch.ops.texocr.generate(img, self.id, int(max_len), e, bool(return_logits))
        else:
            toks, n, logits = torch.ops.texocr.generate_from_enc(enc, self.id, int(max_len), e, bool(return_logits))
        n = int(n.item())
        toks = toks[:, :n]
        return (toks, logits[:, :n]) if return_logits else toks

    def generate_beam(self, img: torch.Tensor, beams: int, max_len: int, eos: Optional[int], return_beams: bool = False):
        """Beam search (build extension; the reference has none).  Returns the best beam's tokens (B, n), or with
        return_beams=True (tokens (B, beams, n), scores (B, beams)) sorted best first."""
        self._ensure()
        toks, scores, allt, n = torch.ops.texocr.generate_beam(img, self.id, int(beams), int(max_len),
                                                              -1 if eos is None else int(eos), bool(return_beams))
        n = int(n.item())
        if return_beams:
            return allt[:, :n].reshape(img.shape[0], beams, n), scores
        return toks[:, :n]

    def set_sampling(self, on: bool, temp: float = 1.0, seed: int = 0, threshold: float = 0.9) -> None:
        """on=True: the reference sampler (top-k with k = int((1 - threshold) * vocab), utils.py:85-91 -- 99 for
        vocab 1000 because of float rounding -- then softmax(/temp) and one multinomial draw, decoder.py:104-108)."""
        self._ensure()
        k = int((1 - threshold) * self.dims.vocab)
        _lib.check(self.lib.txo_set_sampling(self.handle, 1 if on else 0, max(k, 1), float(temp), int(seed) & (2**64 - 1)))

    def set_stop_mode(self, stop: str) -> None:
        """'global': the reference's loop (decoder.py:115-116: rows keep producing tokens after their eos, one break when every row
        contains it).  'row' (build extension): a row is finished at its first eos -- later tokens are the pad id -- and finished rows
        stop costing work on the launch path (txo_set_stop_mode)."""
        if stop not in ("global", "row"):
            raise ValueError("stop must be 'global' or 'row'")
        self._ensure()
        _lib.check(self.lib.txo_set_stop_mode(self.handle, 1 if stop == "row" else 0))

    def query(self, what: int) -> int:
        """txo_engine_query: 0 = the last generate() ran as one persistent launch, 1 = persistent launches that fell back,
        2 = row ranges (streams) of the last launch-path decode, 3 = the last decode's cross attention ran in latent form,
        4 = (not a question) re-read the TXO_* development knobs of generate() from the environment, 5 = live-row compactions of the
        last generate (stop='row' on the launch path)."""
        out = C.c_int64(0)
        _lib.check(self.lib.txo_engine_query(self.handle, int(what), C.byref(out)))
        return out.value

    # profiling hooks used by bench.py
    def profile(self, on) -> None:
        """0/False off, 1/True full (markers around every encode and step), 2 cross-attention dispatch events only."""
        self._ensure()
        _lib.check(self.lib.txo_profile_enable(self.handle, int(on)))

    def profile_read(self, kind: int):
        ms, n = C.c_double(0), C.c_int64(0)
        _lib.check(self.lib.txo_profile_read(self.handle, kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value


# --------------------------------------------------------------------------------------------------
# parameters in the reference's module tree
# --------------------------------------------------------------------------------------------------
class _Node(nn.Module):
    """A name-only container: the reference's module tree is mirrored for its parameter NAMES."""


def _attach(root: nn.Module, dotted: str, p: nn.Parameter) -> None:
    parts = dotted.split(".")
    m = root
    for name in parts[:-1]:
        if name not in m._modules:
            m.add_module(name, _Node())
        m = m._modules[name]
    m.register_parameter(parts[-1], p)


_NORM_KEY = re.compile(r"(layers\.\d+\.0|\.norm|stem\.1|\.block(_list)?\.[135])\.(weight|bias)$")


def _default_init(key: str, shape, fan_in_of: Dict[str, int], gen: torch.Generator) -> torch.Tensor:
    """torch's default initialisers of the reference's layers (values only matter until load_state_dict)."""
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith(("cls_token", "pos_embed")):
        return torch.zeros(shape)                                            # encoder.py:106-107 (never initialised there)
    if "embedding" in key:
        return torch.randn(shape, generator=gen)                             # nn.Embedding
    if _NORM_KEY.search(key):
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    fan_in = fan_in_of.get(key[: -len(leaf)] + "weight", 1)
    bound = 1.0 / math.sqrt(max(fan_in, 1))                                  # kaiming_uniform(a=sqrt(5)) and the bias bound
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def _build_params(root: nn.Module, dims: Dims, prefix: str, shared: Dict[str, nn.Parameter], device) -> None:
    layout = state_dict_layout(dims)
    fan_in = {k: int(np.prod(s[1:])) for k, s, _ in layout if len(s) > 1}
    gen = torch.Generator().manual_seed(0)
    for key, shape, canon in layout:
        if not key.startswith(prefix):
            continue
        if canon not in shared:
            shared[canon] = nn.Parameter(_default_init(canon, shape, fan_in, gen).to(device), requires_grad=False)
        _attach(root, key[len(prefix):], shared[canon])


# --------------------------------------------------------------------------------------------------
# reference-shaped facades
# --------------------------------------------------------------------------------------------------
class VisionEncoder(nn.Module):
    """model.encoder: callable (B,C,H,W) -> (B, N, D); CLS token at index 0 (encoder.py:128-152)."""

    def __init__(self, engine: HipEngine, _shared: Optional[dict] = None):
        super().__init__()
        self._engine = engine
        d = engine.dims
        self.height, self.width = d.canvas_hw
        self.patch_size = d.patch
        _build_params(self, d, "encoder.", _shared if _shared is not None else {}, torch.device("cuda", engine.device))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._engine.encode(x)


def _check_token_ids(tokens: torch.Tensor, vocab: int) -> None:
    """The reference's nn.Embedding raises IndexError for an id outside the table (decoder.py:51); so does this facade (the C ABI
    forces such ids into the table instead: it cannot raise from the device)."""
    if tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= vocab):
        raise IndexError(f"token id outside the vocabulary [0, {vocab})")


class Transformer(nn.Module):
    """model.decoder.net: (B,t) int64 tokens -> (B,t,V) logits over the whole prefix (decoder.py:41-67): ONE causal
    multi-position pass (txo_decode_prefill), which also leaves the K/V cache filled for the positions given."""

    def __init__(self, engine: HipEngine, _shared: Optional[dict] = None):
        super().__init__()
        self._engine = engine
        self.max_len = engine.dims.max_len
        _build_params(self, engine.dims, "decoder.net.", _shared if _shared is not None else {}, torch.device("cuda", engine.device))

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, enc: Optional[torch.Tensor] = None, **kw):
        if kw:
            raise ValueError(f"unsupported arguments for the inference path: {sorted(kw)}")
        if enc is None:
            raise ValueError("Must provide enc (cross-attending decoder)")       # attention.py:232-233
        padded = mask is not None and not bool(mask.all())
        if padded and tuple(mask.shape) != tuple(x.shape):
            raise ValueError("mask must have the shape of x")
        if x.ndim != 2 or x.dtype != torch.int64 or not x.is_cuda:
            raise ValueError("x must be an int64 GPU tensor of shape (B, t)")
        if x.shape[1] > self.max_len:
            raise ValueError("prefix longer than decoder.max_len")
        eng = self._engine
        _check_token_ids(x, eng.dims.vocab)
        eng.decode_begin(enc)
        one_pass = eng.dims.vocab % 8 == 0 and x.shape[1] <= eng.max_batch * eng.max_tokens and os.environ.get("TXO_NET_STEPWISE") is None
        if padded:
            # attention.py:130-155: a padded position is never attended by a query that is not padding.  Logits AT padded positions are
            # unspecified here (the reference softmaxes such a row uniformly over all keys, future ones included; nothing reads it)
            eng.set_key_mask(mask.to(x.device))
            try:
                if one_pass:
                    return eng.decode_prefill(x)              # one causal multi-position pass with the key mask (csrc/prefill.h)
                out = torch.empty((x.shape[0], x.shape[1], eng.dims.vocab), device=x.device, dtype=torch.float32)
                xt = x.t().contiguous()
                for t in range(x.shape[1]):
                    out[:, t] = eng.decode_step(t, xt[t])[0]
                return out
            finally:
                eng.set_key_mask(None)
        # the one-pass prefill needs a vocabulary that is a multiple of 8 and a prefix that fits the engine's workspace
        # (max_batch * max_tokens rows); anything else takes the single-position steps
        if one_pass:
            return eng.decode_prefill(x)
        out = torch.empty((x.shape[0], x.shape[1], eng.dims.vocab), device=x.device, dtype=torch.float32)
        xt = x.t().contiguous()                                   # fallback (odd vocabulary sizes; tests): one cached step per position
        for t in range(x.shape[1]):
            out[:, t] = eng.decode_step(t, xt[t])[0]
        return out


class AutoRegressiveDecoder(nn.Module):
    """model.decoder (decoder.py:70-122)."""

    def __init__(self, engine: HipEngine, _shared: Optional[dict] = None):
        super().__init__()
        self._engine = engine
        self.net = Transformer(engine, _shared)
        self.max_len = self.net.max_len
        self._resample = None

    def forward(self, *a, **k):
        raise NotImplementedError("AutoRegressiveDecoder.forward is the training loss (decoder.py:124-145); this engine "
                                  "implements the generate() inference path only")

    @torch.no_grad()
    def generate(self, start_tokens: torch.Tensor, eos_tok: Optional[int], max_len: int, temp: float = 1.0,
                 decode: str = "greedy", generator: Optional[torch.Generator] = None, seed: Optional[int] = None,
                 **kwargs) -> torch.Tensor:
        enc = kwargs.pop("enc", None)
        mask = kwargs.pop("mask", None)
        return_logits = bool(kwargs.pop("return_logits", False))   # build extension: also the logits every token was picked from
        stop = kwargs.pop("stop", "global")                        # build extension: 'row' = per-row stop, pad behind a row's first eos
        pad = kwargs.pop("pad", None)
        if stop not in ("global", "row"):
            raise ValueError("stop must be 'global' or 'row'")
        if kwargs:
            raise ValueError(f"unsupported arguments: {sorted(kwargs)}")
        if enc is None:
            raise ValueError("Must provide enc (cross-attending decoder)")
        if mask is not None and mask.ndim == 1:
            mask = mask[None, :]
        padded = mask is not None and not bool(mask.all())
        if padded and tuple(mask.shape) != tuple(start_tokens.shape if start_tokens.ndim == 2 else start_tokens[None, :].shape):
            raise ValueError("mask must have the shape of start_tokens")
        if decode not in ("greedy", "sample"):
            raise ValueError("decode must be 'greedy' or 'sample'")
        squeeze = start_tokens.ndim == 1
        st = start_tokens[None, :] if squeeze else start_tokens                  # decoder.py:88
        B, T0 = st.shape
        eng = self._engine
        _check_token_ids(st, eng.dims.vocab)
        self._resample = None
        if decode == "sample":
            if seed is None:
                seed = generator.initial_seed() if generator is not None else int(torch.randint(0, 2**62, (1,)).item())
            eng.set_sampling(True, temp=temp, seed=seed)
            self._resample = (temp, seed)
        if stop == "row":
            eng.set_stop_mode("row")
        try:
            # beyond the positional table the engine slides the window with its multi-position forward, which needs a vocabulary
            # that is a multiple of 8 and a table that fits its workspace -- else the general stepwise loop below
            window_ok = max_len <= self.max_len or (eng.dims.vocab % 8 == 0 and self.max_len <= eng.max_batch * eng.max_tokens)
            if padded:
                if return_logits:
                    raise ValueError("return_logits is not available with a padding mask")
                out = self._generate_stepwise(st, eos_tok, max_len, enc, mask=mask)
            elif T0 == 1 and bool((st == eng.dims.bos).all()) and (window_ok or return_logits):
                out = eng.generate(None, max_len, eos_tok, enc=enc, return_logits=return_logits)
            elif return_logits:
                raise ValueError("return_logits needs a BOS start inside the positional table (max_len <= decoder.max_len)")
            else:
                out = self._generate_stepwise(st, eos_tok, max_len, enc)
        finally:
            if decode == "sample":
                eng.set_sampling(False)
            if stop == "row":
                eng.set_stop_mode("global")
        if stop == "row" and eos_tok is not None:
            # (the engine's own generate has padded already; the stepwise loop -- arbitrary start prefix, padding mask -- and the
            # return_logits form are padded here: the same rule, tokens behind a row's first eos, start tokens included in the test as
            # decoder.py:115 does.  Logits behind a row's eos are what the finished row kept producing: unspecified.)
            p_id = eng.dims.pad if pad is None else int(pad)
            if return_logits:
                out = (_pad_after_eos(out[0], st.to(out[0].device), eos_tok, p_id), out[1])
            else:
                out = _pad_after_eos(out, st.to(out.device), eos_tok, p_id)
        if return_logits:
            return (out[0].squeeze(0), out[1].squeeze(0)) if squeeze else out
        return out.squeeze(0) if squeeze else out

    def _generate_stepwise(self, st, eos_tok, max_len, enc, mask=None):
        """General form (arbitrary start prefix, any max_len): one engine step per position with the reference's per-step
        host-side eos check (decoder.py:115-116); the engine picks the token (argmax or its sampler).

        While the output fits the positional table the KV cache is extended by one position per token.  Beyond it the
        reference feeds ``output[:, -max_len:]`` through the whole decoder with positions re-indexed from 0
        (decoder.py:99-100): no cached key survives that shift, so every further token re-runs its window through the
        cached path (max_len engine steps per token).  Exact and slow -- the reference is as slow there."""
        eng = self._engine
        st = st.to(enc.device)
        B, T0 = st.shape
        L = self.max_len
        eng.decode_begin(enc)
        output = st
        # decoder.py:95-101,112: the mask covers the start tokens, every generated token extends it with True, and it slides with the window
        m = None if mask is None else mask.to(device=enc.device, dtype=torch.bool)
        if m is not None:
            eng.set_key_mask(m[:, -L:])
        valid = 0                                                  # positions of the CURRENT window held by the cache
        for i in range(max_len):
            window = output[:, -L:]                                # decoder.py:99-100
            n = window.shape[1]
            if output.shape[1] > L:
                valid = 0                                          # every position shifted: nothing cached is reusable
                if m is not None:
                    eng.set_key_mask(m[:, -L:])
            wt = window.t().contiguous()
            if n - 1 - valid > 1 and eng.dims.vocab % 8 == 0 and n - 1 <= eng.max_batch * eng.max_tokens:   # (with or without a padding mask)
                eng.decode_prefill(window[:, :n - 1].contiguous(), want_logits=False)   # positions 0..n-2 in one pass
            else:
                for p in range(valid, n - 1):
                    eng.decode_step(p, wt[p], want_logits=False)
            if self._resample is not None and output.shape[1] > L:
                # the device sampler draws from a counter RNG keyed by (seed, row, position); once the window slides the
                # position stays at L - 1, so the seed advances with the token index instead
                eng.set_sampling(True, temp=self._resample[0], seed=self._resample[1] + i)
            _, tok = eng.decode_step(n - 1, wt[n - 1], want_logits=False)
            valid = n
            output = torch.cat((output, tok[:, None]), dim=-1)
            if m is not None:
                m = torch.nn.functional.pad(m, (0, 1), value=True)
            if eos_tok is not None and bool((output == eos_tok).any(dim=1).all()):
                break
        if m is not None:
            eng.set_key_mask(None)
        return output[:, T0:]


def _pad_after_eos(tokens: torch.Tensor, start: torch.Tensor, eos: int, pad: int) -> torch.Tensor:
    """stop='row': every token behind a row's first eos (looked for in start tokens + output, decoder.py:115) becomes `pad`."""
    if tokens.numel() == 0:
        return tokens
    seen_before = (start == eos).any(dim=1, keepdim=True)
    is_eos = tokens == eos
    prior = (torch.cumsum(is_eos.to(torch.int32), dim=1) - is_eos.to(torch.int32)) > 0     # an eos strictly before this column
    return torch.where(prior | seen_before, torch.full_like(tokens, pad), tokens)


class OCRModel(nn.Module):
    """TeXOCR model for image-to-LaTeX conversion -- inference surface (ocr_model.py:14-66)."""

    def __init__(self, encoder: VisionEncoder, decoder: AutoRegressiveDecoder, bos_token: int, eos_token: int,
                 trg_pad_idx: int, device: torch.device):
        super().__init__()
        if encoder._engine is not decoder._engine:
            raise ValueError("encoder and decoder must share one engine")
        self.encoder, self.decoder = encoder, decoder
        self.bos_token, self.eos_token, self.trg_pad_idx = bos_token, eos_token, trg_pad_idx
        self.device = device
        self._engine = encoder._engine
        self._engine._provider = self._export_weights
        self._engine._stale = True
        self.eval()

    # ---- weights: the reference's state_dict layout, aliases included ----
    def _export_weights(self) -> Dict[str, torch.Tensor]:
        return dict(self.state_dict())

    def sync_weights(self) -> "OCRModel":
        """Re-upload the parameters to the engine at the next call (after editing them in place)."""
        self._engine._stale = True
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """nn.Module.load_state_dict for the reference's key layout.  Accepts tensors or numpy arrays; the aliased
        shared-LayerNorm keys layers.{s}.0.* (s > 0) may be omitted, and when given must equal layers.0.0.* -- the
        reference holds ONE LayerNorm per stack (attention.py:200,221)."""
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in state_dict.items()}
        own = self.state_dict()
        for k in own:
            if _is_ln_alias(k):
                base = _LN_ALIAS.sub(lambda m: f".layers.0.0.{m.group(2)}", k)
                if k in sd and base in sd and not torch.equal(sd[k].float().cpu(), sd[base].float().cpu()):
                    raise ValueError(f"{k} differs from {base}: the reference shares ONE LayerNorm per stack")
                if k not in sd and base in sd:
                    sd[k] = sd[base]
        for k, v in sd.items():
            if k in own and tuple(v.shape) != tuple(own[k].shape):
                raise RuntimeError(f"load_state_dict: size mismatch for {k}: got {tuple(v.shape)}, expected {tuple(own[k].shape)}")
        res = super().load_state_dict(sd, strict=strict)
        self._engine._stale = True
        return res

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if dev is not None and not isinstance(dev, torch.dtype) and torch.device(dev).type != "cuda":
            raise ValueError("this model only runs on the GPU")
        return self

    @torch.no_grad()
    def generate(self, src: torch.Tensor, max_len: int, temp: float = 0.3, *, decode: str = "greedy",
                 generator: Optional[torch.Generator] = None, seed: Optional[int] = None, return_logits: bool = False,
                 beam: int = 0, return_beams: bool = False, stop: str = "global"):
        if stop not in ("global", "row"):
            raise ValueError("stop must be 'global' or 'row'")
        if beam:                                           # build extension (BASELINE config 5); engine max_batch >= B * beam
            if max_len > self.decoder.max_len:
                raise ValueError(f"beam search needs max_len <= decoder.max_len ({self.decoder.max_len})")
            return self._engine.generate_beam(src, beam, max_len, self.eos_token, return_beams=return_beams)
        if decode == "greedy" and self.bos_token == self._engine.dims.bos:
            # (max_len > decoder.max_len: txo_generate slides the window like the reference, decoder.py:99-100)
            if stop == "row":
                # (with return_logits the engine does not compact -- a finished row's logits would be missing -- and only pads)
                self._engine.set_stop_mode("row")
                try:
                    return self._engine.generate(src, max_len, self.eos_token, return_logits=return_logits)
                finally:
                    self._engine.set_stop_mode("global")
            return self._engine.generate(src, max_len, self.eos_token, return_logits=return_logits)
        enc = self.encoder(src)
        start = torch.full((src.shape[0], 1), self.bos_token, dtype=torch.int64, device=src.device)   # ocr_model.py:57
        return self.decoder.generate(start_tokens=start, eos_tok=self.eos_token, max_len=max_len, temp=temp,
                                     decode=decode, generator=generator, seed=seed, enc=enc, return_logits=return_logits,
                                     stop=stop, pad=self.trg_pad_idx)

    def forward(self, *a, **k):
        raise NotImplementedError("OCRModel.forward is the training loss (ocr_model.py:38-44); this engine "
                                  "implements the generate() inference path only")


def _assemble(dims: Dims, dtype: str, max_batch: int, max_tokens: int, device: torch.device) -> OCRModel:
    eng = HipEngine(dims, dtype=dtype, max_batch=max_batch, max_tokens=max_tokens)
    shared: Dict[str, nn.Parameter] = {}
    return OCRModel(VisionEncoder(eng, shared), AutoRegressiveDecoder(eng, shared), dims.bos, dims.eos, dims.pad, device)


def create_model(config: dict, dtype: str = "fp32", max_batch: int = 64, max_tokens: int = 0) -> OCRModel:
    """create_model(config) (ocr_model.py:113-130).  As in the reference the model comes back with default-initialised
    parameters; load_state_dict replaces them."""
    dims = Dims.from_config(config)
    device = torch.device(config.get("device", "cuda"))
    if device.type != "cuda":
        device = torch.device("cuda")
    return _assemble(dims, dtype, max_batch, max_tokens, device)


def model_from_dims(dims: Dims, dtype: str = "fp32", max_batch: int = 64, max_tokens: int = 0) -> OCRModel:
    return _assemble(dims, dtype, max_batch, max_tokens, torch.device("cuda"))
