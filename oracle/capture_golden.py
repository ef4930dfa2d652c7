#!/usr/bin/env python3
"""ORACLE TOOLING -- runs ONLY in the build container (needs /root/reference).

Imports the real reference (olibridge01/TeXOCR, read-only at /root/reference), fills it with
the deterministic synthetic weights of ``texocr_amd.synth`` and records golden input/output
vectors for the OCRModel.generate() path under ``tests/golden/``.  The fixtures are plain
arrays + JSON: no reference source, bytecode or pickled module travels.

How the reference is driven:
  * the tree uses absolute imports ``TeXOCR.*`` and has no top-level __init__, so it is
    exposed as a namespace package through a symlink in a temp dir;
  * ``torchvision`` (absent here) is imported at ocr_model.py:4 / dataset.py:15 -> stubbed;
  * the north-star front end is the plain PatchEmbedding, reached by constructing
    ``VisionEncoder(img_size=<int>, patch_size=16, in_channels=C, ...)`` directly
    (encoder.py:86 default embed_layer);
  * greedy := ``torch.multinomial`` replaced by argmax for the duration of generate()
    (decoder.py:104-108; argmax survives top-k and softmax(/temp), both monotone).

Usage:  python oracle/capture_golden.py [--only NAME]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from texocr_amd.config import Dims, default_config, reference_config          # noqa: E402
from texocr_amd import synth                                 # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    sys.dont_write_bytecode = True
    ns = tempfile.mkdtemp(prefix="texocr_ref_ns_")
    os.symlink(REF, os.path.join(ns, "TeXOCR"))
    sys.path.insert(0, ns)
    tv = types.ModuleType("torchvision")
    tv.transforms = MagicMock()
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
    from TeXOCR.model.encoder import VisionEncoder
    from TeXOCR.model.decoder import create_decoder
    from TeXOCR.model.ocr_model import OCRModel, create_model
    import TeXOCR.model.decoder as dec_mod
    import TeXOCR.utils as ref_utils
    return VisionEncoder, create_decoder, OCRModel, dec_mod, ref_utils, create_model


VisionEncoder, create_decoder, OCRModel, ref_dec_mod, ref_utils, ref_create_model = import_reference()


def build_reference(d: Dims, seed: int):
    """Reference OCRModel with PatchEmbedding front end, weights from the synth recipe."""
    cfg = default_config(
        encoder={"embed_dim": d.embed_dim, "heads": d.enc_heads, "num_layers": d.enc_layers},
        decoder={"embed_dim": d.embed_dim, "heads": d.dec_heads, "num_layers": d.dec_layers,
                 "exp_factor": d.dec_exp},
        max_length=d.max_len, vocab_size=d.vocab)
    enc = VisionEncoder(img_size=d.canvas, patch_size=d.patch, in_channels=d.in_channels,
                        embed_dim=d.embed_dim, num_layers=d.enc_layers, heads=d.enc_heads)
    model = OCRModel(enc, create_decoder(cfg), d.bos, d.eos, d.pad, torch.device("cpu")).eval()
    sd_np = synth.synth_state_dict(d, seed)
    ref_sd = model.state_dict()
    # pin the key layout the build's importer must accept
    assert set(ref_sd.keys()) == set(sd_np.keys()), set(ref_sd.keys()) ^ set(sd_np.keys())
    for k, v in ref_sd.items():
        assert tuple(v.shape) == sd_np[k].shape, (k, v.shape, sd_np[k].shape)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    # aliases really are one tensor in the reference
    L = model.encoder.attn_layers.layers
    assert all(L[i][0] is L[0][0] for i in range(len(L)))
    return model, sd_np


class greedy_patch:
    """multinomial -> argmax; records the raw last-position logits of every step through the
    decoder module's ``topk`` name (decoder.py:104)."""

    def __init__(self):
        self.logits = []

    def __enter__(self):
        self._mn, self._tk = torch.multinomial, ref_dec_mod.topk
        torch.multinomial = lambda p, n: p.argmax(-1, keepdim=True)

        def rec(logits, *a, **k):
            self.logits.append(logits.detach().clone())
            return self._tk(logits, *a, **k)
        ref_dec_mod.topk = rec
        return self

    def __exit__(self, *exc):
        torch.multinomial, ref_dec_mod.topk = self._mn, self._tk


def margins(step_logits: torch.Tensor) -> np.ndarray:
    top2 = step_logits.topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)


def top5(step_logits: torch.Tensor):
    v, i = step_logits.topk(5, dim=-1)
    return i.numpy().astype(np.int16), v.numpy().astype(np.float32)


def tf_logits(model, tokens: torch.Tensor, enc: torch.Tensor, bos: int) -> torch.Tensor:
    prefix = torch.cat([torch.full((tokens.shape[0], 1), bos, dtype=torch.long), tokens[:, :-1]], 1)
    return model.decoder.net(prefix, mask=torch.ones_like(prefix, dtype=torch.bool), enc=enc)


def save(name: str, meta: dict, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **arrays)
    with open(os.path.join(GOLD, f"{name}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    sz = os.path.getsize(os.path.join(GOLD, f"{name}.npz"))
    print(f"[golden] {name}: {sz/1024:.0f} KiB  {', '.join(f'{k}{tuple(v.shape)}' for k, v in arrays.items())}")


# ------------------------------------------------------------------------------------------
TINY = Dims(canvas=64, in_channels=3, embed_dim=64, enc_heads=2, enc_layers=2, dec_heads=2,
            dec_layers=2, vocab=64, max_len=24, bos=62, eos=61, pad=63)


@torch.no_grad()
def cap_tiny():
    """Tiny model, everything stored in full: pins the LN sandwich per sub-layer (G4), the
    encoder, teacher-forced logits, greedy tokens and the state_dict layout."""
    d, seed, img_seed = TINY, 7, 11
    model, sd = build_reference(d, seed)
    img = torch.from_numpy(synth.synth_images(2, 3, 32, 48, img_seed))
    enc = model.encoder(img)
    # per-sub-layer trace = input of every following sub-layer; AttentionLayers returns the x
    # entering each 'self' sub-layer as 'hiddens' (attention.py:239-240)
    x = model.encoder.patch_embed(img)
    _, inter = None, None
    hid_enc = []
    orig = model.encoder.attn_layers.forward

    def wrap(*a, **k):
        out, inter = orig(*a, **k)
        hid_enc.extend(inter["hiddens"])
        hid_enc.append(out)
        return out, inter
    model.encoder.attn_layers.forward = wrap
    model.encoder(img)
    model.encoder.attn_layers.forward = orig

    with greedy_patch() as gp:
        toks = model.generate(img, max_len=16)
    step_logits = torch.stack(gp.logits, 1)
    tfl = tf_logits(model, toks, enc, d.bos)
    hid_dec = []
    orig_d = model.decoder.net.attn_layers.forward

    def wrap_d(*a, **k):
        out, inter = orig_d(*a, **k)
        hid_dec.extend(inter["hiddens"])
        hid_dec.append(out)
        return out, inter
    model.decoder.net.attn_layers.forward = wrap_d
    tf_logits(model, toks, enc, d.bos)
    model.decoder.net.attn_layers.forward = orig_d

    layout = [[k, list(s), c] for k, s, c in synth.state_dict_layout(d)]
    save("tiny", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                  "image_shape": [2, 3, 32, 48], "max_len": 16, "state_dict_layout": layout,
                  "note": "enc_hiddens[i] = x entering encoder layer i's self sub-layer, last = stack output "
                          "(before final LN); dec_hiddens likewise for the teacher-forced decoder run"},
         patch_embed=x.numpy(), enc=enc.numpy(), enc_hiddens=torch.stack(hid_enc).numpy(),
         dec_hiddens=torch.stack(hid_dec).numpy(), tokens=toks.numpy().astype(np.int16),
         step_logits=step_logits.numpy(), tf_logits=tfl.numpy(), margin=margins(step_logits))


@torch.no_grad()
def cap_cfg1():
    """BASELINE config 1 shape: default dims, canvas 224, B=4 3x224x224, greedy 256 steps."""
    d = Dims(canvas=224)
    seed, img_seed = 0, 1234
    model, sd = build_reference(d, seed)
    img = torch.from_numpy(synth.synth_images(4, 3, 224, 224, img_seed))
    enc = model.encoder(img)
    with greedy_patch() as gp:
        toks = model.generate(img, max_len=256)
    step_logits = torch.stack(gp.logits, 1)            # (4, 256, 1000)
    assert toks.shape == (4, 256)
    t5i, t5v = top5(step_logits)
    tfl = tf_logits(model, toks, enc, d.bos)
    print("   cfg1: max |step - teacher-forced| logits", float((tfl - step_logits).abs().max()),
          " min margin", float(margins(step_logits).min()))
    save("cfg1_b4_224x224", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                             "image_shape": [4, 3, 224, 224], "max_len": 256,
                             "enc_checksum_note": "enc_sum[b] = float64 sum over (N, D) of enc[b]"},
         enc0=enc[0].numpy(), enc_sum=enc.double().sum((1, 2)).numpy(),
         enc_abs_sum=enc.double().abs().sum((1, 2)).numpy(),
         tokens=toks.numpy().astype(np.int16), margin=margins(step_logits),
         top5_ids=t5i, top5_vals=t5v, logits_first16=step_logits[:2, :16].numpy(),
         logits_last4=step_logits[:2, -4:].numpy())


@torch.no_grad()
def cap_cfg2():
    """BASELINE config 2 image shape (3x224x672, N=589) at B=2, 48 greedy steps."""
    d = Dims(canvas=672)
    seed, img_seed = 0, 4321
    model, sd = build_reference(d, seed)
    img = torch.from_numpy(synth.synth_images(2, 3, 224, 672, img_seed))
    enc = model.encoder(img)
    with greedy_patch() as gp:
        toks = model.generate(img, max_len=48)
    step_logits = torch.stack(gp.logits, 1)
    t5i, t5v = top5(step_logits)
    save("cfg2_b2_224x672", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                             "image_shape": [2, 3, 224, 672], "max_len": 48,
                             "enc_rows_note": "enc_rows = enc[:, ::8] (every 8th token, CLS included)"},
         enc_rows=enc[:, ::8].numpy(), enc_sum=enc.double().sum((1, 2)).numpy(),
         enc_abs_sum=enc.double().abs().sum((1, 2)).numpy(),
         tokens=toks.numpy().astype(np.int16), margin=margins(step_logits),
         top5_ids=t5i, top5_vals=t5v, logits_first8=step_logits[:, :8].numpy())


@torch.no_grad()
def cap_posids():
    """G2: the position-id vector the reference builds for (H, W, canvas) (encoder.py:136-141)."""
    out, meta = {}, []
    for (H, W, canvas) in [(224, 224, 224), (224, 672, 672), (224, 448, 896), (64, 320, 896), (16, 16, 64)]:
        max_h = max_w = canvas // 16
        h, w = H // 16, W // 16
        grid_indices = torch.arange(max_h * max_w).reshape(max_h, max_w)
        ids = grid_indices[:h, :w].reshape(-1)
        ids = torch.cat((torch.zeros(1), ids + 1), dim=0).long()
        out[f"ids_{H}_{W}_{canvas}"] = ids.numpy().astype(np.int32)
        meta.append([H, W, canvas])
    # and check the formula against a live module for one odd shape
    d = Dims(canvas=96, embed_dim=64, enc_heads=1, enc_layers=1, dec_heads=1, dec_layers=1, vocab=32, max_len=8,
             bos=30, eos=29, pad=31)
    model, sd = build_reference(d, 3)
    img = torch.from_numpy(synth.synth_images(1, 3, 32, 80, 5))
    out["enc_32_80_96"] = model.encoder(img).numpy()
    save("posids", {"cases": meta, "live_case": {"dims": d.to_dict(), "weight_seed": 3, "image_seed": 5,
                                                  "image_shape": [1, 3, 32, 80]}}, **out)


@torch.no_grad()
def cap_eos():
    """G8: the GLOBAL eos break (decoder.py:115-116).  eos ids are picked from tokens the tiny
    model emits greedily, so that (a) every row has emitted it by some step -> early break,
    (b) only one row ever emits it -> runs to max_len."""
    d, seed, img_seed = TINY, 7, 11
    model, sd = build_reference(d, seed)
    img = torch.from_numpy(synth.synth_images(2, 3, 32, 48, img_seed))
    with greedy_patch():
        free = model.decoder.generate(torch.full((2, 1), d.bos, dtype=torch.long), eos_tok=None, max_len=20,
                                      enc=model.encoder(img))
    r0, r1 = set(free[0].tolist()), set(free[1].tolist())
    both = sorted(r0 & r1)
    only0 = sorted(r0 - r1)
    cases, arrays = [], {"free_tokens": free.numpy().astype(np.int16)}
    for name, eos in (("both", both[0] if both else None), ("only_row0", only0[0] if only0 else None)):
        if eos is None:
            continue
        model.eos_token = eos
        with greedy_patch():
            t = model.generate(img, max_len=20)
        arrays[f"tokens_{name}"] = t.numpy().astype(np.int16)
        cases.append({"name": name, "eos": int(eos), "n_steps": int(t.shape[1])})
    # bos == eos: the BOS column itself satisfies the check -> break after one step
    model.eos_token = d.bos
    with greedy_patch():
        t = model.generate(img, max_len=20)
    arrays["tokens_eos_is_bos"] = t.numpy().astype(np.int16)
    cases.append({"name": "eos_is_bos", "eos": int(d.bos), "n_steps": int(t.shape[1])})
    save("eos_break", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                       "image_shape": [2, 3, 32, 48], "max_len": 20, "cases": cases}, **arrays)


@torch.no_grad()
def cap_ragged():
    """A padded multi-token start prefix with its mask (decoder.py:95-101,112; attention.py:130-155): decoder.generate(start_tokens
    (B,T0), mask=(B,T0)) and decoder.net(x, mask=) from the reference -- left padding, an interior hole, and an all-True row."""
    d, seed, img_seed = TINY, 17, 19
    model, sd = build_reference(d, seed)
    img = torch.from_numpy(synth.synth_images(3, 3, 32, 48, img_seed))
    enc = model.encoder(img)
    start = torch.tensor([[d.pad, d.pad, d.bos, 5, 9], [d.bos, 7, 3, 11, 2], [d.bos, 4, d.pad, 6, 8]], dtype=torch.long)
    mask = torch.tensor([[0, 0, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 0, 1, 1]], dtype=torch.bool)
    with greedy_patch() as gp:
        toks = model.decoder.generate(start, eos_tok=None, max_len=12, enc=enc, mask=mask.clone())
    step_logits = torch.stack(gp.logits, 1)
    net = model.decoder.net(start, mask=mask.clone(), enc=enc)
    # the same start tokens WITHOUT the mask must give other tokens somewhere (else the fixture pins nothing)
    with greedy_patch():
        plain = model.decoder.generate(start, eos_tok=None, max_len=12, enc=enc)
    assert not torch.equal(plain, toks), "the mask changes nothing for these tokens: pick other start tokens"
    save("ragged_prefix", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed, "image_shape": [3, 3, 32, 48], "max_len": 12},
         start=start.numpy().astype(np.int16), mask=mask.numpy(), tokens=toks.numpy().astype(np.int16),
         step_logits=step_logits.numpy().astype(np.float32), margin=margins(step_logits), net_logits=net.numpy().astype(np.float32),
         tokens_unmasked=plain.numpy().astype(np.int16))


@torch.no_grad()
def cap_window():
    """G9: sliding window (decoder.py:99-100): max_length=8 but max_len=20."""
    d = Dims(canvas=64, in_channels=3, embed_dim=64, enc_heads=2, enc_layers=1, dec_heads=2, dec_layers=1,
             vocab=64, max_len=8, bos=62, eos=61, pad=63)
    model, sd = build_reference(d, 9)
    img = torch.from_numpy(synth.synth_images(1, 3, 32, 32, 13))
    with greedy_patch():
        t = model.generate(img, max_len=20)
    save("sliding_window", {"dims": d.to_dict(), "weight_seed": 9, "image_seed": 13,
                            "image_shape": [1, 3, 32, 32], "max_len": 20}, tokens=t.numpy().astype(np.int16))


@torch.no_grad()
def cap_sampling():
    """G11: the reference's sampling distribution at one step: top-k support and softmax(/temp)
    (utils.py:85-91, decoder.py:104-107).  V=1000 -> k = int((1-0.9)*1000) = 99 (float rounding)."""
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(3, 1000, generator=g)
    filt = ref_utils.topk(logits)
    probs = torch.softmax(filt / 0.3, dim=-1)
    save("sampling", {"temp": 0.3, "threshold": 0.9, "support": int((filt[0] > -float("inf")).sum())},
         logits=logits.numpy(), probs=probs.numpy())


@torch.no_grad()
def cap_hybrid():
    """N1: the default factory create_model(config/config.yml): hybrid ResNetV2 [2,4,6] embedder, 1 channel,
    (160, 1008) canvas.  Two 1x32x96 images (12 tokens + CLS), 12 greedy steps; per-stage backbone features."""
    cfg = reference_config()
    cfg["device"] = "cpu"
    d = Dims.from_config(cfg)
    seed, img_seed = 5, 77
    model = ref_create_model({k: v for k, v in cfg.items() if k not in ("embed",)}).eval()
    sd_np = synth.synth_state_dict(d, seed)
    ref_sd = model.state_dict()
    assert set(ref_sd.keys()) == set(sd_np.keys()), set(ref_sd.keys()) ^ set(sd_np.keys())
    for k, v in ref_sd.items():
        assert tuple(v.shape) == sd_np[k].shape, (k, v.shape, sd_np[k].shape)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    img = torch.from_numpy(synth.synth_images(2, 1, 32, 96, img_seed))
    bb = model.encoder.patch_embed.backbone_net
    stem = bb.stem(img)
    s0 = bb.stages[0](stem)
    s1 = bb.stages[1](s0)
    s2 = bb.stages[2](s1)
    emb = model.encoder.patch_embed(img)
    enc = model.encoder(img)
    with greedy_patch() as gp:
        toks = model.generate(img, max_len=12)
    step_logits = torch.stack(gp.logits, 1)
    layout = [[k, list(s), c] for k, s, c in synth.state_dict_layout(d)]
    save("hybrid_b2_32x96", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed, "image_shape": [2, 1, 32, 96],
                             "max_len": 12, "n_state_dict_keys": len(layout),
                             "note": "stem/stage features are NCHW as the reference produces them"},
         stem=stem.numpy(), stage0=s0.numpy(), stage1=s1.numpy(), stage2=s2.numpy(), embed=emb.numpy(), enc=enc.numpy(),
         tokens=toks.numpy().astype(np.int16), step_logits=step_logits.numpy(), margin=margins(step_logits))


@torch.no_grad()
def cap_hybrid_full():
    """N1 at the default factory's REAL size (VERDICT r05 missing 3): create_model(config/config.yml) on its full 1x160x1008 canvas
    (encoder.py:172-191: 10 x 63 patches + CLS = 631 tokens), two images, 32 greedy steps from the reference; encoder output in full."""
    cfg = reference_config()
    cfg["device"] = "cpu"
    d = Dims.from_config(cfg)
    seed, img_seed = 5, 78
    model = ref_create_model({k: v for k, v in cfg.items() if k not in ("embed",)}).eval()
    sd_np = synth.synth_state_dict(d, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    img = torch.from_numpy(synth.synth_images(2, 1, 160, 1008, img_seed))
    emb = model.encoder.patch_embed(img)
    enc = model.encoder(img)
    with greedy_patch() as gp:
        toks = model.generate(img, max_len=32)
    step_logits = torch.stack(gp.logits, 1)
    save("hybrid_b2_160x1008", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed, "image_shape": [2, 1, 160, 1008],
                                "max_len": 32, "tokens_per_image": int(enc.shape[1]),
                                "note": "embed_every8 = every 8th patch row of HybridEmbedding's output (backbone + 1x1 projection)"},
         embed_every8=emb.numpy()[:, ::8].astype(np.float32), enc=enc.numpy(), tokens=toks.numpy().astype(np.int16),
         step_logits=step_logits.numpy(), margin=margins(step_logits))


@torch.no_grad()
def cap_cfg4():
    """G10 / BASELINE config 4: ViT-Base encoder (12L/768d/12h) + 6-layer decoder (768d/12h: encoder and decoder widths
    must match), 3x224x672 (N=589), B=2, 8 greedy steps from the reference (encoder.py:75-121, decoder.py:148-173)."""
    d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
    seed, img_seed = 0, 2468
    model, sd = build_reference(d, seed)
    img = torch.from_numpy(synth.synth_images(2, 3, 224, 672, img_seed))
    enc = model.encoder(img)
    with greedy_patch() as gp:
        toks = model.generate(img, max_len=8)
    step_logits = torch.stack(gp.logits, 1)
    t5i, t5v = top5(step_logits)
    print("   cfg4: min margin", float(margins(step_logits).min()))
    save("cfg4_b2_224x672", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                             "image_shape": [2, 3, 224, 672], "max_len": 8,
                             "enc_rows_note": "enc_rows = enc[:, ::16, ::4] (every 16th token, every 4th feature)"},
         enc_rows=enc[:, ::16, ::4].numpy(), enc_sum=enc.double().sum((1, 2)).numpy(),
         enc_abs_sum=enc.double().abs().sum((1, 2)).numpy(),
         tokens=toks.numpy().astype(np.int16), margin=margins(step_logits),
         top5_ids=t5i, top5_vals=t5v, step_logits=step_logits.numpy())


@torch.no_grad()
def cap_cfg2_full():
    """The benchmark's own shape for the WHOLE decode: default dims, canvas 672, 3x224x672 (N = 589: two
    cross-attention passes of the HIP kernels), B=2, 256 greedy steps from the reference (self history > 128 keys together with the
    two-pass cross panel).  The image seed is the first whose smallest top-1/top-2 margin over all 2 x 256 steps is >= 1e-4, so
    that an fp32 implementation with another summation order (~5e-6) must reproduce every token."""
    d = Dims(canvas=672)
    seed = 0
    model, sd = build_reference(d, seed)
    for img_seed in range(9001, 9040):
        img = torch.from_numpy(synth.synth_images(2, 3, 224, 672, img_seed))
        with greedy_patch() as gp:
            toks = model.generate(img, max_len=256)
        step_logits = torch.stack(gp.logits, 1)            # (2, 256, 1000)
        mm = float(margins(step_logits).min())
        print(f"   cfg2_full: image seed {img_seed}: min margin {mm:.2e}")
        if mm >= 1e-4:
            break
    assert toks.shape == (2, 256)
    enc = model.encoder(img)
    t5i, t5v = top5(step_logits)
    save("cfg2_b2_224x672_t256", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                                  "image_shape": [2, 3, 224, 672], "max_len": 256, "min_margin": mm,
                                  "enc_rows_note": "enc_rows = enc[:, ::8] (every 8th token, CLS included)"},
         enc_rows=enc[:, ::8].numpy(), enc_sum=enc.double().sum((1, 2)).numpy(),
         enc_abs_sum=enc.double().abs().sum((1, 2)).numpy(),
         tokens=toks.numpy().astype(np.int16), margin=margins(step_logits),
         top5_ids=t5i, top5_vals=t5v, logits_first8=step_logits[:, :8].numpy(), logits_last4=step_logits[:, -4:].numpy())


@torch.no_grad()
def cap_cfg4_t64():
    """BASELINE config 4 (ViT-Base 12L/768d/12h + 6L decoder) for 64 greedy steps, B=2, 3x224x672: pins the 768-wide decode beyond
    the first few positions.  Image seed chosen like cap_cfg2_full (min margin >= 1e-4)."""
    d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
    seed = 0
    model, sd = build_reference(d, seed)
    for img_seed in range(2468, 2500):
        img = torch.from_numpy(synth.synth_images(2, 3, 224, 672, img_seed))
        with greedy_patch() as gp:
            toks = model.generate(img, max_len=64)
        step_logits = torch.stack(gp.logits, 1)
        mm = float(margins(step_logits).min())
        print(f"   cfg4_t64: image seed {img_seed}: min margin {mm:.2e}")
        if mm >= 1e-4:
            break
    t5i, t5v = top5(step_logits)
    save("cfg4_b2_224x672_t64", {"dims": d.to_dict(), "weight_seed": seed, "image_seed": img_seed,
                                 "image_shape": [2, 3, 224, 672], "max_len": 64, "min_margin": mm},
         tokens=toks.numpy().astype(np.int16), margin=margins(step_logits),
         top5_ids=t5i, top5_vals=t5v, logits_first4=step_logits[:, :4].numpy(), logits_last4=step_logits[:, -4:].numpy())


@torch.no_grad()
def cap_wrapper():
    """N3: the reference's TeXOCRWrapper.__call__ (ocr_model.py:94-110) on a drawn image, through the default factory
    (hybrid embedder).  torchvision is absent, so the wrapper object is assembled by hand around the reference's own
    create_model / RegExTokenizer / process_output, and its ``img_transform`` is the inference part of
    data_wrangling/dataset.py:365-371 (ToTensor -> Grayscale(1) -> Invert, no RandomAffine) written with numpy.  Two calls:
    one inside the positional table, one with max_len > max_length, where the reference slides its window
    (decoder.py:99-100)."""
    from PIL import Image, ImageDraw
    from TeXOCR.model.ocr_model import TeXOCRWrapper as RefWrapper
    from TeXOCR.tokenizer.tokenizer import RegExTokenizer as RefTok
    cfg = reference_config(max_length=24)
    cfg["device"] = "cpu"
    tok = RefTok()
    tok.load(os.path.join(REF, "tokenizer", "tokenizer_clean_1k.txt"))
    cfg["vocab_size"] = tok.vocab_size
    d = Dims.from_config(cfg)
    model = ref_create_model({k: v for k, v in cfg.items() if k not in ("embed",)}).eval()
    img = Image.new("RGB", (160, 48), (255, 255, 255))
    dr = ImageDraw.Draw(img)
    dr.line((10, 40, 60, 8), fill=(0, 0, 0), width=2)
    dr.ellipse((70, 10, 110, 40), outline=(40, 40, 200), width=2)
    dr.rectangle((120, 12, 150, 36), outline=(200, 30, 30), width=1)
    pixels = np.asarray(img, dtype=np.uint8)

    def transform(im):
        a = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
        g = 0.2989 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]
        return torch.from_numpy(np.ascontiguousarray(1.0 - g))[None]
    w = object.__new__(RefWrapper)
    w.tokenizer, w.model, w.img_transform = tok, model, transform
    # random weights give razor-thin top-1/top-2 margins now and then: take the first weight seed whose
    # smallest margin over both calls leaves room for fp32 reduction-order noise (~5e-6)
    for seed in range(21, 60):
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(d, seed).items()})
        out = {"pixels": pixels, "tensor": transform(img).numpy()}
        meta = {"dims": d.to_dict(), "weight_seed": seed, "cases": []}
        worst = 1e9
        for name, max_len in (("inside", 20), ("sliding", 40)):
            with greedy_patch() as gp:
                toks, text = w(img, max_len=max_len)
            m = margins(torch.stack(gp.logits, 1))
            worst = min(worst, float(m.min()))
            out[f"tokens_{name}"] = np.asarray(toks, dtype=np.int16)
            out[f"margin_{name}"] = m[0]
            meta["cases"].append({"name": name, "max_len": max_len, "text": text, "n_tokens": len(toks)})
        print(f"   wrapper: weight seed {seed}: min margin {worst:.2e}")
        if worst > 2e-3:
            break
    meta["min_margin"] = worst
    save("wrapper_160x48", meta, **out)


def cap_tokenizer():
    """N4: encode/decode vectors from the reference RegExTokenizer on its own vocabulary file (its merge table is
    exported as data to tests/golden/tokenizer_vocab_1k.json), and process_output (utils.py:73-79) known answers."""
    from TeXOCR.tokenizer.tokenizer import RegExTokenizer as RefTok
    src = os.path.join(REF, "tokenizer", "tokenizer_clean_1k.txt")
    ref = RefTok()
    ref.load(src)
    # the vocabulary as data (merge table extracted from the loaded reference object), not a copy of the file
    with open(os.path.join(GOLD, "tokenizer_vocab_1k.json"), "w") as f:
        json.dump({"vocab_size": ref.vocab_size, "special_tokens": ref.special_tokens,
                   "merges": [[a, b, t] for (a, b), t in ref.bp_merges.items()]}, f)
    texts = [r"\int _ { 0 } ^ { 1 } x ^ 2 d x", r"\frac { \partial f } { \partial x } = \lambda \sum _ { i = 1 } ^ { n } a _ i ,",
             r"E = m c ^ 2", r"\left( \begin{array} { c c } 1 & 0 \\ 0 & 1 \end{array} \right)", "a  b\n\n c",
             "x_{12345} + 1000000", r"\alpha\beta <EOS>", "\u00e9\u03b1 \u4e2d"]
    cases = [{"text": t, "ids": ref.encode(t), "decoded": ref.decode(ref.encode(t))} for t in texts]
    with open(os.path.join(GOLD, "tokenizer_cases.json"), "w") as f:
        json.dump({"vocab_size": ref.vocab_size, "special_tokens": ref.special_tokens, "cases": cases,
                   "process_output": [[c["text"], ref_utils.process_output(c["text"])] for c in cases]}, f, indent=1,
                  ensure_ascii=True)
    print("[golden] tokenizer_cases.json", len(cases), "cases")


CAPS = {"ragged": cap_ragged, "cfg2_full": cap_cfg2_full, "cfg4_t64": cap_cfg4_t64, "tokenizer": cap_tokenizer, "cfg4": cap_cfg4, "wrapper": cap_wrapper, "hybrid": cap_hybrid, "hybrid_full": cap_hybrid_full, "tiny": cap_tiny, "cfg1": cap_cfg1, "cfg2": cap_cfg2, "posids": cap_posids, "eos": cap_eos,
        "window": cap_window, "sampling": cap_sampling}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    for name, fn in CAPS.items():
        if a.only in (None, name):
            fn()
