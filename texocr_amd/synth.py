"""Deterministic synthetic weights and images (there is no checkpoint and no network).

The reference ships no checkpoint and leaves ``cls_token`` / ``pos_embed`` at zeros
(reference ``model/encoder.py:106-107``), so benches, smoke runs and golden fixtures all
use this recipe: every parameter is drawn from ``numpy.random.Generator(PCG64(seed ^
crc32(canonical_key)))`` so the same bytes are produced in this container and on the GPU
box (same image, same numpy) without committing 60 MB of weights.

``synth_state_dict`` returns the *reference's* ``state_dict()`` key layout, including the
aliased shared-LayerNorm keys (reference ``model/attention.py:200,221``: one ``nn.LayerNorm``
object is appended to every ``[norm, layer, residual]`` triple, so ``layers.{s}.0.weight``
appears for every ``s`` and they are all the same tensor).
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, List, Tuple

import numpy as np

from .config import Dims, RESNET_CHANNELS, RESNET_DEPTHS

Shape = Tuple[int, ...]


def enc_kinds(d: Dims) -> List[str]:
    return ["self", "mlp"] * d.enc_layers                 # attention.py:207-210


def dec_kinds(d: Dims) -> List[str]:
    return ["self", "cross", "mlp"] * d.dec_layers        # attention.py:204-205


def _stack_keys(prefix: str, kinds: List[str], D: int, inner: int, ffn: int) -> List[Tuple[str, Shape, str]]:
    """(key, shape, canonical_key) for one AttentionLayers stack."""
    out: List[Tuple[str, Shape, str]] = []
    for s, kind in enumerate(kinds):
        p = f"{prefix}.layers.{s}"
        out.append((f"{p}.0.weight", (D,), f"{prefix}.layers.0.0.weight"))   # shared LN
        out.append((f"{p}.0.bias", (D,), f"{prefix}.layers.0.0.bias"))
        if kind in ("self", "cross"):
            for n in ("q", "k", "v"):
                out.append((f"{p}.1.{n}.weight", (inner, D), f"{p}.1.{n}.weight"))
            out.append((f"{p}.1.fc_out.0.weight", (2 * D, inner), f"{p}.1.fc_out.0.weight"))
            out.append((f"{p}.1.fc_out.0.bias", (2 * D,), f"{p}.1.fc_out.0.bias"))
        else:
            out.append((f"{p}.1.fc_in.fc.weight", (2 * ffn, D), f"{p}.1.fc_in.fc.weight"))
            out.append((f"{p}.1.fc_in.fc.bias", (2 * ffn,), f"{p}.1.fc_in.fc.bias"))
            out.append((f"{p}.1.fc_out.weight", (D, ffn), f"{p}.1.fc_out.weight"))
            out.append((f"{p}.1.fc_out.bias", (D,), f"{p}.1.fc_out.bias"))
    return out


def resnet_blocks():
    """(stage, block, in_ch, mid_ch, out_ch, stride, has_downsample) for ResNetV2 [2,4,6] (resnet.py:152-254)."""
    out, prev = [], 64
    for s, (depth, ch) in enumerate(zip(RESNET_DEPTHS, RESNET_CHANNELS)):
        for i in range(depth):
            out.append((s, i, prev, ch // 4, ch, (2 if s > 0 else 1) if i == 0 else 1, i == 0))
            prev = ch
    return out


def _resnet_keys(p: str) -> List[Tuple[str, Shape, str]]:
    """ResNetV2 state_dict keys.  Every Bottleneck registers its six layers twice (`block_list.N` and the
    nn.Sequential `block.N` built from the same modules, resnet.py:132-141): aliased keys, one storage."""
    keys: List[Tuple[str, Shape, str]] = [
        (f"{p}.stem.0.weight", (64, 1, 7, 7), f"{p}.stem.0.weight"),
        (f"{p}.stem.1.weight", (64,), f"{p}.stem.1.weight"), (f"{p}.stem.1.bias", (64,), f"{p}.stem.1.bias")]
    for s, i, cin, mid, cout, stride, ds in resnet_blocks():
        b = f"{p}.stages.{s}.stage_blocks.{i}"
        if ds:
            keys += [(f"{b}.downsample.conv.weight", (cout, cin, 1, 1), f"{b}.downsample.conv.weight"),
                     (f"{b}.downsample.norm.weight", (cout,), f"{b}.downsample.norm.weight"),
                     (f"{b}.downsample.norm.bias", (cout,), f"{b}.downsample.norm.bias")]
        shapes = [(mid, cin, 1, 1), (mid,), (mid, mid, 3, 3), (mid,), (cout, mid, 1, 1), (cout,)]
        for alias in ("block_list", "block"):
            for j, shp in enumerate(shapes):
                canon = f"{b}.block_list.{j}"
                keys.append((f"{b}.{alias}.{j}.weight", shp, f"{canon}.weight"))
                if j % 2 == 1:
                    keys.append((f"{b}.{alias}.{j}.bias", shp, f"{canon}.bias"))
    return keys


def state_dict_layout(d: Dims) -> List[Tuple[str, Shape, str]]:
    """Every key of the reference OCRModel.state_dict() for these dims, with its shape and
    the canonical (de-aliased) key that owns the storage."""
    D = d.embed_dim
    keys: List[Tuple[str, Shape, str]] = [
        ("encoder.cls_token", (1, 1, D), "encoder.cls_token"),
        ("encoder.pos_embed", (1, d.n_pos, D), "encoder.pos_embed"),
    ]
    if d.embed == "hybrid":
        keys += _resnet_keys("encoder.patch_embed.backbone_net")
        keys += [("encoder.patch_embed.proj.weight", (D, RESNET_CHANNELS[-1], 1, 1), "encoder.patch_embed.proj.weight"),
                 ("encoder.patch_embed.proj.bias", (D,), "encoder.patch_embed.proj.bias")]
    else:
        keys += [("encoder.patch_embed.proj.weight", (D, d.in_channels, d.patch, d.patch), "encoder.patch_embed.proj.weight"),
                 ("encoder.patch_embed.proj.bias", (D,), "encoder.patch_embed.proj.bias")]
    keys += _stack_keys("encoder.attn_layers", enc_kinds(d), D, d.enc_inner, d.enc_ffn)
    keys += [("encoder.norm.weight", (D,), "encoder.norm.weight"),
             ("encoder.norm.bias", (D,), "encoder.norm.bias"),
             ("decoder.net.token_embedding.weight", (d.vocab, D), "decoder.net.token_embedding.weight"),
             ("decoder.net.pos_embedding.embedding.weight", (d.max_len, D),
              "decoder.net.pos_embedding.embedding.weight")]
    keys += _stack_keys("decoder.net.attn_layers", dec_kinds(d), D, d.dec_inner, d.dec_ffn)
    keys += [("decoder.net.norm.weight", (D,), "decoder.net.norm.weight"),
             ("decoder.net.norm.bias", (D,), "decoder.net.norm.bias"),
             ("decoder.net.to_logits.weight", (d.vocab, D), "decoder.net.to_logits.weight"),
             ("decoder.net.to_logits.bias", (d.vocab,), "decoder.net.to_logits.bias")]
    return keys


_LN_KEY = re.compile(r"(layers\.\d+\.0|\.norm|backbone_net\.stem\.1|\.block_list\.[135])\.(weight|bias)$")


def _draw(key: str, shape: Shape, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64((seed & 0xFFFFFFFF) ^ zlib.crc32(key.encode())))
    return rng.standard_normal(shape)


def synth_param(key: str, shape: Shape, seed: int) -> np.ndarray:
    """One parameter, fp32. Scales: matrices 1/sqrt(fan_in); LayerNorm gamma 1+0.1n, beta 0.1n;
    biases 0.05n; embeddings / cls / pos O(0.5) so that a wrong position id is a gross error."""
    n = _draw(key, shape, seed)
    leaf = key.rsplit(".", 1)[-1]
    is_ln = bool(_LN_KEY.search(key))
    if key.endswith(("cls_token", "pos_embed")) or "embedding" in key:
        v = 0.5 * n
    elif is_ln and leaf == "weight":
        v = 1.0 + 0.1 * n
    elif is_ln and leaf == "bias":
        v = 0.1 * n
    elif leaf == "bias":
        v = 0.05 * n
    else:
        fan_in = int(np.prod(shape[1:]))
        v = n / np.sqrt(fan_in)
    return np.ascontiguousarray(v, dtype=np.float32)


def synth_state_dict(d: Dims, seed: int = 0) -> Dict[str, np.ndarray]:
    """Reference-layout state dict (aliased LN keys share one ndarray object)."""
    cache: Dict[str, np.ndarray] = {}
    out: Dict[str, np.ndarray] = {}
    for key, shape, canon in state_dict_layout(d):
        if canon not in cache:
            cache[canon] = synth_param(canon, shape, seed)
        out[key] = cache[canon]
    return out


def synth_images(batch: int, channels: int, h: int, w: int, seed: int = 1234) -> np.ndarray:
    """U[0,1) fp32 NCHW images (host side; fixtures need identical bytes on both machines)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((batch, channels, h, w), dtype=np.float32)
