#!/usr/bin/env python3
"""Writes the input of examples/generate_tiny.c: the tiny fixture (tests/golden/tiny.{json,npz}: a 2-layer 64-d model whose
greedy tokens were captured from the reference) as one flat little-endian file -- engine config, every state_dict tensor
(from the deterministic weight recipe the fixture names), the images and the expected tokens.  numpy only.

  python examples/make_tiny_blob.py OUT.bin
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from texocr_amd.config import Dims  # noqa: E402
from texocr_amd import synth  # noqa: E402


def main(out):
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "tiny.json")))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "tiny.npz"))
    d = Dims(**meta["dims"])
    sd = synth.synth_state_dict(d, meta["weight_seed"])
    B, C, H, W = meta["image_shape"]
    img = synth.synth_images(B, C, H, W, seed=meta["image_seed"])
    toks = gold["tokens"].astype(np.int64)
    ch, cw = d.canvas_hw
    cfg = [ch, cw, 0, d.in_channels, d.embed_dim, d.enc_heads, d.enc_layers, d.dec_heads, d.dec_layers, d.enc_exp, d.dec_exp,
           d.vocab, d.max_len, d.bos, d.eos, d.pad, 0, B, 0]          # txo_config field order; dtype TXO_F32, max_batch B
    with open(out, "wb") as f:
        f.write(b"TXOB")
        f.write(struct.pack("<19i", *cfg))
        f.write(struct.pack("<i", len(sd)))
        for k, v in sd.items():
            kb = k.encode()
            f.write(struct.pack("<i", len(kb))); f.write(kb)
            f.write(struct.pack("<i", v.ndim)); f.write(struct.pack(f"<{v.ndim}q", *v.shape))
            f.write(np.ascontiguousarray(v, dtype="<f4").tobytes())
        f.write(struct.pack("<4i", B, C, H, W)); f.write(np.ascontiguousarray(img, dtype="<f4").tobytes())
        f.write(struct.pack("<2i", meta["max_len"], toks.shape[1])); f.write(np.ascontiguousarray(toks, dtype="<i8").tobytes())
    print(f"wrote {out}: {len(sd)} tensors, images {B}x{C}x{H}x{W}, {toks.shape[1]} expected steps")


if __name__ == "__main__":
    main(sys.argv[1])
