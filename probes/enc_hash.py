"""sha1 of the bf16 encoder's output on fixed inputs (two builds of the library must agree bit for bit: TXO_LIB_PATH=... python probes/enc_hash.py)"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
for tag, d, B, H, W in (("config.yml dims 224x672", Dims(canvas=672), 6, 224, 672), ("ViT-Base 224x672", Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=3, dec_heads=12, dec_layers=1), 5, 224, 672),
                        ("config.yml dims 64x208", Dims(canvas=224), 9, 64, 208)):
    m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=d.n_tokens(H, W))
    m.load_state_dict(synth.synth_state_dict(d, 0))
    img = torch.from_numpy(synth.synth_images(B, 3, H, W, seed=5)).cuda()
    e = m.encoder(img)
    print(tag, hashlib.sha1(e.cpu().numpy().tobytes()).hexdigest(), flush=True)
