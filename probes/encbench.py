"""encoder-only A/B: bf16 attention variant 2 vs old (TXO_ENC_ATTN_OLD), cfg2 dims B=64 and cfg4 dims B=256"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
import bench

def run(d, B, tag):
    res = {}
    img = torch.rand((B, 3, 224, 672), device="cuda")
    outs = {}
    for old in (False, True):
        if old: os.environ["TXO_ENC_ATTN_OLD"] = "1"
        else: os.environ.pop("TXO_ENC_ATTN_OLD", None)
        m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
        m.load_state_dict(synth.synth_state_dict(d, 0))
        for _ in range(3): e = m.encoder(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): e = m.encoder(img)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        outs[old] = e
        tf = bench.enc_flop(d, B, 589) / dt / 1e12
        print(f"{tag} B={B} {'old' if old else 'v2 '}: encoder {dt*1e3:8.3f} ms = {tf:7.1f} TFLOP/s = {tf/2500:.3f} of bf16 peak", flush=True)
        del m
    os.environ.pop("TXO_ENC_ATTN_OLD", None)
    diff = (outs[False] - outs[True]).abs()
    print(f"   v2 vs old encoder output: max |diff| {float(diff.max()):.4f}, mean {float(diff.mean()):.5f}, scale {float(outs[True].abs().mean()):.3f}")

run(Dims(canvas=672), 64, "cfg2")
run(Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6), 256, "cfg4")
