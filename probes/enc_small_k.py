"""ViT-Small encoder (config.yml dims: K = 256 / 512 / 1024 GEMMs) at batch 64 / 256: the 256x256 LDS-DMA GEMM (default) against the
   128x128 register-staged one (TXO_GEMM_OLD=1), per-kernel times by HIP events around encode()"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
import bench
d = Dims(canvas=672)
for B in (64, 256):
    img = torch.rand((B, 3, 224, 672), device="cuda")
    for env in ({}, {"TXO_GEMM_OLD": "1"}):
        for k, v in env.items(): os.environ[k] = v
        m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
        m.load_state_dict(synth.synth_state_dict(d, 0))
        for k in env: os.environ.pop(k)
        for _ in range(3): e = m.encoder(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): e = m.encoder(img)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        tf = bench.enc_flop(d, B, 589) / dt / 1e12
        print(f"ViT-Small B={B} {env or 'default'}: encoder {dt*1e3:8.3f} ms = {tf:7.1f} TFLOP/s = {tf/2500:.3f} of bf16 peak", flush=True)
        del m
