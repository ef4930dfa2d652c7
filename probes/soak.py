"""Soak: many back-to-back generate() calls on the default (persistent) path; every call must return the same tokens, with
no fall-back to launches.  python probes/soak.py [calls] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=d.n_tokens(224, 672))
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
ref = m.generate(img, 256).clone()
lat = []
bad = 0
for i in range(calls):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.generate(img, 256)
    torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    bad += int(not torch.equal(out, ref))
lat.sort()
print(f"soak B={B}: {calls} calls, mismatching outputs {bad}, fallbacks {m._engine.query(1)}, persistent last {m._engine.query(0)}, "
      f"latency ms p50 {lat[len(lat)//2]*1e3:.2f} p99 {lat[int(len(lat)*0.99)]*1e3:.2f} max {lat[-1]*1e3:.2f}")
