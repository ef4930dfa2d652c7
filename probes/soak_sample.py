"""Soak of the SAMPLED decode on the persistent launch: the same seed must give the same tokens on every call, no fall-back.
   python probes/soak_sample.py [calls] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=d.n_tokens(224, 672))
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
ref = m.generate(img, 256, temp=0.3, decode="sample", seed=9).clone()
bad = 0
t0 = time.perf_counter()
for i in range(calls):
    out = m.generate(img, 256, temp=0.3, decode="sample", seed=9)
    bad += int(not torch.equal(out, ref))
torch.cuda.synchronize()
print(f"sampled soak B={B}: {calls} calls, mismatching outputs {bad}, fallbacks {m._engine.query(1)}, persistent last {m._engine.query(0)}, "
      f"{(time.perf_counter() - t0) / calls * 1e3:.2f} ms per call")
