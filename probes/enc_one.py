import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
import bench
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=1)
B = 256
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
for _ in range(3): m.encoder(img)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): e = m.encoder(img)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
tf = bench.enc_flop(d, B, 589) / dt / 1e12
print(f"ViT-Base encoder B=256: {dt*1e3:8.3f} ms = {tf:7.1f} TFLOP/s = {tf/2500:.3f} of the bf16 peak  (checksum {float(e.double().sum()):.3f})")
