"""Why is the ViT-Base encode inside generate() ~1.3 ms slower than back-to-back encodes?  Times one encode (CUDA events) after different predecessors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
m = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0)); m.eos_token = None
img = torch.rand((256, 3, 224, 672), device="cuda")
def enc_ms():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); m.encoder(img); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(3): m.encoder(img)
def stat(v): return f"min {min(v):.2f} med {sorted(v)[len(v)//2]:.2f} max {max(v):.2f}"
print("after an encode            :", stat([enc_ms() for _ in range(8)]))
v = []
for _ in range(6): m.generate(img, 64); v.append(enc_ms())
print("after generate(64 steps)   :", stat(v))
v = []
for _ in range(6): m.generate(img, 256); v.append(enc_ms())
print("after generate(256 steps)  :", stat(v))
v = []
for _ in range(6): torch.cuda.synchronize(); time.sleep(0.25); v.append(enc_ms())
print("after 250 ms of idle       :", stat(v))
v = []
for _ in range(6):
    x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize(); del x
    v.append(enc_ms())
print("after a 1 GiB fill         :", stat(v))
