import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=64, max_tokens=589); m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((64, 3, 224, 672), device="cuda")
lat = []
t00 = time.perf_counter()
for i in range(int(sys.argv[1])):
    t0 = time.perf_counter(); m.generate(img, 256); torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
print("elapsed", time.perf_counter() - t00)
print(" ".join(f"{x:.1f}" for x in lat))
