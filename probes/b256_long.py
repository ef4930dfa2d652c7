"""batch 256, one process, many rounds: does the slow mode of a fresh process go away with time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((256, 3, 224, 672), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
ms = []
for rnd in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): out = m.generate(img, 256)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 2 * 1e3)
print(" ".join(f"{x:5.1f}" for x in ms), flush=True)
