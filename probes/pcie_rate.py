"""Headline workload with the images starting in (pinned) HOST memory: the PCIe-inclusive rate the bench contract asks to be noted (never `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=64, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
host = torch.rand((64, 3, 224, 672)).pin_memory()
pageable = torch.rand((64, 3, 224, 672))
dev = host.cuda()
for _ in range(5): m.generate(dev, 256)
def run(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
a = run(lambda: m.generate(dev, 256))
b = run(lambda: m.generate(host.cuda(non_blocking=True), 256))
c = run(lambda: m.generate(pageable.cuda(), 256))
t = run(lambda: m.generate(dev, 256).cpu())
print(f"images resident in HBM        : {a*1e3:7.3f} ms = {64/a:7.1f} img/s")
print(f"from pinned host memory (H2D) : {b*1e3:7.3f} ms = {64/b:7.1f} img/s   (38.5 MB per batch)")
print(f"from pageable host memory     : {c*1e3:7.3f} ms = {64/c:7.1f} img/s")
print(f"resident + tokens back to host: {t*1e3:7.3f} ms = {64/t:7.1f} img/s   (131 KB per batch)")
