"""The reference's default decode (decode='sample': top-k 99, temperature 0.3, one multinomial draw per step) on the benchmark
workload, persistent launch vs launch-per-stage: python probes/sample_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672)
for dtype in ("bf16", "fp32"):
    B = 64
    m = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=589)
    m.load_state_dict(synth.synth_state_dict(d, 0))
    img = torch.rand((B, 3, 224, 672), device="cuda")
    for mode in ("1", "0"):
        os.environ["TXO_PERSIST"] = mode
        for _ in range(3): out = m.generate(img, 256, temp=0.3, decode="sample", seed=5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): out = m.generate(img, 256, temp=0.3, decode="sample", seed=5)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
        print(f"decode='sample' {dtype} B={B} 224x672 T=256: {'persistent launch' if m._engine.query(0) == 1 else 'launch per stage '} {dt*1e3:7.2f} ms = {B/dt:7.1f} images/s", flush=True)
    os.environ.pop("TXO_PERSIST")
