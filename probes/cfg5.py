import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=896)
k = 5
for B in (16, 128):
    m = model_from_dims(d, dtype="bf16", max_batch=B * k, max_tokens=785); m.load_state_dict(synth.synth_state_dict(d, 0))
    tot_t, tot_n = 0.0, 0
    for W in (224, 448, 672, 896):
        img = torch.rand((B, 3, 224, W), device="cuda")
        m.generate(img, 16, beam=k); torch.cuda.synchronize()
        t0 = time.perf_counter(); t = m.generate(img, 256, beam=k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tot_t += dt; tot_n += B
        print(f"cfg5-shape bf16 beam={k} B={B} 224x{W}: {dt*1e3:.1f} ms = {B/dt:.0f} img/s (steps {t.shape[1]})", flush=True)
    print(f"  mixed widths: {tot_n/tot_t:.0f} img/s", flush=True)
    del m
