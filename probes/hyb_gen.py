import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims, reference_config
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims.from_config(reference_config())
B = 64
sd = synth.synth_state_dict(d, 0)
for dtype in sys.argv[1].split(","):
    m = model_from_dims(d, dtype=dtype, max_batch=B); m.load_state_dict(sd)
    img = torch.rand((B, 1, 160, 1008), device="cuda")
    for _ in range(2): m.generate(img, 256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m.encoder(img)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): m.generate(img, 256)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
    print(f"hybrid {dtype} B={B}: encoder {te*1e3:.2f} ms, generate {tg*1e3:.1f} ms = {B/tg:.0f} img/s", flush=True)
    del m
