"""where does the launch path with latent cross attention overtake the persistent launch?  bf16 greedy, 224x672, 256 steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672)
def t(m, img, reps=4):
    for _ in range(2): m.generate(img, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): m.generate(img, 256)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,96,112,128,160,192,256").split(",")]:
    img = torch.rand((B, 3, 224, 672), device="cuda")
    res = {}
    os.environ["TXO_LATENT"] = "0"; mk = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589); mk.load_state_dict(synth.synth_state_dict(d, 0))
    os.environ["TXO_LATENT"] = "1"; ml = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589); ml.load_state_dict(synth.synth_state_dict(d, 0))
    os.environ.pop("TXO_LATENT")
    os.environ["TXO_PERSIST"] = "1"; res["persistent (K/V)"] = t(mk, img); os.environ.pop("TXO_PERSIST")
    for lanes in ("1", "2"):
        os.environ["TXO_LANES"] = lanes
        res[f"launches latent, {lanes} range(s)"] = t(ml, img)
        os.environ["TXO_PERSIST"] = "0"; res[f"launches K/V, {lanes} range(s)"] = t(mk, img); os.environ.pop("TXO_PERSIST")
        os.environ.pop("TXO_LANES")
    print(f"B={B:4d}: " + " | ".join(f"{k} {v*1e3:6.2f} ms ({B/v:6.0f}/s)" for k, v in res.items()), flush=True)
    del mk, ml
