"""persistent vs launches, latent vs K/V form (engine knobs are read at engine creation)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

def run(B, dtype, latent, T=256, reps=6):
    d = Dims(canvas=672)
    os.environ["TXO_LATENT"] = str(latent)
    m = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=589)
    m.load_state_dict(synth.synth_state_dict(d, 0))
    os.environ.pop("TXO_LATENT")
    img = torch.rand((B, 3, 224, 672), device="cuda")
    res = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["TXO_PERSIST"] = mode
        for _ in range(3): m.generate(img, T)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): out = m.generate(img, T)
        torch.cuda.synchronize(); res.setdefault(mode, []).append((time.perf_counter() - t0) / reps)
        res.setdefault("p" + mode, []).append(m._engine.query(0))
    os.environ.pop("TXO_PERSIST", None)
    p, l = min(res["1"]), min(res["0"])
    print(f"B={B:4d} {dtype} latent={latent}: persistent {p*1e3:8.2f} ms ({B/p:8.1f} img/s) ran={res['p1']} | launches {l*1e3:8.2f} ms ({B/l:8.1f} img/s) | fallbacks {m._engine.query(1)}", flush=True)

if __name__ == "__main__":
    for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64").split(",")]:
        for latent in (0, 1, 0, 1):
            run(B, "bf16", latent)
