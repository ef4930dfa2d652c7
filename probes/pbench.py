"""quick A/B: persistent launch vs launch-per-stage, B x dtype"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

def run(B, dtype, H=224, W=672, T=256, reps=6, dims=None):
    d = dims or Dims(canvas=672)
    m = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=d.n_tokens(H, W))
    m.load_state_dict(synth.synth_state_dict(d, 0))
    img = torch.rand((B, 3, H, W), device="cuda")
    res = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["TXO_PERSIST"] = mode
        for _ in range(3):
            m.generate(img, T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = m.generate(img, T)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res.setdefault(mode, []).append(dt)
        assert m._engine.query(0) == int(mode), (m._engine.query(0), mode)
    os.environ.pop("TXO_PERSIST", None)
    p, l = min(res["1"]), min(res["0"])
    print(f"B={B:4d} {dtype} D={d.embed_dim}: persistent {p*1e3:8.2f} ms ({B/p:8.1f} img/s) | launches {l*1e3:8.2f} ms ({B/l:8.1f} img/s) | fallbacks {m._engine.query(1)}", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    run(64, "bf16")
    if which == "all":
        run(64, "fp32")
        run(256, "bf16")
        run(8, "bf16")
        run(1, "bf16")
    if os.environ.get("TXO_PSTAMPS"):
        pass
