"""A/B of run-time knobs of the persistent decode launch in ONE process (same box, interleaved):
   python probes/envab.py B dtype "TXO_PS_POLL=0" "TXO_PS_POLL=1" "TXO_PS_POLL=1,TXO_PS_STAGGER_US=14"
Each argument is one configuration (comma-separated VAR=value pairs, or "-" for none); prints ms per generate (min / median of rounds)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

B = int(sys.argv[1]); dtype = sys.argv[2]; cfgs = sys.argv[3:]
H, W, T = 224, 672, 256
d = Dims(canvas=672)
m = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=d.n_tokens(H, W))
m.load_state_dict(synth.synth_state_dict(d, 0))
m.eos_token = None if os.environ.get("AB_NO_EOS") else m.eos_token
img = torch.rand((B, 3, H, W), device="cuda")
res = {c: [] for c in cfgs}
ref = None
for rnd in range(4):
    for c in cfgs:
        kv = [] if c == "-" else [x.split("=") for x in c.split(",")]
        for k, v in kv: os.environ[k] = v
        for _ in range(2): out = m.generate(img, T)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = m.generate(img, T)
        torch.cuda.synchronize(); res[c].append((time.perf_counter() - t0) / 5)
        if ref is None: ref = out.clone()
        same = bool(torch.equal(out, ref))
        for k, v in kv: os.environ.pop(k)
        if not same: print("TOKENS DIFFER under", c)
for c in cfgs:
    r = res[c]
    print(f"{c:60s} min {min(r)*1e3:7.2f} ms  median {statistics.median(r)*1e3:7.2f} ms  ({B/min(r):7.1f} img/s)  persistent={m._engine.query(0)} fallbacks={m._engine.query(1)}", flush=True)
