"""Decode A/B of engines BUILT under different environments (knobs the engine reads once, at creation), interleaved in one process:
   python probes/dec_ab.py <what> <B> "-" "TXO_LAT_NW=4" ...      what = greedy | sample | beam5 | cfg4   (bf16, 224x672, 256 positions; cfg4 = greedy at the ViT-Base dims)
Prints min / median ms per generate and whether the tokens equal the first configuration's."""
import os, sys, time, statistics, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

what, B, cfgs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
H, W, T = 224, 672, 256
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6) if what == "cfg4" else Dims(canvas=672)
sd = synth.synth_state_dict(d, 0)
rows = B * 5 if what == "beam5" else B
ms = {}; envs = {}
for c in cfgs:
    kv = dict(x.split("=", 1) for x in c.split(",") if "=" in x)
    os.environ.update(kv)
    m = model_from_dims(d, dtype="bf16", max_batch=rows, max_tokens=d.n_tokens(H, W))
    for k in kv: os.environ.pop(k)
    m.load_state_dict(sd); m.eos_token = None
    ms[c] = m; envs[c] = kv
torch.manual_seed(0)
img = torch.rand((B, 3, H, W), device="cuda")
def run(m):
    if what in ("greedy", "cfg4"): return m.generate(img, T)
    if what == "sample": return m.generate(img, T, decode="sample", temp=0.3, seed=7)
    return m.generate(img, T, beam=5)
res = {c: [] for c in cfgs}; outs = {}
for rnd in range(3):
    for c, m in ms.items():
        os.environ.update(envs[c])          # (the per-call knobs are re-read when the TXO_* environment changes: texocr_amd/model.py)
        run(m); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = run(m)
        torch.cuda.synchronize(); res[c].append((time.perf_counter() - t0) / 3); outs[c] = out
        for k in envs[c]: os.environ.pop(k)
for c in cfgs:
    eq = float((outs[c] == outs[cfgs[0]]).float().mean())
    print(f"{what} B={B} [{c:28s}] min {min(res[c])*1e3:8.2f} ms  median {statistics.median(res[c])*1e3:8.2f} ms  = {B/min(res[c]):7.1f} img/s   tokens equal to the first: {eq:.4f}  sha1 {hashlib.sha1(outs[c].cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
