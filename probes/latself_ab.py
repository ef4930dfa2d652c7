"""Self attention in latent form (z history) against the K/V history, engines built with TXO_LATENT_SELF=1 / 0, interleaved in one process:
   greedy batch B (default 256), sampled batch B, beam 5 x 128 images at 224x672.   python probes/latself_ab.py [B]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W, T = 224, 672, 256
d = Dims(canvas=672)
sd = synth.synth_state_dict(d, 0)
def make(env, mb):
    os.environ.update(env)
    m = model_from_dims(d, dtype="bf16", max_batch=mb, max_tokens=d.n_tokens(H, W))
    for k in env: os.environ.pop(k)
    m.load_state_dict(sd); m.eos_token = None
    return m
def ab(tag, run, models, rounds=3, reps=3):
    res = {k: [] for k in models}
    outs = {}
    for rnd in range(rounds):
        for k, m in models.items():
            run(m); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): out = run(m)
            torch.cuda.synchronize(); res[k].append((time.perf_counter() - t0) / reps)
            outs[k] = out
    ks = list(models)
    for k in ks:
        agree = float((outs[ks[0]] == outs[k]).float().mean())
        print(f"{tag:16s} {k:20s} min {min(res[k])*1e3:8.2f} ms  median {statistics.median(res[k])*1e3:8.2f} ms   tokens equal to {ks[0]}: {agree:.4f}", flush=True)

VARIANTS = {"default": {}, "sk0": {"TXO_SK": "0"}, "kvself": {"TXO_LATENT_SELF": "0"}, "kvself+sk0 (r04)": {"TXO_LATENT_SELF": "0", "TXO_SK": "0"},
            "sk4": {"TXO_SK": "4"}, "sk8,rt2": {"TXO_SK_RT": "2"}, "sk8,rt1": {"TXO_SK_RT": "1"}}
img = torch.rand((B, 3, H, W), device="cuda")
ms = {k: make(v, B) for k, v in VARIANTS.items()}
ab(f"greedy B={B}", lambda m: m.generate(img, T), ms)
ab(f"sampled B={B}", lambda m: m.generate(img, T, decode="sample", temp=0.3, seed=7), {k: ms[k] for k in ("default", "kvself+sk0 (r04)")})
del ms
img2 = torch.rand((128, 3, H, W), device="cuda")
mb = {k: make(VARIANTS[k], 640) for k in ("default", "sk0", "kvself", "kvself+sk0 (r04)")}
ab("beam 5 x 128", lambda m: m.generate(img2, T, beam=5), mb, rounds=2, reps=2)
