"""persistent launch with the cross attention in latent form (persist.h LATENT) against the launch path in latent form (same tile functions:
   tokens must be identical) and against the other default paths: ms per generate.
   python probes/plat_check.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

Bs = [int(x) for x in sys.argv[1:]] or [256, 192, 128, 96, 64]
d = Dims(canvas=672)


def make(B, env):
    for k, v in env.items(): os.environ[k] = v
    m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
    m.load_state_dict(synth.synth_state_dict(d, 0))
    for k in env: os.environ.pop(k)
    return m


def run(m, img, env, reps=4):
    for k, v in env.items(): os.environ[k] = v
    for _ in range(2): out = m.generate(img, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = m.generate(img, 256)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    q = [m._engine.query(i) for i in (0, 1, 2, 3)]
    for k in env: os.environ.pop(k)
    return out, dt, q


for B in Bs:
    img = torch.from_numpy(synth.synth_images(B, 3, 224, 672, seed=11)).cuda()
    res = {}
    for name, cenv, renv in (("persist latent", {"TXO_PERSIST_LATENT": "1"}, {"TXO_PERSIST": "1"}),
                             ("launch latent 1 range", {"TXO_LATENT": "1", "TXO_PERSIST_LATENT": "0"}, {"TXO_LANES": "1"}),
                             ("persist K/V", {"TXO_PERSIST_LATENT": "0"}, {"TXO_PERSIST": "1"}),
                             ("default (r04 rules)", {"TXO_PERSIST_LATENT": "0"}, {})):
        m = make(B, cenv)
        out, dt, q = run(m, img, renv)
        res[name] = out.cpu()
        print(f"B={B:3d} {name:24s}: {dt*1e3:7.2f} ms = {B/dt:7.1f} img/s  persistent={q[0]} fallbacks={q[1]} ranges={q[2]} latent={q[3]}", flush=True)
        del m
    a, b = res["persist latent"], res["launch latent 1 range"]
    n = min(a.shape[1], b.shape[1])
    print(f"B={B:3d} persist latent == launch latent: {bool(torch.equal(a[:, :n], b[:, :n]))} ({(a[:, :n] == b[:, :n]).float().mean().item():.4f}); "
          f"vs persist K/V agreement {(a[:, :n] == res['persist K/V'][:, :n]).float().mean().item():.4f}", flush=True)
