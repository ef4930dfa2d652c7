"""sampled decode (the reference's default) beyond 128 images, bf16: the persistent launch (K/V form) against launches on ONE row range in the
K/V form and in latent form (TXO_LATENT=1 engines): ms per generate"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672)
for B in [int(x) for x in sys.argv[1:]] or [128, 160, 192, 256]:
    img = torch.rand((B, 3, 224, 672), device="cuda")
    for name, cenv, renv in (("persistent K/V", {}, {"TXO_PERSIST": "1"}), ("launches K/V", {"TXO_LATENT": "0"}, {"TXO_PERSIST": "0"}),
                             ("launches latent", {"TXO_LATENT": "1"}, {"TXO_PERSIST": "0"}), ("default", {}, {})):
        for k, v in cenv.items(): os.environ[k] = v
        m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
        m.load_state_dict(synth.synth_state_dict(d, 0))
        for k in cenv: os.environ.pop(k)
        for k, v in renv.items(): os.environ[k] = v
        for _ in range(2): out = m.generate(img, 256, temp=0.3, decode="sample", seed=5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): out = m.generate(img, 256, temp=0.3, decode="sample", seed=5)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        q = [m._engine.query(i) for i in (0, 2, 3)]
        for k in renv: os.environ.pop(k)
        print(f"sampled bf16 B={B:3d} {name:16s}: {dt*1e3:7.2f} ms = {B/dt:7.1f} img/s  persistent={q[0]} ranges={q[1]} latent={q[2]}", flush=True)
        del m
