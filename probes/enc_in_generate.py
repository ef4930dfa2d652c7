"""ViT-Base encoder time INSIDE generate() (marker events, as bench.py's cfg4.encoder_mfma), per environment, engines side by side:
   python probes/enc_in_generate.py "-" "TXO_PP_SB_MB=0" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
sd = synth.synth_state_dict(d, 0)
img = torch.rand((256, 3, 224, 672), device="cuda")
models = []
for e in sys.argv[1:] or ["-"]:
    kv = dict(x.split("=", 1) for x in e.split(",") if "=" in x)
    os.environ.update(kv)
    m = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
    for k in kv: os.environ.pop(k)
    m.load_state_dict(sd); m.eos_token = None
    models.append((e, m))
for rnd in range(2):
    for e, m in models:
        m.generate(img, 32)
        m._engine.profile(True)
        for _ in range(3): m.generate(img, 32)
        torch.cuda.synchronize()
        ems, n = m._engine.profile_read(1)
        m._engine.profile(False)
        tf = bench.enc_flop(d, 256, 589) / (ems * 1e-3) / 1e12
        print(f"[{e}] encoder inside generate: {ems:.3f} ms (avg of {n}) = {tf:.1f} TFLOP/s = {tf/2500:.4f}", flush=True)
