"""BASELINE configs[3] (ViT-Base 12L/768d + 6L decoder, B=256, bf16): cross attention in K/V form vs latent form, engines built side by side.
   python probes/cfg4_latent_ab.py [max_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
sd = synth.synth_state_dict(d, 0)
img = torch.rand((256, 3, 224, 672), device="cuda", generator=torch.Generator(device="cuda").manual_seed(4321))
ms = {}
for lat in ("0", "1"):
    os.environ["TXO_LATENT"] = lat
    m = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
    os.environ.pop("TXO_LATENT")
    m.load_state_dict(sd)
    m.eos_token = None
    for lanes in ("2", "1"):
        os.environ["TXO_LANES"] = lanes
        for _ in range(1): out = m.generate(img, T)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m.generate(img, T)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        os.environ.pop("TXO_LANES")
        print(f"cfg4 B=256 T={T} latent={lat} ranges={m._engine.query(2)} latent_used={m._engine.query(3)}: {best*1e3:8.2f} ms = {256/best:7.1f} img/s", flush=True)
    del m
    torch.cuda.empty_cache()
