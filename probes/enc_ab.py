"""ViT-Base (BASELINE cfg 4) and config.yml-dims encoders under several environments, interleaved in ONE process:
   python probes/enc_ab.py "-" "TXO_PP_TR=0" "TXO_PP_TR=1" ...     ("-" = default environment)
Prints ms per encode, TFLOP/s and the fraction of the 2.5 PF bf16 peak per environment (best of 3 rounds of 10)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
import bench

def main():
    envs = sys.argv[1:] or ["-"]
    cfgs = [("cfg4 ViT-Base B=256", Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6), 256),
            ("cfg2 config.yml B=64", Dims(canvas=672), 64)]
    for tag, d, B in cfgs:
        img = torch.rand((B, 3, 224, 672), device="cuda")
        sd = synth.synth_state_dict(d, 0)
        models = []
        for e in envs:
            kv = dict(x.split("=", 1) for x in e.split(",") if "=" in x)
            os.environ.update(kv)
            m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
            for k in kv: os.environ.pop(k)
            m.load_state_dict(sd)
            models.append(m)
        best = [1e9] * len(envs)
        ref = None
        for rnd in range(3):
            for i, m in enumerate(models):
                for _ in range(2): out = m.encoder(img)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): out = m.encoder(img)
                torch.cuda.synchronize(); best[i] = min(best[i], (time.perf_counter() - t0) / 10)
                if rnd == 0:
                    if ref is None: ref = out.clone()
                    else: assert torch.equal(out, ref), f"{envs[i]}: encoder output differs from {envs[0]}"
        for e, dt in zip(envs, best):
            tf = bench.enc_flop(d, B, 589) / dt / 1e12
            print(f"{tag} [{e}]: {dt*1e3:8.3f} ms = {tf:7.1f} TFLOP/s = {tf/2500:.4f} of 2.5 PF", flush=True)
        del models

main()
