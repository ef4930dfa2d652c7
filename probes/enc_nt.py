"""encoder-only A/B of the GEMM epilogues' store policy (TXO_ENC_NT=0 plain / 1 non-temporal / unset = by output size), interleaved"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
import bench

def run(d, B, tag, dtype="bf16"):
    img = torch.rand((B, 3, 224, 672), device="cuda")
    ms = {}
    outs = {}
    for nt in ("0", "1", None):
        if nt is None: os.environ.pop("TXO_ENC_NT", None)
        else: os.environ["TXO_ENC_NT"] = nt
        ms[nt] = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=589)
        ms[nt].load_state_dict(synth.synth_state_dict(d, 0))
    os.environ.pop("TXO_ENC_NT", None)
    for rnd in range(2):
        for nt, m in ms.items():
            for _ in range(3): e = m.encoder(img)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): e = m.encoder(img)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            outs[nt] = e
            tf = bench.enc_flop(d, B, 589) / dt / 1e12
            peak = 2500 if dtype == "bf16" else 157.3
            print(f"{tag} {dtype} B={B} TXO_ENC_NT={nt}: encoder {dt*1e3:8.3f} ms = {tf:7.1f} TFLOP/s = {tf/peak:.3f} of peak", flush=True)
    print("   outputs identical:", bool(torch.equal(outs["0"], outs["1"])))

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "cfg4"): run(Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6), 256, "cfg4")
if which in ("all", "cfg2"):
    run(Dims(canvas=672), 64, "cfg2")
    run(Dims(canvas=672), 256, "cfg2")
if which == "all": run(Dims(canvas=672), 64, "cfg2", "fp32")
