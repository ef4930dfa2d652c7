"""decoder.net() as ONE multi-position pass (txo_decode_prefill) against single-position steps, and the sliding window through
txo_generate: python probes/prefill_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

def clock(f, n):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

for dtype in ("fp32", "bf16"):
    for B, T in ((4, 256), (64, 256)):
        d = Dims(canvas=672)
        m = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=589)
        m.load_state_dict(synth.synth_state_dict(d, 0))
        img = torch.rand((B, 3, 224, 672), device="cuda")
        enc = m.encoder(img)
        x = torch.randint(0, d.vocab - 3, (B, T), device="cuda"); x[:, 0] = d.bos
        one = clock(lambda: m.decoder.net(x, enc=enc), 5)
        os.environ["TXO_NET_STEPWISE"] = "1"
        steps = clock(lambda: m.decoder.net(x, enc=enc), 2)
        os.environ.pop("TXO_NET_STEPWISE")
        print(f"decoder.net {dtype} {B}x{T} (589 encoder tokens): one pass {one*1e3:8.2f} ms | {T} steps {steps*1e3:8.2f} ms | {steps/one:5.1f}x", flush=True)
    # sliding window: a 64-entry positional table, 96 tokens -> 32 tokens beyond the table, each one prefill of 64 rows per image
    d = Dims(canvas=224, max_len=64)
    m = model_from_dims(d, dtype=dtype, max_batch=16)
    m.load_state_dict(synth.synth_state_dict(d, 0))
    img = torch.rand((16, 3, 64, 224), device="cuda")
    m.eos_token = None
    inside = clock(lambda: m.generate(img, 64), 3)
    beyond = clock(lambda: m.generate(img, 96), 3)
    print(f"sliding window {dtype} B=16, table 64: 64 tokens {inside*1e3:.2f} ms, 96 tokens {beyond*1e3:.2f} ms -> {(beyond-inside)/32*1e6:.0f} us per token beyond the table", flush=True)
