import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
N = 589
enc_flop = 2*(N-1)*3*256*768 + 12*(N*(10*768*768 + 6*768*3072) + 4*N*N*768)
sd = synth.synth_state_dict(d, 0)
for dtype in sys.argv[1].split(","):
    m = model_from_dims(d, dtype=dtype, max_batch=B, max_tokens=N); m.load_state_dict(sd)
    img = torch.rand((B, 3, 224, 672), device="cuda")
    m.encoder(img); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.encoder(img); torch.cuda.synchronize(); te = time.perf_counter() - t0
    m.generate(img, 16); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.generate(img, 256); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    m._engine.profile(True); m.generate(img, 32); torch.cuda.synchronize()
    ms, n = m._engine.profile_read(0); m._engine.profile(False)
    by = B * 12 * 2 * N * 64 * (2 if dtype == "bf16" else 4)
    print(f"cfg4 {dtype} B={B}: encoder {te*1e3:.1f} ms = {B*enc_flop/te/1e12:.0f} TFLOP/s | generate(256) {tg*1e3:.0f} ms = {B/tg:.0f} img/s | cross-attn {ms*1e3:.0f} us = {by/ms/1e6:.0f} GB/s", flush=True)
    del m
