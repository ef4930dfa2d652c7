"""Race screen for gemm_pp_kernel: bf16 encoder features with the LDS-DMA GEMM must equal the register-staged GEMM's bit for bit,
over many launches and shapes (the two kernels accumulate in the same order).  python probes/race_screen.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for name, d in [("config.yml dims", Dims(canvas=672)),
                ("ViT-Base 2 layers", Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=2, dec_heads=12, dec_layers=1))]:
    sd = synth.synth_state_dict(d, 3)
    os.environ["TXO_GEMM_OLD"] = "1"
    m_old = model_from_dims(d, dtype="bf16", max_batch=64, max_tokens=589); m_old.load_state_dict(sd)
    del os.environ["TXO_GEMM_OLD"]
    m_new = model_from_dims(d, dtype="bf16", max_batch=64, max_tokens=589); m_new.load_state_dict(sd)
    for (B, H, W) in [(1, 224, 672), (3, 224, 448), (7, 64, 672), (12, 224, 672), (33, 224, 224), (64, 224, 672)]:
        img = torch.rand((B, 3, H, W), device="cuda")
        ref = m_old.encoder(img)
        n_bad = sum(int(not torch.equal(m_new.encoder(img), ref)) for _ in range(iters))
        bad += n_bad
        print(f"{name}: B={B} {H}x{W} (M={B * (1 + H // 16 * (W // 16))}): {iters - n_bad}/{iters} identical", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} mismatches)")
sys.exit(0 if bad == 0 else 1)
