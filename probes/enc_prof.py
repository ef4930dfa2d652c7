import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=1)
B = 256
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
for _ in range(4): m.encoder(img)
torch.cuda.synchronize()
