import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
m = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0)); m.eos_token = None
img = torch.rand((256, 3, 224, 672), device="cuda")
for _ in range(2): m.generate(img, 32)
torch.cuda.synchronize()
