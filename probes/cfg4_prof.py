import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
B = 256
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
sd = synth.synth_state_dict(d, 0)
m = model_from_dims(d, dtype=sys.argv[1], max_batch=B, max_tokens=589); m.load_state_dict(sd)
img = torch.rand((B, 3, 224, 672), device="cuda")
m.generate(img, 64); torch.cuda.synchronize()
