import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=896); k = 5; B = 128
m = model_from_dims(d, dtype="bf16", max_batch=B * k, max_tokens=785); m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
m.generate(img, 16, beam=k); torch.cuda.synchronize()
m.generate(img, 256, beam=k); torch.cuda.synchronize()
