import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
B = int(sys.argv[1]); out = sys.argv[2]; mode = sys.argv[3] if len(sys.argv) > 3 else "greedy"
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
m.eos_token = None
kw = dict(decode="sample", temp=0.3, seed=7) if mode == "sample" else {}
m.generate(img, 256, **kw)
os.environ["TXO_PSTAMPS"] = out
m.generate(img, 256, **kw); torch.cuda.synchronize()
