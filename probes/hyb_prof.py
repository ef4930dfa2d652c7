"""default-factory (hybrid) encoder under rocprofv3: python probes/hyb_prof.py [dtype] [batch]   (TXO_BACKBONE_BF16=1 for the bf16 backbone)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd import synth
from texocr_amd.config import Dims, reference_config
from texocr_amd.model import model_from_dims
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = Dims.from_config(reference_config())
H, W = d.canvas_hw
m = model_from_dims(d, dtype=dt, max_batch=B, max_tokens=d.n_tokens(H, W))
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 1, H, W), device="cuda")
for _ in range(4): m.encoder(img)
torch.cuda.synchronize()
