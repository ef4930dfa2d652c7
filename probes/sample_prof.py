"""sampled decode on the launch path for rocprofv3 (probes/prof_py.sh): per-launch time of sample_step_kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TXO_PERSIST"] = "0"
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0)); m.eos_token = None
torch.manual_seed(0)
img = torch.rand((B, 3, 224, 672), device="cuda")
for _ in range(2): m.generate(img, 256, decode="sample", temp=0.3, seed=7)
torch.cuda.synchronize()
