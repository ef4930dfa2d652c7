"""hash of sampled tokens (decode='sample', fixed seeds) on both decode paths: two builds of the sampler must print the same line"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
d = Dims(canvas=224)
m = model_from_dims(d, dtype="fp32", max_batch=37)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.from_numpy(synth.synth_images(37, 3, 64, 224, seed=3)).cuda()
h = hashlib.sha256()
for mode in ("1", "0"):
    os.environ["TXO_PERSIST"] = mode
    for seed, temp in ((1, 0.3), (2, 1.0), (3, 3.0)):
        h.update(m.generate(img, 48, temp=temp, decode="sample", seed=seed).cpu().numpy().tobytes())
print("sampled tokens sha256", h.hexdigest()[:16])
