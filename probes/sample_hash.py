"""hash of sampled tokens (decode='sample', fixed seeds) on both decode paths and several vocabulary sizes (16-byte and scalar row requests, the
LDS form beyond 1024 entries): two builds of the sampler must print the same lines (e.g. a -DTXO_SAMPLER_LDS=1 build against the default one)"""
import os, sys, hashlib, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
for vocab in (1000, 999, 950, 512, 1100):
    d = dataclasses.replace(Dims(canvas=224), vocab=vocab, bos=vocab - 2, eos=vocab - 3, pad=vocab - 1)
    m = model_from_dims(d, dtype="fp32", max_batch=37)
    m.load_state_dict(synth.synth_state_dict(d, 0))
    img = torch.from_numpy(synth.synth_images(37, 3, 64, 224, seed=3)).cuda()
    h = hashlib.sha256()
    for mode in ("1", "0"):
        os.environ["TXO_PERSIST"] = mode
        for seed, temp in ((1, 0.3), (2, 1.0), (3, 3.0)):
            h.update(m.generate(img, 48, temp=temp, decode="sample", seed=seed).cpu().numpy().tobytes())
    print(f"vocab {vocab}: sampled tokens sha256", h.hexdigest()[:16], flush=True)
