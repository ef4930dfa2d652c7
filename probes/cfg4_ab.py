import sys, os; sys.path.insert(0, "probes")
import pbench
from texocr_amd.config import Dims
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
for b in (256, 64, 128): pbench.run(b, "bf16", reps=2, dims=d)
