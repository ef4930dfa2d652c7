"""config-4 dims (768-wide, 12 heads, 6 layers): persistent launch vs launches at one batch size, optional stamps:
   python probes/cfg4_persist.py 256 [stamps_file]"""
import sys, os; sys.path.insert(0, "probes")
import pbench
from texocr_amd.config import Dims
d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
if len(sys.argv) > 2: os.environ["TXO_PSTAMPS"] = sys.argv[2]
pbench.run(int(sys.argv[1]), "bf16", reps=2, dims=d)
