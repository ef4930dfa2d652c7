"""Does bench.py's row_stop_b256 extra depend on what ran before it in the process?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from texocr_amd.config import Dims
class A: height, width, max_len, dtype = 224, 672, 256, "bf16"
dev = torch.device("cuda")
d = Dims(canvas=672)
def rs(tag):
    r = bench.row_stop_measurement(d, "bf16", A, dev)
    print(tag, "global", r["global_break"]["ms_per_step"], "row", r["row_stop"]["ms_per_step"], "speedup", r["speedup"], flush=True)
rs("fresh      ")
rs("again      ")
r = bench.side_measurement(d, "bf16", 256, A, dev, 2, 4, True, False); print("b256", r["value"], flush=True)
rs("after b256 ")
r = bench.beam_measurement("bf16", A, dev); print("beam", r["value"], flush=True)
rs("after beam ")
d4 = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)
r = bench.side_measurement(d4, "bf16", 256, A, dev, 1, 3, True, True); print("cfg4", r["value"], flush=True)
rs("after cfg4 ")
