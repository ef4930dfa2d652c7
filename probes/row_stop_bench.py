"""bench.py's row_stop_b256 extra on its own:  python probes/row_stop_bench.py [batch] [target_median]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from texocr_amd.config import Dims
class A: height, width, max_len = 224, 672, 256
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
med = int(sys.argv[2]) if len(sys.argv) > 2 else 90
print(json.dumps(bench.row_stop_measurement(Dims(canvas=672), "bf16", A, torch.device("cuda"), B=B, target_median=med), indent=1))
