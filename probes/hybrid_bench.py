"""bench.py's hybrid_default extra on its own:  python probes/hybrid_bench.py [dtype] [batch]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
class A: height, width, max_len = 224, 672, 256
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
print(json.dumps(bench.hybrid_measurement(dt, A, torch.device("cuda"), B=B), indent=1))
