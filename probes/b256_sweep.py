"""batch-256 bf16 greedy: row ranges x graph replay (environment knobs are read per generate)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda")
ref = None
for rnd in range(2):
    for lanes in ("-", "1", "2", "3", "4"):
        for graph in ("-", "1"):
            if lanes != "-": os.environ["TXO_LANES"] = lanes
            if graph != "-": os.environ["TXO_GRAPH"] = graph
            for _ in range(2): out = m.generate(img, 256)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): out = m.generate(img, 256)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            os.environ.pop("TXO_LANES", None); os.environ.pop("TXO_GRAPH", None)
            if ref is None: ref = out.clone()
            print(f"B={B} lanes={lanes} graph={graph}: {dt*1e3:7.2f} ms = {B/dt:7.1f} img/s  same_tokens={bool(torch.equal(out, ref))} persistent={m._engine.query(0)}", flush=True)
