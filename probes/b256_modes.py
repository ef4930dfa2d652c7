"""batch-256 bf16 greedy, ONE measurement per process (the two row ranges' mode differs between processes):
   python probes/b256_modes.py [batch]      environment: TXO_CU_SPLIT=k, TXO_LANES=n
Prints ms per generate for 3 rounds of 4 and a token hash."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=B, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((B, 3, 224, 672), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for _ in range(3): out = m.generate(img, 256)
ms = []
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): out = m.generate(img, 256)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 4 * 1e3)
h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]
print(f"B={B} split={os.environ.get('TXO_CU_SPLIT','-')} lanes={os.environ.get('TXO_LANES','-')} ranges={m._engine.query(2)}: " + " ".join(f"{x:6.2f}" for x in ms) + f" ms  best {B/min(ms)*1e3:7.1f} img/s  tokens {h}", flush=True)
