"""Is the batch-256 slow mode the two row ranges landing on ONE hardware queue?  k extra HIP streams are created before the engine:
   python probes/b256_queues.py k"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
k = int(sys.argv[1])
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
keep = []
for i in range(k):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    keep.append(s)
d = Dims(canvas=672)
m = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
m.load_state_dict(synth.synth_state_dict(d, 0))
img = torch.rand((256, 3, 224, 672), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for _ in range(3): out = m.generate(img, 256)
ms = []
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): out = m.generate(img, 256)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 4 * 1e3)
print(f"extra streams {k}: " + " ".join(f"{x:6.2f}" for x in ms) + " ms", flush=True)
