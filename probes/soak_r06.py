"""Round-6 soak: the new paths back to back in one process -- per-row stop at batch 256 (compaction + replayed steps), the global break, beam
search (tuned stream pair), the persistent launch with stop='row' -- token hashes must not move, nothing may hang."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texocr_amd.config import Dims
from texocr_amd import synth
from texocr_amd.model import model_from_dims
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = Dims(canvas=672)
sd = synth.synth_state_dict(d, 0)
b = sd["decoder.net.to_logits.bias"].copy(); b[d.eos] += 2.05; sd["decoder.net.to_logits.bias"] = b
m = model_from_dims(d, dtype="bf16", max_batch=640, max_tokens=589)
m.load_state_dict(sd)
g = torch.Generator(device="cuda").manual_seed(777)
img = torch.rand((256, 3, 224, 672), generator=g, device="cuda") * torch.linspace(0.4, 1.6, 256, device="cuda")[:, None, None, None]
def h(t): return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10]
ref = {}
t0 = time.time()
for i in range(N):
    cases = {"row256": lambda: m.generate(img, 256, stop="row"), "glob256": lambda: m.generate(img, 256),
             "row64": lambda: m.generate(img[:64], 256, stop="row"), "beam": lambda: m.generate(img[:128], 64, beam=5),
             "row200": lambda: m.generate(img[:200], 256, stop="row"), "sample256row": lambda: m.generate(img, 128, decode="sample", seed=5, temp=0.5, stop="row")}
    for k, f in cases.items():
        out = f()
        hh = h(out)
        if k not in ref: ref[k] = hh; print(k, tuple(out.shape), hh, flush=True)
        assert ref[k] == hh, (i, k, hh, ref[k])
print(f"{N} rounds x {len(ref)} cases stable in {time.time()-t0:.1f} s; compactions of the last row-stop run: {m._engine.query(5)}")
