"""Where do the persistent launch and the launch-per-stage path first differ? (diagnostic for the bit-identity tests)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import Dims, build
for dtype in ("fp32", "bf16"):
    for B in (1, 5):
        d = Dims(canvas=224)
        d, sd, m = build(d, seed=3, dtype=dtype, max_batch=B)
        g = torch.Generator(device="cuda").manual_seed(77 + B)
        img = torch.rand((B, 3, 64, 224), generator=g, device="cuda")
        outs = []
        for mode in ("1", "0"):
            os.environ["TXO_PERSIST"] = mode
            outs.append(m.generate(img, 40, return_logits=True))
        (tp, lp), (tl, ll) = outs
        diff = (lp - ll).abs().amax(dim=(0, 2)).cpu()
        print(dtype, B, "tokens equal", bool(torch.equal(tp, tl)), "max |dlogit| per position:", " ".join(f"{x:.1e}" for x in diff.tolist()[:12]), flush=True)
