"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI via the
reference-shaped facades, against (a) golden vectors captured from the reference and (b) the CPU oracle on
seeded inputs; plus size-independent properties at the BASELINE size.

Tolerances (north_star): greedy token ids bit-exact, logits within 1e-3 in fp32 mode.  The fp32 assertions
below are much tighter than 1e-3 (observed error ~5e-6; the reference's own step-vs-teacher-forced noise is
4.8e-6).  Token exactness is asserted on fixtures whose smallest top-1/top-2 logit margin (stored in the
fixture) is >= 1e-4; on free seeds it is asserted up to the first step whose oracle margin is < 2e-5."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from texocr_amd import synth
from texocr_amd.config import Dims

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import cpu_ref
    return cpu_ref


def build(meta_or_dims, seed=None, dtype="fp32", max_batch=8, max_tokens=0, latent=None, sd=None, env=None):
    """latent: None = the engine's default choice of the cross-attention form; 0 / 1 = the projected K/V panels / the raw encoder
    rows (csrc/lat_attn.h) on every decode with launches (TXO_LATENT, read once when the engine is created).
    env: further knobs the engine reads once at creation (TXO_LAT_G, TXO_LATENT_SELF, ...)."""
    import os
    from texocr_amd.model import model_from_dims
    if isinstance(meta_or_dims, dict):
        d, seed = Dims(**meta_or_dims["dims"]), meta_or_dims["weight_seed"]
    else:
        d = meta_or_dims
    sd = sd if sd is not None else synth.synth_state_dict(d, seed)
    knobs = dict(env or {})
    if latent is not None:
        knobs["TXO_LATENT"] = str(int(latent))
    os.environ.update(knobs)
    try:
        m = model_from_dims(d, dtype=dtype, max_batch=max_batch, max_tokens=max_tokens)
    finally:
        for k in knobs:
            os.environ.pop(k, None)
    m.load_state_dict(sd)
    return d, sd, m


def images(meta):
    return torch.from_numpy(synth.synth_images(*meta["image_shape"], seed=meta["image_seed"]))


def assert_tokens_exact_up_to_margin(tok, ref_tok, ref_logits, thr=2e-5):
    """tokens must be identical until (per row) the first step whose oracle top1-top2 margin < thr."""
    top2 = ref_logits.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).numpy()
    for b in range(tok.shape[0]):
        small = np.nonzero(margin[b] < thr)[0]
        upto = int(small[0]) if small.size else tok.shape[1]
        assert np.array_equal(tok[b, :upto], ref_tok[b, :upto]), (b, upto)


# ------------------------------------------------------------------------------------------------
# golden fixtures captured from the reference
# ------------------------------------------------------------------------------------------------
def test_tiny_golden_every_interface():
    meta, g = load_golden("tiny")
    d, sd, m = build(meta)
    img = images(meta).cuda()
    enc = m.encoder(img)
    np.testing.assert_allclose(enc.cpu().numpy(), g["enc"], atol=2e-5)
    toks = torch.from_numpy(g["tokens"].astype(np.int64)).cuda()
    prefix = torch.cat([torch.full((2, 1), d.bos, dtype=torch.long, device="cuda"), toks[:, :-1]], 1)
    logits = m.decoder.net(prefix, mask=torch.ones_like(prefix, dtype=torch.bool), enc=enc)   # teacher forced
    np.testing.assert_allclose(logits.cpu().numpy(), g["tf_logits"], atol=3e-5)
    out, step_logits = m.generate(img, meta["max_len"], return_logits=True)
    assert out.dtype == torch.int64 and out.is_cuda
    assert np.array_equal(out.cpu().numpy(), g["tokens"])
    np.testing.assert_allclose(step_logits.cpu().numpy(), g["step_logits"], atol=3e-5)
    # decoder.generate with explicit start tokens (reference call form, decoder.py:77-85)
    start = torch.full((2, 1), d.bos, dtype=torch.long, device="cuda")
    out2 = m.decoder.generate(start_tokens=start, eos_tok=d.eos, max_len=meta["max_len"], temp=0.3, enc=enc)
    assert np.array_equal(out2.cpu().numpy(), g["tokens"])


def test_cfg1_golden_tokens_bit_exact_256_steps():
    meta, g = load_golden("cfg1_b4_224x224")
    assert float(g["margin"].min()) >= 1e-4
    d, sd, m = build(meta)
    img = images(meta).cuda()
    enc = m.encoder(img)
    np.testing.assert_allclose(enc[0].cpu().numpy(), g["enc0"], atol=5e-5)
    np.testing.assert_allclose(enc.double().sum((1, 2)).cpu().numpy(), g["enc_sum"], atol=5e-2)
    toks, logits = m.generate(img, 256, return_logits=True)
    assert toks.shape == (4, 256)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    lg = logits.cpu()
    np.testing.assert_allclose(lg[:2, :16].numpy(), g["logits_first16"], atol=1e-4)
    np.testing.assert_allclose(lg[:2, -4:].numpy(), g["logits_last4"], atol=1e-4)
    v = torch.gather(lg, 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
    assert float((v - torch.from_numpy(g["top5_vals"])).abs().max()) < 1e-3          # north_star bound
    np.testing.assert_allclose(v.numpy(), g["top5_vals"], atol=1e-4)


def test_cfg2_shape_golden():
    meta, g = load_golden("cfg2_b2_224x672")
    d, sd, m = build(meta)
    img = images(meta).cuda()
    enc = m.encoder(img)
    assert enc.shape == (2, 589, 256)
    np.testing.assert_allclose(enc[:, ::8].cpu().numpy(), g["enc_rows"], atol=5e-5)
    toks, logits = m.generate(img, 48, return_logits=True)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    np.testing.assert_allclose(logits[:, :8].cpu().numpy(), g["logits_first8"], atol=1e-4)


@pytest.mark.parametrize("path", ["persistent", "launches"])
def test_cfg2_benchmark_shape_golden_256_steps_inside_batch_64(path, monkeypatch):
    """The benchmark's own shape pinned to the reference for the WHOLE decode: 3x224x672 (589 keys: two cross-attention passes
    together with a self history beyond 128 positions), 256 greedy steps captured from the reference at B=2 -- decoded here as
    rows 41 and 6 of a 64-image batch (rows never interact; decoder.py:115 is the only cross-row operation) on the fp32 engine,
    through the persistent launch and through launches: every token exact (fixture margin >= 1e-4), logits within 1e-3."""
    meta, g = load_golden("cfg2_b2_224x672_t256")
    assert float(g["margin"].min()) >= 1e-4 and g["tokens"].shape == (2, 256)
    d, sd, m = build(meta, max_batch=64, max_tokens=589)
    gen = torch.Generator(device="cuda").manual_seed(21)
    img = torch.rand((64, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[41], img[6] = fix[0], fix[1]
    enc = m.encoder(img[[41, 6]].contiguous())
    np.testing.assert_allclose(enc[:, ::8].cpu().numpy(), g["enc_rows"], atol=5e-5)
    monkeypatch.setenv("TXO_PERSIST", "1" if path == "persistent" else "0")
    toks, logits = m.generate(img, 256, return_logits=True)
    assert m._engine.query(0) == (1 if path == "persistent" else 0) and m._engine.query(1) == 0
    assert toks.shape == (64, 256)
    assert np.array_equal(toks[[41, 6]].cpu().numpy(), g["tokens"])
    lg = logits[[41, 6]].cpu()
    v = torch.gather(lg, 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
    err = float((v - torch.from_numpy(g["top5_vals"])).abs().max())
    print(f"cfg2 t256 ({path}): max |dlogit| over the reference's top-5 of all 2x256 steps {err:.2e}")
    assert err < 1e-3                                                                  # north_star bound
    np.testing.assert_allclose(lg[:, :8].numpy(), g["logits_first8"], atol=1e-4)
    np.testing.assert_allclose(lg[:, -4:].numpy(), g["logits_last4"], atol=1e-4)


def test_position_ids_smaller_than_canvas():
    meta, g = load_golden("posids")
    lc = meta["live_case"]
    d, sd, m = build(lc)
    enc = m.encoder(images(lc).cuda())
    np.testing.assert_allclose(enc.cpu().numpy(), g["enc_32_80_96"], atol=2e-5)


def test_eos_global_break_golden():
    meta, g = load_golden("eos_break")
    d, sd, m = build(meta)
    img = images(meta).cuda()
    for case in meta["cases"]:
        m.eos_token = case["eos"]
        if case["name"] == "eos_is_bos":
            pass
        t = m.generate(img, meta["max_len"])
        assert t.shape[1] == case["n_steps"], case
        assert np.array_equal(t.cpu().numpy(), g[f"tokens_{case['name']}"]), case
    m.eos_token = None
    enc = m.encoder(img)
    start = torch.full((2, 1), d.bos, dtype=torch.long, device="cuda")
    free = m.decoder.generate(start_tokens=start, eos_tok=None, max_len=meta["max_len"], enc=enc)
    assert np.array_equal(free.cpu().numpy(), g["free_tokens"])


def test_sliding_window_matches_reference():
    """max_len > decoder.max_len: the reference slides its window (decoder.py:99-100).  The one-call C entry point (txo_generate)
    follows it -- the first max_length positions through the KV cache, every later token through one multi-position forward of
    its window (txo_decode_prefill's pass) -- on both decode paths, and so do the facades."""
    import os
    meta, g = load_golden("sliding_window")
    d, sd, m = build(meta)
    img = images(meta).cuda()
    assert meta["max_len"] > d.max_len
    t = m.generate(img, meta["max_len"])
    assert t.shape == (1, meta["max_len"])
    assert np.array_equal(t.cpu().numpy(), g["tokens"])
    # straight through the C entry point, with logits, persistent launch and launches
    for mode in ("1", "0"):
        os.environ["TXO_PERSIST"] = mode
        try:
            tk, lg = m._engine.generate(img, meta["max_len"], d.eos, return_logits=True)
        finally:
            os.environ.pop("TXO_PERSIST")
        assert np.array_equal(tk.cpu().numpy(), g["tokens"]) and lg.shape == (1, meta["max_len"], d.vocab)
        assert bool((lg.argmax(-1) == tk).all())
    # the same through decoder.generate with explicit start tokens, and with a longer start prefix (general stepwise form)
    enc = m.encoder(img)
    start = torch.full((1, 1), d.bos, dtype=torch.int64, device="cuda")
    t2 = m.decoder.generate(start, d.eos, meta["max_len"], enc=enc)
    assert np.array_equal(t2.cpu().numpy(), g["tokens"])
    pre = torch.cat([start, t[:, :3]], 1)
    t3 = m.decoder.generate(pre, d.eos, meta["max_len"] - 3, enc=enc)
    assert np.array_equal(t3.cpu().numpy(), g["tokens"][:, 3:])
    with pytest.raises(ValueError, match="slide|position"):
        m._engine.decode_step(d.max_len, torch.zeros(1, dtype=torch.int64, device="cuda"))
    # an eos inside the slid part: the GLOBAL break looks at the whole output (decoder.py:115), window or not
    late = int(g["tokens"][0, d.max_len + 3])
    if late not in g["tokens"][0, :d.max_len + 3].tolist():
        m.eos_token = late
        tb = m.generate(img, meta["max_len"])
        assert tb.shape[1] == d.max_len + 4 and np.array_equal(tb.cpu().numpy(), g["tokens"][:, :d.max_len + 4])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_prefill_equals_cached_steps_and_continues(dtype, monkeypatch):
    """txo_decode_prefill: decoder.net() over a whole prefix in one causal pass.  Against t single-position cached steps of the
    same engine (fp32: < 2e-5, i.e. summation-order noise; bf16: bf16 noise), batch 5 x 40 positions at 224x224 and the benchmark
    width (589 keys); and the K/V cache it leaves behind continues with txo_decode_step exactly like the cache the steps built."""
    d = Dims(canvas=672)
    d, sd, m = build(d, seed=6, dtype=dtype, max_batch=5, max_tokens=589)
    g = torch.Generator(device="cuda").manual_seed(2)
    for W in (224, 672):
        img = torch.rand((5, 3, 224, W), generator=g, device="cuda")
        toks, step_logits = m.generate(img, 41, return_logits=True)
        enc = m.encoder(img)
        prefix = torch.cat([torch.full((5, 1), d.bos, dtype=torch.int64, device="cuda"), toks[:, :39]], 1)   # 40 positions
        lp = m.decoder.net(prefix, enc=enc)                                   # one pass
        monkeypatch.setenv("TXO_NET_STEPWISE", "1")
        ls = m.decoder.net(prefix, enc=enc)                                   # 40 cached steps
        monkeypatch.delenv("TXO_NET_STEPWISE")
        err = float((lp - ls).abs().max())
        print(f"prefill vs steps ({dtype}, width {W}): max |dlogit| {err:.2e}")
        assert lp.shape == (5, 40, d.vocab)
        assert err < (2e-5 if dtype == "fp32" else 0.08)
        if dtype == "fp32":
            assert float((lp - step_logits[:, :40]).abs().max()) < 2e-5 and bool((lp.argmax(-1) == toks[:, :40]).all())
        # continue behind the prefill: position 40 fed with token 39 gives the logits of the free-running decode
        m._engine.decode_begin(enc)
        m._engine.decode_prefill(prefix, want_logits=False)
        l40, _ = m._engine.decode_step(40, toks[:, 39].contiguous())
        assert float((l40 - step_logits[:, 40]).abs().max()) < (2e-5 if dtype == "fp32" else 0.08)


def test_prefill_in_image_chunks_when_the_workspace_is_smaller_than_the_prefix_block(monkeypatch):
    """The multi-position forward runs in the encoder's workspace (max_batch * max_tokens rows); a prefix block with more rows
    than that is processed in image chunks (here 4 images x 20 positions = 80 rows against 4 x 17 = 68: chunks of 3 + 1 images).
    Same logits as single-position steps, and as the reference's teacher-forced fixture for the first two images."""
    meta, g = load_golden("tiny")
    d, sd, m = build(meta, max_batch=4)
    assert m._engine.max_batch * m._engine.max_tokens < 4 * 20
    img2 = images(meta)
    img = torch.cat([img2, img2.flip(0)]).cuda()                       # 4 images
    enc = m.encoder(img)
    toks = torch.from_numpy(g["tokens"].astype(np.int64)).cuda()
    rows = torch.cat([toks, toks.flip(0)])
    extra = torch.tensor([[3, 9, 27]], dtype=torch.long, device="cuda").expand(4, -1)       # causal: later tokens do not touch earlier logits
    prefix = torch.cat([torch.full((4, 1), d.bos, dtype=torch.long, device="cuda"), rows, extra], 1)
    assert prefix.shape == (4, 20)
    one = m.decoder.net(prefix, enc=enc)
    monkeypatch.setenv("TXO_NET_STEPWISE", "1")
    steps = m.decoder.net(prefix, enc=enc)
    assert one.shape == (4, 20, d.vocab)
    assert float((one - steps).abs().max()) < 2e-5
    np.testing.assert_allclose(one[:2, :16].cpu().numpy(), g["tf_logits"][:, :16], atol=3e-5)
    np.testing.assert_allclose(one[2:, :16].flip(0).cpu().numpy(), g["tf_logits"][:, :16], atol=3e-5)


def test_prefill_net_is_an_order_of_magnitude_faster_than_steps(monkeypatch):
    """decoder.net() on 4 x 256 tokens: one multi-position pass against 256 single-position steps driven from Python (measured
    16x in fp32 at 197 encoder tokens, 50-90x at 589: profiles/r03_prefill_bench.txt).  Timing on a shared box is noisy: the best
    of three rounds is taken and the assertion is a loose 5x."""
    import time
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=0, max_batch=4)
    img = torch.from_numpy(synth.synth_images(4, 3, 224, 224, seed=1234)).cuda()
    enc = m.encoder(img)
    prefix = torch.randint(0, d.vocab - 3, (4, 256), device="cuda")
    prefix[:, 0] = d.bos

    def clock(n):
        best, out = None, None
        for _ in range(3):
            m.decoder.net(prefix, enc=enc); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = m.decoder.net(prefix, enc=enc)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            best = dt if best is None else min(best, dt)
        return best, out
    fast, a = clock(5)
    monkeypatch.setenv("TXO_NET_STEPWISE", "1")
    slow, b = clock(1)
    print(f"decoder.net 4x256: one pass {fast * 1e3:.2f} ms, 256 steps {slow * 1e3:.2f} ms ({slow / fast:.1f}x)")
    assert float((a - b).abs().max()) < 5e-5
    assert slow / fast >= 5


def test_hybrid_resnet_embedder_golden():
    """N1: the default factory's model (hybrid ResNetV2 embedder, 1 channel, 160x1008 canvas) against the
    reference fixture: token grid, encoder output, greedy tokens, logits."""
    from texocr_amd.config import reference_config
    from texocr_amd.model import create_model
    meta, g = load_golden("hybrid_b2_32x96")
    d, sd, m = build(meta, max_batch=2, max_tokens=64)
    assert d.embed == "hybrid"
    img = images(meta).cuda()
    enc = m.encoder(img)
    assert enc.shape == (2, 13, 256)
    np.testing.assert_allclose(enc.cpu().numpy(), g["enc"], atol=2e-4)
    toks, logits = m.generate(img, meta["max_len"], return_logits=True)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    assert float(np.abs(logits.cpu().numpy() - g["step_logits"]).max()) < 1e-3
    # create_model(reference config dict) builds the same engine kind
    m2 = create_model(reference_config(), max_batch=2, max_tokens=64)
    m2.load_state_dict(synth.synth_state_dict(d, meta["weight_seed"]))
    assert torch.equal(m2.generate(img, meta["max_len"]), toks)
    with pytest.raises(ValueError):
        m.encoder(torch.zeros(1, 3, 32, 96, device="cuda"))          # single-channel only


def test_hybrid_oracle_wider_image_and_bf16():
    cpu_ref = _oracle()
    from texocr_amd.config import reference_config
    d = Dims.from_config(reference_config())
    d, sd, m = build(d, seed=9, max_batch=3, max_tokens=1 + 4 * 20)
    img = torch.from_numpy(synth.synth_images(3, 1, 64, 320, seed=31))
    sdt = cpu_ref.to_torch_sd(sd)
    enc_ref = cpu_ref.encode(sdt, img, grid_w=d.grid)
    enc = m.encoder(img.cuda())
    assert enc.shape == (3, 81, 256)
    assert float((enc.cpu() - enc_ref).abs().max()) < 5e-4
    ref_t, ref_l = cpu_ref.generate_cached(sdt, img, d.bos, d.eos, 16, collect_logits=True, enc=enc_ref)
    toks, logits = m.generate(img.cuda(), 16, return_logits=True)
    assert_tokens_exact_up_to_margin(toks.cpu().numpy(), ref_t.numpy(), ref_l, thr=1e-4)
    # bf16 mode (r06 policy): the BACKBONE stores and multiplies in fp32 also in the bf16 engine -- with these 45 random conv / GroupNorm
    # layers any 2^-9 perturbation, rounding the input pixels alone, moves the backbone's output by 10-20 %
    # (tests/test_oracle_golden.py) -- so the bf16 engine's encoder output must sit within 2 % of the fp32 ORACLE: what is asserted.
    d2, sd2, mb = build(d, seed=9, dtype="bf16", max_batch=3, max_tokens=1 + 4 * 20)
    encb = mb.encoder(img.cuda()).cpu()
    scale = float(enc_ref.abs().mean())
    rel_fp32 = float((encb - enc_ref).abs().mean()) / scale
    print(f"hybrid bf16 engine (fp32 backbone) encoder output vs fp32 oracle, mean|diff|/mean|ref|: {rel_fp32:.4f}")
    assert rel_fp32 < 0.02, rel_fp32
    # the opt-in bf16 backbone (TXO_BACKBONE_BF16=1) is judged against a CPU emulation that only ROUNDS what that mode stores as bf16
    # (cpu_ref.resnet_backbone(q=bf16_round)): close to the emulation, closer by far than either sits to the fp32 result
    d3, sd3, mq = build(d, seed=9, dtype="bf16", max_batch=3, max_tokens=1 + 4 * 20, env={"TXO_BACKBONE_BF16": "1"})
    encq = mq.encoder(img.cuda()).cpu()
    enc_emu = cpu_ref.encode(sdt, img, grid_w=d.grid, backbone_q=cpu_ref.bf16_round)
    rel_q = float((encq - enc_ref).abs().mean()) / scale
    rel_emu = float((encq - enc_emu).abs().mean()) / scale
    emu_fp32 = float((enc_emu - enc_ref).abs().mean()) / scale
    print(f"opt-in bf16 backbone: engine vs fp32 oracle {rel_q:.4f}, bf16-storage emulation vs fp32 oracle {emu_fp32:.4f}, engine vs emulation {rel_emu:.4f}")
    assert rel_emu < 0.1, rel_emu
    assert rel_emu < 0.5 * rel_q and abs(rel_q - emu_fp32) < 0.5 * emu_fp32


def test_split_fp32_backbone_gemm_against_exact_f32():
    """bf16 engine, hybrid embedder: the fp32 backbone's convolutions run as fp32 operands SPLIT onto the bf16 matrix pipe
    (csrc/gemm_split.h: a = hi + lo, three bf16 MFMAs per product term, ~2^-16 per product).  Against the same engine with the exact-f32 MFMA
    kernel (TXO_BACKBONE_EXACT=1) the 45-layer backbone + ViT output moves by far less than bf16 storage anywhere would (0.1-0.2), and
    stays within the 2 % of the fp32 reference that the mode promises."""
    from texocr_amd.config import reference_config
    cpu_ref = _oracle()
    d = Dims.from_config(reference_config())
    img = torch.from_numpy(synth.synth_images(3, 1, 64, 320, seed=31))
    _, sd, m_split = build(d, seed=9, dtype="bf16", max_batch=3, max_tokens=1 + 4 * 20)
    _, _, m_exact = build(d, seed=9, dtype="bf16", max_batch=3, max_tokens=1 + 4 * 20, env={"TXO_BACKBONE_EXACT": "1"})
    a, b = m_split.encoder(img.cuda()).cpu(), m_exact.encoder(img.cuda()).cpu()
    ref = cpu_ref.encode(cpu_ref.to_torch_sd(sd), img, grid_w=d.grid)
    scale = float(ref.abs().mean())
    rel_ab = float((a - b).abs().mean()) / scale
    rel_a, rel_b = float((a - ref).abs().mean()) / scale, float((b - ref).abs().mean()) / scale
    print(f"hybrid encoder, bf16 engine: split backbone vs exact-f32 backbone {rel_ab:.5f}; vs fp32 oracle: split {rel_a:.5f}, exact {rel_b:.5f}")
    assert rel_ab < 0.01 and rel_a < 0.02 and rel_b < 0.02
    assert not torch.equal(a, b)                                      # (two different kernels did run)


def test_hybrid_default_factory_full_canvas_golden():
    """N1 at its real size: create_model(config.yml) on the full 1x160x1008 canvas (631 tokens), two images, 32 greedy steps captured from
    the reference (tests/golden/hybrid_b2_160x1008, oracle/capture_golden.py: cap_hybrid_full).  fp32: tokens exact, logits < 1e-3;
    bf16 (fp32 backbone): encoder output within 2 % of the reference, tokens equal wherever the reference's margin exceeds the bf16 bound."""
    meta, g = load_golden("hybrid_b2_160x1008")
    d, sd, m = build(meta, max_batch=2)
    assert d.embed == "hybrid" and d.canvas_hw == (160, 1008) and meta["tokens_per_image"] == 631
    img = images(meta).cuda()
    enc = m.encoder(img)
    assert enc.shape == (2, 631, 256)
    np.testing.assert_allclose(enc.cpu().numpy(), g["enc"], atol=5e-4)
    toks, logits = m.generate(img, meta["max_len"], return_logits=True)
    assert float(g["margin"].min()) > 1e-3
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    assert float(np.abs(logits.cpu().numpy() - g["step_logits"]).max()) < 1e-3
    _, _, mb = build(meta, dtype="bf16", max_batch=2)
    encb = mb.encoder(img).cpu().numpy()
    rel = float(np.abs(encb - g["enc"]).mean() / np.abs(g["enc"]).mean())
    tb, lb = mb.generate(img, meta["max_len"], return_logits=True)
    assert rel < 0.02, rel
    tb, lb = tb.cpu().numpy(), lb.cpu().numpy()
    # per row: logits are compared on the common prefix (behind a different token the two decodes see different inputs), and a token may
    # differ only where the reference's own top-1 / top-2 margin is inside twice the deviation measured on that prefix
    worst = 0.0
    for b in range(2):
        diff = np.nonzero(tb[b] != g["tokens"][b])[0]
        k = int(diff[0]) if diff.size else tb.shape[1] - 1
        dl = float(np.abs(lb[b, :k + 1] - g["step_logits"][b, :k + 1]).max())
        worst = max(worst, dl)
        if diff.size:
            assert g["margin"][b, k] < 2 * dl, (b, k, float(g["margin"][b, k]), dl)
    print(f"hybrid full canvas, bf16 engine vs reference: encoder mean-relative {rel:.4f}, max |dlogit| on the common prefixes {worst:.4f}")
    assert worst < 0.06, worst                         # (twice the measured 0.030; the plain-ViT bf16 engine sits at 0.025)


# ------------------------------------------------------------------------------------------------
# oracle on free seeds / shapes (ragged batch, variable width, single image)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,seed", [(1, 224, 224, 5), (3, 64, 448, 6), (5, 224, 672, 7), (2, 16, 16, 8)])
def test_oracle_parity_free_seeds(B, H, W, seed):
    cpu_ref = _oracle()
    d = Dims(canvas=672)
    d, sd, m = build(d, seed=seed)
    img = torch.from_numpy(synth.synth_images(B, 3, H, W, seed=100 + seed))
    sdt = cpu_ref.to_torch_sd(sd)
    enc_ref = cpu_ref.encode(sdt, img)
    enc = m.encoder(img.cuda())
    assert float((enc.cpu() - enc_ref).abs().max()) < 1e-4
    T = 40
    ref_t, ref_l = cpu_ref.generate_cached(sdt, img, d.bos, d.eos, T, collect_logits=True, enc=enc_ref)
    toks, logits = m.generate(img.cuda(), T, return_logits=True)
    assert_tokens_exact_up_to_margin(toks.cpu().numpy(), ref_t.numpy(), ref_l)
    same = (toks.cpu() == ref_t).all(dim=1)
    if bool(same.all()):
        assert float((logits.cpu() - ref_l).abs().max()) < 1e-3


def test_net_equals_generate_cache_equals_recompute():
    """KV-cache == full-prefix recompute: the teacher-forced net() logits for the generated
    prefix equal the step logits of generate(), and both equal the oracle's recompute-mode decoder_net."""
    cpu_ref = _oracle()
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=3)
    img = torch.from_numpy(synth.synth_images(2, 3, 96, 128, seed=17))
    toks, step_logits = m.generate(img.cuda(), 24, return_logits=True)
    enc = m.encoder(img.cuda())
    prefix = torch.cat([torch.full((2, 1), d.bos, dtype=torch.long, device="cuda"), toks[:, :-1]], 1)
    tf = m.decoder.net(prefix, enc=enc)
    assert float((tf - step_logits).abs().max()) < 2e-5
    ref = cpu_ref.decoder_net(cpu_ref.to_torch_sd(sd), prefix.cpu(), enc.cpu())
    assert float((tf.cpu() - ref).abs().max()) < 1e-4


def test_vit_base_width_768():
    """BASELINE config 4 dims at reduced depth: 768-d / 12 heads (3-vector LayerNorm rows, K=768 GEMMs)."""
    cpu_ref = _oracle()
    d = Dims(canvas=224, embed_dim=768, enc_heads=12, enc_layers=2, dec_heads=12, dec_layers=2)
    d, sd, m = build(d, seed=4, max_batch=2)
    img = torch.from_numpy(synth.synth_images(2, 3, 64, 224, seed=21))
    sdt = cpu_ref.to_torch_sd(sd)
    enc_ref = cpu_ref.encode(sdt, img)
    enc = m.encoder(img.cuda())
    assert float((enc.cpu() - enc_ref).abs().max()) < 2e-4
    ref_t, ref_l = cpu_ref.generate_cached(sdt, img, d.bos, d.eos, 12, collect_logits=True, enc=enc_ref)
    toks, logits = m.generate(img.cuda(), 12, return_logits=True)
    assert_tokens_exact_up_to_margin(toks.cpu().numpy(), ref_t.numpy(), ref_l)
    if bool((toks.cpu() == ref_t).all()):
        assert float((logits.cpu() - ref_l).abs().max()) < 1e-3


# ------------------------------------------------------------------------------------------------
# bf16 perf mode: not token-exact by construction; bounded logits error, teacher forced
# ------------------------------------------------------------------------------------------------
def test_bf16_mode_logits_error_bounded():
    meta, g = load_golden("cfg1_b4_224x224")
    d, sd, m = build(meta, dtype="bf16")
    img = images(meta).cuda()
    enc = m.encoder(img)
    assert float(np.abs(enc[0].cpu().numpy() - g["enc0"]).max()) < 0.15
    toks = torch.from_numpy(g["tokens"].astype(np.int64)).cuda()
    prefix = torch.cat([torch.full((4, 1), d.bos, dtype=torch.long, device="cuda"), toks[:, :-1]], 1)
    tf = m.decoder.net(prefix[:, :64], enc=enc).cpu()
    ref_top = torch.from_numpy(g["top5_vals"][:, :64])
    got_top = torch.gather(tf, 2, torch.from_numpy(g["top5_ids"][:, :64].astype(np.int64)))
    err = float((got_top - ref_top).abs().max())
    agree = float((tf.argmax(-1) == toks[:, :64].cpu()).float().mean())
    print(f"cfg1 bf16 vs reference fixture, teacher-forced 4x64: max |dlogit| over the reference's top-5 {err:.4f}, top-1 agreement {agree:.4f}")
    # measured on MI355X: max |dlogit| ~0.04 on logits spanning about [-3, 3], agreement ~0.98 (disagreements sit on the
    # thinnest margins of this random-weight model: the fixture's smallest top-1/top-2 margin is 1e-4)
    # bounds = ~1.5x the measurements (r03, the one-pass decoder forward: 0.023 / 1.000; r02, single-position steps: 0.029-0.04 / 0.98-0.996)
    assert err < 0.06, err
    assert agree > 0.97, agree                               # teacher-forced top-1 agreement


# ------------------------------------------------------------------------------------------------
# BASELINE size (batch 64, 3x224x672, 256 steps): size-independent properties
# ------------------------------------------------------------------------------------------------
def test_full_size_batch_independence_and_determinism():
    d = Dims(canvas=672)
    d, sd, m = build(d, seed=0, max_batch=64, max_tokens=589)
    g = torch.Generator(device="cuda").manual_seed(1234)
    img = torch.rand((64, 3, 224, 672), generator=g, device="cuda")
    t1 = m.generate(img, 256)
    t2 = m.generate(img, 256)
    assert t1.shape == (64, 256)                              # eos never fires for every row -> full length
    assert torch.equal(t1, t2)                                # deterministic
    # rows do not interact: decoding images 10..12 alone gives the same rows (any divergence must sit on a
    # razor-thin margin, which fp32 batch-invariant kernels do not produce)
    sub = m.generate(img[10:13].contiguous(), 256)
    assert torch.equal(sub, t1[10:13])
    # a permutation of the batch permutes the output
    perm = torch.randperm(64, device="cuda", generator=g)
    t3 = m.generate(img[perm].contiguous(), 256)
    assert torch.equal(t3, t1[perm])
    assert int(t1.min()) >= 0 and int(t1.max()) < d.vocab
    # opt-in execution modes must give the same tokens: hipGraph replay of the step, two concurrent lanes, fused self QKV
    import os
    for env in ({"TXO_GRAPH": "1"}, {"TXO_GRAPH": "1", "TXO_LANES": "2"}, {"TXO_LANES": "4"}):
        os.environ.update(env)
        try:
            assert torch.equal(m.generate(img, 256), t1), env
        finally:
            for k in env:
                os.environ.pop(k)
    # the fused-QKV self-attention variant (read at engine creation) is a different kernel: same tokens up to thin margins
    os.environ["TXO_SELF_FUSED"] = "1"
    try:
        _, _, mfu = build(d, seed=0, max_batch=8, max_tokens=589)
    finally:
        os.environ.pop("TXO_SELF_FUSED")
    tf = mfu.generate(img[:8].contiguous(), 64)
    assert float((tf == t1[:8, :64]).float().mean()) > 0.98
    # bucketed variable-width input (BASELINE config 5's input shape): per-image rows equal the fixed-width result
    from texocr_amd.dist import generate_bucketed
    mixed = [img[0], img[1, :, :, :448].contiguous(), img[2], img[3, :, :, :448].contiguous()]
    rows = generate_bucketed(lambda b: m.generate(b, 32), mixed, max_batch=64)
    assert torch.equal(torch.stack([rows[0], rows[2]]), m.generate(img[[0, 2]].contiguous(), 32))
    assert rows[1].shape == (32,) and torch.equal(rows[1], m.generate(mixed[1][None].contiguous(), 32)[0])


# ------------------------------------------------------------------------------------------------
# error behaviour, weight loading, sampling mode, collective smoke
# ------------------------------------------------------------------------------------------------
def test_errors_and_state_dict_checks():
    from texocr_amd.model import model_from_dims
    d = Dims(canvas=64, embed_dim=64, enc_heads=1, enc_layers=1, dec_heads=1, dec_layers=1, vocab=32, max_len=8,
             bos=30, eos=29, pad=31)
    sd = synth.synth_state_dict(d, 1)
    m = model_from_dims(d, max_batch=2)
    # as in the reference, a freshly created model is usable (default-initialised parameters)
    assert bool(torch.isfinite(m.encoder(torch.rand(1, 3, 32, 32, device="cuda"))).all())
    bad = dict(sd)
    bad["encoder.attn_layers.layers.1.0.weight"] = sd["encoder.attn_layers.layers.0.0.weight"] + 1
    with pytest.raises(ValueError, match="shares ONE LayerNorm"):
        m.load_state_dict(bad)
    m = model_from_dims(d, max_batch=2)
    miss = {k: v for k, v in sd.items() if k != "decoder.net.to_logits.bias"}
    with pytest.raises(RuntimeError, match="(?i)missing"):
        m.load_state_dict(miss)
    m = model_from_dims(d, max_batch=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})     # torch tensors, aliases included
    for shape in [(1, 1, 32, 32), (1, 3, 30, 32), (1, 3, 32, 80), (3, 3, 32, 32)]:
        with pytest.raises(ValueError):
            m.encoder(torch.zeros(*shape, device="cuda"))
    with pytest.raises(ValueError, match="CUDA"):
        m.encoder(torch.zeros(1, 3, 32, 32))
    assert m.generate(torch.zeros(1, 3, 32, 32, device="cuda"), 9).shape == (1, 9)   # beyond max_length: sliding window
    with pytest.raises(ValueError, match="enc"):
        m.decoder.net(torch.zeros(1, 2, dtype=torch.long, device="cuda"))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 32, 32, device="cuda"), torch.zeros(1, 4, dtype=torch.long))
    out = m.generate(torch.rand(2, 3, 32, 48, device="cuda"), 8)
    assert out.shape == (2, 8)


def test_sampling_mode_support_reproducibility_and_distribution():
    """N2: the reference's sampler (top-k 99 -> softmax(/0.3) -> multinomial, decoder.py:104-108) on the device.
    A different RNG stream than torch.multinomial, so parity is statistical: support, reproducibility, and a
    chi-square test of 4096 first-step draws against the oracle's sampling distribution."""
    cpu_ref = _oracle()
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=0, max_batch=64)
    img1 = torch.from_numpy(synth.synth_images(1, 3, 64, 64, seed=2))
    img = img1.cuda().expand(2, -1, -1, -1).contiguous()
    a = m.generate(img, 12, temp=0.3, decode="sample", seed=7)
    b = m.generate(img, 12, temp=0.3, decode="sample", seed=7)
    c = m.generate(img, 12, temp=0.3, decode="sample", seed=8)
    assert torch.equal(a, b) and a.shape == (2, 12) and not torch.equal(a, c)
    g = torch.Generator(device="cuda").manual_seed(7)
    assert torch.equal(m.generate(img, 12, temp=0.3, decode="sample", generator=g), a)
    # every sampled token lies in the reference's top-k support (k = int(0.1 * 1000) = 99, utils.py:85-91)
    enc = m.encoder(img)
    prefix = torch.cat([torch.full((2, 1), d.bos, dtype=torch.long, device="cuda"), a[:, :-1]], 1)
    logits = m.decoder.net(prefix, enc=enc)
    top = logits.topk(99, dim=-1).indices
    assert bool((top == a[:, :, None]).any(-1).all())
    # temperature -> 0 degenerates to greedy
    assert torch.equal(m.generate(img, 12, temp=1e-4, decode="sample", seed=3), m.generate(img, 12))
    # distribution of the first token: 64 identical rows x 64 seeds, temp 1.0 to spread the mass
    big = img1.cuda().expand(64, -1, -1, -1).contiguous()
    draws = torch.cat([m.generate(big, 1, temp=1.0, decode="sample", seed=1000 + s)[:, 0] for s in range(64)]).cpu()
    step0 = cpu_ref.generate_cached(cpu_ref.to_torch_sd(sd), img1, d.bos, d.eos, 1, collect_logits=True)[1][:, 0]
    probs = cpu_ref.sample_probs(step0, 1.0)[0].double()
    assert int((probs > 0).sum()) == 99
    n = draws.numel()
    counts = torch.bincount(draws, minlength=d.vocab).double()
    assert float(counts[probs == 0].sum()) == 0                      # nothing outside the support
    exp = probs * n
    keep = exp >= 5
    chi = float((((counts - exp) ** 2 / exp)[keep]).sum() + (counts[~keep].sum() - exp[~keep].sum()) ** 2 / max(float(exp[~keep].sum()), 1e-9))
    dof = int(keep.sum())
    assert chi < dof + 6 * (2 * dof) ** 0.5, (chi, dof)             # ~6 sigma of the chi-square law


def test_non_finite_input_policy():
    """Weights: NaN / inf refused when they are uploaded (TXO_E_INVALID naming the key).  Pixels: a NaN pixel makes the rows of its own image
    unspecified -- token ids still inside the vocabulary, no fault -- and changes nothing for the other images of the batch (rows never
    interact; the reference would return NaN logits for that image).  Token ids outside the vocabulary: IndexError like nn.Embedding."""
    d = Dims(canvas=224)
    sd = synth.synth_state_dict(d, 0)
    for dtype, B in (("fp32", 4), ("bf16", 4), ("bf16", 130)):           # persistent launch (fp32, bf16) and the launch path (130 rows)
        _, _, m = build(d, sd=sd, dtype=dtype, max_batch=B)
        img = torch.rand((B, 3, 64, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        clean = m.generate(img, 24).cpu()
        bad = img.clone()
        bad[1, 0, 5, 7] = float("nan")
        bad[2, 1, 9, 3] = float("inf")
        for kw in ({}, {"decode": "sample", "seed": 5, "temp": 0.7}, {"beam": 2} if B * 2 <= 8 else {}):
            if kw.get("beam"):
                _, _, mb = build(d, sd=sd, dtype=dtype, max_batch=B * 2)
                ref, got = mb.generate(img, 24, **kw).cpu(), mb.generate(bad, 24, **kw).cpu()
            else:
                ref, got = (clean if not kw else m.generate(img, 24, **kw).cpu()), m.generate(bad, 24, **kw).cpu()
            assert got.shape[0] == B and int(got.min()) >= 0 and int(got.max()) < d.vocab, (dtype, B, kw)
            keep = [i for i in range(B) if i not in (1, 2)]
            n = min(ref.shape[1], got.shape[1])
            assert torch.equal(got[keep, :n], ref[keep, :n]), (dtype, B, kw)
        assert torch.isfinite(m.encoder(bad)[keep]).all()
    bad_sd = dict(sd)
    w = sd["decoder.net.to_logits.weight"].copy()
    w[3, 4] = np.nan
    bad_sd["decoder.net.to_logits.weight"] = w
    with pytest.raises(ValueError, match="non-finite value in weight decoder.net.to_logits.weight"):
        build(d, sd=bad_sd)[2].generate(torch.rand(1, 3, 32, 32, device="cuda"), 2)
    _, _, m = build(d, sd=sd, max_batch=2)
    enc = m.encoder(torch.rand(2, 3, 32, 32, device="cuda"))
    with pytest.raises(IndexError):
        m.decoder.net(torch.tensor([[d.bos, d.vocab]] * 2, device="cuda"), enc=enc)
    with pytest.raises(IndexError):
        m.decoder.generate(torch.tensor([[-1]] * 2, device="cuda"), d.eos, 4, enc=enc)
    # straight through the C ABI an id outside the table is forced into it (no fault, finite logits)
    m._engine.decode_begin(enc)
    lg, _ = m._engine.decode_step(0, torch.tensor([d.vocab + 5, -7], device="cuda"))
    assert torch.isfinite(lg).all()


def test_world1_rccl_collectives_really_run():
    """One-rank "nccl" (= RCCL) group on the one GPU a test box has: every collective of texocr_amd/dist.py is ISSUED (force_collective) and
    checked to have gone through torch.distributed.all_gather_into_tensor with an output buffer that is not its input -- the token
    gather, the logits gather north_star names, and the bucketed beam-search gather of BASELINE configs[4].  No 1 -> 8 GPU curve exists
    for this repository (no multi-GPU hardware in the pool): world sizes > 1 are covered by the gloo tests in tests/test_host_cpu.py."""
    import os
    import torch.distributed as dist
    from texocr_amd import dist as tdist
    from texocr_amd.dist import sharded_generate, sharded_generate_bucketed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = []
    real = dist.all_gather_into_tensor

    def spy(out, inp, group=None, async_op=False):
        calls.append((out.data_ptr(), inp.data_ptr(), tuple(out.shape), tuple(inp.shape), out.dtype))
        return real(out, inp, group=group, async_op=async_op)
    tdist.dist.all_gather_into_tensor = spy
    try:
        assert dist.get_backend() == "nccl"
        d = Dims(canvas=64, embed_dim=64, enc_heads=1, enc_layers=1, dec_heads=1, dec_layers=1, vocab=32, max_len=8,
                 bos=30, eos=29, pad=31)
        d, sd, m = build(d, seed=1, max_batch=20)
        img = torch.rand(4, 3, 32, 32, device="cuda")
        want, want_lg = m._engine.generate(img, 8, d.eos, return_logits=True)
        # 1. token ids
        got = sharded_generate(lambda x, n: m._engine.generate(x, n, None), img, 8, eos=d.eos, bos=d.bos, force_collective=True)
        assert torch.equal(got, want)
        assert len(calls) == 1 and calls[0][0] != calls[0][1] and calls[0][2] == (4, 8) and calls[0][4] == torch.int64
        # 2. tokens + per-step logits (north_star: "all-gather of logits")
        got, lg = sharded_generate(lambda x, n: m._engine.generate(x, n, None, return_logits=True), img, 8, eos=d.eos, bos=d.bos,
                                   gather_logits=True, force_collective=True)
        assert torch.equal(got, want) and torch.equal(lg, want_lg[:, :got.shape[1]])
        assert len(calls) == 3 and calls[2][2] == (4, 8, d.vocab) and calls[2][4] == torch.float32 and calls[2][0] != calls[2][1]
        # 3. bucketed widths, greedy and beam search (BASELINE configs[4]): one gather for all buckets
        imgs = [torch.rand(3, 32, w, device="cuda") for w in (32, 48, 32, 64, 48, 32)]
        ref_rows = tdist.generate_bucketed(lambda x: m.generate(x, 8), imgs, max_batch=4)
        rows = sharded_generate_bucketed(lambda x, n: m._engine.generate(x, n, None), imgs, 8, eos=d.eos, bos=d.bos, max_batch=4,
                                         force_collective=True)
        assert len(calls) == 4 and calls[3][0] != calls[3][1]
        assert all(torch.equal(a, b) for a, b in zip(rows, ref_rows))
        beam_ref = tdist.generate_bucketed(lambda x: m.generate(x, 8, beam=3), imgs, max_batch=4)
        beam_rows = sharded_generate_bucketed(lambda x, n: m.generate(x, n, beam=3), imgs, 8, eos=d.eos, bos=d.bos, max_batch=4, beam=True,
                                              force_collective=True)
        assert len(calls) == 5 and calls[4][0] != calls[4][1]
        assert all(torch.equal(a, b) for a, b in zip(beam_rows, beam_ref))
    finally:
        tdist.dist.all_gather_into_tensor = real
        dist.destroy_process_group()


def test_wrapper_end_to_end(tmp_path):
    """N3: TeXOCRWrapper(config)(PIL image) -> (tokens, LaTeX string): tokenizer file, checkpoint file in the
    reference's save_checkpoint layout (utils.py:52-61) with a shorter positional table (ocr_model.py:84-88)."""
    import os
    from PIL import Image
    from texocr_amd.config import reference_config
    from texocr_amd.wrapper import TeXOCRWrapper
    cfg = reference_config(max_length=256)
    d = Dims.from_config({**cfg, "max_length": 96})
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(d, 3).items()}
    ckpt = tmp_path / "checkpoint_e1.pth"
    torch.save({"model_state_dict": sd, "optimizer_state_dict": {}, "epoch": 1}, ckpt)
    import json
    from texocr_amd.tokenizer import RegExTokenizer
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_vocab_1k.json")))
    RegExTokenizer.from_tables(v["vocab_size"], v["special_tokens"], v["merges"]).save(str(tmp_path / "vocab.txt"))
    cfg["tokenizer_path"] = str(tmp_path / "vocab.txt")
    cfg["model_path"] = str(ckpt)
    w = TeXOCRWrapper(cfg)
    assert w.model.decoder.max_len == 96                      # taken from the checkpoint's positional table
    img = Image.new("RGB", (200, 40), (255, 255, 255))
    for x in range(20, 180, 7):
        img.putpixel((x, 20), (0, 0, 0))
    toks, text = w(img, max_len=20, decode="greedy")
    assert len(toks) == 19 and isinstance(text, str)
    toks2, text2 = w(img, max_len=20, decode="greedy")
    assert toks2 == toks and text2 == text
    toks3, _ = w(img, max_len=20, temp=0.3, seed=5)           # the reference's default decode: sampling
    assert len(toks3) == 19


def test_wrapper_matches_reference_wrapper_fixture(tmp_path):
    """N3 pinned: the reference's TeXOCRWrapper.__call__ (ocr_model.py:94-110) was run on a drawn image through the default
    factory (hybrid embedder); this build's wrapper must return the same tokens and the same LaTeX string, inside the
    positional table and beyond it (sliding window, max_len 40 > max_length 24)."""
    import json, os
    from PIL import Image
    from texocr_amd.config import reference_config
    from texocr_amd.tokenizer import RegExTokenizer
    from texocr_amd.wrapper import TeXOCRWrapper
    meta, g = load_golden("wrapper_160x48")
    d = Dims(**meta["dims"])
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(d, meta["weight_seed"]).items()}
    ckpt = tmp_path / "checkpoint_e1.pth"
    torch.save({"model_state_dict": sd, "optimizer_state_dict": {}, "epoch": 1}, ckpt)
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenizer_vocab_1k.json")))
    RegExTokenizer.from_tables(v["vocab_size"], v["special_tokens"], v["merges"]).save(str(tmp_path / "vocab.txt"))
    cfg = reference_config(max_length=256)                    # the checkpoint's positional table (24) overrides this
    cfg["tokenizer_path"], cfg["model_path"] = str(tmp_path / "vocab.txt"), str(ckpt)
    w = TeXOCRWrapper(cfg)
    assert w.model.decoder.max_len == d.max_len == 24
    img = Image.fromarray(g["pixels"])
    for case in meta["cases"]:
        toks, text = w(img, max_len=case["max_len"], decode="greedy")
        assert toks == g[f"tokens_{case['name']}"].tolist(), case["name"]
        assert text == case["text"]


def test_custom_ops_and_module_state_dict():
    """The boundary as north_star words it: torch.ops.texocr.* over the C ABI, and OCRModel as an nn.Module whose
    state_dict() has the reference's key layout (aliased shared-LayerNorm keys included) and round-trips."""
    from texocr_amd.model import model_from_dims
    meta, g = load_golden("tiny")
    d, sd, m = build(meta)
    assert isinstance(m, torch.nn.Module) and isinstance(m.encoder, torch.nn.Module) and isinstance(m.decoder.net, torch.nn.Module)
    own = m.state_dict()
    assert list(own.keys()) == [k for k, _, _ in synth.state_dict_layout(d)] == [k for k, _, _ in meta["state_dict_layout"]]
    for k, v in own.items():
        assert v.is_cuda and np.array_equal(v.cpu().numpy(), sd[k]), k
    # aliases are ONE parameter (attention.py:200,221): parameters() counts it once
    assert own["encoder.attn_layers.layers.3.0.weight"].data_ptr() == own["encoder.attn_layers.layers.0.0.weight"].data_ptr()
    n_unique = sum(int(np.prod(s)) for k, s, c in synth.state_dict_layout(d) if k == c)
    assert sum(p.numel() for p in m.parameters()) == n_unique
    # the operators, called directly
    img = images(meta).cuda()
    eid = m._engine.id
    enc = torch.ops.texocr.encode(img, eid)
    np.testing.assert_allclose(enc.cpu().numpy(), g["enc"], atol=2e-5)
    toks, n, logits = torch.ops.texocr.generate(img, eid, 16, d.eos, True)
    assert n.device.type == "cpu" and int(n) == 16 and toks.shape == (2, 16) and logits.shape == (2, 16, d.vocab)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    np.testing.assert_allclose(logits.cpu().numpy(), g["step_logits"], atol=3e-5)
    toks2, n2, lg2 = torch.ops.texocr.generate_from_enc(enc, eid, 16, -1, False)
    assert torch.equal(toks2, toks) and lg2.shape[0] == 0
    torch.ops.texocr.decode_begin(enc, eid)
    ref = torch.from_numpy(g["tokens"].astype(np.int64)).cuda()
    tok = torch.full((2,), d.bos, dtype=torch.int64, device="cuda")
    for t in range(4):
        lg, nxt = torch.ops.texocr.decode_step(tok, eid, t, 2, True)
        np.testing.assert_allclose(lg.cpu().numpy(), g["step_logits"][:, t], atol=3e-5)
        assert torch.equal(nxt, ref[:, t])
        tok = nxt
    with pytest.raises((ValueError, RuntimeError)):
        torch.ops.texocr.encode(img.cpu(), eid)
    # state_dict round trip into a second model; in-place edits need sync_weights()
    m2 = model_from_dims(d, max_batch=8)
    assert not torch.equal(m2.generate(img, 16), toks)            # default-initialised weights: another model
    res = m2.load_state_dict(m.state_dict())
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m2.generate(img, 16), toks)
    with torch.no_grad():
        m2.decoder.net.to_logits.bias[5] += 100.0
    m2.sync_weights()
    assert bool((m2.generate(img, 4) == 5).all())


@pytest.mark.parametrize("latent", [None, 1], ids=["kv_form", "latent_form"])
@pytest.mark.parametrize("vocab", [1000, 1100], ids=["rows_in_registers", "general_path"])
def test_beam_search_extension(vocab, latent):
    """BASELINE config 5 asks for beam search; the reference has none, so parity is anchored at
    beams=1 == greedy, plus agreement with the oracle's independent CPU restatement of the same definition.
    Both forms of beam_select_kernel (csrc/step.h): an image's k rows in registers up to 1024 entries, the general path beyond."""
    cpu_ref = _oracle()
    d = Dims(canvas=224, vocab=vocab)
    # latent_form: the cross attention against the raw encoder rows, an image's k beams as ONE row of k * heads heads (16 per tile)
    d, sd, m = build(d, seed=0, max_batch=12, latent=latent)
    img = torch.from_numpy(synth.synth_images(3, 3, 64, 96, seed=41))
    m.eos_token = None
    greedy = m.generate(img.cuda(), 24)
    assert torch.equal(m.generate(img.cuda(), 24, beam=1), greedy)
    assert m._engine.query(3) == (1 if latent else 0)
    sdt = cpu_ref.to_torch_sd(sd)
    enc = cpu_ref.encode(sdt, img)
    for k, eos in ((4, None), (3, int(greedy[0, 5]))):
        m.eos_token = eos
        toks, scores = m.generate(img.cuda(), 24, beam=k, return_beams=True)
        ref_t, ref_s = cpu_ref.beam_search_cached(sdt, enc, d.bos, eos, 24, k)
        assert toks.shape[:2] == (3, k) and toks.shape[2] == ref_t.shape[2]
        np.testing.assert_allclose(scores.cpu().numpy(), ref_s.numpy(), atol=2e-3)
        assert bool((scores[:, :-1] >= scores[:, 1:]).all())                  # best first
        assert torch.equal(toks.cpu(), ref_t), (k, eos)
        best = m.generate(img.cuda(), 24, beam=k)
        assert torch.equal(best, toks[:, 0])
        assert torch.equal(m.generate(img.cuda(), 24, beam=k), best)          # deterministic
        # two row ranges on two streams (the default from 256 beam rows on): ranges are whole images, slots range-local -> the same bits
        import os
        os.environ["TXO_LANES"] = "2"
        try:
            toks2, scores2 = m.generate(img.cuda(), 24, beam=k, return_beams=True)
            assert m._engine.query(2) == 2
        finally:
            os.environ.pop("TXO_LANES")
        assert torch.equal(toks2, toks) and torch.equal(scores2, scores)
    with pytest.raises(ValueError):
        m.generate(img.cuda(), 24, beam=9)
    with pytest.raises(ValueError):
        m.generate(torch.rand(5, 3, 64, 96, device="cuda"), 24, beam=3)       # 15 rows > max_batch 12


def test_beam_search_two_row_ranges_at_300_rows_bit_identical_to_one():
    """From 256 beam rows on, beam search decodes two row ranges (whole images each) on two streams, in bf16 with the latent cross
    attention whose tiles take up to 16 heads = two beams of an image: same tokens and scores as ONE range."""
    import os
    d = Dims(canvas=224, max_len=16)
    _, _, m = build(d, seed=9, dtype="bf16", max_batch=300)
    img = torch.from_numpy(synth.synth_images(60, 3, 32, 64, seed=17)).cuda()
    m.eos_token = None
    t2, s2 = m.generate(img, 12, beam=5, return_beams=True)
    assert m._engine.query(2) == 2 and m._engine.query(3) == 1
    os.environ["TXO_LANES"] = "1"
    try:
        t1, s1 = m.generate(img, 12, beam=5, return_beams=True)
        assert m._engine.query(2) == 1
    finally:
        os.environ.pop("TXO_LANES")
    assert torch.equal(t1, t2) and torch.equal(s1, s2)
    # with an eos (the most frequent token of the best beams): finished beams repeat eos at no cost, the loop stops only when EVERY
    # range's beams are finished -- same length, same beams, same scores on one range and on two
    vals, counts = np.unique(t2[:, 0].cpu().numpy(), return_counts=True)
    m.eos_token = int(vals[counts.argmax()])
    e2 = m.generate(img, 12, beam=5, return_beams=True)
    os.environ["TXO_LANES"] = "1"
    try:
        e1 = m.generate(img, 12, beam=5, return_beams=True)
    finally:
        os.environ.pop("TXO_LANES")
    assert e1[0].shape == e2[0].shape and torch.equal(e1[0], e2[0]) and torch.equal(e1[1], e2[1])
    assert not torch.equal(e2[0], t2[:, :, :e2[0].shape[2]]) or e2[0].shape[2] < 12      # the eos changed something


def test_multipass_attention_panels():
    """Panels longer than one register pass: a full 448x448 canvas (N = 785 encoder tokens -> several cross-attention passes
    per launch in both storage types) and a positional table of 300 (> 256 cached self-attention keys -> two passes)."""
    cpu_ref = _oracle()
    d = Dims(canvas=448, embed_dim=64, enc_heads=2, enc_layers=1, dec_heads=2, dec_layers=2, vocab=96, max_len=300,
             bos=94, eos=93, pad=95)
    d, sd, m = build(d, seed=11, max_batch=2)
    img = torch.from_numpy(synth.synth_images(2, 3, 448, 448, seed=51))
    sdt = cpu_ref.to_torch_sd(sd)
    enc_ref = cpu_ref.encode(sdt, img)
    enc = m.encoder(img.cuda())
    assert enc.shape == (2, 785, 64)
    assert float((enc.cpu() - enc_ref).abs().max()) < 1e-4
    m.eos_token = None
    ref_t, ref_l = cpu_ref.generate_cached(sdt, img, d.bos, None, 290, collect_logits=True, enc=enc_ref)
    toks, logits = m.generate(img.cuda(), 290, return_logits=True)
    assert toks.shape == (2, 290)
    assert_tokens_exact_up_to_margin(toks.cpu().numpy(), ref_t.numpy(), ref_l)
    if bool((toks.cpu() == ref_t).all()):
        assert float((logits.cpu() - ref_l).abs().max()) < 1e-3
    # bf16 engine: teacher-forced on the fp32 tokens, logits stay close through all passes
    d2, sd2, mb = build(d, seed=11, dtype="bf16", max_batch=2)
    encb = mb.encoder(img.cuda())
    prefix = torch.cat([torch.full((2, 1), d.bos, dtype=torch.long, device="cuda"), toks[:, :-1]], 1)
    import os
    os.environ["TXO_NET_STEPWISE"] = "1"            # the single-position kernels' multi-pass panels are what this test is about
    try:
        lb = mb.decoder.net(prefix[:, :280], enc=encb).cpu()
    finally:
        os.environ.pop("TXO_NET_STEPWISE")
    errb = float((lb - ref_l[:, :280]).abs().max())
    print(f"multi-pass panels, bf16 steps vs oracle (64-d toy, 280 positions): max |dlogit| {errb:.4f}")
    assert errb < 0.04                              # measured 0.025


def test_large_ragged_batch_rows_independent():
    """More decode rows than one launch round of 16-row tiles at the config.yml widths, ragged (130 rows: nine tiles, the
    last one with 2 rows): fp32 vs the oracle, and in both storage types the rows of the 130-row run must equal the same
    images decoded in 64-row runs bit for bit (a row's arithmetic never depends on the batch around it)."""
    cpu_ref = _oracle()
    d = Dims(canvas=64, max_len=16)
    img = torch.from_numpy(synth.synth_images(130, 3, 32, 64, seed=77))
    d, sd, m = build(d, seed=21, max_batch=130)
    sdt = cpu_ref.to_torch_sd(sd)
    m.eos_token = None
    ref_t, ref_l = cpu_ref.generate_cached(sdt, img, d.bos, None, 12, collect_logits=True)
    toks, logits = m.generate(img.cuda(), 12, return_logits=True)
    assert toks.shape == (130, 12)
    assert_tokens_exact_up_to_margin(toks.cpu().numpy(), ref_t.numpy(), ref_l)
    if bool((toks.cpu() == ref_t).all()):
        assert float((logits.cpu() - ref_l).abs().max()) < 1e-3
    for dtype, latent in (("fp32", 0), ("bf16", 0), ("fp32", 1), ("bf16", 1)):
        # the property is asserted per FORM of the cross attention (r05: both forms): a form's bits do not depend on the batch
        _, _, mm = build(d, seed=21, dtype=dtype, max_batch=130, latent=latent)
        mm.eos_token = None
        big_t, big_l = mm.generate(img.cuda(), 12, return_logits=True)
        small_t, small_l = mm.generate(img[:64].cuda(), 12, return_logits=True)
        assert mm._engine.query(3) == latent
        assert torch.equal(big_t[:64], small_t), (dtype, latent)
        assert torch.equal(big_l[:64], small_l), (dtype, latent)
        tail_t, tail_l = mm.generate(img[66:].cuda(), 12, return_logits=True)      # rows 66..129 as a 64-row batch
        assert torch.equal(big_t[66:], tail_t), (dtype, latent)
        assert torch.equal(big_l[66:], tail_l), (dtype, latent)
    # The DEFAULT bf16 engine switches forms at 129 rows (launches in latent form above, the persistent launch on K/V panels up to 128):
    # the same image then decodes with different low bits in a 130-row and in a 64-row batch.  What IS asserted across the switch:
    # position 0 (same prefix in both runs) differs by at most bf16 noise, both runs stay within the bf16 bound of the fp32 ORACLE on
    # their common prefix with it, and where the two runs pick different tokens the oracle's margin at that position is thinner than
    # twice the bound.  The fp32 parity mode has ONE form at every batch size (first loop).
    _, _, md = build(d, seed=21, dtype="bf16", max_batch=130)
    md.eos_token = None
    big_t, big_l = md.generate(img.cuda(), 12, return_logits=True)
    assert md._engine.query(0) == 0 and md._engine.query(3) == 1          # 130 rows: launches, latent form
    small_t, small_l = md.generate(img[:64].cuda(), 12, return_logits=True)
    assert md._engine.query(3) == 0                                       # 64 rows: K/V form
    first = float((big_l[:64, 0] - small_l[:, 0]).abs().max())
    BOUND = 0.07                                                          # 1.5x the measurement (0.031 / 0.047)
    ref_sorted = ref_l.sort(-1, descending=True).values
    margin = (ref_sorted[..., 0] - ref_sorted[..., 1]).numpy()            # oracle top-1 minus top-2 per (row, position)
    worst = 0.0
    for run_t, run_l, rows in ((big_t.cpu(), big_l.cpu(), slice(0, 130)), (small_t.cpu(), small_l.cpu(), slice(0, 64))):
        rt, rl = ref_t[rows], ref_l[rows]
        for r in range(run_t.shape[0]):
            diff = np.nonzero((run_t[r] != rt[r]).numpy())[0]
            k = int(diff[0]) if diff.size else 11
            worst = max(worst, float((run_l[r, :k + 1] - rl[r, :k + 1]).abs().max()))
            if diff.size:
                assert margin[rows][r, k] < 2 * BOUND, (r, k, float(margin[rows][r, k]))
    print(f"default bf16 engine across the 128-row form switch: position-0 logits of the same image differ by {first:.4f}; "
          f"max |dlogit| vs the fp32 oracle on the common prefixes {worst:.4f}")
    assert first < BOUND and worst < BOUND


def test_bf16_dma_gemm_bit_identical_to_register_staged(monkeypatch):
    """The 256x256 LDS-DMA GEMM (gemm_pp.h) and the 128x128 register-staged GEMM (gemm_big.h) accumulate every output element in
    the same order, so the bf16 engine must give bit-identical encoder features and tokens with either (TXO_GEMM_OLD=1 forces
    the latter) -- and with either EPILOGUE form of the former: TXO_PP_TR=1 runs every epilogue straight from the transposed
    accumulators (weight rows permuted on their way into LDS), =0 stages every one through LDS, unset picks per epilogue (r05).
    Ragged row count (B * 589 is not a multiple of 256: the tile-seam's relaxed wait must fall back on the last row panel) and
    every epilogue (heads scatter, GLU + residual, GeGLU, bias + residual, cross K/V store) are on the path; repeated to screen
    for timing-dependent races (the seam's early request / counted wait)."""
    def build_with(env, *a, **kw):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            return build(*a, **kw)[2]
        finally:
            for k in env:
                monkeypatch.delenv(k)
    forms = ({}, {"TXO_PP_TR": "1"}, {"TXO_PP_TR": "0"})
    d = Dims(canvas=672)
    img = torch.from_numpy(synth.synth_images(12, 3, 224, 672, seed=91)).cuda()
    m_old = build_with({"TXO_GEMM_OLD": "1"}, d, seed=4, dtype="bf16", max_batch=12, max_tokens=589)
    m_old.eos_token = None
    enc_old = m_old.encoder(img)
    t_old, l_old = m_old.generate(img, 24, return_logits=True)
    for env in forms:
        m_new = build_with(env, d, seed=4, dtype="bf16", max_batch=12, max_tokens=589)
        m_new.eos_token = None
        for _ in range(25):
            assert torch.equal(m_new.encoder(img), enc_old), env
        t_new, l_new = m_new.generate(img, 24, return_logits=True)
        assert torch.equal(t_new, t_old) and torch.equal(l_new, l_old), env
    # ViT-Base widths (K = 768 / 3072: 12 and 48 K tiles); 20 images: 47 row panels, several tiles per workgroup (the seam between tiles)
    d2 = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=2, dec_heads=12, dec_layers=1)
    img20 = torch.from_numpy(synth.synth_images(20, 3, 224, 672, seed=93)).cuda()
    b_old = build_with({"TXO_GEMM_OLD": "1"}, d2, seed=6, dtype="bf16", max_batch=20, max_tokens=589)
    ref, ref20 = b_old.encoder(img), b_old.encoder(img20)
    for env in forms:
        b_new = build_with(env, d2, seed=6, dtype="bf16", max_batch=20, max_tokens=589)
        for _ in range(10):
            assert torch.equal(b_new.encoder(img), ref), env
            assert torch.equal(b_new.encoder(img20), ref20), env
    # default-factory model: the hybrid embedder's 1x1 projection (position rows added in the epilogue, K = 1024)
    from texocr_amd.config import reference_config
    dh = Dims.from_config(reference_config())
    imgh = torch.from_numpy(synth.synth_images(4, 1, 160, 1008, seed=92)).cuda()
    # (with the opt-in bf16 backbone: the default keeps the backbone and this projection in fp32, engine.hip: bk_fp32)
    h_old = build_with({"TXO_GEMM_OLD": "1", "TXO_BACKBONE_BF16": "1"}, dh, seed=8, dtype="bf16", max_batch=4)
    ref_h = h_old.encoder(imgh)
    for env in forms:
        h_new = build_with(dict(env, TXO_BACKBONE_BF16="1"), dh, seed=8, dtype="bf16", max_batch=4)
        assert torch.equal(h_new.encoder(imgh), ref_h), env


def test_encoder_tile_walk_and_image_chunks_do_not_change_a_bit(monkeypatch):
    """r06: (a) the 256x256 GEMM's persistent tile walk runs its column bands inside ROW SUPER-BLOCKS (gemm_pp.h: `sbr`; the A operand of a
    super-block stays in the Infinity Cache across the bands) and (b) Engine::encode can push IMAGE CHUNKS through the stack one after the
    other (TXO_ENC_CHUNK, an experiment knob).  Neither may change a bit: a tile's arithmetic does not depend on when it is computed, and rows
    of different images never meet.  ViT-Base widths, 50 images = 116 row panels: super-blocks of 10 panels (eleven full, one of six), with
    the walk direction alternating; chunks of 16 and of 7 images (ragged last chunk)."""
    def build_with(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            return build(d, seed=6, dtype="bf16", max_batch=50, max_tokens=589)[2]
        finally:
            for k in env:
                monkeypatch.delenv(k)
    d = Dims(canvas=672, embed_dim=768, enc_heads=12, enc_layers=2, dec_heads=12, dec_layers=1)
    img = torch.from_numpy(synth.synth_images(50, 3, 224, 672, seed=95)).cuda()
    ref = build_with({"TXO_PP_SB_MB": "0"}).encoder(img)                 # the r05 walk: every band over all row panels
    assert bool(torch.isfinite(ref).all())
    for env in ({}, {"TXO_PP_SB_MB": "4"}, {"TXO_PP_SB_MB": "4", "TXO_PP_CT": "2"}, {"TXO_ENC_CHUNK": "16"}, {"TXO_ENC_CHUNK": "7", "TXO_PP_SB_MB": "4"}):
        m = build_with(env)
        for _ in range(3):
            assert torch.equal(m.encoder(img), ref), env


def test_wide_decoder_large_batch_ffn_path():
    """Perf mode routes the FFN-in of wide decoders (rows >= 768 wide) at >= 128 decode rows through one LayerNorm launch + the
    large-GEMM kernel instead of the 16-row kernel.  Same math, different accumulation order: logits of a 130-row bf16 run stay
    within bf16 noise of the 64-row run of the same images (which uses the 16-row kernel) and close to the fp32 engine."""
    d = Dims(canvas=64, embed_dim=768, enc_heads=12, enc_layers=1, dec_heads=12, dec_layers=2, max_len=16)
    img = torch.from_numpy(synth.synth_images(130, 3, 32, 64, seed=78)).cuda()
    _, _, mf = build(d, seed=23, dtype="fp32", max_batch=130)
    _, _, mb = build(d, seed=23, dtype="bf16", max_batch=130)
    mf.eos_token = None; mb.eos_token = None
    tf, lf = mf.generate(img, 10, return_logits=True)
    prefix = torch.cat([torch.full((130, 1), d.bos, dtype=torch.long, device="cuda"), tf[:, :-1]], 1)
    enc_big = mb.encoder(img)
    import os
    os.environ["TXO_NET_STEPWISE"] = "1"            # single-position steps: the launch path's FFN-in routing is what this test is about
    try:
        lb_big = mb.decoder.net(prefix, enc=enc_big)                       # 130 rows: large-batch path
        lb_small = mb.decoder.net(prefix[:64], enc=enc_big[:64])          # 64 rows: 16-row kernel
    finally:
        os.environ.pop("TXO_NET_STEPWISE")
    e1, e2 = float((lb_big[:64] - lb_small).abs().max()), float((lb_big.float() - lf).abs().max())
    a1 = float((lb_big[:64].argmax(-1) == lb_small.argmax(-1)).float().mean())
    print(f"wide decoder FFN routes: 130-row vs 64-row bf16 max |dlogit| {e1:.4f} (top-1 agreement {a1:.4f}); bf16 vs fp32 {e2:.4f}")
    assert e1 < 0.02                                # measured 0.013
    assert e2 < 0.05                                # measured 0.033
    assert a1 > 0.99                                # measured 0.998


def test_folded_out_projection_on_32_row_blocks_bit_identical_to_16_row_blocks():
    """Beyond 256 decode rows per range (a beam search's rows), the latent form's folded output projection (K = heads * D = 2048) runs on
    32-row blocks that share their weight fragments (dec_gemm_wide_kernel<EPI_GLU_RES, 16, 32, 2>): same K split, same reduction
    order, same epilogue as the 16-row tile -> the same bits.  520 rows on ONE row range against the engine with that routing off."""
    import os
    d = Dims(canvas=224, max_len=16)
    img = torch.from_numpy(synth.synth_images(520, 3, 32, 64, seed=91)).cuda()
    outs = []
    for off in (False, True):
        if off: os.environ["TXO_DEC_WIDE_OFF"] = "1"
        try:
            _, _, m = build(d, seed=5, dtype="bf16", max_batch=520)
        finally:
            os.environ.pop("TXO_DEC_WIDE_OFF", None)
        m.eos_token = None
        os.environ["TXO_LANES"] = "1"
        try:
            outs.append(m.generate(img, 12, return_logits=True))
        finally:
            os.environ.pop("TXO_LANES")
        assert m._engine.query(0) == 0 and m._engine.query(3) == 1          # launches, latent form
        del m
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])                                # logits bit for bit


# ------------------------------------------------------------------------------------------------
# the decode loop as ONE persistent launch (csrc/persist.h) against the launch-per-stage path
# ------------------------------------------------------------------------------------------------
def _both_paths(m, img, max_len, **kw):
    """generate() through the persistent launch (TXO_PERSIST=1) and through launches (TXO_PERSIST=0); asserts which path ran."""
    import os
    outs = []
    for mode in ("1", "0"):
        os.environ["TXO_PERSIST"] = mode
        try:
            outs.append(m.generate(img, max_len, **kw))
            assert m._engine.query(0) == int(mode), f"TXO_PERSIST={mode}: the other decode path ran"
        finally:
            os.environ.pop("TXO_PERSIST")
    assert m._engine.query(1) == 0, "a persistent launch fell back to launches"
    return outs[0], outs[1]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B", [1, 5, 64, 100, 200])
def test_persistent_decode_bit_identical_to_launches(dtype, B):
    """Same tile functions, same reduction orders: tokens AND per-step logits are bit-identical for every batch
    size (1 row in one team ... 13 rows in each of 8 teams with a ragged last one ... 25 rows per team: two 16-row
    tiles per GEMM stage and four rounds of attention pairs)."""
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=3, dtype=dtype, max_batch=B, latent=0)     # (beyond 128 rows bf16 launches default to the latent form)
    g = torch.Generator(device="cuda").manual_seed(77 + B)
    img = torch.rand((B, 3, 64, 224), generator=g, device="cuda")
    (tp, lp), (tl, ll) = _both_paths(m, img, 40, return_logits=True)
    assert tp.shape == (B, 40)
    assert torch.equal(tp, tl)
    assert torch.equal(lp, ll)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_persistent_decode_bit_identical_to_launches_full_width(dtype):
    """The same at the benchmark's image size (224 x 672: 589 keys, i.e. TWO passes of the cross attention in bf16 and
    four in fp32) and a batch that leaves second groups without attention pairs (24 images: 3 rows per team)."""
    d = Dims(canvas=672)
    d, sd, m = build(d, seed=4, dtype=dtype, max_batch=24)
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand((24, 3, 224, 672), generator=g, device="cuda")
    (tp, lp), (tl, ll) = _both_paths(m, img, 12, return_logits=True)
    assert tp.shape == (24, 12)
    assert torch.equal(tp, tl)
    assert torch.equal(lp, ll)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B", [3, 64])
def test_persistent_decode_samples_like_the_launch_path(dtype, B):
    """The reference's DEFAULT decode samples (ocr_model.py:47, decoder.py:104-108).  decode='sample' runs inside the persistent
    launch too: the sampler is the position's last stage, drawing from the same counter RNG keyed by (seed; row, position), so
    the tokens AND the logits they were drawn from equal the launch path's bit for bit; another seed gives other tokens; with an
    eos the GLOBAL break happens at the same position on both paths."""
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=3, dtype=dtype, max_batch=B)
    g = torch.Generator(device="cuda").manual_seed(31 + B)
    img = torch.rand((B, 3, 64, 224), generator=g, device="cuda")
    kw = dict(temp=0.3, decode="sample", seed=11, return_logits=True)
    (tp, lp), (tl, ll) = _both_paths(m, img, 40, **kw)
    assert tp.shape == (B, 40) and torch.equal(tp, tl) and torch.equal(lp, ll)
    greedy = m.generate(img, 40)
    assert not torch.equal(tp, greedy), "sampling at temperature 0.3 never left the arg-max path"
    (t2, _), _ = _both_paths(m, img, 40, temp=0.3, decode="sample", seed=12, return_logits=True)
    assert not torch.equal(t2, tp)
    # support: every drawn token is among the 99 largest logits of its step
    top = lp.topk(99, dim=-1).indices
    assert bool((top == tp[:, :, None]).any(-1).all())
    # an eos that the sampled sequences hit: same early break on both paths
    flat = tp.cpu().numpy()
    cands = [tok for tok in range(d.vocab) if (flat == tok).any(axis=1).all()]
    if cands:
        m.eos_token = cands[0]
        a, b = _both_paths(m, img, 40, temp=0.3, decode="sample", seed=11)
        assert torch.equal(a, b) and a.shape[1] <= 40


@pytest.mark.parametrize("vocab", [999, 950, 512, 1100])
def test_sampler_row_request_forms_and_the_oracle_top_k(vocab):
    """The register-form sampler requests a row in 16-byte pieces when the vocabulary and a lane's chunk are multiples of 4 (1000, 512), one
    logit at a time otherwise (999: V % 4, 950: 15 slots per lane), and the LDS form serves vocabularies beyond 1024 (1100).  On every
    form: persistent launch and launch path draw the same tokens from the same logits, the draws are inside the reference's top-k support
    (utils.py:85-91: k = int(0.1 V) largest), and they are draws (not all arg-maxes)."""
    import dataclasses
    d = dataclasses.replace(Dims(canvas=224), vocab=vocab, bos=vocab - 2, eos=vocab - 3, pad=vocab - 1)
    d, sd, m = build(d, seed=3, dtype="fp32", max_batch=21)
    m.eos_token = None
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand((21, 3, 64, 224), generator=g, device="cuda")
    kw = dict(temp=0.7, decode="sample", seed=4, return_logits=True)
    if vocab <= 1024:
        (tp, lp), (tl, ll) = _both_paths(m, img, 24, **kw)
        assert torch.equal(tp, tl) and torch.equal(lp, ll)
    else:                                                    # (the persistent launch's sampler keeps the row in registers: launches serve this one)
        tp, lp = m.generate(img, 24, **kw)
        assert m._engine.query(0) == 0
        t2, l2 = m.generate(img, 24, **kw)
        assert torch.equal(tp, t2) and torch.equal(lp, l2)
    k = max(int((1 - 0.9) * vocab), 1)
    top = lp.topk(k, dim=-1).indices
    assert bool((top == tp[:, :, None]).any(-1).all())
    assert not torch.equal(tp, lp.argmax(-1))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sampled_decode_draws_do_not_depend_on_the_row_ranges(dtype):
    """A draw is keyed by (seed; row of the BATCH, position): two row ranges on two streams (the default for bf16 beyond 223 images,
    sampled like greedy) sample the same tokens from the same logits as one range."""
    import os
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=3, dtype=dtype, max_batch=48)
    g = torch.Generator(device="cuda").manual_seed(77)
    img = torch.rand((48, 3, 64, 224), generator=g, device="cuda")
    outs = []
    for lanes in ("1", "2"):
        os.environ["TXO_LANES"] = lanes
        try:
            outs.append(m.generate(img, 24, temp=0.3, decode="sample", seed=11, return_logits=True))
            assert m._engine.query(0) == 0 and m._engine.query(2) == int(lanes)
        finally:
            os.environ.pop("TXO_LANES")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], m.generate(img, 24))              # and they are draws, not arg-maxes


def test_persistent_decode_gives_up_cleanly_and_launches_take_over():
    """The persistent launch waits on other workgroups with bounded spins; when a hand-off times out (or a team turns out to
    span two XCDs) every workgroup leaves, the engine reports it and redoes the decode with launches.  A test hook makes
    team 0 report a time-out at position 3: same tokens and logits as the launch path, one fallback counted."""
    import os
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=5, dtype="bf16", max_batch=16)
    g = torch.Generator(device="cuda").manual_seed(11)
    img = torch.rand((16, 3, 64, 224), generator=g, device="cuda")
    os.environ["TXO_PERSIST"] = "0"
    try:
        tl, ll = m.generate(img, 24, return_logits=True)
    finally:
        os.environ.pop("TXO_PERSIST")
    before = m._engine.query(1)
    os.environ["TXO_PERSIST"] = "1"; os.environ["TXO_PERSIST_INJECT_FAIL"] = "3"
    try:
        tp, lp = m.generate(img, 24, return_logits=True)
    finally:
        os.environ.pop("TXO_PERSIST"); os.environ.pop("TXO_PERSIST_INJECT_FAIL")
    assert m._engine.query(1) == before + 1, "the injected time-out was not counted as a fallback"
    assert m._engine.query(0) == 0, "the fallback decode must be the launch path"
    assert torch.equal(tp, tl) and torch.equal(lp, ll)
    os.environ["TXO_PERSIST"] = "1"
    try:
        tp2, _ = m.generate(img, 24, return_logits=True)     # and the engine is usable afterwards, persistent again
    finally:
        os.environ.pop("TXO_PERSIST")
    assert m._engine.query(0) == 1 and torch.equal(tp2, tl)


@pytest.mark.parametrize("held_cus,hold_ms", [(96, 80), (200, 80), (128, 1)])
def test_persistent_decode_under_real_cu_contention(held_cus, hold_ms):
    """The persistent launch needs its 256 workgroups co-resident.  A second stream holds `held_cus` compute units (a spin kernel
    with a whole CU's LDS per block, tests/hooks/hold_cus.hip) while generate() runs: for 80 ms -- twenty times the hand-off time-out, so
    the resident workgroups give up, the launch ends as soon as the rest have started and seen the fail word, and launches take
    over beside the filler -- or for 1 ms, which the launch simply waits out.  Tokens and logits equal the undisturbed launch
    path's either way, nothing hangs, the fall-back is counted, and the engine goes back to the persistent launch afterwards."""
    import os, time
    import ctypes as C
    from texocr_amd import build as tb
    hooks = C.CDLL(tb.build_test_hooks(verbose=False))          # tests/hooks/hold_cus.hip: not part of the product library
    hooks.txo_test_hold_cus.restype, hooks.txo_test_hold_cus.argtypes = C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    d = Dims(canvas=224)
    d, sd, m = build(d, seed=5, dtype="bf16", max_batch=32)
    g = torch.Generator(device="cuda").manual_seed(12)
    img = torch.rand((32, 3, 64, 224), generator=g, device="cuda")
    os.environ["TXO_PERSIST"] = "0"
    try:
        tl, ll = m.generate(img, 48, return_logits=True)
    finally:
        os.environ.pop("TXO_PERSIST")
    # the filler must run BESIDE the decode: two HIP streams that the runtime mapped onto one hardware queue run one after the other
    # (profiles/r06_b256_stream_pairs.txt) -- the filler would then simply finish first and nothing would be contended.  Take a side stream
    # that demonstrably runs side by side with the current one (a 3 ms one-CU hold on it must not delay a trivial launch here).
    side, tried = None, []
    for _ in range(12):
        cand = torch.cuda.Stream()
        torch.cuda.synchronize()
        assert hooks.txo_test_hold_cus(1, 1024, 3000, C.c_void_p(cand.cuda_stream)) == 0
        t0 = time.perf_counter()
        _ = torch.zeros(16, device="cuda") + 1
        torch.cuda.current_stream().synchronize()
        beside = time.perf_counter() - t0 < 1.5e-3
        cand.synchronize()
        tried.append(cand)                                      # (kept alive: the pool then hands out another stream)
        if beside:
            side = cand
            break
    if side is None:
        pytest.skip("no side stream runs beside the current stream in this process")
    before = m._engine.query(1)
    os.environ["TXO_PERSIST"] = "1"
    try:
        torch.cuda.synchronize()
        assert hooks.txo_test_hold_cus(held_cus, 160 * 1024, hold_ms * 1000, C.c_void_p(side.cuda_stream)) == 0
        time.sleep(0.002)                                       # the filler is on its CUs before the decode launch is enqueued
        t0 = time.perf_counter()
        tp, lp = m.generate(img, 48, return_logits=True)
        dt = time.perf_counter() - t0
        side.synchronize()
        fell_back = m._engine.query(1) - before
        print(f"held {held_cus} CUs for {hold_ms} ms: generate took {dt * 1e3:.1f} ms, fall-backs {fell_back}, persistent={m._engine.query(0)}")
        assert torch.equal(tp, tl) and torch.equal(lp, ll)
        assert dt < 0.5, "generate() must not hang on a busy GPU"
        if hold_ms >= 40:
            assert fell_back == 1 and m._engine.query(0) == 0, "co-residency was impossible: the launch must have given up"
            assert dt < 0.060, "the give-up must not wait for the filler to finish"
        tp2, lp2 = m.generate(img, 48, return_logits=True)      # idle GPU again: persistent, same result
        assert m._engine.query(0) == 1 and torch.equal(tp2, tl) and torch.equal(lp2, ll)
    finally:
        os.environ.pop("TXO_PERSIST")


def test_persistent_decode_global_eos_break():
    """Small vocabulary so that eos fires at different positions in different rows and teams: the persistent launch must
    return exactly the columns of the reference's GLOBAL break (decoder.py:115-116), like the launch path and the oracle."""
    cpu_ref = _oracle()
    d = Dims(canvas=224, vocab=16, bos=14, eos=5, pad=15, max_len=256)
    d, sd, m = build(d, seed=11, max_batch=20)
    img = torch.from_numpy(synth.synth_images(20, 3, 32, 96, seed=5)).cuda()
    # pick as eos a token that every row emits, the last row as late as possible (rows and teams finish at different positions)
    import os
    os.environ["TXO_PERSIST"] = "1"
    m.eos_token = None
    free = m.generate(img, 256).cpu().numpy()
    os.environ.pop("TXO_PERSIST")
    best = None
    for tok in range(d.vocab):
        hit = free == tok
        if hit.any(axis=1).all():
            first = hit.argmax(axis=1)
            if best is None or first.max() > best[1]:
                best = (tok, int(first.max()), int(first.min()))
    assert best is not None and best[1] > best[2] + 2, best
    eos = best[0]
    m.eos_token = eos
    tp, tl = _both_paths(m, img, 256)
    assert torch.equal(tp, tl)
    ref = cpu_ref.generate_cached(cpu_ref.to_torch_sd(sd), img.cpu(), d.bos, eos, 256)
    assert tp.shape[1] == best[1] + 1 and 1 < tp.shape[1] < 256, "the crafted case must break early"
    assert tp.shape == tuple(ref.shape)
    assert np.array_equal(tp.cpu().numpy(), ref.numpy())
    assert np.array_equal(tp.cpu().numpy(), free[:, :tp.shape[1]])
    assert bool((tp == eos).any(dim=1).all()) and not bool((tp[:, :-1] == eos).any(dim=1).all())
    # eos_tok = None: never breaks
    m.eos_token = None
    os.environ["TXO_PERSIST"] = "1"
    try:
        assert m.generate(img, 48).shape == (20, 48) and m._engine.query(0) == 1
    finally:
        os.environ.pop("TXO_PERSIST")
    # bos == eos: the BOS column already satisfies the test -> one step (decoder.py:115 looks at the whole output)
    d2 = Dims(canvas=224, vocab=16, bos=5, eos=5, pad=15, max_len=64)
    d2, sd2, m2 = build(d2, seed=11, max_batch=20)
    t2p, t2l = _both_paths(m2, img, 64)
    assert t2p.shape == (20, 1) and torch.equal(t2p, t2l)


@pytest.mark.parametrize("stagger_us", ["150", "600"])
def test_persistent_eos_break_when_teams_drift_apart(stagger_us, monkeypatch):
    """Teams of the persistent launch are not synchronised with each other.  Started 150 / 600 us apart (TXO_PS_STAGGER_US: several
    positions of drift between neighbouring teams, dozens between the first and the last) with a ragged batch, a team that has
    seen the chip-wide done mask must still decode up to the position at which the LAST row of the batch first produced eos:
    every returned column of every row equals the launch path's and the oracle's (no column left as allocated)."""
    cpu_ref = _oracle()
    d = Dims(canvas=224, vocab=16, bos=14, eos=5, pad=15, max_len=256)
    d, sd, m = build(d, seed=11, max_batch=20)
    img = torch.from_numpy(synth.synth_images(20, 3, 32, 96, seed=5)).cuda()
    m.eos_token = None
    monkeypatch.setenv("TXO_PERSIST", "0")
    free = m.generate(img, 256).cpu().numpy()
    # eos candidates: every row emits the token; take those whose LAST first-occurrence is in the first team (rows 0..2) or in the
    # last one (rows 18..19) -- with the stagger the last team is the furthest behind, the first the furthest ahead
    cands = []
    for tok in range(d.vocab):
        hit = free == tok
        if hit.any(axis=1).all():
            first = hit.argmax(axis=1)
            if first.max() > first.min() + 2 and first.max() < 200:
                cands.append((tok, int(first.max()), int(first.argmax())))
    assert cands
    for eos, last, row in cands[:4]:
        m.eos_token = eos
        monkeypatch.setenv("TXO_PERSIST", "0")
        tl = m.generate(img, 256)
        monkeypatch.setenv("TXO_PERSIST", "1"); monkeypatch.setenv("TXO_PS_STAGGER_US", stagger_us)
        monkeypatch.setenv("TXO_DEBUG_POISON", "1")                   # ops.py pre-fills the token block with -7: an undecoded column would show
        for _ in range(3):
            tp = m.generate(img, 256)
            assert int(tp.min()) >= 0
            assert m._engine.query(0) == 1 and m._engine.query(1) == 0
            assert tp.shape == (20, last + 1), (eos, last, row, tuple(tp.shape))
            assert torch.equal(tp, tl), (eos, last, row)
        monkeypatch.delenv("TXO_PS_STAGGER_US"); monkeypatch.delenv("TXO_DEBUG_POISON")
        assert np.array_equal(tp.cpu().numpy(), free[:, :last + 1])
    ref = cpu_ref.generate_cached(cpu_ref.to_torch_sd(sd), img.cpu(), d.bos, cands[0][0], 256)
    m.eos_token = cands[0][0]
    monkeypatch.setenv("TXO_PERSIST", "1"); monkeypatch.setenv("TXO_PS_STAGGER_US", stagger_us)
    assert np.array_equal(m.generate(img, 256).cpu().numpy(), ref.numpy())


# ------------------------------------------------------------------------------------------------
# BASELINE configs 2 (bf16), 4 and 5 at their full sizes
# ------------------------------------------------------------------------------------------------
CFG4 = dict(canvas=672, embed_dim=768, enc_heads=12, enc_layers=12, dec_heads=12, dec_layers=6)


def test_cfg4_vit_base_golden():
    """G10: ViT-Base 12L/768d/12h encoder + 6L/768d/12h decoder, B=2, 224x672, 8 greedy steps captured from the
    reference (encoder.py:75-121, decoder.py:148-173); fp32 parity mode must reproduce it, bf16 mode is measured."""
    meta, g = load_golden("cfg4_b2_224x672")
    d, sd, m = build(meta, max_batch=2, max_tokens=589)
    img = images(meta).cuda()
    enc = m.encoder(img)
    assert enc.shape == (2, 589, 768)
    np.testing.assert_allclose(enc[:, ::16, ::4].cpu().numpy(), g["enc_rows"], atol=1e-4)
    np.testing.assert_allclose(enc.double().sum((1, 2)).cpu().numpy(), g["enc_sum"], rtol=0, atol=0.1)
    toks, logits = m.generate(img, 8, return_logits=True)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    err = float(np.abs(logits.cpu().numpy() - g["step_logits"]).max())
    assert err < 1e-3, err                                       # north_star: logits within 1e-3 in fp32
    del m
    _, _, mb = build(meta, dtype="bf16", max_batch=2, max_tokens=589)
    tb, lb = mb.generate(img, 8, return_logits=True)
    agree = float((tb.cpu().numpy() == g["tokens"]).mean())
    lerr = float(np.abs(lb.cpu().numpy() - g["step_logits"]).max())
    print(f"cfg4 bf16 vs reference: free-running token agreement {agree:.3f} over 2x8, max |dlogit| {lerr:.3f} (fixture margin >= 0.032)")
    assert lerr < 0.06 and agree >= 0.93             # measured 0.037-0.042 and 16/16 (one flip in 16 allowed)


def test_cfg4_full_size_b256():
    """BASELINE configs[3] at size: B=256, 224x672.  fp32: the two fixture images as rows 100 and 7 of a 256-image batch
    must decode to the reference's tokens / logits (rows never interact; decoder.py:115 is the only cross-row operation).
    bf16: 256 full steps -- deterministic, permutation-equivariant, token range."""
    meta, g = load_golden("cfg4_b2_224x672")
    d = Dims(**meta["dims"])
    sd = synth.synth_state_dict(d, meta["weight_seed"])
    from texocr_amd.model import model_from_dims
    gen = torch.Generator(device="cuda").manual_seed(4)
    img = torch.rand((256, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[100], img[7] = fix[0], fix[1]
    m = model_from_dims(d, dtype="fp32", max_batch=256, max_tokens=589)
    m.load_state_dict(sd)
    toks, logits = m.generate(img, 8, return_logits=True)
    assert toks.shape == (256, 8)
    assert np.array_equal(toks[[100, 7]].cpu().numpy(), g["tokens"])
    assert float(np.abs(logits[[100, 7]].cpu().numpy() - g["step_logits"]).max()) < 1e-3
    del m
    torch.cuda.empty_cache()
    mb = model_from_dims(d, dtype="bf16", max_batch=256, max_tokens=589)
    mb.load_state_dict(sd)
    t1 = mb.generate(img, 256)
    assert t1.shape == (256, 256) and int(t1.min()) >= 0 and int(t1.max()) < d.vocab
    assert torch.equal(mb.generate(img, 256), t1)
    perm = torch.randperm(256, device="cuda", generator=gen)
    assert torch.equal(mb.generate(img[perm].contiguous(), 256), t1[perm])


def test_cfg4_golden_64_steps_inside_batch_256():
    """ViT-Base + 6L decoder pinned to the reference beyond the first positions: 64 greedy steps at B=2 from the reference, decoded
    as rows 200 and 33 of a 256-image batch on the fp32 engine (tokens exact, logits within 1e-3)."""
    meta, g = load_golden("cfg4_b2_224x672_t64")
    assert float(g["margin"].min()) >= 1e-4 and g["tokens"].shape == (2, 64)
    d = Dims(**meta["dims"])
    sd = synth.synth_state_dict(d, meta["weight_seed"])
    from texocr_amd.model import model_from_dims
    gen = torch.Generator(device="cuda").manual_seed(9)
    img = torch.rand((256, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[200], img[33] = fix[0], fix[1]
    m = model_from_dims(d, dtype="fp32", max_batch=256, max_tokens=589)
    m.load_state_dict(sd)
    toks, logits = m.generate(img, 64, return_logits=True)
    assert np.array_equal(toks[[200, 33]].cpu().numpy(), g["tokens"])
    lg = logits[[200, 33]].cpu()
    v = torch.gather(lg, 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
    err = float((v - torch.from_numpy(g["top5_vals"])).abs().max())
    print(f"cfg4 t64: max |dlogit| over the reference's top-5 of all 2x64 steps {err:.2e}")
    assert err < 1e-3
    np.testing.assert_allclose(lg[:, -4:].numpy(), g["logits_last4"], atol=2e-4)


def test_cfg5_full_size_beam5_bucketed():
    """BASELINE configs[4] at size: 128 images, widths 224..896 in steps of 64 (the renderer pads widths to multiples of
    64, render_data.py:85-86), bucketed by exact size like BucketBatchSampler (dataset.py:281-326), beam search k=5
    (a build extension: the reference has none), max_len 256.  Anchors: the oracle's independent CPU beam search on a
    4-image subset (fp32), and the composition bucket -> shard -> beam -> gather used on a node."""
    cpu_ref = _oracle()
    from texocr_amd.dist import generate_bucketed, sharded_generate_bucketed
    from texocr_amd.model import model_from_dims
    d = Dims(canvas=896)
    sd = synth.synth_state_dict(d, 0)
    rng = np.random.Generator(np.random.PCG64(55))
    widths = rng.choice(np.arange(224, 897, 64), size=128)
    gen = torch.Generator(device="cuda").manual_seed(8)
    imgs = [torch.rand((3, 224, int(w)), generator=gen, device="cuda") for w in widths]
    biggest = max(np.bincount(widths // 64).max(), 1)
    mb = model_from_dims(d, dtype="bf16", max_batch=int(biggest) * 5, max_tokens=785)
    mb.load_state_dict(sd)
    rows = generate_bucketed(lambda b: mb.generate(b, 256, beam=5), imgs, max_batch=int(biggest))
    assert len(rows) == 128 and all(r.ndim == 1 and 1 <= r.shape[0] <= 256 for r in rows)
    assert all(int(r.min()) >= 0 and int(r.max()) < d.vocab for r in rows)
    rows2 = generate_bucketed(lambda b: mb.generate(b, 256, beam=5), imgs, max_batch=int(biggest))
    assert all(torch.equal(a, b) for a, b in zip(rows, rows2))
    # the node-level composition at world size 1 (no process group): same rows
    rows3 = sharded_generate_bucketed(lambda b, n: mb.generate(b, n, beam=5), imgs, 256, d.eos, bos=d.bos, max_batch=int(biggest), beam=True)
    assert all(torch.equal(a, b) for a, b in zip(rows, rows3))
    del mb
    # fp32 against the oracle's beam search: one image of four different widths, 12 steps
    pick = [int(np.nonzero(widths == w)[0][0]) for w in sorted(set(widths.tolist()))[:4]]
    mf = model_from_dims(d, dtype="fp32", max_batch=5, max_tokens=785)
    mf.load_state_dict(sd)
    sdt = cpu_ref.to_torch_sd(sd)
    for i in pick:
        x = imgs[i][None].contiguous()
        beams, scores = mf.generate(x, 12, beam=5, return_beams=True)
        enc = cpu_ref.encode(sdt, x.cpu())
        ref_beams, ref_scores = cpu_ref.beam_search_cached(sdt, enc, d.bos, d.eos, 12, 5)
        assert np.array_equal(beams[0].cpu().numpy(), ref_beams[0].numpy()), i
        np.testing.assert_allclose(scores[0].cpu().numpy(), ref_scores[0].numpy(), atol=2e-3)


def test_cfg2_full_size_bf16():
    """BASELINE configs[1] in the mode the headline number is quoted in: bf16, B=64, 224x672, 256 steps -- deterministic,
    permutation-equivariant; measured agreement with the token-exact fp32 engine (teacher-forced on the fp32 tokens)."""
    from texocr_amd.model import model_from_dims
    d = Dims(canvas=672)
    sd = synth.synth_state_dict(d, 0)
    gen = torch.Generator(device="cuda").manual_seed(1234)
    img = torch.rand((64, 3, 224, 672), generator=gen, device="cuda")
    mb = model_from_dims(d, dtype="bf16", max_batch=64, max_tokens=589)
    mb.load_state_dict(sd)
    t1 = mb.generate(img, 256)
    assert t1.shape == (64, 256) and int(t1.min()) >= 0 and int(t1.max()) < d.vocab
    assert torch.equal(mb.generate(img, 256), t1)
    perm = torch.randperm(64, device="cuda", generator=gen)
    assert torch.equal(mb.generate(img[perm].contiguous(), 256), t1[perm])
    mf = model_from_dims(d, dtype="fp32", max_batch=64, max_tokens=589)
    mf.load_state_dict(sd)
    tf_, lf = mf.generate(img, 64, return_logits=True)
    # teacher-forced bf16 logits on the fp32 engine's tokens
    prefix = torch.cat([torch.full((64, 1), d.bos, dtype=torch.int64, device="cuda"), tf_[:, :-1]], 1)
    lb = mb.decoder.net(prefix, enc=mb.encoder(img))
    agree = float((lb.argmax(-1) == tf_).float().mean())
    err = float((lb - lf).abs().max())
    first = float((t1[:, 0] == tf_[:, 0]).float().mean())
    print(f"cfg2 bf16 vs fp32 engine, B=64 x 64 steps: teacher-forced top-1 agreement {agree:.4f}, max |dlogit| {err:.3f}, "
          f"free-running first-token agreement {first:.3f}")
    assert agree > 0.975 and err < 0.06              # measured 0.988 and 0.035-0.039


def test_torch_free_c_program_on_the_c_abi(tmp_path):
    """examples/generate_tiny.c: plain C + HIP runtime API, no Python in the process: loads the tiny fixture's weights one
    reference key at a time, runs txo_generate and compares the tokens with the reference's."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from texocr_amd import build as b
    exe = b.build_example(verbose=False)
    blob = str(tmp_path / "tiny.bin")
    subprocess.run([sys.executable, os.path.join(root, "examples", "make_tiny_blob.py"), blob], check=True, capture_output=True)
    r = subprocess.run([exe, blob], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tokens match the reference" in r.stdout
    assert "per-row stop" in r.stdout and "match the reference up to each row's eos" in r.stdout     # txo_set_stop_mode from plain C


# ------------------------------------------------------------------------------------------------
# round 4: the cross attention in latent form (csrc/lat_attn.h), and the bf16 mode pinned to the reference at the
# benchmark's own shape over every position
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["cfg1", "tiny64", "width672", "vitbase"])
def test_latent_cross_attention_matches_kv_form_fp32(shape):
    """energy_h[n] = q_h . (Wk_h enc[n]) * 0.125 = (0.125 Wk_h^T q_h) . enc[n] and out_h = Wv_h (sum_n p_h[n] enc[n])
    (reference model/attention.py:124-127,148,166,172-173,248): the latent form re-associates the reference's sums, so in the
    fp32 parity mode its logits must equal the K/V form's to rounding (asserted < 1e-5; measured 3e-6) and the greedy tokens must
    be the same, at 197 keys, at one head of a 64-wide model, at the benchmark's 589 keys, and (bf16 storage only exists there:
    the fp32 tile does not fit 768-wide rows) for a ViT-Base-wide decoder against its own K/V form within the bf16 bound."""
    dtype, bound = "fp32", 1e-5
    if shape == "cfg1":
        d, B, H, W, T = Dims(canvas=224), 4, 224, 224, 48
    elif shape == "tiny64":
        d, B, H, W, T = Dims(canvas=64, embed_dim=64, enc_heads=1, enc_layers=1, dec_heads=1, dec_layers=1, vocab=32, max_len=8,
                             bos=30, eos=29, pad=31), 3, 64, 64, 8
    elif shape == "width672":
        d, B, H, W, T = Dims(canvas=672), 5, 224, 672, 24
    else:
        d, B, H, W, T = Dims(canvas=224, embed_dim=768, enc_heads=12, enc_layers=2, dec_heads=12, dec_layers=2), 4, 224, 224, 16
        dtype, bound = "bf16", 0.08
    img = torch.from_numpy(synth.synth_images(B, 3, H, W, seed=7)).cuda()
    outs = []
    for latent in (0, 1):
        _, _, m = build(d, seed=2, dtype=dtype, max_batch=B, latent=latent)
        m.eos_token = None
        toks, logits = m.generate(img, T, return_logits=True)
        assert m._engine.query(0) == (0 if latent else m._engine.query(0))       # forced latent form decodes with launches
        assert m._engine.query(3) == latent
        if latent:
            assert torch.equal(m.generate(img, T), toks)      # without logits: up to 4 images replay a captured graph of the same launches
        outs.append((toks.cpu(), logits.cpu()))
        del m
    (t0, l0), (t1, l1) = outs
    first = float((l0[:, 0] - l1[:, 0]).abs().max())
    print(f"latent vs K/V form, {shape} {dtype}: max |dlogit| at position 0 {first:.3e}, token agreement {float((t0 == t1).float().mean()):.4f}")
    assert first < bound
    if dtype == "fp32":
        assert torch.equal(t0, t1)
        assert float((l0 - l1).abs().max()) < bound


def test_latent_form_fp32_meets_the_reference_fixture_at_the_benchmark_shape():
    """The latent form against the REFERENCE (not against the build's other form): 3x224x672, 256 greedy steps captured from
    the reference, decoded as rows 41 and 6 of a 64-image batch on the fp32 engine with the latent tile: every token exact
    (fixture margin >= 1e-4), logits over the reference's top-5 of every step within 1e-3."""
    meta, g = load_golden("cfg2_b2_224x672_t256")
    d, sd, m = build(meta, max_batch=64, max_tokens=589, latent=1)
    gen = torch.Generator(device="cuda").manual_seed(21)
    img = torch.rand((64, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[41], img[6] = fix[0], fix[1]
    toks, logits = m.generate(img, 256, return_logits=True)
    assert m._engine.query(0) == 0
    assert np.array_equal(toks[[41, 6]].cpu().numpy(), g["tokens"])
    v = torch.gather(logits[[41, 6]].cpu(), 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
    err = float((v - torch.from_numpy(g["top5_vals"])).abs().max())
    print(f"cfg2 t256, fp32 latent form: max |dlogit| over the reference's top-5 of all 2x256 steps {err:.2e}")
    assert err < 1e-3


def _teacher_forced_stepwise(m, d, img, ref_tokens, monkeypatch):
    """decoder.net() as single-position steps (TXO_NET_STEPWISE=1): the reference's tokens fed through the DECODE kernels of the
    launch path (dec_gemm / dec_attn or lat_attn), every position's logits"""
    toks = torch.from_numpy(ref_tokens.astype(np.int64)).cuda()
    prefix = torch.cat([torch.full((toks.shape[0], 1), d.bos, dtype=torch.long, device="cuda"), toks[:, :-1]], 1)
    monkeypatch.setenv("TXO_NET_STEPWISE", "1")
    try:
        return m.decoder.net(prefix, enc=m.encoder(img)).cpu(), toks.cpu()
    finally:
        monkeypatch.delenv("TXO_NET_STEPWISE")


@pytest.mark.parametrize("form", ["kv", "latent"])
def test_bf16_benchmark_shape_vs_reference_every_position(form, monkeypatch):
    """The mode the headline number is quoted in against the reference at the benchmark's OWN shape: fixture
    cfg2_b2_224x672_t256 (3x224x672 = 589 keys, self history beyond 128 positions), the reference's 2 x 256 tokens teacher-forced
    through the bf16 engine's decode kernels, both forms of the cross attention: logits over the reference's top-5 of every
    position and top-1 agreement, bounds at 1.5x the measurement."""
    meta, g = load_golden("cfg2_b2_224x672_t256")
    d, sd, m = build(meta, dtype="bf16", max_batch=2, max_tokens=589, latent=(1 if form == "latent" else 0))
    img = images(meta).cuda()
    tf, toks = _teacher_forced_stepwise(m, d, img, g["tokens"], monkeypatch)
    got = torch.gather(tf, 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
    err = float((got - torch.from_numpy(g["top5_vals"])).abs().max())
    agree = float((tf.argmax(-1) == toks).float().mean())
    flips = (tf.argmax(-1) != toks).numpy()
    worst = float(g["margin"][flips].max()) if flips.any() else 0.0
    print(f"cfg2 t256 bf16 ({form} form) vs reference, teacher-forced 2x256: max |dlogit| over the reference's top-5 {err:.4f}, "
          f"top-1 agreement {agree:.4f}, largest reference margin among the flipped positions {worst:.4f}")
    assert err < 0.038, err                                    # 1.5x the measurement (r04: K/V form 0.0228 / 0.9941, latent form 0.0251 / 0.9883)
    assert agree > 0.982, agree
    assert worst < 2 * 0.038                                   # a flip needs a margin thinner than twice the logit error (measured: <= 0.0082)


@pytest.mark.parametrize("path", ["persistent", "launches"])
def test_bf16_benchmark_shape_free_running_inside_batch_64(path, monkeypatch):
    """... and free-running on both decode paths (the persistent launch cannot be teacher-forced): the fixture images as rows 41 and
    6 of a 64-image bf16 batch follow the reference's tokens until the first position whose reference margin is thinner than the
    bf16 logit error, and the logits on that common prefix stay within the bound; both paths return the same tokens."""
    meta, g = load_golden("cfg2_b2_224x672_t256")
    d, sd, m = build(meta, dtype="bf16", max_batch=64, max_tokens=589, latent=0)
    gen = torch.Generator(device="cuda").manual_seed(21)
    img = torch.rand((64, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[41], img[6] = fix[0], fix[1]
    monkeypatch.setenv("TXO_PERSIST", "1" if path == "persistent" else "0")
    toks, logits = m.generate(img, 256, return_logits=True)
    assert m._engine.query(0) == (1 if path == "persistent" else 0) and m._engine.query(1) == 0
    got_t, got_l = toks[[41, 6]].cpu().numpy(), logits[[41, 6]].cpu()
    worst_err, shortest = 0.0, 256
    for r in range(2):
        diff = np.nonzero(got_t[r] != g["tokens"][r])[0]
        k = int(diff[0]) if diff.size else 255             # logits of position k were still computed from the reference's prefix
        v = torch.gather(got_l[r, :k + 1], 1, torch.from_numpy(g["top5_ids"][r, :k + 1].astype(np.int64)))
        worst_err = max(worst_err, float((v - torch.from_numpy(g["top5_vals"][r, :k + 1])).abs().max()))
        shortest = min(shortest, k)
        if diff.size:
            assert float(g["margin"][r, k]) < 2 * 0.035, (r, k, float(g["margin"][r, k]))
    print(f"cfg2 t256 bf16 free-running ({path}) inside batch 64: common prefix >= {shortest} positions, max |dlogit| over the top-5 on it {worst_err:.4f}")
    assert worst_err < 0.035 and shortest >= 32                # measured 0.0208 on a common prefix of 53 positions
    monkeypatch.setenv("TXO_PERSIST", "0" if path == "persistent" else "1")
    assert torch.equal(m.generate(img, 256), toks)            # the other path: same tokens


@pytest.mark.parametrize("form", ["kv", "latent"])
def test_bf16_cfg4_vs_reference_64_positions(form, monkeypatch):
    """ViT-Base + 6L decoder in bf16 against the reference's 2 x 64 tokens (fixture cfg4_b2_224x672_t64), teacher-forced through
    the decode kernels of the launch path, both forms (the 768-wide latent tile: 4 waves, one tile of encoder rows per wave)."""
    meta, g = load_golden("cfg4_b2_224x672_t64")
    d, sd, m = build(meta, dtype="bf16", max_batch=2, max_tokens=589, latent=(1 if form == "latent" else 0))
    img = images(meta).cuda()
    tf, toks = _teacher_forced_stepwise(m, d, img, g["tokens"], monkeypatch)
    got = torch.gather(tf, 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
    err = float((got - torch.from_numpy(g["top5_vals"])).abs().max())
    agree = float((tf.argmax(-1) == toks).float().mean())
    print(f"cfg4 t64 bf16 ({form} form) vs reference, teacher-forced 2x64: max |dlogit| over the reference's top-5 {err:.4f}, top-1 agreement {agree:.4f}")
    assert err < 0.039, err                                    # 1.5x the measurement (r04: 0.0258 / 0.0251, top-1 0.9844 both forms)
    assert agree > 0.976, agree


def test_two_row_ranges_bit_identical_to_one(monkeypatch):
    """The default decode beyond 128 images in bf16 (two row ranges on two streams, latent cross attention) against ONE range
    (TXO_LANES=1): rows never interact, so the tokens must be identical -- at batch 256, the benchmark's image size."""
    d = Dims(canvas=672)
    d, sd, m = build(d, seed=0, dtype="bf16", max_batch=256, max_tokens=589)
    gen = torch.Generator(device="cuda").manual_seed(9)
    img = torch.rand((256, 3, 224, 672), generator=gen, device="cuda")
    t2 = m.generate(img, 96)
    assert m._engine.query(0) == 0
    monkeypatch.setenv("TXO_LANES", "1")
    t1 = m.generate(img, 96)
    assert torch.equal(t1, t2)


def test_fp32_fixture_on_two_row_ranges(monkeypatch):
    """the fp32 reference-fixture run of the benchmark's shape (rows 41 and 6 of 64) with the batch decoded as TWO row ranges on
    two streams (TXO_LANES=2): tokens exact, logits within 1e-3"""
    meta, g = load_golden("cfg2_b2_224x672_t256")
    d, sd, m = build(meta, max_batch=64, max_tokens=589)
    gen = torch.Generator(device="cuda").manual_seed(21)
    img = torch.rand((64, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[41], img[6] = fix[0], fix[1]
    monkeypatch.setenv("TXO_LANES", "2")
    toks = m.generate(img, 256)
    assert m._engine.query(0) == 0
    assert np.array_equal(toks[[41, 6]].cpu().numpy(), g["tokens"])


@pytest.mark.parametrize("path", ["persistent", "launches"])
def test_eos_break_at_the_last_table_position_does_not_start_the_window(path, monkeypatch):
    """max_len > decoder.max_len AND the GLOBAL eos break (decoder.py:115-116) fires exactly at the table's last position: the
    reference returns max_length columns and never slides its window.  (steps == max_length alone cannot tell this case from
    'no break'; the engine carries the break out of both decode paths.)"""
    cpu_ref = _oracle()
    d = Dims(canvas=64, vocab=32, max_len=8, bos=30, eos=29, pad=31)     # config.yml widths (the persistent launch exists for them), an 8-entry table
    d, sd, m = build(d, seed=5, max_batch=3)
    sdt = cpu_ref.to_torch_sd(sd)
    hit = None
    for seed in range(60):                                     # rows whose free-running tokens put the batch's LAST first-occurrence of some token at position 7
        img = torch.from_numpy(synth.synth_images(3, 3, 32, 64, seed=seed))
        free = cpu_ref.generate_recompute(sdt, img, d.bos, None, 8).numpy()
        for tok in range(d.vocab):
            occ = free == tok
            if occ.any(axis=1).all() and int(occ.argmax(axis=1).max()) == 7:
                hit = (img, tok)
                break
        if hit:
            break
    assert hit, "no seed puts a common token's last first-occurrence at the table's last position"
    img, eos = hit
    ref = cpu_ref.generate_recompute(sdt, img, d.bos, eos, 20)
    assert ref.shape == (3, 8)
    m.eos_token = eos
    monkeypatch.setenv("TXO_PERSIST", "1" if path == "persistent" else "0")
    out = m.generate(img.cuda(), 20)
    assert m._engine.query(0) == (1 if path == "persistent" else 0)
    assert out.shape == (3, 8), tuple(out.shape)
    assert np.array_equal(out.cpu().numpy(), ref.numpy())
    # and one position later the window does run: the reference's 9th column
    m.eos_token = None
    ref9 = cpu_ref.generate_recompute(sdt, img, d.bos, None, 9)
    assert np.array_equal(m.generate(img.cuda(), 9).cpu().numpy(), ref9.numpy())


def test_ragged_start_prefix_with_padding_mask_matches_reference():
    """A padded multi-token start prefix is a legal decoder.generate call (decoder.py:95-101,112; attention.py:130-155).  Fixture from the
    reference: left padding, an interior hole, an all-True row.  decoder.generate(start_tokens, mask=) returns the reference's tokens;
    decoder.net(x, mask=) its logits at every position that is not padding (rows of padded positions are unspecified: the reference
    softmaxes them uniformly over all keys, nothing reads them); without the mask the tokens differ like the reference's do."""
    meta, g = load_golden("ragged_prefix")
    assert float(g["margin"].min()) >= 1e-4
    d, sd, m = build(meta, max_batch=3)
    img = images(meta).cuda()
    enc = m.encoder(img)
    start = torch.from_numpy(g["start"].astype(np.int64)).cuda()
    mask = torch.from_numpy(g["mask"]).cuda()
    toks = m.decoder.generate(start, None, meta["max_len"], enc=enc, mask=mask)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    net = m.decoder.net(start, mask=mask, enc=enc).cpu()              # ONE masked multi-position pass (r05: csrc/prefill.h, KMASK)
    ref = torch.from_numpy(g["net_logits"])
    keep = torch.from_numpy(g["mask"])
    assert float((net - ref)[keep].abs().max()) < 1e-3
    assert bool(torch.isfinite(net).all())
    import os
    os.environ["TXO_NET_STEPWISE"] = "1"                               # the same call as single-position steps
    try:
        net_steps = m.decoder.net(start, mask=mask, enc=enc).cpu()
    finally:
        os.environ.pop("TXO_NET_STEPWISE")
    assert float((net_steps - ref)[keep].abs().max()) < 1e-3
    assert float((net_steps - net)[keep].abs().max()) < 2e-5
    # perf mode: the masked pass within bf16 noise of the reference at the positions that are not padding, finite everywhere
    _, _, mb = build(meta, dtype="bf16", max_batch=3)
    netb = mb.decoder.net(start, mask=mask, enc=mb.encoder(img)).cpu()
    errb = float((netb - ref)[keep].abs().max())
    print(f"masked one-pass decoder.net in bf16 vs the reference at non-padded positions: max |dlogit| {errb:.4f}")
    assert errb < 0.033 and bool(torch.isfinite(netb).all())          # 1.5x the measurement (0.0214)
    plain = m.decoder.generate(start, None, meta["max_len"], enc=enc)
    assert np.array_equal(plain.cpu().numpy(), g["tokens_unmasked"])
    # an all-True mask is the unmasked call; a 1-d start with a 1-d mask works like the reference's squeeze
    assert torch.equal(m.decoder.generate(start, None, 6, enc=enc, mask=torch.ones_like(mask)), plain[:, :6])
    one = m.decoder.generate(start[0], None, 6, enc=enc[:1], mask=mask[0])
    assert one.shape == (6,) and np.array_equal(one.cpu().numpy(), g["tokens"][0, :6])


def test_latent_tile_head_group_size_does_not_change_a_heads_bits():
    """lat_attn.h claims that a head's arithmetic does not depend on how many heads share its tile: engines built with
    TXO_LAT_G = 2 / 4 / 8 (heads per tile) at batch 8 and the engine's own choice must return bit-identical tokens and logits, in
    both storage types; in a beam search the k beams of an image share ONE row of k * heads heads in tiles of 16 -- TXO_LAT_G=8
    forces one tile per beam instead: same beams, same scores."""
    d = Dims(canvas=224)
    img = torch.from_numpy(synth.synth_images(8, 3, 64, 160, seed=61)).cuda()
    for dtype in ("bf16", "fp32"):
        ref = None
        for g in (None, "2", "4", "8"):
            _, _, m = build(d, seed=12, dtype=dtype, max_batch=40, latent=1, env={"TXO_LAT_G": g} if g else None)
            m.eos_token = None
            toks, logits = m.generate(img, 32, return_logits=True)
            assert m._engine.query(3) == 1
            beams = m.generate(img, 20, beam=5, return_beams=True)
            if ref is None:
                ref = (toks, logits, beams)
            else:
                assert torch.equal(toks, ref[0]) and torch.equal(logits, ref[1]), (dtype, g)
                assert torch.equal(beams[0], ref[2][0]) and torch.equal(beams[1], ref[2][1]), (dtype, g)
            del m


def test_tiled_weights_and_whole_row_requests_do_not_change_a_bit():
    """Round 5 changed the SHAPE of three kinds of memory requests, never their values: the decode projections read tiled copies of their
    weights (dec_gemm.h: w_tiled; also inside the persistent launch), the latent core writes its output in the folded projection's tiled A
    layout (a_tiled) and fetches whole encoder rows.  Engines built with TXO_W_TILED=0 / TXO_A_TILED=0 (row-major everywhere) must return
    bit-identical tokens and logits: persistent launch and launches at batch 7, the latent launch path incl. a beam search at row counts
    that are not multiples of 16 (37 rows; 185 beam rows), both storage types."""
    d = Dims(canvas=224)
    img = torch.from_numpy(synth.synth_images(37, 3, 64, 160, seed=9)).cuda()
    for dtype in ("bf16", "fp32"):
        ref = None
        for env in (None, {"TXO_W_TILED": "0"}, {"TXO_A_TILED": "0"}):
            _, _, m = build(d, seed=5, dtype=dtype, max_batch=185, env=env)
            m.eos_token = None
            (tp, lp), (tl, ll) = _both_paths(m, img[:7], 24, return_logits=True)
            assert torch.equal(tp, tl) and torch.equal(lp, ll)
            del m
            _, _, m = build(d, seed=5, dtype=dtype, max_batch=185, latent=1, env=env)
            m.eos_token = None
            lat = m.generate(img, 24, return_logits=True)
            assert m._engine.query(3) == 1
            beams = m.generate(img, 16, beam=5, return_beams=True)
            out = (tp, lp, lat[0], lat[1], beams[0], beams[1])
            if ref is None:
                ref = out
            else:
                for a, b in zip(out, ref):
                    assert torch.equal(a, b), (dtype, env)
            del m


def test_bf16_benchmark_shape_free_running_inside_batch_256_default_path():
    """The reference fixture of the benchmark's shape (cfg2_b2_224x672_t256) as rows 200 and 33 of a 256-image bf16 batch on the DEFAULT
    path at that size -- launches, cross attention in latent form, two row ranges on two streams, head groups of 8: the rows follow the
    reference's tokens until the first position whose reference margin is thinner than the bf16 logit error; logits on that common
    prefix within the bound of the batch-64 test."""
    meta, g = load_golden("cfg2_b2_224x672_t256")
    d, sd, m = build(meta, dtype="bf16", max_batch=256, max_tokens=589)
    gen = torch.Generator(device="cuda").manual_seed(23)
    img = torch.rand((256, 3, 224, 672), generator=gen, device="cuda")
    fix = images(meta).cuda()
    img[200], img[33] = fix[0], fix[1]
    toks, logits = m.generate(img, 256, return_logits=True)
    assert m._engine.query(0) == 0 and m._engine.query(3) == 1
    got_t, got_l = toks[[200, 33]].cpu().numpy(), logits[[200, 33]].cpu()
    worst_err, shortest = 0.0, 256
    for r in range(2):
        diff = np.nonzero(got_t[r] != g["tokens"][r])[0]
        k = int(diff[0]) if diff.size else 255
        v = torch.gather(got_l[r, :k + 1], 1, torch.from_numpy(g["top5_ids"][r, :k + 1].astype(np.int64)))
        worst_err = max(worst_err, float((v - torch.from_numpy(g["top5_vals"][r, :k + 1])).abs().max()))
        shortest = min(shortest, k)
        if diff.size:
            assert float(g["margin"][r, k]) < 2 * 0.038, (r, k, float(g["margin"][r, k]))
    print(f"cfg2 t256 bf16 free-running inside batch 256 (default path: latent, two ranges): common prefix >= {shortest} positions, "
          f"max |dlogit| over the top-5 on it {worst_err:.4f}")
    assert worst_err < 0.038 and shortest >= 32
    # without per-step logits the batch runs on two row ranges (the default): the same tokens
    t2 = m.generate(img, 256)
    assert m._engine.query(2) == 2
    assert torch.equal(t2, toks)


def test_sampled_decode_at_256_rows_on_its_default_path(monkeypatch):
    """decode='sample' beyond 223 images takes the greedy decode's path (launches in latent form, two row ranges): the draws must not
    depend on the ranges (one range vs two: bit-identical tokens), every drawn token lies inside the top-k support of the logits it
    was drawn from (k = int(0.1 * vocab) = 100 at 1000 entries... the reference keeps int((1 - 0.9) * V), utils.py:85-91), and the
    draws are reproducible."""
    d = Dims(canvas=672)
    d, sd, m = build(d, seed=0, dtype="bf16", max_batch=256, max_tokens=589)
    m.eos_token = None
    gen = torch.Generator(device="cuda").manual_seed(31)
    img = torch.rand((256, 3, 224, 672), generator=gen, device="cuda")
    t2 = m.generate(img, 48, temp=0.3, decode="sample", seed=5)
    assert m._engine.query(0) == 0 and m._engine.query(2) == 2 and m._engine.query(3) == 1
    assert torch.equal(m.generate(img, 48, temp=0.3, decode="sample", seed=5), t2)
    monkeypatch.setenv("TXO_LANES", "1")
    t1, l1 = m.generate(img, 48, temp=0.3, decode="sample", seed=5, return_logits=True)
    assert m._engine.query(2) == 1
    assert torch.equal(t1, t2)
    k = int((1 - 0.9) * d.vocab)
    kth = l1.topk(k, dim=-1).values[..., -1]
    drawn = torch.gather(l1, 2, t1[..., None])[..., 0]
    assert bool((drawn >= kth).all())                                     # support within the top-k of the step's logits
    assert not torch.equal(t1, m.generate(img, 48))                       # draws, not arg-maxes


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_latent_self_attention_history(dtype):
    """Opt-in form (TXO_LATENT_SELF=1): the decoder SELF attention against the history of normalised block inputs z instead of k / v
    (attention.py:114-127 with kv_input = x; folded q' and Wo' as in the cross attention).  fp32: logits within 2e-5 of the K/V history
    and the reference fixture of the benchmark's shape (256 greedy steps) token-exact in this form; bf16: within bf16 noise of the
    K/V history; beam search (per-beam slot tables through the latent core) against the oracle's beam search."""
    cpu_ref = _oracle()
    env = {"TXO_LATENT_SELF": "1"}
    d = Dims(canvas=224)
    img = torch.from_numpy(synth.synth_images(6, 3, 64, 160, seed=71))
    _, sd, m_kv = build(d, seed=14, dtype=dtype, max_batch=30, latent=1)
    _, _, m_z = build(d, seed=14, dtype=dtype, max_batch=30, latent=1, env=env)
    m_kv.eos_token = None; m_z.eos_token = None
    tk, lk = m_kv.generate(img.cuda(), 40, return_logits=True)
    tz, lz = m_z.generate(img.cuda(), 40, return_logits=True)
    same = (tk == tz).all(1)
    k_common = int(((tk == tz).int().cumprod(1).sum(1)).min())             # shortest common prefix over the rows
    err = float((lk[:, :max(k_common, 1)] - lz[:, :max(k_common, 1)]).abs().max())
    print(f"latent self history vs K/V history ({dtype}): common prefix >= {k_common} of 40, max |dlogit| on it {err:.2e}, rows equal {int(same.sum())}/6")
    if dtype == "fp32":
        assert k_common == 40 and err < 2e-5
    else:
        assert k_common >= 8 and err < 0.06
    if dtype == "fp32":
        sdt = cpu_ref.to_torch_sd(sd)
        enc = cpu_ref.encode(sdt, img[:3])
        toks, scores = m_z.generate(img[:3].cuda(), 24, beam=4, return_beams=True)
        ref_t, ref_s = cpu_ref.beam_search_cached(sdt, enc, d.bos, None, 24, 4)
        assert torch.equal(toks.cpu(), ref_t)
        np.testing.assert_allclose(scores.cpu().numpy(), ref_s.numpy(), atol=2e-3)
        del m_kv, m_z
        meta, g = load_golden("cfg2_b2_224x672_t256")
        _, _, m = build(meta, max_batch=16, max_tokens=589, latent=1, env=env)
        gen = torch.Generator(device="cuda").manual_seed(21)
        big = torch.rand((16, 3, 224, 672), generator=gen, device="cuda")
        fix = images(meta).cuda()
        big[11], big[2] = fix[0], fix[1]
        toks, logits = m.generate(big, 256, return_logits=True)
        assert np.array_equal(toks[[11, 2]].cpu().numpy(), g["tokens"])
        v = torch.gather(logits[[11, 2]].cpu(), 2, torch.from_numpy(g["top5_ids"].astype(np.int64)))
        assert float((v - torch.from_numpy(g["top5_vals"])).abs().max()) < 1e-3
    else:
        # bf16 beam search in this form: the width-256 bf16 tile runs on FOUR waves, so its beam slot table covers 16 x 4 x 4 = 256
        # positions.  Inside that range the z history must reproduce the K/V history's beams (within bf16 noise: the same best first tokens);
        # a positional table beyond it (max_length 300) makes the engine decline the form and take the K/V history -- bit for bit the engine
        # built without the knob -- instead of reading another beam's history.
        bk = m_kv.generate(img[:3].cuda(), 24, beam=4).cpu()
        bz = m_z.generate(img[:3].cuda(), 24, beam=4).cpu()
        assert bz.shape == bk.shape and int((bz == bk).int().cumprod(1).sum(1).min()) >= 4
        d300 = Dims(canvas=224, max_len=300)
        _, sd3, a = build(d300, seed=14, dtype="bf16", max_batch=30, latent=1)
        _, _, b = build(d300, seed=14, dtype="bf16", max_batch=30, latent=1, env=env)
        assert torch.equal(a.generate(img[:3].cuda(), 280, beam=4), b.generate(img[:3].cuda(), 280, beam=4))

