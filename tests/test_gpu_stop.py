"""GPU tests of the per-row stop (stop='row': txo_set_stop_mode, SURVEY D7 / 8f N2 -- a build extension: the reference's loop,
model/decoder.py:97-118, only breaks globally and returns whatever finished rows kept producing).

What is pinned to the reference: rows are independent, so with stop='row' every row's tokens up to and including its first eos, and
the position of the loop's end, must be the reference's (the `eos_break` fixture captured from it); behind a row's first eos there is
`pad`.  The oracle's stop='row' is the reference algorithm plus that rewrite (oracle/cpu_ref.py: pad_after_eos).  On the launch path
the engine also COMPACTS the live rows of a row range (Engine::compact_lane): the tests force compactions every other position and
check that nothing a row produces depends on the slot it sits in -- fp32 bit-exact against the oracle, sampled draws unchanged."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import build, images, _oracle
from texocr_amd import synth
from texocr_amd.config import Dims

pytestmark = pytest.mark.gpu

Q_LAST_PERSISTENT, Q_LAST_RANGES, Q_LAST_COMPACTIONS = 0, 2, 5


class knobs:
    """TXO_PERSIST / TXO_LANES / TXO_GRAPH are re-read by the binding whenever the environment changes between two calls: set them
    around the calls (the TXO_STOP_* periods are read once, when the engine is created: build(env=...))."""
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k in self.kv:
            os.environ.pop(k, None)


STOP_ENV = {"TXO_STOP_EVERY": "2", "TXO_STOP_GAIN": "1"}


def _first_eos(tok, eos):
    return [int(np.nonzero(r == eos)[0][0]) if (r == eos).any() else -1 for r in tok]


def test_row_stop_reference_fixture_both_paths():
    """eos_break fixture (captured from the reference): stop='row' returns the reference's tokens up to each row's first eos, pad behind
    it, the reference's number of steps (the fixture's model is 64 wide: launch path, with and without a captured step graph; the persistent
    launch is covered by test_row_stop_persistent_launch_and_stepwise_loop)."""
    ref = _oracle()
    meta, g = load_golden("eos_break")
    d, sd, m = build(meta)
    img = images(meta).cuda()
    for mode in ("1", "0"):
        os.environ["TXO_GRAPH"] = mode
        try:
            for case in meta["cases"]:
                m.eos_token = case["eos"]
                want_ref = torch.from_numpy(g[f"tokens_{case['name']}"].astype(np.int64))
                full = torch.cat([torch.full((want_ref.shape[0], 1), d.bos, dtype=torch.long), want_ref], 1)
                want = ref.pad_after_eos(full, 1, case["eos"], d.pad).numpy()
                t = m.generate(img, meta["max_len"], stop="row")
                assert t.shape[1] == case["n_steps"], (mode, case)
                assert np.array_equal(t.cpu().numpy(), want), (mode, case)
                # and the default is untouched by the call before it: the reference's global break, junk and all
                t0 = m.generate(img, meta["max_len"])
                assert np.array_equal(t0.cpu().numpy(), g[f"tokens_{case['name']}"]), (mode, case)
        finally:
            os.environ.pop("TXO_GRAPH")


STOP_DIMS = Dims(canvas=64, in_channels=3, embed_dim=64, enc_heads=2, enc_layers=2, dec_heads=2, dec_layers=2, vocab=64, max_len=48,
                 bos=62, eos=61, pad=63)


def _stop_case(rows=40, bias=1.2):
    """40 tiny images of different contrast, eos favoured by a logit bias: rows produce their first eos anywhere between position 0
    and 46, some never (tuned on the oracle; its smallest top-1/top-2 margin is 5e-4, a hundred times the fp32 error)."""
    d = STOP_DIMS
    sd = synth.synth_state_dict(d, 7)
    b = sd["decoder.net.to_logits.bias"].copy()
    b[d.eos] += bias
    sd["decoder.net.to_logits.bias"] = b
    img = torch.from_numpy(synth.synth_images(rows, 3, 32, 48, seed=11)) * torch.linspace(0.2, 3.0, rows)[:, None, None, None]
    return d, sd, img


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("latent", [0, 1])
@pytest.mark.parametrize("lanes", [1, 2])
def test_row_stop_compaction_matches_oracle_fp32(latent, lanes, graph):
    """graph = 1: the default -- the shrinking ranges replay captured steps, one per row count (multiples of 16 rows); 0: eager launches,
    any row count."""
    ref = _oracle()
    d, sd, img = _stop_case()
    want = ref.generate_cached(ref.to_torch_sd(sd), img, d.bos, d.eos, d.max_len, stop="row", pad=d.pad).numpy()
    first = _first_eos(want, d.eos)
    assert len(set(first)) >= 6 and min(f for f in first if f >= 0) < 8 and max(first) > 30 and -1 in first   # the schedule the test is about
    _, _, m = build(d, sd=sd, max_batch=40, latent=latent, env=dict(STOP_ENV, TXO_STOP_GRAPH=str(graph)))
    min_comp = 1 if graph else 3                                  # (40 rows: at most two multiples of 16 to step down to)
    with knobs(TXO_LANES=lanes):
        t = m.generate(img.cuda(), d.max_len, stop="row")
    assert m._engine.query(Q_LAST_PERSISTENT) == 0 and m._engine.query(Q_LAST_RANGES) == lanes
    assert m._engine.query(Q_LAST_COMPACTIONS) >= min_comp
    assert np.array_equal(t.cpu().numpy(), want)
    # the rows that do finish, alone: now the loop ends at the last row's eos, with compactions on the way
    keep = [i for i, f in enumerate(first) if f >= 0]
    want2 = ref.generate_cached(ref.to_torch_sd(sd), img[keep], d.bos, d.eos, d.max_len, stop="row", pad=d.pad).numpy()
    assert want2.shape[1] == max(first) + 1 < d.max_len
    with knobs(TXO_LANES=lanes):
        t2 = m.generate(img[keep].cuda(), d.max_len, stop="row")
        t2b = m.generate(img[keep].cuda(), d.max_len, stop="row")          # (again: the captured steps of the first call are replayed)
    assert m._engine.query(Q_LAST_COMPACTIONS) >= min_comp
    assert np.array_equal(t2.cpu().numpy(), want2) and torch.equal(t2, t2b)
    # a compacted session is not continued: the next call starts from scratch and the default mode is the reference's again
    t3 = m.generate(img[keep].cuda(), d.max_len)
    glob = ref.generate_cached(ref.to_torch_sd(sd), img[keep], d.bos, d.eos, d.max_len).numpy()
    assert m._engine.query(Q_LAST_COMPACTIONS) == 0 and np.array_equal(t3.cpu().numpy(), glob)


def test_row_stop_sampled_draws_do_not_depend_on_the_slot():
    """decode='sample' with stop='row': a draw is keyed by (seed; row of the BATCH, position), and the key follows a row through every
    compaction -- the tokens equal the uncompacted run's up to each row's first eos."""
    d, sd, img = _stop_case()
    _, _, m = build(d, sd=sd, max_batch=40, env=STOP_ENV)
    x = img.cuda()
    with knobs(TXO_LANES=2):
        glob = m.generate(x, d.max_len, temp=0.5, decode="sample", seed=123).cpu()
        row = m.generate(x, d.max_len, temp=0.5, decode="sample", seed=123, stop="row").cpu()
    assert m._engine.query(Q_LAST_RANGES) == 2
    assert m._engine.query(Q_LAST_COMPACTIONS) >= 2
    ref = _oracle()
    full = torch.cat([torch.full((glob.shape[0], 1), d.bos, dtype=torch.long), glob], 1)
    assert np.array_equal(row.numpy(), ref.pad_after_eos(full, 1, d.eos, d.pad).numpy())


WIDE_DIMS = Dims(canvas=64, in_channels=3, embed_dim=256, enc_heads=8, enc_layers=1, dec_heads=8, dec_layers=2, vocab=200, max_len=40,
                 bos=198, eos=197, pad=199)


@pytest.mark.parametrize("bias,breaks", [(2.5, True), (2.0, False)])
def test_row_stop_persistent_launch_and_stepwise_loop(bias, breaks):
    """A 256-wide decoder (the persistent launch exists for it), 24 rows whose first eos falls anywhere in 0..18 (bias 2.5: the loop
    breaks at 19 steps) or 1..30 with one row never finishing (bias 2.0: runs to max_len).  Persistent launch (no compaction there: the
    rewrite behind the decode), launch path with compactions, and the facade's general stepwise loop: the oracle's tokens, exactly."""
    ref = _oracle()
    d = WIDE_DIMS
    sd = synth.synth_state_dict(d, 5)
    b = sd["decoder.net.to_logits.bias"].copy()
    b[d.eos] += bias
    sd["decoder.net.to_logits.bias"] = b
    img = torch.from_numpy(synth.synth_images(24, 3, 32, 64, seed=3)) * torch.linspace(0.2, 3.0, 24)[:, None, None, None]
    want = ref.generate_cached(ref.to_torch_sd(sd), img, d.bos, d.eos, d.max_len, stop="row", pad=d.pad).numpy()
    assert (want.shape[1] < d.max_len) == breaks and len(set(_first_eos(want, d.eos))) >= 8
    x = img.cuda()
    _, _, m = build(d, sd=sd, max_batch=24, env=STOP_ENV)
    with knobs(TXO_PERSIST=1):
        t = m.generate(x, d.max_len, stop="row")
    assert m._engine.query(Q_LAST_PERSISTENT) == 1 and m._engine.query(Q_LAST_COMPACTIONS) == 0
    assert np.array_equal(t.cpu().numpy(), want)
    with knobs(TXO_PERSIST=0):
        t = m.generate(x, d.max_len, stop="row")
    assert m._engine.query(Q_LAST_PERSISTENT) == 0 and m._engine.query(Q_LAST_COMPACTIONS) >= 1
    assert np.array_equal(t.cpu().numpy(), want)
    enc = m.encoder(x)
    start = torch.full((x.shape[0], 1), d.bos, dtype=torch.long, device="cuda")
    t2 = m.decoder.generate(start, d.eos, d.max_len, enc=enc, stop="row", pad=d.pad)
    assert np.array_equal(t2.cpu().numpy(), want)
    pre = torch.cat([start, torch.from_numpy(want[:, :1]).cuda()], 1)                 # a two-token prefix: the stepwise loop
    t3 = m.decoder.generate(pre, d.eos, d.max_len - 1, enc=enc, stop="row", pad=d.pad)
    assert np.array_equal(t3.cpu().numpy(), want[:, 1:])


def test_row_stop_bf16_batch_256_compacts_and_keeps_every_row():
    """config.yml dims, 256 images, bf16, the launch path's two row ranges with the default compaction period: against the SAME engine's
    global-break run, every row's tokens up to its first eos are identical (a row's arithmetic does not depend on how many rows its
    range still has) and the rest is pad."""
    d = Dims(canvas=224)
    sd = synth.synth_state_dict(d, 0)
    b = sd["decoder.net.to_logits.bias"].copy()
    b[d.eos] += 2.6
    sd["decoder.net.to_logits.bias"] = b
    _, _, m = build(d, sd=sd, dtype="bf16", max_batch=256)
    img = torch.rand((256, 3, 64, 224), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    img = img * torch.linspace(0.3, 2.0, 256, device="cuda")[:, None, None, None]
    glob = m.generate(img, 128).cpu()
    row = m.generate(img, 128, stop="row").cpu()
    first = _first_eos(glob.numpy(), d.eos)
    assert m._engine.query(Q_LAST_PERSISTENT) == 0
    n_fin = sum(f >= 0 for f in first)
    assert n_fin >= 32, f"schedule too thin for the test: {n_fin} rows finish ({sorted(first)})"
    assert m._engine.query(Q_LAST_COMPACTIONS) >= 1
    ref = _oracle()
    full = torch.cat([torch.full((256, 1), d.bos, dtype=torch.long), glob], 1)
    want = ref.pad_after_eos(full, 1, d.eos, d.pad).numpy()
    assert row.shape == glob.shape
    assert np.array_equal(row.numpy(), want)


@pytest.mark.parametrize("bias,breaks", [(1.4, True), (1.0, False)])
def test_row_stop_beyond_the_positional_table(bias, breaks):
    """max_len > decoder.max_len: the reference slides its window (decoder.py:99-100); rows whose first eos falls inside the window region
    (positions 8..) are padded behind it like any other, and the loop ends where the reference's does.  Against the oracle's RECOMPUTE mode
    (the reference algorithm as written) with stop='row'; persistent-launch and launch-path engines alike take the window's multi-position pass."""
    ref = _oracle()
    d = Dims(canvas=64, in_channels=3, embed_dim=64, enc_heads=2, enc_layers=2, dec_heads=2, dec_layers=2, vocab=64, max_len=8, bos=62, eos=61, pad=63)
    sd = synth.synth_state_dict(d, 7)
    b = sd["decoder.net.to_logits.bias"].copy()
    b[d.eos] += bias
    sd["decoder.net.to_logits.bias"] = b
    img = torch.from_numpy(synth.synth_images(6, 3, 32, 48, seed=11)) * torch.linspace(0.2, 3.0, 6)[:, None, None, None]
    want = ref.generate_recompute(ref.to_torch_sd(sd), img, d.bos, d.eos, 24, stop="row", pad=d.pad).numpy()
    first = _first_eos(want, d.eos)
    assert (want.shape[1] < 24) == breaks and max(first) >= d.max_len            # at least one row finishes beyond the table
    _, _, m = build(d, sd=sd, max_batch=6, env=STOP_ENV)
    t = m.generate(img.cuda(), 24, stop="row")
    assert m._engine.query(Q_LAST_COMPACTIONS) == 0                              # (no compaction when the window may be needed)
    assert np.array_equal(t.cpu().numpy(), want)
    glob = m.generate(img.cuda(), 24).cpu()
    full = torch.cat([torch.full((6, 1), d.bos, dtype=torch.long), glob], 1)
    assert np.array_equal(ref.pad_after_eos(full, 1, d.eos, d.pad).numpy(), want)


def test_row_stop_with_return_logits_pads_and_does_not_compact():
    d, sd, img = _stop_case()
    ref = _oracle()
    want = ref.generate_cached(ref.to_torch_sd(sd), img, d.bos, d.eos, d.max_len, stop="row", pad=d.pad).numpy()
    _, _, m = build(d, sd=sd, max_batch=40, env=STOP_ENV)
    t, lg = m.generate(img.cuda(), d.max_len, stop="row", return_logits=True)
    assert m._engine.query(Q_LAST_COMPACTIONS) == 0 and lg.shape == (40, d.max_len, d.vocab)
    assert np.array_equal(t.cpu().numpy(), want)
    enc = m.encoder(img.cuda())
    start = torch.full((40, 1), d.bos, dtype=torch.long, device="cuda")
    t2, lg2 = m.decoder.generate(start, d.eos, d.max_len, enc=enc, stop="row", pad=d.pad, return_logits=True)
    assert np.array_equal(t2.cpu().numpy(), want) and torch.equal(lg2, lg)


def test_stop_mode_argument_checks():
    meta, g = load_golden("eos_break")
    d, sd, m = build(meta)
    with pytest.raises(ValueError):
        m.generate(images(meta).cuda(), 8, stop="never")
    from texocr_amd import _lib
    assert m._engine.lib.txo_set_stop_mode(m._engine.handle, 7) == _lib.TXO_E_INVALID
    assert m._engine.lib.txo_set_stop_mode(None, 0) == _lib.TXO_E_INVALID
