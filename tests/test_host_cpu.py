"""Host-side logic that needs no GPU: config mapping, the synthetic-weight recipe, the data-parallel
sharding helpers (incl. a 2-process gloo run of the all-gather + global-eos trim)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from texocr_amd import synth
from texocr_amd.config import Dims, default_config
from texocr_amd.dist import all_gather_rows, global_eos_steps, shard_bounds, sharded_generate


def test_dims_from_reference_config():
    d = Dims.from_config(default_config())
    assert (d.embed_dim, d.enc_heads, d.enc_layers, d.dec_layers, d.vocab, d.max_len) == (256, 8, 4, 4, 1000, 256)
    assert (d.bos, d.eos, d.pad) == (998, 997, 999)
    assert d.enc_inner == 512 and d.enc_ffn == 1024 and d.n_tokens(224, 672) == 589 and d.n_pos == 1 + 42 * 42
    cfg = default_config()
    del cfg["max_length"]
    with pytest.raises(ValueError, match="max_length"):
        Dims.from_config(cfg)
    with pytest.raises(ValueError, match="embed_dim must match"):
        Dims.from_config(default_config(encoder={"embed_dim": 768}))
    with pytest.raises(ValueError):
        Dims(canvas=224).check_image(3, 224, 230)
    with pytest.raises(ValueError):
        Dims(canvas=224).check_image(3, 224, 448)
    with pytest.raises(ValueError):
        Dims(canvas=224).check_image(1, 224, 224)


def test_synth_is_deterministic_and_param_count():
    d = Dims(canvas=672)
    a, b = synth.synth_state_dict(d, 0), synth.synth_state_dict(d, 0)
    assert len(a) == 144                                   # default config has 144 state_dict keys
    assert all(np.array_equal(a[k], b[k]) for k in a)
    uniq = {id(v): v for v in a.values()}
    assert sum(v.size for v in uniq.values()) == 15_409_896   # north-star shape, unique params
    assert sum(v.size for v in a.values()) == 15_419_112       # state_dict elements incl. aliased LN keys
    c = synth.synth_state_dict(d, 1)
    assert not np.array_equal(a["decoder.net.to_logits.weight"], c["decoder.net.to_logits.weight"])
    img = synth.synth_images(2, 3, 32, 48, 11)
    assert img.dtype == np.float32 and img.min() >= 0 and img.max() < 1


def test_shard_bounds_cover():
    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _ref_break(tokens, eos, bos):
    """decoder.py:97-116 restated as a loop over an already decoded token matrix."""
    out = torch.full((tokens.shape[0], 1), bos, dtype=torch.long)
    for i in range(tokens.shape[1]):
        out = torch.cat([out, tokens[:, i:i + 1]], 1)
        if eos is not None and (out == eos).any(dim=1).all():
            break
    return out.shape[1] - 1


def test_global_eos_steps_matches_reference_loop():
    g = torch.Generator().manual_seed(0)
    for _ in range(200):
        B, T = int(torch.randint(1, 5, (1,), generator=g)), int(torch.randint(1, 12, (1,), generator=g))
        toks = torch.randint(0, 4, (B, T), generator=g)
        for eos, bos in ((2, 3), (3, 3), (None, 3), (7, 3)):
            assert global_eos_steps(toks, eos, bos) == _ref_break(toks, eos, bos), (toks, eos, bos)


def _fake_generate(img, max_len):
    # deterministic per-image "decode": depends only on that image -> rows are independent like the real path
    base = (img.reshape(img.shape[0], -1).sum(1) * 1000).long()
    return (base[:, None] + torch.arange(max_len)[None, :] * 7) % 13


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        imgs = torch.from_numpy(synth.synth_images(total, 1, 4, 4, seed=3))
        out = sharded_generate(_fake_generate, imgs, 10, eos=5, bos=12)
        lo, hi = shard_bounds(total, rank, world)
        out2 = sharded_generate(_fake_generate, imgs[lo:hi], 10, eos=5, bos=12, images_are_local=True, global_batch=total)
        tk, lg = sharded_generate(lambda x, n: (_fake_generate(x, n), _fake_generate(x, n)[..., None].float() * torch.ones(3)),
                                  imgs, 10, eos=5, bos=12, gather_logits=True)
        assert torch.equal(tk, out) and lg.shape == (out.shape[0], out.shape[1], 3) and torch.equal(lg[..., 0].long(), out)
        rows = all_gather_rows(torch.full((hi - lo, 2), rank), [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0]
                                                               for r in range(world)])
        # per-row stop on the gathered batch: the global break's step count, pad behind every row's first eos
        out_row = sharded_generate(_fake_generate, imgs, 10, eos=5, bos=12, stop="row", pad=99)
        assert out_row.shape == out.shape
        for b in range(out.shape[0]):
            hit = (out[b] == 5).nonzero()
            n = int(hit[0]) + 1 if hit.numel() else out.shape[1]
            assert torch.equal(out_row[b, :n], out[b, :n]) and bool((out_row[b, n:] == 99).all())
        q.put((rank, out.numpy(), out2.numpy(), rows.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_sharded_generate_gloo_world2(total):
    world, port = 2, 29500 + (os.getpid() % 2000) + total
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    imgs = torch.from_numpy(synth.synth_images(total, 1, 4, 4, seed=3))
    full = _fake_generate(imgs, 10)
    want = full[:, :global_eos_steps(full, 5, 12)].numpy()
    for rank, out, out2, rows in res:
        assert np.array_equal(out, want) and np.array_equal(out2, want)
        exp_rows = np.concatenate([np.full((shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0], 2), r)
                                   for r in range(world)])
        assert np.array_equal(rows, exp_rows)


def test_tokenizer_matches_reference_vectors(tmp_path):
    """N4: byte-level BPE encode/decode against vectors produced by the reference's RegExTokenizer with its 1k vocabulary
    (merge table exported as data by oracle/capture_golden.py), and the reference's 3-line vocabulary file format
    (tokenizer.py:110-125) through a save()/load() round trip."""
    import json
    from texocr_amd.tokenizer import RegExTokenizer, process_output
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(root, "tokenizer_cases.json")))
    v = json.load(open(os.path.join(root, "tokenizer_vocab_1k.json")))
    tk0 = RegExTokenizer.from_tables(v["vocab_size"], v["special_tokens"], v["merges"])
    path = tmp_path / "vocab.txt"
    tk0.save(str(path))
    lines = open(path).read().splitlines()
    assert lines[0] == "1000" and lines[1].startswith("{'<PAD>': 999") and lines[2].startswith("{(115, 115): 256, (32, 95): 257")
    tk = RegExTokenizer()
    tk.load(str(path))
    assert tk.bp_merges == tk0.bp_merges
    assert tk.vocab_size == g["vocab_size"] == 1000
    assert tk.special_tokens == {"<PAD>": 999, "<BOS>": 998, "<EOS>": 997} == g["special_tokens"]
    assert len(tk.bp_merges) == 741                       # 256 bytes + 741 merges + 3 specials
    for c in g["cases"]:
        assert tk.encode(c["text"]) == c["ids"], c["text"]
        assert tk.decode(c["ids"]) == c["decoded"]
    for text, want in g["process_output"]:
        assert process_output(text) == want
    # the survey's known answer
    assert tk.encode(r"\int _ { 0 } ^ { 1 } x ^ 2 d x") == [615, 257, 258, 381, 261, 266, 258, 259, 261, 334, 266, 260, 264, 334]
    assert process_output(r"\int _ { 0 } ^ { 1 } x ^ 2 d x") == r"\int_{0}^{1}x^2dx"
    with pytest.raises(ValueError):
        tk.decode([5000])


def test_wrapper_preprocess():
    from PIL import Image
    from texocr_amd.wrapper import preprocess_image
    img = Image.new("RGB", (50, 20), (255, 255, 255))
    img.putpixel((3, 4), (0, 0, 0))
    x = preprocess_image(img)
    assert x.shape == (1, 32, 64) and x.dtype == torch.float32
    # white -> 1 - (0.2989 + 0.587 + 0.114) = 1e-4 (torchvision's luma weights), black -> 1, padding = 0
    assert float(x[0, 4, 3]) > 0.99 and float(x.sum()) < 1.2 and float(x[0, 25, 55]) == 0.0


def test_generate_bucketed_groups_by_exact_size():
    from texocr_amd.dist import generate_bucketed
    g = torch.Generator().manual_seed(1)
    sizes = [(16, 32), (16, 64), (16, 32), (32, 32), (16, 64), (16, 32), (16, 32)]
    imgs = [torch.rand((1, h, w), generator=g) for h, w in sizes]
    calls = []

    def fn(batch):
        calls.append(tuple(batch.shape))
        return (batch.reshape(batch.shape[0], -1).sum(1, keepdim=True) * 100).long().expand(-1, 3)

    out = generate_bucketed(fn, imgs, max_batch=3)
    assert sorted(calls) == sorted([(3, 1, 16, 32), (1, 1, 16, 32), (2, 1, 16, 64), (1, 1, 32, 32)])
    for im, o in zip(imgs, out):
        assert int(o[0]) == int(im.sum() * 100) and o.shape == (3,)


def test_custom_ops_registered_with_fake_impls():
    """torch.ops.texocr.* exist without a GPU and their fake implementations give the output shapes (what torch.compile /
    FakeTensorMode need); real calls on CPU tensors are refused (no CPU fallback)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from texocr_amd import ops
    d = Dims(canvas=672)

    class Stub:                                                  # what a fake implementation needs of an engine: its dims
        dims, device = d, 0
    stub = Stub()
    eid = ops.register_engine(stub)
    for name in ("encode", "decode_begin", "decode_step", "generate", "generate_from_enc", "generate_beam"):
        assert hasattr(torch.ops.texocr, name)
    with FakeTensorMode():
        img = torch.empty((4, 3, 224, 672))
        enc = torch.ops.texocr.encode(img, eid)
        assert enc.shape == (4, 589, 256) and enc.dtype == torch.float32
        toks, n, logits = torch.ops.texocr.generate(img, eid, 256, d.eos, True)
        assert toks.shape == (4, 256) and toks.dtype == torch.int64 and n.shape == (1,) and logits.shape == (4, 256, 1000)
        toks, n, logits = torch.ops.texocr.generate_from_enc(enc, eid, 32, -1, False)
        assert toks.shape == (4, 32) and logits.shape == (0, 32, 1000)
        lg, nxt = torch.ops.texocr.decode_step(torch.empty((4,), dtype=torch.int64), eid, 0, 4, True)
        assert lg.shape == (4, 1000) and nxt.shape == (4,) and nxt.dtype == torch.int64
        best, scores, allt, n = torch.ops.texocr.generate_beam(img, eid, 5, 64, d.eos, True)
        assert best.shape == (4, 64) and scores.shape == (4, 5) and allt.shape == (20, 64)
    with pytest.raises((ValueError, RuntimeError)):
        torch.ops.texocr.encode(torch.zeros(1, 3, 224, 224), eid)            # CPU tensor: refused, never computed on the host
    ops.unregister_engine(eid)
    with pytest.raises(RuntimeError, match="no live engine"):
        torch.ops.texocr.encode(torch.zeros(1, 3, 224, 224), eid)


def _fake_rows(batch, max_len, eos):
    """per-image token rows that depend on the image only (so sharding cannot change them): eos appears at a position
    derived from the image, junk continues behind it as in the reference's greedy loop"""
    out = torch.empty((batch.shape[0], max_len), dtype=torch.int64)
    for b in range(batch.shape[0]):
        key = int(batch[b].sum().item() * 1000) % 97
        row = (torch.arange(max_len) * 7 + key) % 50 + 10
        row[3 + key % (max_len - 4)] = eos
        out[b] = row
    return out


def _fake_greedy(batch, max_len):
    return _fake_rows(batch, max_len, 5)


def _fake_beam(batch, max_len):
    """beam-search shape: finished rows repeat eos; the call returns only as many columns as its LOCAL batch needs"""
    rows = _fake_rows(batch, max_len, 5)
    first = (rows == 5).float().argmax(dim=1)
    for b in range(rows.shape[0]):
        rows[b, first[b] + 1:] = 5
    return rows[:, : int(first.max()) + 1]


def _bucket_worker(rank, world, port, beam, q):
    import torch.distributed as dist
    from texocr_amd.dist import sharded_generate_bucketed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        imgs = _bucket_images()
        rows = sharded_generate_bucketed(_fake_beam if beam else _fake_greedy, imgs, 24, 5, max_batch=2, beam=beam)
        q.put((rank, [r.tolist() for r in rows]))
    finally:
        dist.destroy_process_group()


def _bucket_images():
    g = torch.Generator().manual_seed(3)
    widths = [32, 64, 32, 96, 64, 32, 32, 64, 96, 32, 32]          # three buckets: 6 / 3 / 2 images
    return [torch.rand((1, 16, w), generator=g) for w in widths]


@pytest.mark.parametrize("beam", [False, True])
def test_sharded_generate_bucketed_gloo_world2(beam):
    """BASELINE config 5's composition (bucket -> shard within the bucket -> greedy / beam -> ONE gather) on 2 ranks gives
    what one process gives; with max_batch=2 per rank the 6-image bucket becomes two global batches of 4 and 2."""
    import torch.multiprocessing as mp
    from texocr_amd.dist import sharded_generate_bucketed, bucket_plan
    imgs = _bucket_images()
    plan1 = bucket_plan(imgs, 4)
    assert [len(c) for c in plan1] == [4, 2, 3, 2] and sorted(sum(plan1, [])) == list(range(len(imgs)))
    # the single-process result uses the SAME global batches (max_batch * world images): that is what the ranks shard
    single = sharded_generate_bucketed(_fake_beam if beam else _fake_greedy, imgs, 24, 5, max_batch=4, beam=beam)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (1 if beam else 0) + os.getpid() % 200
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, beam, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == [r.tolist() for r in single]
    # rows of one global batch share their length (GLOBAL break / longest beam); different batches differ
    lens = [len(r) for r in got[0]]
    for chunk in plan1:
        assert len({lens[i] for i in chunk}) == 1
    assert all(5 in r for r in got[0])


# ------------------------------------------------------------------------------------------------
# bench.py's own step composition on two gloo ranks (the N > 1 path of the benchmark, without GPUs)
# ------------------------------------------------------------------------------------------------
_BENCH_DIMS = dict(canvas=64, in_channels=3, embed_dim=64, enc_heads=2, enc_layers=1, dec_heads=2, dec_layers=1, vocab=16,
                   max_len=24, bos=14, eos=5, pad=15)


def _bench_step_worker(rank, world, port, per_rank, eos, q):
    import importlib.util
    import torch.distributed as dist
    from oracle import cpu_ref
    from texocr_amd.config import Dims
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        d = Dims(**_BENCH_DIMS)
        sd = cpu_ref.to_torch_sd(synth.synth_state_dict(d, 11))
        torch.set_num_threads(2)
        # the rank's own images, as bench.py draws them (seed + rank)
        imgs = [torch.from_numpy(synth.synth_images(per_rank, 3, 32, 48, seed=100 + rank + 10 * k)) for k in range(2)]
        step = bench.make_step(True, lambda x, n: cpu_ref.generate_cached(sd, x, d.bos, None, n), None, imgs, 24, eos, d.bos,
                               per_rank * world, expect_full=False)
        out = [step(i).numpy() for i in range(2)]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_bench_step_composition_gloo_world2():
    """bench.py --gpus N (N > 1): every rank decodes its shard with the eos test off, ONE all-gather, the GLOBAL break on the
    gathered batch.  Two gloo ranks running bench.make_step with the oracle as the per-rank generator must return, on every
    rank, what ONE process returns for the concatenated batch with the reference's eos break -- for an eos that breaks early
    and for one that never fires for all rows."""
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    d = Dims(**_BENCH_DIMS)
    sd = cpu_ref.to_torch_sd(synth.synth_state_dict(d, 11))
    per_rank, world = 3, 2
    both = [torch.cat([torch.from_numpy(synth.synth_images(per_rank, 3, 32, 48, seed=100 + r + 10 * k)) for r in range(world)]) for k in range(2)]
    free = cpu_ref.generate_cached(sd, both[0], d.bos, None, 24).numpy()
    common = [t for t in range(d.vocab) if (free == t).any(axis=1).all()]
    assert common, "no token emitted by every row: pick another weight seed"
    never = [t for t in range(d.vocab) if not (free == t).any()]
    cases = [common[0]] + never[:1]
    ctx = mp.get_context("spawn")
    for ci, eos in enumerate(cases):
        q = ctx.Queue()
        port = 29800 + ci + os.getpid() % 150
        procs = [ctx.Process(target=_bench_step_worker, args=(r, world, port, per_rank, eos, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for k in range(2):
            want = cpu_ref.generate_cached(sd, both[k], d.bos, eos, 24).numpy()       # one process, the reference's break
            for r in range(world):
                assert np.array_equal(got[r][k], want), (eos, k, r, got[r][k].shape, want.shape)
        if eos == common[0]:
            assert got[0][0].shape[1] < 24, "the crafted eos must break early"
