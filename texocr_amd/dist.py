"""Multi-GPU sharding of OCRModel.generate(): one process per GPU, images sharded across ranks, ONE
collective at the end (all-gather of the generated token ids; optionally of the logits).

The reference has no multi-device code.  The path shards naturally: every image's
encode + decode is independent of every other image except for the GLOBAL eos break of
AutoRegressiveDecoder.generate (reference model/decoder.py:115-116).  To keep that break exact without a
per-step collective, every rank decodes its shard to ``max_len`` with the eos check disabled, the token
blocks are all-gathered (RCCL over xGMI on GPUs: backend "nccl"; gloo on CPU in the tests), and the global
break is applied afterwards by trimming at the first step where every row of the GLOBAL batch contains eos
(rows never interact, so the trimmed result equals the single-device result).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` rows for `rank`; sizes differ by at most one row."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_eos_steps(tokens: torch.Tensor, eos: Optional[int], bos: Optional[int] = None) -> int:
    """Number of columns the reference would return for this GLOBAL batch: it stops after the first step
    at which every row contains eos (the BOS column counts too, decoder.py:115)."""
    B, T = tokens.shape
    if eos is None or T == 0:
        return T
    seen = (tokens == eos).cumsum(dim=1) > 0
    if bos is not None and bos == eos:
        seen = torch.ones_like(seen)
    all_seen = seen.all(dim=0)
    idx = torch.nonzero(all_seen)
    return int(idx[0].item()) + 1 if idx.numel() else T


def pad_after_eos(tokens: torch.Tensor, eos: Optional[int], pad: int, bos: Optional[int] = None) -> torch.Tensor:
    """stop='row' (build extension, see texocr.h: txo_set_stop_mode) on a gathered batch: every token behind a row's first eos becomes `pad`
    (a row whose BOS already is eos is finished from the start, decoder.py:115 looks at the whole output)."""
    if eos is None or tokens.numel() == 0:
        return tokens
    is_eos = (tokens == eos).to(torch.int32)
    prior = (torch.cumsum(is_eos, dim=1) - is_eos) > 0
    if bos is not None and bos == eos:
        prior = torch.ones_like(prior)
    return torch.where(prior, torch.full_like(tokens, pad), tokens)


def all_gather_rows(local: torch.Tensor, counts: List[int], group=None, force: bool = False) -> torch.Tensor:
    """All-gather row blocks of possibly different heights (pads to the tallest, one collective).  force: issue the
    collective even in a one-rank group (bench.py under torchrun --nproc-per-node 1 times the RCCL call itself)."""
    world = len(counts)
    if world == 1 and not (force and dist.is_initialized()):
        return local
    tall = max(counts)
    if local.shape[0] < tall:
        pad = torch.zeros((tall - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    out = torch.empty((world * tall,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    if all(c == tall for c in counts):
        return out
    return torch.cat([out[r * tall: r * tall + c] for r, c in enumerate(counts)], dim=0)


def sharded_generate(generate_local: Callable[[torch.Tensor, int], torch.Tensor], images: torch.Tensor, max_len: int,
                     eos: Optional[int], bos: Optional[int] = None, group=None,
                     images_are_local: bool = False, global_batch: Optional[int] = None, gather_logits: bool = False,
                     force_collective: bool = False, stop: str = "global", pad: Optional[int] = None):
    """Data-parallel generate.

    generate_local(img_shard, max_len) -> (b_local, max_len) int64 tokens decoded WITHOUT the eos break
    (e.g. ``lambda x, n: model._engine.generate(x, n, eos=None)``).
    `images` is either the global batch (every rank holds it; the rank's shard is sliced out) or, with
    images_are_local=True, already this rank's shard (then `global_batch` gives the total row count).
    Returns the GLOBAL (B, n_steps) token tensor on every rank.  With gather_logits=True, generate_local returns
    (tokens, logits (b_local, max_len, V)) and the per-step logits are all-gathered as well (one more collective,
    B/G * T * V * 4 bytes per rank -- 65 MB at 64 x 256 x 1000), returning (tokens, logits).
    stop='row' (needs `pad`): the per-row stop's rewrite -- pad behind every row's first eos -- applied to the gathered, trimmed batch; the
    number of steps is the global break's (what one device returns for the whole batch with stop='row')."""
    if stop not in ("global", "row") or (stop == "row" and pad is None):
        raise ValueError("stop must be 'global' or 'row' (with the pad id)")

    def unpack(r):
        return r if gather_logits else (r, None)

    def finish(t):
        return pad_after_eos(t, eos, pad, bos) if stop == "row" else t
    if not dist.is_initialized():
        toks, lg = unpack(generate_local(images, max_len))
        n = global_eos_steps(toks, eos, bos)
        return (finish(toks[:, :n]), lg[:, :n]) if gather_logits else finish(toks[:, :n])
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if images_are_local:
        total = global_batch if global_batch is not None else images.shape[0] * world
        local = images
    else:
        total = images.shape[0]
        lo, hi = shard_bounds(total, rank, world)
        local = images[lo:hi]
    counts = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} images, expected {counts[rank]}")
    if not local.shape[0]:
        raise ValueError("every rank needs at least one image")
    toks, lg = unpack(generate_local(local, max_len))
    if toks.shape[1] != max_len:
        raise ValueError("generate_local must decode exactly max_len steps (eos break disabled)")
    full = all_gather_rows(toks, counts, group, force=force_collective)
    n = global_eos_steps(full, eos, bos)
    if gather_logits:
        return finish(full[:, :n]), all_gather_rows(lg, counts, group, force=force_collective)[:, :n]
    return finish(full[:, :n])


def generate_bucketed(generate_fn: Callable[[torch.Tensor], torch.Tensor], images, max_batch: int = 64):
    """Variable-size inputs: group images of identical (H, W) into batches, exactly like the reference's
    BucketBatchSampler (data_wrangling/dataset.py:281-326 buckets by exact (w, h), :306-310), run each bucket through
    ``generate_fn(batch) -> (b, n) tokens`` and return the per-image token rows in the original order.
    Each bucket keeps the reference's GLOBAL eos break among its own rows, as a reference batch would."""
    buckets = {}
    for i, im in enumerate(images):
        if im.ndim != 3:
            raise ValueError("each image must be (C, H, W)")
        buckets.setdefault((int(im.shape[1]), int(im.shape[2])), []).append(i)
    out = [None] * len(images)
    for _, idx in sorted(buckets.items()):
        for lo in range(0, len(idx), max_batch):
            part = idx[lo: lo + max_batch]
            toks = generate_fn(torch.stack([images[i] for i in part]).contiguous())
            for j, i in enumerate(part):
                out[i] = toks[j]
    return out


def bucket_plan(images, global_batch: int) -> List[List[int]]:
    """Indices of `images` grouped like the reference's BucketBatchSampler (data_wrangling/dataset.py:281-326: one bucket per
    exact (w, h), :306-310) and cut into batches of at most `global_batch` images; buckets in sorted (H, W) order."""
    buckets = {}
    for i, im in enumerate(images):
        if im.ndim != 3:
            raise ValueError("each image must be (C, H, W)")
        buckets.setdefault((int(im.shape[1]), int(im.shape[2])), []).append(i)
    plan = []
    for _, idx in sorted(buckets.items()):
        for lo in range(0, len(idx), global_batch):
            plan.append(idx[lo: lo + global_batch])
    return plan


def sharded_generate_bucketed(generate_local: Callable[[torch.Tensor, int], torch.Tensor], images, max_len: int,
                              eos: Optional[int], bos: Optional[int] = None, group=None, max_batch: int = 64,
                              beam: bool = False, force_collective: bool = False):
    """BASELINE config 5 across the ranks of one node: variable-width images are bucketed by exact size, every batch of a
    bucket is sharded ACROSS the ranks (so all ranks run the same shapes at the same time), each rank decodes its shard
    (greedy, or beam search whose beams stay on their image's rank), and ONE all-gather returns every batch's token rows
    to every rank.  Returns one 1-D token tensor per image, in the order of `images`.

    generate_local(batch (b, C, H, W), max_len) -> (b, n) int64 tokens.
      beam=False: greedy WITHOUT the eos break (n == max_len, e.g. ``lambda x, n: engine.generate(x, n, eos=None)``); the
        reference's GLOBAL break (decoder.py:115-116) is applied per batch after the gather, over the rows of ALL ranks.
      beam=True: the engine's beam search with its eos handling on (n <= max_len: the local loop stops when every beam of
        every LOCAL image is finished; a finished beam only repeats eos, at no cost, so rows are padded with eos up to the
        batch's longest shard -- what an unsharded run of the whole batch returns).
    `images` is the GLOBAL list (every rank holds it, or at least the images of its own shards).
    force_collective: issue the all-gather even in a one-rank group (tests / bench on one GPU exercise the RCCL call itself)."""
    if beam and eos is None:
        raise ValueError("beam search needs an eos token")
    ready = dist.is_initialized()
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if ready else (1, 0)
    plan = bucket_plan(images, max_batch * world)
    dev = images[0].device
    blocks, talls = [], []
    for chunk in plan:
        tall = (len(chunk) + world - 1) // world
        lo, hi = shard_bounds(len(chunk), rank, world)
        block = torch.full((tall, max_len + 1), eos if (beam and eos is not None) else 0, dtype=torch.int64, device=dev)
        block[:, max_len] = 0
        if hi > lo:
            toks = generate_local(torch.stack([images[i] for i in chunk[lo:hi]]).contiguous(), max_len)
            if toks.shape[0] != hi - lo or toks.shape[1] > max_len or (not beam and toks.shape[1] != max_len):
                raise ValueError("generate_local returned an unexpected shape")
            block[: hi - lo, : toks.shape[1]] = toks
            block[: hi - lo, max_len] = toks.shape[1]
        blocks.append(block)
        talls.append(tall)
    local = torch.cat(blocks, dim=0)
    if world > 1 or (force_collective and ready):                     # (force_collective: the RCCL call itself also in a one-rank group)
        full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        full = full.view(world, local.shape[0], max_len + 1)
    else:
        full = local[None]
    out = [None] * len(images)
    row0 = 0
    for chunk, tall in zip(plan, talls):
        rows, lens = [], []
        for r in range(world):
            lo, hi = shard_bounds(len(chunk), r, world)
            rows.append(full[r, row0: row0 + (hi - lo), :max_len])
            lens.append(full[r, row0: row0 + (hi - lo), max_len])
        toks = torch.cat(rows, dim=0)                                 # the batch's rows in chunk order
        n = int(torch.cat(lens).max().item()) if beam else global_eos_steps(toks, eos, bos)
        for j, i in enumerate(chunk):
            out[i] = toks[j, :n]
        row0 += tall
    return out
