"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A CPU (torch fp32) restatement of the arithmetic of the reference's ``OCRModel.generate()``
hot path, written from the behaviour of the cited reference lines.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module,
and there only as the checker / the timed CPU baseline.  ``texocr_amd`` never imports it.

Parity status: PINNED.  The reference has no tests or golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against outputs of the reference
itself, produced in the build container by ``oracle/capture_golden.py`` (which imports
``/root/reference``) and committed as fixtures under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every function here against those fixtures.

Third-party arithmetic the reference relies on (un-vendored, unpinned in its
requirements.txt): torch ATen CPU kernels -- ``F.layer_norm`` (eps 1e-5, biased variance),
``F.gelu`` (exact erf), ``F.softmax`` (fp32), ``nn.GLU`` (a * sigmoid(b)), ``nn.Conv2d`` --
and einops reshapes.  Captured with torch 2.10.0, einops 0.8.2.

All tensors fp32, token ids int64.  ``sd`` is a reference-layout state dict of torch
tensors (see ``texocr_amd.synth.state_dict_layout``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

DIM_HEAD = 64           # reference model/attention.py:76
SCALE = DIM_HEAD ** -0.5  # attention.py:80 (0.125 exactly)
SD = Dict[str, torch.Tensor]


def to_torch_sd(sd_np) -> SD:
    return {k: torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v for k, v in sd_np.items()}


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm(D): eps 1e-5, affine, biased variance (attention.py:200, encoder.py:121,
    decoder.py:30)."""
    return F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


def _split_heads(t: torch.Tensor) -> torch.Tensor:
    # 'b n (h d) -> b h n d' (attention.py:127): feature f = head*64 + d
    B, n, inner = t.shape
    return t.view(B, n, inner // DIM_HEAD, DIM_HEAD).permute(0, 2, 1, 3)


def mha(sd: SD, p: str, xq: torch.Tensor, src: torch.Tensor, causal: bool,
        kv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, q_mask: Optional[torch.Tensor] = None,
        k_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MultiHeadAttention.forward (attention.py:101-180).  q_mask (B,nq) / k_mask (B,nk) bool: the reference's padding masks
    (:130-155): energy[b,:,i,j] is filled with -FLT_MAX where not (q_mask[b,i] and k_mask[b,j]) -- a fully masked query row
    therefore softmaxes uniformly over ALL keys.  None = all True.

    ``xq`` (B,nq,D) queries' input, ``src`` (B,nk,D) keys'/values' input (== xq for self
    attention, the raw encoder output for cross attention, attention.py:113-126).  If ``kv``
    is given it is the already projected, head-split (k, v) -- used by the cached decoder.
    Causal fill: S[i, j] = -FLT_MAX for j > i + (nk - nq) (attention.py:158-163; the
    ``F.pad(mask, (j-i, 0))`` makes it correct for nq < nk)."""
    q = _split_heads(xq @ sd[f"{p}.q.weight"].t())                         # :124,127 (no bias :89)
    if kv is None:
        k = _split_heads(src @ sd[f"{p}.k.weight"].t())                    # :125
        v = _split_heads(src @ sd[f"{p}.v.weight"].t())                    # :126
    else:
        k, v = kv
    energy = torch.matmul(q, k.transpose(-1, -2)) * SCALE                  # :148
    if q_mask is not None or k_mask is not None:                           # :130-155
        B_, nq_, nk_ = energy.shape[0], energy.shape[-2], energy.shape[-1]
        qm = q_mask if q_mask is not None else torch.ones((B_, nq_), dtype=torch.bool)
        km = k_mask if k_mask is not None else torch.ones((B_, nk_), dtype=torch.bool)
        energy = energy.masked_fill(~(qm[:, None, :, None] & km[:, None, None, :]), -torch.finfo(energy.dtype).max)
    if causal:
        nq, nk = energy.shape[-2:]
        i = torch.arange(nq).view(nq, 1)
        j = torch.arange(nk).view(1, nk)
        energy = energy.masked_fill(j > i + (nk - nq), -torch.finfo(energy.dtype).max)  # :149,163
    attn = F.softmax(energy, dim=-1)                                       # :166
    out = torch.matmul(attn, v)                                            # :172
    B, H, nq, dh = out.shape
    out = out.permute(0, 2, 1, 3).reshape(B, nq, H * dh)                   # :173
    y = out @ sd[f"{p}.fc_out.0.weight"].t() + sd[f"{p}.fc_out.0.bias"]    # :96-97
    a, g = y.chunk(2, dim=-1)                                              # nn.GLU :98
    return a * torch.sigmoid(g)


def project_kv(sd: SD, p: str, src: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return (_split_heads(src @ sd[f"{p}.k.weight"].t()), _split_heads(src @ sd[f"{p}.v.weight"].t()))


def ffn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """MLP with GeGLU (attention.py:9-17, 41-67): value = first half, gelu gate = second half."""
    u = x @ sd[f"{p}.fc_in.fc.weight"].t() + sd[f"{p}.fc_in.fc.bias"]
    a, g = u.chunk(2, dim=-1)
    return (a * F.gelu(g)) @ sd[f"{p}.fc_out.weight"].t() + sd[f"{p}.fc_out.bias"]


def stack(sd: SD, prefix: str, kinds: List[str], x: torch.Tensor, enc: Optional[torch.Tensor],
          causal: bool, trace: Optional[list] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """AttentionLayers.forward (attention.py:223-269): ONE shared LayerNorm (:200,221), applied
    before every block and again after every residual add except the last (:242-259)."""
    g, b = sd[f"{prefix}.layers.0.0.weight"], sd[f"{prefix}.layers.0.0.bias"]
    last = len(kinds) - 1
    for s, kind in enumerate(kinds):
        p = f"{prefix}.layers.{s}.1"
        r = x
        z = layer_norm(x, g, b)
        if kind == "self":
            o = mha(sd, p, z, z, causal, q_mask=mask, k_mask=mask)        # attention.py:136-141: k_mask = q_mask for self attention
        elif kind == "cross":
            o = mha(sd, p, z, enc, False, q_mask=mask)                    # ... and enc_mask (None -> all True) for cross attention
        else:
            o = ffn(sd, p, z)
        x = o + r
        if s != last:
            x = layer_norm(x, g, b)
        if trace is not None:
            trace.append(x)
    return x


def kinds_of(sd: SD, prefix: str) -> List[str]:
    kinds, s = [], 0
    while f"{prefix}.layers.{s}.0.weight" in sd:
        p = f"{prefix}.layers.{s}.1"
        if f"{p}.fc_in.fc.weight" in sd:
            kinds.append("mlp")
        elif prefix.startswith("decoder") and s % 3 == 1:
            kinds.append("cross")
        else:
            kinds.append("self")
        s += 1
    return kinds


# --------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------
def pos_ids(h: int, w: int, canvas_grid: int) -> torch.Tensor:
    """[0] ++ (r * max_w + c + 1) over the MAX-canvas grid (encoder.py:136-141)."""
    r = torch.arange(h).view(h, 1)
    c = torch.arange(w).view(1, w)
    return torch.cat([torch.zeros(1, dtype=torch.long), (r * canvas_grid + c + 1).reshape(-1)])


def patch_embed(sd: SD, img: torch.Tensor) -> torch.Tensor:
    """Conv2d(C, D, k=16, s=16)+bias, flatten(2).transpose(1,2) (encoder.py:22,25-28)."""
    w = sd["encoder.patch_embed.proj.weight"]
    x = F.conv2d(img, w, sd["encoder.patch_embed.proj.bias"], stride=w.shape[-1])
    return x.flatten(2).transpose(1, 2)


# ---- hybrid CNN embedder (the default factory's front end) ------------------------------------------
def _same_pad(x: torch.Tensor, k: int, s: int, value: float = 0.0) -> torch.Tensor:
    """utils.pad_same / get_same_padding (utils.py:97-99,112-123): TF-style SAME, extra pixel bottom/right."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + (k - 1) + 1 - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + (k - 1) + 1 - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to bfloat16 and back: what one store of the bf16 engine does to an fp32 value"""
    return x.to(torch.bfloat16).to(torch.float32)


def std_conv(x: torch.Tensor, w: torch.Tensor, stride: int, q=None) -> torch.Tensor:
    """StdConv2d.forward (resnet.py:54-66): weights standardised per output channel (biased variance, eps 1e-6);
    static symmetric padding when stride == 1 and (k-1) even (utils.py:101-110), dynamic SAME otherwise.
    q: optional rounding applied where the bf16 engine stores (standardised weights, conv output) -- see resnet_backbone."""
    k = w.shape[-1]
    # same ATen call as the reference, so the standardised weights are bit-identical to its
    ws = F.batch_norm(w.reshape(1, w.shape[0], -1), None, None, training=True, momentum=0.0, eps=1e-6).reshape_as(w)
    if q is not None:
        ws = q(ws)
    if stride == 1 and (k - 1) % 2 == 0:
        y = F.conv2d(x, ws, None, 1, (k - 1) // 2)
    else:
        y = F.conv2d(_same_pad(x, k, stride), ws, None, stride, 0)
    return q(y) if q is not None else y


def group_norm_act(sd: SD, p: str, x: torch.Tensor, act: bool, q=None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNormAct (resnet.py:14-35): 32 groups, eps 1e-5, optional ReLU.  With `res` the residual add and the ReLU
    behind it (Bottleneck.forward, resnet.py:143-149) are applied before the (optional) rounding, as the engine's fused
    normalise kernel does."""
    x = F.group_norm(x, 32, sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-5)
    if res is not None:
        x = x + res
    x = F.relu(x) if act else x
    return q(x) if q is not None else x


def resnet_backbone(sd: SD, p: str, img: torch.Tensor, q=None) -> torch.Tensor:
    """ResNetV2(depths=[2,4,6]).forward (resnet.py:200-254) with Bottleneck.forward (:143-149).

    q = bf16_round gives a CPU EMULATION of the engine's bf16 mode for this backbone: every tensor the engine stores as bf16
    (input pixels as a GEMM operand, standardised weights, every conv output, every GroupNorm output) is rounded there, all
    arithmetic stays fp32.  It is not the reference's arithmetic; it exists to show what bf16 storage alone does to these 45
    layers with random weights (tests/test_oracle_golden.py, tests/test_gpu_parity.py)."""
    x = std_conv(q(img) if q is not None else img, sd[f"{p}.stem.0.weight"], 2, q)
    x = group_norm_act(sd, f"{p}.stem.1", x, True, q)
    x = F.max_pool2d(_same_pad(x, 3, 2, value=-float("inf")), 3, 2)                 # resnet.py:69-79
    for st, depth in enumerate((2, 4, 6)):
        for i in range(depth):
            b = f"{p}.stages.{st}.stage_blocks.{i}"
            stride = 2 if (i == 0 and st > 0) else 1
            res = x
            if f"{b}.downsample.conv.weight" in sd:
                res = group_norm_act(sd, f"{b}.downsample.norm", std_conv(x, sd[f"{b}.downsample.conv.weight"], stride, q), False, q)
            y = group_norm_act(sd, f"{b}.block_list.1", std_conv(x, sd[f"{b}.block_list.0.weight"], 1, q), True, q)
            y = group_norm_act(sd, f"{b}.block_list.3", std_conv(y, sd[f"{b}.block_list.2.weight"], stride, q), True, q)
            x = group_norm_act(sd, f"{b}.block_list.5", std_conv(y, sd[f"{b}.block_list.4.weight"], 1, q), True, q, res=res)
    return x


def hybrid_embed(sd: SD, img: torch.Tensor, q=None) -> torch.Tensor:
    """HybridEmbedding.forward (encoder.py:65-72): backbone, 1x1 proj conv (+bias), flatten(2).transpose(1,2)."""
    f = resnet_backbone(sd, "encoder.patch_embed.backbone_net", img, q)
    w = sd["encoder.patch_embed.proj.weight"]
    f = F.conv2d(f, q(w) if q is not None else w, sd["encoder.patch_embed.proj.bias"])
    return f.flatten(2).transpose(1, 2)


def encode(sd: SD, img: torch.Tensor, trace: Optional[list] = None, grid_w: Optional[int] = None, backbone_q=None) -> torch.Tensor:
    """VisionTransformer.forward (encoder.py:128-152), head = Identity.  grid_w = patches per row of the max
    canvas (canvas_w / 16); defaults to a square canvas.  backbone_q: see resnet_backbone (bf16 emulation, hybrid only)."""
    B, C, H, W = img.shape
    hybrid = "encoder.patch_embed.backbone_net.stem.0.weight" in sd
    P = 16 if hybrid else sd["encoder.patch_embed.proj.weight"].shape[-1]
    grid = grid_w or int(round(math.sqrt(sd["encoder.pos_embed"].shape[1] - 1)))
    x = hybrid_embed(sd, img, backbone_q) if hybrid else patch_embed(sd, img)
    x = torch.cat([sd["encoder.cls_token"].expand(B, -1, -1), x], dim=1)    # :133-134
    x = x + sd["encoder.pos_embed"][:, pos_ids(H // P, W // P, grid)]       # :143
    x = stack(sd, "encoder.attn_layers", kinds_of(sd, "encoder.attn_layers"), x, None, False, trace)
    return layer_norm(x, sd["encoder.norm.weight"], sd["encoder.norm.bias"])  # :148


# --------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------
def decoder_net(sd: SD, tokens: torch.Tensor, enc: torch.Tensor, trace: Optional[list] = None,
                mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Transformer.forward (decoder.py:41-67): logits for ALL positions; mask (B,t) bool = its `mask` argument (None: all True)."""
    T = tokens.shape[1]
    x = sd["decoder.net.token_embedding.weight"][tokens]                    # :51
    x = x + sd["decoder.net.pos_embedding.embedding.weight"][:T][None]      # :52, attention.py:30-32
    x = stack(sd, "decoder.net.attn_layers", kinds_of(sd, "decoder.net.attn_layers"), x, enc, True, trace, mask=mask)
    x = layer_norm(x, sd["decoder.net.norm.weight"], sd["decoder.net.norm.bias"])   # :57
    return x @ sd["decoder.net.to_logits.weight"].t() + sd["decoder.net.to_logits.bias"]  # :60


def topk_filter(logits: torch.Tensor, threshold: float = 0.9) -> torch.Tensor:
    """utils.topk (utils.py:85-91): keep k = int((1-threshold)*V) largest, others -inf."""
    k = int((1 - threshold) * logits.shape[-1])
    val, ind = torch.topk(logits, k)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(1, ind, val)
    return out


def _all_rows_have_eos(output: torch.Tensor, eos: Optional[int]) -> bool:
    # decoder.py:115-116 -- a GLOBAL break: every row must contain eos somewhere (BOS column included)
    return eos is not None and bool((output == eos).any(dim=1).all())


def pad_after_eos(output: torch.Tensor, T0: int, eos: Optional[int], pad: int) -> torch.Tensor:
    """stop='row' -- a BUILD EXTENSION the reference does not have (its loop, decoder.py:97-118, only breaks globally and returns
    whatever the finished rows kept producing): a row is finished at its first eos (start tokens included, as the test at :115 looks
    at the whole output) and every token behind it is `pad`.  Rows are independent (attention.py:101-180 never mixes batch rows), so the
    tokens up to each row's first eos -- and the position of the global break -- are the reference's.  output: (B, T0 + n) with the
    start tokens; returns the (B, n) generated part."""
    toks = output[:, T0:].clone()
    if eos is None:
        return toks
    for b in range(output.shape[0]):
        hits = (output[b] == eos).nonzero()
        if hits.numel():
            first = int(hits[0])                                            # column of the whole output
            toks[b, max(first + 1 - T0, 0):] = pad
    return toks


@torch.no_grad()
def generate_recompute(sd: SD, img: torch.Tensor, bos: int, eos: Optional[int], max_len: int,
                       net_max_len: Optional[int] = None, collect_logits: bool = False, grid_w: Optional[int] = None,
                       start_tokens: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                       stop: str = "global", pad: int = 0):
    """The reference algorithm as written: full-prefix recompute each step, cross K/V
    re-projected each step, sliding window (ocr_model.py:46-66, decoder.py:77-122), with the
    sampler replaced by argmax (greedy; argmax survives top-k and softmax(/temp))."""
    enc = encode(sd, img, grid_w=grid_w)
    B = img.shape[0]
    net_max_len = net_max_len or sd["decoder.net.pos_embedding.embedding.weight"].shape[0]
    output = torch.full((B, 1), bos, dtype=torch.long) if start_tokens is None else start_tokens.clone()   # ocr_model.py:57 / decoder.generate's start_tokens
    T0 = output.shape[1]
    m = torch.ones_like(output, dtype=torch.bool) if mask is None else mask.clone()                       # decoder.py:95
    steps = []
    for _ in range(max_len):                                                # decoder.py:97
        x = output[:, -net_max_len:]                                        # :99
        m = m[:, -net_max_len:]                                             # :100
        logits = decoder_net(sd, x, enc, mask=None if mask is None else m)[:, -1, :]   # :103
        m = F.pad(m, (0, 1), value=True)                                    # :112
        if collect_logits:
            steps.append(logits)
        nxt = logits.argmax(dim=-1, keepdim=True)
        output = torch.cat([output, nxt], dim=-1)                           # :111
        if _all_rows_have_eos(output, eos):                                 # :115
            break
    toks = output[:, T0:] if stop == "global" else pad_after_eos(output, T0, eos, pad)   # :118
    return (toks, torch.stack(steps, 1)) if collect_logits else toks


class CachedDecoder:
    """KV-cached form of the same arithmetic: per step only the new position is
    pushed through the stack; self-attention K/V of the *normed* sub-layer input are cached,
    cross-attention K/V are projected once from the raw encoder output.  Valid while the
    prefix length stays <= the positional table (no sliding window, decoder.py:99-100)."""

    def __init__(self, sd: SD, enc: torch.Tensor):
        self.sd, self.enc = sd, enc
        self.prefix = "decoder.net.attn_layers"
        self.kinds = kinds_of(sd, self.prefix)
        self.cross = {s: project_kv(sd, f"{self.prefix}.layers.{s}.1", enc)
                      for s, k in enumerate(self.kinds) if k == "cross"}
        self.self_kv: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.t = 0

    def step(self, tok: torch.Tensor) -> torch.Tensor:
        """tok (B,) int64 = token at position self.t; returns logits (B, V) for position self.t."""
        sd = self.sd
        if self.t >= sd["decoder.net.pos_embedding.embedding.weight"].shape[0]:
            raise ValueError("prefix longer than the positional table: the reference would slide its window")
        x = (sd["decoder.net.token_embedding.weight"][tok] +
             sd["decoder.net.pos_embedding.embedding.weight"][self.t])[:, None, :]
        g, b = sd[f"{self.prefix}.layers.0.0.weight"], sd[f"{self.prefix}.layers.0.0.bias"]
        last = len(self.kinds) - 1
        for s, kind in enumerate(self.kinds):
            p = f"{self.prefix}.layers.{s}.1"
            r = x
            z = layer_norm(x, g, b)
            if kind == "self":
                k1, v1 = project_kv(sd, p, z)
                if s in self.self_kv:
                    k0, v0 = self.self_kv[s]
                    k1, v1 = torch.cat([k0, k1], 2), torch.cat([v0, v1], 2)
                self.self_kv[s] = (k1, v1)
                o = mha(sd, p, z, z, True, kv=(k1, v1))
            elif kind == "cross":
                o = mha(sd, p, z, self.enc, False, kv=self.cross[s])
            else:
                o = ffn(sd, p, z)
            x = o + r
            if s != last:
                x = layer_norm(x, g, b)
        x = layer_norm(x, sd["decoder.net.norm.weight"], sd["decoder.net.norm.bias"])
        self.t += 1
        return (x @ sd["decoder.net.to_logits.weight"].t() + sd["decoder.net.to_logits.bias"])[:, 0, :]


@torch.no_grad()
def generate_cached(sd: SD, img: torch.Tensor, bos: int, eos: Optional[int], max_len: int,
                    collect_logits: bool = False, enc: Optional[torch.Tensor] = None, stop: str = "global", pad: int = 0):
    enc = encode(sd, img) if enc is None else enc
    B = enc.shape[0]
    dec = CachedDecoder(sd, enc)
    tok = torch.full((B,), bos, dtype=torch.long)
    output = tok[:, None]
    steps = []
    for _ in range(max_len):
        logits = dec.step(tok)
        if collect_logits:
            steps.append(logits)
        tok = logits.argmax(dim=-1)
        output = torch.cat([output, tok[:, None]], dim=-1)
        if _all_rows_have_eos(output, eos):
            break
    toks = output[:, 1:] if stop == "global" else pad_after_eos(output, 1, eos, pad)
    return (toks, torch.stack(steps, 1)) if collect_logits else toks


@torch.no_grad()
def teacher_forced_logits_cached(sd: SD, enc: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
    """Cached-form equivalent of decoder_net(): logits (B, T, V) for a given prefix."""
    dec = CachedDecoder(sd, enc)
    return torch.stack([dec.step(tokens[:, t]) for t in range(tokens.shape[1])], 1)


def sample_probs(logits: torch.Tensor, temp: float) -> torch.Tensor:
    """The reference's sampling distribution for one step (decoder.py:104-107)."""
    return F.softmax(topk_filter(logits) / temp, dim=-1)


# --------------------------------------------------------------------------------------
# beam search -- NOT in the reference: parity is unpinned except k = 1 == greedy.
# Definition used by the build: length-unnormalised sum of log_softmax(logits); k beams per image, only beam 0 live at
# step 0; a beam that has emitted eos is finished and continues with eos at no cost; candidates are ranked by score with
# ties broken by the lower flat index (beam * V + token); stop when every beam of every image is finished.
# --------------------------------------------------------------------------------------
@torch.no_grad()
def beam_search_cached(sd: SD, enc: torch.Tensor, bos: int, eos: Optional[int], max_len: int, k: int):
    """Returns (tokens (B, k, n), scores (B, k)) sorted by score, best first."""
    B = enc.shape[0]
    V = sd["decoder.net.to_logits.weight"].shape[0]
    dec = CachedDecoder(sd, enc.repeat_interleave(k, dim=0))
    tok = torch.full((B * k,), bos, dtype=torch.long)
    score = torch.full((B, k), -float("inf"))
    score[:, 0] = 0.0
    fin = torch.zeros((B, k), dtype=torch.bool)
    hist = torch.zeros((B, k, 0), dtype=torch.long)
    for _ in range(max_len):
        logp = F.log_softmax(dec.step(tok), dim=-1).view(B, k, V)
        if eos is not None:
            frozen = torch.full_like(logp, -float("inf"))
            frozen[..., eos] = 0.0
            logp = torch.where(fin[..., None], frozen, logp)
        cand = (score[..., None] + logp).view(B, k * V)
        # top-k with lowest-flat-index tie break: stable sort on (-value, index)
        order = torch.sort(cand, dim=1, descending=True, stable=True).indices[:, :k]
        score = torch.gather(cand, 1, order)
        parent, token = order // V, order % V
        hist = torch.cat([torch.gather(hist, 1, parent[..., None].expand(-1, -1, hist.shape[2])), token[..., None]], dim=2)
        fin = torch.gather(fin, 1, parent) | ((token == eos) if eos is not None else torch.zeros_like(fin))
        rows = (torch.arange(B)[:, None] * k + parent).reshape(-1)
        dec.self_kv = {s_: (kk[rows], vv[rows]) for s_, (kk, vv) in dec.self_kv.items()}
        tok = token.reshape(-1)
        if eos is not None and bool(fin.all()):
            break
    return hist, score
