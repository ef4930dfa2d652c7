"""Model dimensions for the OCRModel.generate() path.

The reference builds its module tree from a flat YAML dict (reference
``utils.py:24-28``; keys read at ``model/encoder.py:174-190``,
``model/decoder.py:150-172``, ``model/ocr_model.py:115-128``).  ``Dims.from_config``
accepts that same dict.  Keys the reference does not have (``img_size``,
``in_channels``) select the plain ``PatchEmbedding`` front end that the
north-star path uses (reference ``model/encoder.py:11-28``; the default factory
hard-wires the hybrid ResNet embedder instead, which is a "next" row).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

DIM_HEAD = 64  # reference model/attention.py:76 -- never overridden by AttentionLayers (:214-216)
PATCH = 16


RESNET_DEPTHS = (2, 4, 6)          # create_encoder hard-wires ResNetV2(depths=[2,4,6]) (encoder.py:177-180)
RESNET_CHANNELS = (256, 512, 1024)  # ResNetV2 default channels (resnet.py:204)
HYBRID_CANVAS = (160, 1008)        # create_encoder's img_size (encoder.py:183)


@dataclass(frozen=True)
class Dims:
    canvas: int = 672          # square max canvas in pixels (VisionTransformer img_size, encoder.py:95) ...
    canvas_w: int = 0          # ... or canvas x canvas_w when non-zero (the hybrid factory uses 160 x 1008)
    embed: str = "patch"       # "patch": PatchEmbedding (encoder.py:11-28) | "hybrid": ResNetV2 [2,4,6] + 1x1 proj (:31-72,162-191)
    in_channels: int = 3
    embed_dim: int = 256       # encoder and decoder width must match (no enc->dec projection, attention.py:89-91)
    enc_heads: int = 8
    enc_layers: int = 4
    dec_heads: int = 8
    dec_layers: int = 4
    enc_exp: int = 4           # encoder always gets the MLP defaults (encoder.py:182-190 passes no ff_kwargs)
    dec_exp: int = 4
    vocab: int = 1000
    max_len: int = 256         # decoder positional table length (config 'max_length')
    bos: int = 998
    eos: int = 997
    pad: int = 999
    patch: int = PATCH

    # derived ---------------------------------------------------------------
    @property
    def canvas_hw(self):
        return (self.canvas, self.canvas_w or self.canvas)

    @property
    def grid_h(self) -> int:
        return self.canvas // self.patch

    @property
    def grid(self) -> int:
        """patches per row of the max canvas (the stride of the position-id grid, encoder.py:137-139)"""
        return (self.canvas_w or self.canvas) // self.patch

    @property
    def n_pos(self) -> int:
        return 1 + self.grid_h * self.grid

    @property
    def enc_inner(self) -> int:
        return self.enc_heads * DIM_HEAD

    @property
    def dec_inner(self) -> int:
        return self.dec_heads * DIM_HEAD

    @property
    def enc_ffn(self) -> int:
        return self.enc_exp * self.embed_dim

    @property
    def dec_ffn(self) -> int:
        return self.dec_exp * self.embed_dim

    def n_tokens(self, h: int, w: int) -> int:
        return 1 + (h // self.patch) * (w // self.patch)

    def check_image(self, c: int, h: int, w: int) -> None:
        if c != self.in_channels:
            raise ValueError(f"image has {c} channels, model expects {self.in_channels}")
        if h % self.patch or w % self.patch or h <= 0 or w <= 0:
            raise ValueError(f"image size {h}x{w} must be positive multiples of {self.patch}")
        ch, cw = self.canvas_hw
        if h > ch or w > cw:
            raise ValueError(f"image size {h}x{w} exceeds the {ch}x{cw} canvas")

    def to_dict(self) -> dict:
        return asdict(self)

    @staticmethod
    def from_config(config: dict) -> "Dims":
        """Build from a reference-style config dict (config/config.yml layout)."""
        for key in ("max_length", "vocab_size"):
            if key not in config:  # reference asserts these at decoder.py:150-151
                raise ValueError(f"{key} not loaded into config")
        enc, dec = config["encoder"], config["decoder"]
        if enc["embed_dim"] != dec["embed_dim"]:
            raise ValueError("encoder and decoder embed_dim must match: cross-attention K/V are "
                             "Linear(dec_embed_dim, .) applied to the encoder output (attention.py:89-91)")
        if not dec.get("cross_attend", True):
            raise ValueError("decoder.cross_attend=false has no encoder input; not an OCR model")
        if not config.get("glu", True):
            raise ValueError("glu=false (plain GELU FFN) is not built; the shipped config uses glu: true")
        embed = config.get("embed", "hybrid" if "img_size" not in config else "patch")
        if embed not in ("patch", "hybrid"):
            raise ValueError("embed must be 'patch' or 'hybrid'")
        size = config.get("img_size", HYBRID_CANVAS if embed == "hybrid" else 672)
        ch, cw = (size, 0) if isinstance(size, int) else (int(size[0]), int(size[1]))
        if embed == "hybrid" and int(config.get("in_channels", enc.get("n_channels", 1))) != 1:
            raise ValueError("the hybrid ResNetV2 embedder is single-channel (encoder.py:177-180)")
        return Dims(
            canvas=int(ch), canvas_w=int(cw) if cw != ch else 0, embed=embed,
            in_channels=int(config.get("in_channels", enc.get("n_channels", 1))),
            embed_dim=int(enc["embed_dim"]),
            enc_heads=int(enc["heads"]), enc_layers=int(enc["num_layers"]),
            dec_heads=int(dec["heads"]), dec_layers=int(dec["num_layers"]),
            enc_exp=4, dec_exp=int(dec.get("exp_factor", 4)),
            vocab=int(config["vocab_size"]), max_len=int(config["max_length"]),
            bos=int(config.get("bos_token", 998)), eos=int(config.get("eos_token", 997)),
            pad=int(config.get("trg_pad_idx", 999)),
            patch=int(config.get("patch_size", PATCH)),
        )


def reference_config(**over) -> dict:
    """config/config.yml as create_model(config) consumes it: hybrid ResNetV2 embedder, 1 channel,
    (160, 1008) canvas -- what every trained TeXOCR checkpoint uses."""
    cfg = default_config(**over)
    for k in ("img_size", "in_channels", "embed"):
        if k not in over:
            cfg.pop(k, None)
    cfg.setdefault("embed", "hybrid")
    return cfg


def default_config(**over) -> dict:
    """The values of the reference's config/config.yml that the path reads, plus the
    run-time injected keys (train.py:33-34) and this build's front-end keys."""
    cfg = {
        "bos_token": 998, "eos_token": 997, "trg_pad_idx": 999, "patch_size": 16, "glu": True,
        "device": "cuda",
        "encoder": {"embed_dim": 256, "heads": 8, "num_layers": 4, "n_channels": 1,
                    "exp_factor": 4, "dropout": 0.1},
        "decoder": {"embed_dim": 256, "heads": 8, "num_layers": 4, "cross_attend": True,
                    "exp_factor": 4, "dropout": 0.1},
        "max_length": 256, "vocab_size": 1000,
        "img_size": 672, "in_channels": 3, "embed": "patch",
    }
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k] = {**cfg[k], **v}
        else:
            cfg[k] = v
    return cfg
