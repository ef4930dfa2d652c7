"""Serving front end around the engine (reference ``TeXOCRWrapper``, model/ocr_model.py:69-110): tokenizer +
checkpoint load + PIL image -> LaTeX string.

Differences from the reference, all on the host side of the path:
  * preprocessing is the inference part of ``img_transform`` (data_wrangling/dataset.py:365-371: ToTensor ->
    Grayscale(1) -> Invert) WITHOUT the train-time ``RandomAffine``; torchvision is not needed;
  * images whose sides are not multiples of 16 are padded (bottom/right, background = 0 after inversion); the
    reference relies on the dataset renderer having padded them already (render_data.py:81-92);
  * ``decode`` selects the reference's sampler ('sample', its default, temp 0.3) or greedy;
  * ``max_len`` is passed through unchanged (default 350, ocr_model.py:94): beyond the positional table the decoder
    slides its window as the reference does.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .config import Dims
from .model import OCRModel, create_model
from .tokenizer import RegExTokenizer, process_output


def preprocess_image(img, patch: int = 16) -> torch.Tensor:
    """PIL image (any mode) -> (1, H', W') float32 in [0,1], white background -> 0, sides padded to multiples of 16."""
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0                    # ToTensor
    g = 0.2989 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]                  # Grayscale (ITU-R 601-2 luma)
    g = 1.0 - g                                                                      # Invert (dataset.py:65-76)
    h, w = g.shape
    ph, pw = (-h) % patch, (-w) % patch
    if ph or pw:
        g = np.pad(g, ((0, ph), (0, pw)), constant_values=0.0)
    return torch.from_numpy(np.ascontiguousarray(g))[None]


class TeXOCRWrapper:
    def __init__(self, config: dict, dtype: str = "fp32", max_batch: int = 1):
        self.tokenizer = RegExTokenizer()
        self.tokenizer.load(config["tokenizer_path"])                                # ocr_model.py:74-75
        config = dict(config)
        config["vocab_size"] = self.tokenizer.vocab_size                             # :78
        sd = None
        if config.get("model_path"):
            sd = torch.load(config["model_path"], map_location="cpu", weights_only=True)
            if "model_state_dict" in sd:                                             # utils.save_checkpoint layout (:52-61)
                sd = sd["model_state_dict"]
            key = "decoder.net.pos_embedding.embedding.weight"
            if key in sd:                                                            # ocr_model.py:84-88: the checkpoint decides
                config["max_length"] = int(sd[key].shape[0])
        self.model: OCRModel = create_model(config, dtype=dtype, max_batch=max_batch)
        if sd is not None:
            self.model.load_state_dict(sd)
        self.dims: Dims = self.model._engine.dims

    def __call__(self, img, max_len: int = 350, temp: float = 0.3, decode: str = "sample",
                 seed: Optional[int] = None) -> Tuple[list, str]:
        x = preprocess_image(img, self.dims.patch)
        if self.dims.in_channels != 1:
            x = x.expand(self.dims.in_channels, -1, -1)
        x = x[None].contiguous().cuda()
        # max_len may exceed the positional table (the reference's default 350 does for short tables): the model then
        # slides its window exactly as the reference does (decoder.py:99-100), at window-length engine steps per token
        toks = self.model.generate(x, max_len=max_len, temp=temp, decode=decode, seed=seed)
        out_tokens = toks.squeeze(0).tolist()[:-1]                                   # ocr_model.py:104 (drops the EOS)
        return out_tokens, process_output(self.tokenizer.decode(out_tokens))         # :105-108
