"""texocr_amd -- MI355X-native engine for TeXOCR's OCRModel.generate() hot path.

Importing the package does not need a GPU; creating a model does (there is no CPU fallback)."""
from .config import Dims, default_config  # noqa: F401

__all__ = ["Dims", "default_config", "create_model", "OCRModel"]


def __getattr__(name):
    if name in ("create_model", "OCRModel", "model_from_dims", "HipEngine", "VisionEncoder",
                "AutoRegressiveDecoder", "Transformer"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
