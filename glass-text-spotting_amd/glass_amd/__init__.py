"""glass_amd — MI355X-native GLASS (scene-text spotting) inference hot path.

Same module layout and registry names as the reference's `glass` package
(reference glass/__init__.py:4-9 registers by import side effect; so does this)."""
from .utils import registry  # noqa: F401
from .modeling.meta_arch import glass_rcnn as _meta  # noqa: F401  (registers meta-archs, RPN, ROI heads)
from .modeling.meta_arch.glass_rcnn import build_model  # noqa: F401

__all__ = ["build_model"]
