from .post_processor_rotated_boxes import POST_PROCESSOR_REGISTRY, build_post_processor  # noqa: F401
from . import post_processor_academic  # noqa: F401  (registers PostProcessorAcademic)
