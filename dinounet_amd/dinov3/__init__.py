from .vision_transformer import DinoVisionTransformer, VIT_CONFIGS, build_backbone  # noqa: F401
from .adapter import DINOv3_Adapter, MSDeformAttn  # noqa: F401
