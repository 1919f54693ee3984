from .dinounet import (DinoUNet, DINOv3EncoderAdapter, FAPM, UNetDecoder, SqueezeExcitation, DepthwiseSeparableConv,  # noqa: F401
                       LearnableUpsampleBlock, StackedConvBlocks, ConvDropoutNormReLU, DINOv3_MODEL_INFO,
                       DINOv3_INTERACTION_INDEXES, DINOv3_MODEL_FACTORIES, load_dinov3_model)
from .utils import softmax_helper, to_cuda  # noqa: F401
