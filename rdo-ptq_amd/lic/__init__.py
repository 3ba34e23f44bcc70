"""Learned-image-compression building blocks the quantisation package wraps (GDN, Cheng2020 residual blocks, entropy
models, the Cheng2020-anchor topology).  The reference takes these from CompressAI 1.2.4 (requirements.txt:1), which is
not vendored; if a real `compressai` is importable its classes are recognised as well (quantization/quant_block.py)."""
from .layers import (AttentionBlock, GDN, MaskedConv2d, NonNegativeParametrizer, ResidualBlock, ResidualBlockUpsample,
                     ResidualBlockWithStride, conv1x1, conv3x3, subpel_conv3x3)
from .entropy import EntropyBottleneck, GaussianConditional
from .cheng2020 import Cheng2020Anchor, Cheng2020Attention
from .minnen2018 import JointAutoregressiveHierarchicalPriors, MeanScaleHyperprior
from .nic import NIC, RSTB

__all__ = ["AttentionBlock", "Cheng2020Attention", "GDN", "MaskedConv2d", "NonNegativeParametrizer", "ResidualBlock", "ResidualBlockUpsample",
           "ResidualBlockWithStride", "conv1x1", "conv3x3", "subpel_conv3x3", "EntropyBottleneck", "GaussianConditional",
           "Cheng2020Anchor", "JointAutoregressiveHierarchicalPriors", "MeanScaleHyperprior", "NIC", "RSTB"]
