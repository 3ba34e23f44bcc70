"""Drop-in `quantization` package of RDO-PTQ's task-oriented PTQ (reference: quantization/__init__.py:1-6), MI355X-native:
the module surface is the reference's, the arithmetic runs on librdoptq_hip (HIP/gfx950) only."""
from quantization.quant_block import BaseQuantBlock
from quantization.quant_layer import QuantModule
from quantization.quant_model import QuantModel

from quantization.layer_opt import layer_reconstruction
from quantization.block_opt import block_reconstruction

__all__ = ["BaseQuantBlock", "QuantModule", "QuantModel", "layer_reconstruction", "block_reconstruction"]
