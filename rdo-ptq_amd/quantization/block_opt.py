"""`block_reconstruction`: joint AdaRound optimisation of every QuantModule inside a block
(reference: quantization/block_opt.py:176-324)."""
import torch

from .quant_block import BaseQuantBlock
from .quant_model import QuantModel
from .recon import LossFunction, find_unquantized_module, reconstruct  # noqa: F401  (surface compatibility)
from .utils import set_mode  # noqa: F401


def block_reconstruction(model: QuantModel, block: BaseQuantBlock, block_name: str, cali_data: torch.Tensor,
                         batch_size: int = 32, iters: int = 20000, weight: float = 0.01, opt_mode: str = "mse",
                         asym: bool = False, include_act_func: bool = True, b_range: tuple = (20, 2), warmup: float = 0.0,
                         input_prob: float = 1.0, act_quant: bool = False, lr: float = 4e-5, p: float = 2.0, config=None,
                         args=None):
    """Returns the calibration engine of the unit (the reference returns None): its per-iteration loss logs and the mini-batch index
    table stay readable after the run; None for a PixelShuffle pseudo-unit."""
    return reconstruct(model, block, block_name, cali_data, batch_size, iters, weight, opt_mode, asym, include_act_func, b_range,
                       warmup, input_prob, act_quant, lr, p, config, args, is_block=True)
