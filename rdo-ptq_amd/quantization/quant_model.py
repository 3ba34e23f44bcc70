"""`QuantModel`: module surgery that wraps a LIC model's layers in QuantModule / quant blocks
(reference surface: quantization/quant_model.py:10-98)."""
import torch.nn as nn

from lic import EntropyBottleneck

from .fold_bn import search_fold_and_remove_bn
from .quant_block import BaseQuantBlock, specials
from .quant_layer import GDN_TYPES, QuantModule, StraightThrough

_WRAPPABLE = (nn.Conv2d, nn.ConvTranspose2d, nn.Linear, nn.LayerNorm, nn.PixelShuffle) + GDN_TYPES
_FUSABLE_ACT = (nn.LeakyReLU, nn.GELU, nn.ReLU, nn.ReLU6)


class QuantModel(nn.Module):
    def __init__(self, model: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {}, is_fusing=True,
                 is_cheng=False):
        super().__init__()
        if is_fusing:
            search_fold_and_remove_bn(model)
        self.model = model
        self.quant_module_refactor(self.model, weight_quant_params, act_quant_params, is_cheng)

    def quant_module_refactor(self, module: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {},
                              is_cheng=False):
        """Depth-first replacement.  Exact-type matches in `specials` become blocks; wrappable leaves become QuantModules;
        an activation that directly follows a QuantModule *at the same nesting level* is fused into it and replaced by a
        StraightThrough (so the unit schedule and fused activations equal the reference's, quant_model.py:34-62)."""
        last = None
        for name, child in module.named_children():
            if type(child) in specials:
                setattr(module, name, specials[type(child)](child, weight_quant_params, act_quant_params))
            elif isinstance(child, _WRAPPABLE):
                last = QuantModule(child, weight_quant_params, act_quant_params)
                setattr(module, name, last)
            elif isinstance(child, _FUSABLE_ACT):
                if last is not None:
                    last.activation_function = child
                    setattr(module, name, StraightThrough())
            elif isinstance(child, StraightThrough):
                continue
            else:
                self.quant_module_refactor(child, weight_quant_params, act_quant_params, is_cheng)

    def _quant_modules(self):
        return [m for m in self.model.modules() if isinstance(m, QuantModule)]

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        for m in self.model.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.set_quant_state(weight_quant, act_quant)

    def forward(self, input):
        return self.model(input)

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def set_first_last_layer_to_8bit(self):
        mods = self._quant_modules()
        mods[0].weight_quantizer.bitwidth_refactor(8)
        mods[0].act_quantizer.bitwidth_refactor(8)
        mods[-1].weight_quantizer.bitwidth_refactor(8)
        mods[-2].act_quantizer.bitwidth_refactor(8)

    def disable_network_output_quantization(self):
        self._quant_modules()[-1].disable_act_quant = True
