"""Quantised Cheng2020 blocks (reference surface: quantization/quant_block.py:77-102, 219-328, 645-657).

A block owns `QuantModule`s for its convs / GDN and applies the dynamic activation quantiser after the element-wise
joins once it is trained.  The Lu2022 Swin blocks (QuantRSTB & co., quant_block.py:330-641) are not built yet."""
import torch
import torch.nn as nn

from hipops import ops
import lic

from .quant_layer import QuantModule, _nhwc, _nchw_view
from .quantizer import ActQuantizer, StraightThrough, UniformAffineQuantizer


class BaseQuantBlock(nn.Module):
    """State shared by all block wrappers; `set_quant_state` fans out to the inner QuantModules."""

    def __init__(self, act_quant_params: dict = {}):
        super().__init__()
        self.use_weight_quant = False
        self.use_act_quant = False
        self.trained = False
        self.act_quantizer = UniformAffineQuantizer(act=True, **act_quant_params)
        self.activation_function = StraightThrough()
        self.ignore_reconstruction = False

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, QuantModule):
                m.set_quant_state(weight_quant, act_quant)

    def _aq(self, x):
        return ActQuantizer(x) if (self.use_act_quant and self.trained) else x


def _lrelu(x):
    return _nchw_view(ops.lrelu(_nhwc(x)))


def _add(a, b):
    return _nchw_view(ops.add(_nhwc(a), _nhwc(b)))


class QuantRBWS(BaseQuantBlock):
    """ResidualBlockWithStride: conv3x3(s) -> LReLU -> conv3x3 -> GDN, plus a 1x1(s) skip."""
    unit_kind = "rbws"

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.conv1 = QuantModule(basic_block.conv1, weight_quant_params, act_quant_params, disable_act_quant=True)
        self.leaky_relu = basic_block.leaky_relu
        self.conv2 = QuantModule(basic_block.conv2, weight_quant_params, act_quant_params)
        self.gdn = QuantModule(basic_block.gdn, weight_quant_params, act_quant_params)
        self.skip = None if basic_block.skip is None else QuantModule(basic_block.skip, weight_quant_params, act_quant_params)

    def forward(self, x):
        out = self._aq(_lrelu(self.conv1(x)))
        out = self.gdn(self.conv2(out))
        out = _add(out, x if self.skip is None else self.skip(x))
        return self._aq(out)


class QuantRBU(BaseQuantBlock):
    """ResidualBlockUpsample: subpel conv -> LReLU -> conv3x3 -> IGDN, plus a subpel-conv upsample branch."""
    unit_kind = "rbu"

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.subpel_conv = nn.Sequential(
            QuantModule(basic_block.subpel_conv[0], weight_quant_params, act_quant_params, disable_act_quant=True),
            basic_block.subpel_conv[1])
        self.leaky_relu = basic_block.leaky_relu
        self.conv = QuantModule(basic_block.conv, weight_quant_params, act_quant_params)
        self.igdn = QuantModule(basic_block.igdn, weight_quant_params, act_quant_params)
        self.upsample = nn.Sequential(QuantModule(basic_block.upsample[0], weight_quant_params, act_quant_params),
                                      basic_block.upsample[1])

    @staticmethod
    def _subpel(seq, x):
        y = seq[0](x)
        return _nchw_view(ops.pixel_shuffle(_nhwc(y), seq[1].upscale_factor))

    def forward(self, x):
        out = self._aq(_lrelu(self._subpel(self.subpel_conv, x)))
        out = self.igdn(self.conv(out))
        out = _add(out, self._subpel(self.upsample, x))
        return self._aq(out)


class QuantRB(BaseQuantBlock):
    """ResidualBlock: (conv3x3 -> LReLU) x2 plus identity / 1x1 skip."""
    unit_kind = "rb"

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.conv1 = QuantModule(basic_block.conv1, weight_quant_params, act_quant_params, disable_act_quant=True)
        self.leaky_relu = basic_block.leaky_relu
        self.conv2 = QuantModule(basic_block.conv2, weight_quant_params, act_quant_params, disable_act_quant=True)
        self.skip = None if basic_block.skip is None else QuantModule(basic_block.skip, weight_quant_params, act_quant_params)

    def forward(self, x):
        out = self._aq(_lrelu(self.conv1(x)))
        out = self._aq(_lrelu(self.conv2(out)))
        out = _add(out, x if self.skip is None else self.skip(x))
        return self._aq(out)


class QuantSC(BaseQuantBlock):
    """subpel_conv3x3 + LeakyReLU.  Kept for surface compatibility: the reference keys it on a function object
    (quant_block.py:656) so it never matches a module during surgery."""

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.subpel_conv = nn.Sequential(
            QuantModule(basic_block[0], weight_quant_params, act_quant_params, disable_act_quant=True), basic_block[1],
            nn.LeakyReLU(inplace=True))

    def forward(self, x):
        return self.subpel_conv(x)


specials = {lic.ResidualBlockWithStride: QuantRBWS, lic.ResidualBlockUpsample: QuantRBU, lic.ResidualBlock: QuantRB}
try:  # real CompressAI blocks, when that package is installed (main2.py:160 un-pickles such a model)
    from compressai.layers.layers import ResidualBlock as _CRB, ResidualBlockUpsample as _CRBU, \
        ResidualBlockWithStride as _CRBWS
    specials.update({_CRBWS: QuantRBWS, _CRBU: QuantRBU, _CRB: QuantRB})
except Exception:  # pragma: no cover
    pass
