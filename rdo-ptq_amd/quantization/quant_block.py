"""Quantised Cheng2020 blocks (reference surface: quantization/quant_block.py:77-102, 219-328, 645-657).

A block owns `QuantModule`s for its convs / GDN and applies the dynamic activation quantiser after the element-wise
joins once it is trained.  The Lu2022 Swin wrappers (QuantRSTB & co., quant_block.py:330-641) keep tokens in natural pixel
order end to end: the cyclic shift and the window partition are address arithmetic inside the attention kernel, and every
activation quantiser on the path is per-channel over the whole tensor, hence order-independent."""
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
        return ActQuantizer(x, self.act_quantizer.dynamic_bits) if (self.use_act_quant and self.trained) else x


def _tracked(*ts):
    return torch.is_grad_enabled() and any(t.requires_grad for t in ts)


def _lrelu(x):
    if _tracked(x):                       # on torch's tape (hipops.autograd): element-wise glue between the HIP Functions
        return torch.nn.functional.leaky_relu(x, 0.01)
    return _nchw_view(ops.lrelu(_nhwc(x)))


def _add(a, b):
    if _tracked(a, b):
        return a + b
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
        if _tracked(y):
            return torch.nn.functional.pixel_shuffle(y, seq[1].upscale_factor)
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


# ----------------------------------------------------------------------------- Lu2022: Swin transformer wrappers
def _GeluFn():
    from hipops.autograd import GeluFn
    return GeluFn



class PatchEmbed(nn.Module):
    def forward(self, x):
        return x.flatten(2).transpose(1, 2)


class PatchUnEmbed(nn.Module):
    def forward(self, x, x_size):
        return x.transpose(1, 2).reshape(x.shape[0], -1, x_size[0], x_size[1])


class QuantMlp(BaseQuantBlock):
    """fc1 -> GELU -> [AQ] -> fc2 (quant_block.py:330-348); fc1's own output quantiser is disabled."""

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.fc1 = QuantModule(basic_block.fc1, weight_quant_params, act_quant_params, disable_act_quant=True)
        self.act = basic_block.act
        self.fc2 = QuantModule(basic_block.fc2, weight_quant_params, act_quant_params)

    def forward(self, x):
        x = self.fc1(x)
        if isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none":
            x = _GeluFn().apply(x) if _tracked(x) else ops.gelu(x.contiguous())
        else:
            x = self.act(x)
        return self.fc2(self._aq(x))


class QuantWindowAttention(BaseQuantBlock):
    """(S)W-MSA with quantised qkv / proj linears; once trained, the probabilities and attn @ v are activation-quantised
    (quant_block.py:350-420)."""

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.dim, self.window_size, self.num_heads = basic_block.dim, tuple(basic_block.window_size), basic_block.num_heads
        self.scale = basic_block.scale
        self.qkv = QuantModule(basic_block.qkv, weight_quant_params, act_quant_params)
        self.attn_drop = basic_block.attn_drop
        self.proj = QuantModule(basic_block.proj, weight_quant_params, act_quant_params)
        self.proj_drop = basic_block.proj_drop
        self.softmax = basic_block.softmax
        self.relative_position_bias_table = basic_block.relative_position_bias_table
        self.register_buffer("relative_position_index", basic_block.relative_position_index)

    def position_bias(self):
        """[heads, N, N] fp32, contiguous: table rows gathered by the (query, key) offset index."""
        n = self.window_size[0] * self.window_size[1]
        t = self.relative_position_bias_table.detach()[self.relative_position_index.view(-1)]
        return t.view(n, n, -1).permute(2, 0, 1).contiguous()

    def attend(self, tokens, B, H, W, window, shift):
        """tokens [B, H*W, C] in natural order -> same shape: qkv linear, attention core, projection."""
        C = tokens.shape[-1]
        qkv = self.qkv(tokens).contiguous()
        d = ops.attn_desc(B, H, W, C, self.num_heads, window, shift, self.scale)
        bias = self.position_bias()
        if self.use_act_quant and self.trained:
            if _tracked(qkv):
                # the activation-quantised attention below runs raw kernels (no grad_fn): under torch's tape (loss_mode='rd' behind a unit,
                # a second calibration pass) the gradient through the attention path would be cut silently
                raise NotImplementedError("QuantWindowAttention: a trained, activation-quantised attention cannot sit on torch's tape "
                                          "(its quantised probabilities carry no gradient); run the R + lambda*D task loss with act_quant=False")
            n = window * window
            probs = torch.empty((B * (H // window) * (W // window), n, n, self.num_heads), device=qkv.device, dtype=torch.float32)
            ops.window_attention(d, qkv, bias, probs=probs, compute_out=False)
            nb = self.act_quantizer.dynamic_bits
            probs = ops.actquant_perchannel(probs, n_bits=nb)            # per head, as ActQuantizer on [B_, heads, N, N]
            o = ActQuantizer(ops.window_attention_pv(d, qkv, probs).view(B, H * W, C), nb)
        elif _tracked(qkv):
            from hipops.autograd import WindowAttentionFn
            o = WindowAttentionFn.apply(qkv.view(B, H, W, 3 * C), d, bias).view(B, H * W, C)
        else:
            o = ops.window_attention(d, qkv, bias).view(B, H * W, C)
        return self.proj(o)

    def forward(self, x, mask=None):
        """Reference signature: x [num_windows*B, N, C] already partitioned.  Without a mask every window is independent and
        is handled as its own ws x ws image; shifted windows go through QuantSwinTransformerBlock (geometry decides the mask)."""
        if mask is not None:
            raise NotImplementedError("masked window attention is driven by QuantSwinTransformerBlock.forward on this build")
        ws = self.window_size[0]
        return self.attend(x, x.shape[0], ws, ws, ws, 0)

    def extra_repr(self):
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"


class QuantSwinTransformerBlock(BaseQuantBlock):
    """norm1 -> (S)W-MSA -> + shortcut -> norm2 -> MLP -> + -> [AQ]   (quant_block.py:427-553)."""

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.dim, self.input_resolution = basic_block.dim, tuple(basic_block.input_resolution)
        self.num_heads, self.mlp_ratio = basic_block.num_heads, basic_block.mlp_ratio
        self.window_size, self.shift_size = basic_block.window_size, basic_block.shift_size
        if min(self.input_resolution) <= self.window_size:
            self.shift_size, self.window_size = 0, min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = QuantModule(basic_block.norm1, weight_quant_params, act_quant_params)
        self.attn = QuantWindowAttention(basic_block.attn, weight_quant_params, act_quant_params)
        self.norm2 = QuantModule(basic_block.norm2, weight_quant_params, act_quant_params)
        self.mlp = QuantMlp(basic_block.mlp, weight_quant_params, act_quant_params)
        self.attn_mask = basic_block.attn_mask

    def forward(self, x, x_size):
        H, W = int(x_size[0]), int(x_size[1])
        B, L, C = x.shape
        x = x.contiguous()
        a = self.attn.attend(self.norm1(x), B, H, W, self.window_size, self.shift_size)
        if _tracked(x, a):                             # torch's tape is driving (the R + lambda*D tail of the calibration loop): its own adds
            x = x + a
            x = x + self.mlp(self.norm2(x))
            return self._aq(x)
        x = ops.add(x, a.contiguous())
        x = ops.add(x, self.mlp(self.norm2(x)).contiguous())
        return self._aq(x)

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")


class QuantBasicLayer(BaseQuantBlock):
    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.dim, self.input_resolution = basic_block.dim, tuple(basic_block.input_resolution)
        self.depth, self.use_checkpoint = basic_block.depth, basic_block.use_checkpoint
        self.blocks = nn.ModuleList(QuantSwinTransformerBlock(b, weight_quant_params, act_quant_params) for b in basic_block.blocks)

    def forward(self, x, x_size):
        for blk in self.blocks:
            x = blk(x, x_size)
        return x

    def extra_repr(self):
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"


class QuantRSTB(BaseQuantBlock):
    """Residual Swin Transformer Block: tokens = pixels of the NCHW map, `depth` Swin blocks, + input, [AQ]
    (quant_block.py:603-637)."""
    unit_kind = "rstb"

    def __init__(self, basic_block, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.dim, self.input_resolution = basic_block.dim, tuple(basic_block.input_resolution)
        self.residual_group = QuantBasicLayer(basic_block.residual_group, weight_quant_params, act_quant_params)
        self.patch_embed = PatchEmbed()
        self.patch_unembed = PatchUnEmbed()

    def forward(self, x, x_size):
        H, W = int(x_size[0]), int(x_size[1])
        xn = _nhwc(x)                                            # [B, H, W, C]: the token matrix, no copy for channels_last
        B, _, _, C = xn.shape
        t = self.residual_group(xn.view(B, H * W, C), (H, W))
        if _tracked(t, xn):
            return self._aq(_nchw_view(t.reshape(B, H, W, C) + xn))
        out = _nchw_view(ops.add(t.contiguous().view(B, H, W, C), xn))
        return self._aq(out)


specials = {lic.ResidualBlockWithStride: QuantRBWS, lic.ResidualBlockUpsample: QuantRBU, lic.ResidualBlock: QuantRB,
            lic.RSTB: QuantRSTB}
try:  # the reference's own models package (models/layers.py), when it is importable next to this one
    from models.layers import RSTB as _RefRSTB
    specials[_RefRSTB] = QuantRSTB
except Exception:  # pragma: no cover
    pass
try:  # real CompressAI blocks, when that package is installed (main2.py:160 un-pickles such a model)
    from compressai.layers.layers import ResidualBlock as _CRB, ResidualBlockUpsample as _CRBU, \
        ResidualBlockWithStride as _CRBWS
    specials.update({_CRBWS: QuantRBWS, _CRBU: QuantRBU, _CRB: QuantRB})
except Exception:  # pragma: no cover
    pass
