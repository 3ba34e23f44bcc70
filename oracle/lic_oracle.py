"""ORACLE (test infrastructure, not product code).

CPU/torch-fp32 restatement of the CompressAI 1.2.4 pieces that the reference's
hot path touches.  CompressAI is a pinned third-party dependency
(`/root/reference/requirements.txt:1`, `compressai==1.2.4`) whose source is NOT
under /root/reference and is not installed in this image, so everything in this
file is restated from the published algorithm (Balle et al. GDN, Minnen et al.
2018 joint autoregressive prior, Cheng et al. 2020 anchor) and is

    **parity unpinned**  (no reference-side golden vectors exist for it).

Call sites in the reference that consume these classes:
  * GDN                       -> quantization/quant_layer.py:7,51-57,142-154
  * ResidualBlock*/subpel     -> quantization/quant_block.py:8,219-328
  * EntropyBottleneck / GaussianConditional -> quantization/quant_model.py:7,
    models/nic_cvt.py:221-222,297-308
  * MaskedConv2d              -> models/nic_cvt.py:223
  * the Cheng2020Anchor model object itself is un-pickled at main2.py:160.

Only `tests/`, `tools/make_golden.py`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- GDN
class _LowerBoundFn(torch.autograd.Function):
    """max(x, bound) whose gradient also passes when it pushes x upward."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (g < 0)
        return keep.type(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    """gamma' = max(gamma, sqrt(min + 2^-36))^2 - 2^-36."""

    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((self.minimum + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = self.lower_bound(x)
        return out ** 2 - self.pedestal


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        C = x.size(1)
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(C, C, 1, 1)
        norm = F.conv2d(x ** 2, gamma, beta)
        norm = torch.sqrt(norm) if self.inverse else torch.rsqrt(norm)
        return x * norm


# ----------------------------------------------------------------------------- conv helpers / blocks
def conv3x3(in_ch, out_ch, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def conv1x1(in_ch, out_ch, stride=1):
    return nn.Conv2d(in_ch, out_ch, kernel_size=1, stride=stride)


def subpel_conv3x3(in_ch, out_ch, r=1):
    return nn.Sequential(nn.Conv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), nn.PixelShuffle(r))


class ResidualBlockWithStride(nn.Module):
    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def forward(self, x):
        identity = x
        out = self.gdn(self.conv2(self.leaky_relu(self.conv1(x))))
        if self.skip is not None:
            identity = self.skip(x)
        out += identity
        return out


class ResidualBlockUpsample(nn.Module):
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def forward(self, x):
        out = self.igdn(self.conv(self.leaky_relu(self.subpel_conv(x))))
        out += self.upsample(x)
        return out


class ResidualBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def forward(self, x):
        identity = x
        out = self.leaky_relu(self.conv2(self.leaky_relu(self.conv1(x))))
        if self.skip is not None:
            identity = self.skip(x)
        return out + identity


class MaskedConv2d(nn.Conv2d):
    """PixelCNN-style causal conv (mask type 'A'): taps at/after the centre are zeroed."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.size()
        self.mask[:, :, h // 2, w // 2 + (mask_type == "B"):] = 0
        self.mask[:, :, h // 2 + 1:] = 0

    def forward(self, x):
        self.weight.data *= self.mask
        return super().forward(x)


# ----------------------------------------------------------------------------- entropy models (forward likelihood only)
# Straight-through rounding for the differentiable EVALUATION forward used by the build's opt-in R + lambda*D task loss (the
# reference sketches that loss and comments it out, layer_opt.py:146-148; CompressAI itself trains with additive noise).  Values
# are unchanged; only the gradient of round() becomes the identity.  Off by default.
STE_ROUND = False


def _round(x):
    if STE_ROUND and torch.is_grad_enabled() and x.requires_grad:
        return x + (torch.round(x) - x).detach()
    return torch.round(x)


class EntropyBottleneck(nn.Module):
    """Factorised prior of Balle et al. 2018 (filters (3,3,3,3), init_scale 10)."""

    def __init__(self, channels, init_scale=10.0, filters=(3, 3, 3, 3), likelihood_bound=1e-9):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.likelihood_bound = float(likelihood_bound)
        f = (1,) + self.filters + (1,)
        scale = float(init_scale) ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = math.log(math.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), init)))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        q = torch.Tensor([-float(init_scale), 0.0, float(init_scale)])
        self.quantiles = nn.Parameter(q.repeat(channels, 1, 1))

    def _logits_cumulative(self, v):
        logits = v
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(getattr(self, f"_matrix{i}")), logits) + getattr(self, f"_bias{i}")
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, f"_factor{i}")) * torch.tanh(logits)
        return logits

    def forward(self, x):
        perm = [1, 0] + list(range(2, x.dim()))
        xt = x.permute(*perm).contiguous()
        shape = xt.size()
        v = xt.reshape(shape[0], 1, -1)
        med = self.quantiles[:, :, 1:2]
        if self.training:
            out = v + torch.empty_like(v).uniform_(-0.5, 0.5)
        else:
            out = _round(v - med) + med
        lower = self._logits_cumulative(out - 0.5)
        upper = self._logits_cumulative(out + 0.5)
        sign = -torch.sign(lower + upper).detach()
        lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        lik = torch.clamp(lik, min=self.likelihood_bound)
        return out.reshape(shape).permute(*perm).contiguous(), lik.reshape(shape).permute(*perm).contiguous()


class GaussianConditional(nn.Module):
    def __init__(self, scale_table=None, scale_bound=0.11, likelihood_bound=1e-9):
        super().__init__()
        self.scale_bound = float(scale_bound)
        self.likelihood_bound = float(likelihood_bound)

    def quantize(self, inputs, mode, means=None):
        if mode == "noise":
            return inputs + torch.empty_like(inputs).uniform_(-0.5, 0.5)
        out = inputs if means is None else inputs - means
        out = _round(out)
        return out if means is None else out + means

    @staticmethod
    def _std_cum(x):
        return 0.5 * torch.erfc(-(2 ** -0.5) * x)

    def forward(self, inputs, scales, means=None):
        out = self.quantize(inputs, "noise" if self.training else "dequantize", means)
        v = out if means is None else out - means
        s = torch.clamp(scales, min=self.scale_bound)
        v = torch.abs(v)
        lik = self._std_cum((0.5 - v) / s) - self._std_cum((-0.5 - v) / s)
        return out, torch.clamp(lik, min=self.likelihood_bound)


# ----------------------------------------------------------------------------- Cheng2020 anchor
class Cheng2020Anchor(nn.Module):
    """Cheng et al. 2020 'anchor' model (no attention), CompressAI topology.

    Child registration order follows the CompressAI class chain
    CompressionModel -> ScaleHyperprior -> MeanScaleHyperprior -> JointAutoregressiveHierarchicalPriors
    -> Cheng2020Anchor: entropy_bottleneck, g_a, g_s, h_a, h_s, gaussian_conditional,
    entropy_parameters, context_prediction  **[3P-unverified]**.
    """

    def __init__(self, N=192):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = nn.Sequential(
            ResidualBlockWithStride(3, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            conv3x3(N, N, stride=2))
        self.g_s = nn.Sequential(
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), subpel_conv3x3(N, 3, 2))
        self.h_a = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2))
        self.h_s = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), subpel_conv3x3(N, N, 2), nn.LeakyReLU(inplace=True),
            conv3x3(N, N * 3 // 2), nn.LeakyReLU(inplace=True), subpel_conv3x3(N * 3 // 2, N * 3 // 2, 2),
            nn.LeakyReLU(inplace=True), conv3x3(N * 3 // 2, N * 2))
        self.gaussian_conditional = GaussianConditional(None)
        M = N
        self.entropy_parameters = nn.Sequential(
            nn.Conv2d(M * 12 // 3, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.N, self.M = N, M

    def forward(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_lik = self.entropy_bottleneck(z)
        params = self.h_s(z_hat)
        y_hat = self.gaussian_conditional.quantize(y, "noise" if self.training else "dequantize")
        ctx = self.context_prediction(y_hat)
        gp = self.entropy_parameters(torch.cat((params, ctx), dim=1))
        scales_hat, means_hat = gp.chunk(2, 1)
        _, y_lik = self.gaussian_conditional(y, scales_hat, means=means_hat)
        x_hat = self.g_s(y_hat)
        return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}


class AttentionBlock(nn.Module):
    """Simplified (non-local-free) attention of Cheng et al. 2020, CompressAI layout **[3P-unverified]**:
    out = x + conv_a(x) * sigmoid(conv_b(x)); conv_a = 3 residual units, conv_b = 3 residual units + conv1x1;
    residual unit = conv1x1(N, N/2) - ReLU - conv3x3(N/2, N/2) - ReLU - conv1x1(N/2, N), identity add, ReLU."""

    class ResidualUnit(nn.Module):
        def __init__(self, N):
            super().__init__()
            self.conv = nn.Sequential(conv1x1(N, N // 2), nn.ReLU(inplace=True), conv3x3(N // 2, N // 2),
                                      nn.ReLU(inplace=True), conv1x1(N // 2, N))
            self.relu = nn.ReLU(inplace=True)

        def forward(self, x):
            out = self.conv(x)
            out = out + x
            return self.relu(out)

    def __init__(self, N):
        super().__init__()
        RU = AttentionBlock.ResidualUnit
        self.conv_a = nn.Sequential(RU(N), RU(N), RU(N))
        self.conv_b = nn.Sequential(RU(N), RU(N), RU(N), conv1x1(N, N))

    def forward(self, x):
        a = self.conv_a(x)
        b = self.conv_b(x)
        return a * torch.sigmoid(b) + x


class Cheng2020Attention(Cheng2020Anchor):
    """Cheng2020Anchor with attention blocks after the 2nd and the last down-sampling stage (and mirrored in g_s),
    CompressAI `cheng2020_attn` topology **[3P-unverified]**."""

    def __init__(self, N=192):
        super().__init__(N)
        self.g_a = nn.Sequential(
            ResidualBlockWithStride(3, N, stride=2), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), AttentionBlock(N), ResidualBlock(N, N),
            ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
            conv3x3(N, N, stride=2), AttentionBlock(N))
        self.g_s = nn.Sequential(
            AttentionBlock(N), ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2), AttentionBlock(N),
            ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2),
            ResidualBlock(N, N), subpel_conv3x3(N, 3, 2))


# ----------------------------------------------------------------------------- Minnen2018 mean-scale hyperprior
def conv(in_ch, out_ch, kernel_size=5, stride=2):
    return nn.Conv2d(in_ch, out_ch, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_ch, out_ch, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(in_ch, out_ch, kernel_size=kernel_size, stride=stride, output_padding=stride - 1,
                              padding=kernel_size // 2)


class MeanScaleHyperprior(nn.Module):
    """Minnen et al. 2018 without the autoregressive context ('mbt2018-mean'), CompressAI topology and child order
    (entropy_bottleneck, g_a, g_s, h_a, h_s, gaussian_conditional)  **[3P-unverified]**."""

    def __init__(self, N=128, M=192):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = nn.Sequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = nn.Sequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True), deconv(N, N),
                                 GDN(N, inverse=True), deconv(N, 3))
        self.h_a = nn.Sequential(conv(M, N, stride=1, kernel_size=3), nn.LeakyReLU(inplace=True), conv(N, N),
                                 nn.LeakyReLU(inplace=True), conv(N, N))
        self.h_s = nn.Sequential(deconv(N, M), nn.LeakyReLU(inplace=True), deconv(M, M * 3 // 2), nn.LeakyReLU(inplace=True),
                                 conv(M * 3 // 2, M * 2, stride=1, kernel_size=3))
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = N, M

    def forward(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_lik = self.entropy_bottleneck(z)
        scales_hat, means_hat = self.h_s(z_hat).chunk(2, 1)
        y_hat, y_lik = self.gaussian_conditional(y, scales_hat, means=means_hat)
        return {"x_hat": self.g_s(y_hat), "likelihoods": {"y": y_lik, "z": z_lik}}


class JointAutoregressiveHierarchicalPriors(MeanScaleHyperprior):
    """Minnen, Balle, Toderici 2018 with the autoregressive context model ('mbt2018', BASELINE config 5), CompressAI topology and
    child order (..., gaussian_conditional, entropy_parameters, context_prediction)  **[3P-unverified]**: the mean-scale model plus a
    5x5 type-A masked convolution over the (de)quantised latent and a three-layer 1x1 network that maps [hyper | context] features to
    (scales, means).  Training / evaluation forward (parallel masked conv on y_hat), not the sequential decoder."""

    def __init__(self, N=192, M=192):
        super().__init__(N=N, M=M)
        self.entropy_parameters = nn.Sequential(nn.Conv2d(M * 12 // 3, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
                                                nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
                                                nn.Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)

    def forward(self, x):
        y = self.g_a(x)
        z_hat, z_lik = self.entropy_bottleneck(self.h_a(y))
        hyper = self.h_s(z_hat)
        y_hat = self.gaussian_conditional.quantize(y, "noise" if self.training else "dequantize")
        ctx = self.context_prediction(y_hat)
        scales_hat, means_hat = self.entropy_parameters(torch.cat((hyper, ctx), dim=1)).chunk(2, 1)
        _, y_lik = self.gaussian_conditional(y, scales_hat, means=means_hat)
        return {"x_hat": self.g_s(y_hat), "likelihoods": {"y": y_lik, "z": z_lik}}


def mbt2018_forward_w8a8(model: JointAutoregressiveHierarchicalPriors, x, act_quant=True):
    """The forward the reference's QuantModel produces for this model with nearest-rounded W8 weights (channel-wise 'max' scales)
    and, with `act_quant`, its dynamic 8-bit activation quantiser behind every wrapped module (quant_layer.py:107-134): conv /
    transposed conv / GDN, the following LeakyReLU fused in FRONT of the activation quantiser (quant_model.py:51-54), none on the
    last decoder layer (main2.py:258-263) nor on the last module of the child order, context_prediction
    (disable_network_output_quantization, quant_model.py:66-70), whose mask the wrapper bypasses."""
    from .rdo_oracle import act_quant as aq, uaq_fakequant, uaq_init

    def qw(w, tconv=False):
        d, z = uaq_init(w, 8, True, "max", tconv=tconv)
        return uaq_fakequant(w, d, z, 256)

    def run(seq, h, last_plain=False):
        mods = list(seq)
        for i, m in enumerate(mods):
            if isinstance(m, nn.LeakyReLU):
                continue                                   # applied with the module in front of it
            if isinstance(m, nn.ConvTranspose2d):
                h = F.conv_transpose2d(h, qw(m.weight.data, True), m.bias, m.stride, m.padding, m.output_padding)
            elif isinstance(m, nn.Conv2d):
                h = F.conv2d(h, qw(m.weight.data), m.bias, m.stride, m.padding)
            elif isinstance(m, GDN):
                c = h.shape[1]
                gamma = m.gamma_reparam(qw(m.gamma.data)).view(c, c, 1, 1)
                pool = F.conv2d(h * h, gamma, m.beta_reparam(m.beta))
                h = h * (pool.sqrt() if m.inverse else pool.rsqrt())
            else:
                raise TypeError(type(m))
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                h = F.leaky_relu(h, 0.01)
            if act_quant and not (last_plain and i == len(mods) - 1):
                h = aq(h)
        return h
    with torch.no_grad():
        y = run(model.g_a, x)
        z_hat, z_lik = model.entropy_bottleneck(run(model.h_a, y))
        hyper = run(model.h_s, z_hat)
        y_hat = model.gaussian_conditional.quantize(y, "dequantize")
        cp = model.context_prediction
        ctx = F.conv2d(y_hat, qw(cp.weight.data), cp.bias, cp.stride, cp.padding)     # unmasked, no activation quantiser
        scales_hat, means_hat = run(model.entropy_parameters, torch.cat((hyper, ctx), dim=1)).chunk(2, 1)
        _, y_lik = model.gaussian_conditional(y, scales_hat, means=means_hat)
        return {"x_hat": run(model.g_s, y_hat, last_plain=True), "likelihoods": {"y": y_lik, "z": z_lik}}
