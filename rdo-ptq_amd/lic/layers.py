"""GDN and the Cheng2020 residual blocks (CompressAI-compatible attribute names, so `quantization` can wrap them and
pickled CompressAI state_dicts load).  [3P-unverified: restated from the published definitions.]"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_OFFSET = 2.0 ** -18


class _LowerBound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).to(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.tensor([float(bound)]))

    def forward(self, x):
        return _LowerBound.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    """param' = max(param, sqrt(minimum + 2^-36))^2 - 2^-36"""

    def __init__(self, minimum=0.0, reparam_offset=_OFFSET):
        super().__init__()
        self.minimum, self.reparam_offset = float(minimum), float(reparam_offset)
        self.register_buffer("pedestal", torch.tensor([self.reparam_offset ** 2]))
        self.lower_bound = LowerBound((self.minimum + self.reparam_offset ** 2) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return self.lower_bound(x) ** 2 - self.pedestal

    def constants(self):
        """(bound, pedestal) as Python floats -- what the HIP kernels take."""
        return float(self.lower_bound.bound), float(self.pedestal)


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        self.gamma_reparam = NonNegativeParametrizer()
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma_init * torch.eye(in_channels)))

    def forward(self, x):
        c = x.shape[1]
        pool = F.conv2d(x * x, self.gamma_reparam(self.gamma).view(c, c, 1, 1), self.beta_reparam(self.beta))
        return x * (pool.sqrt() if self.inverse else pool.rsqrt())


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1)


def conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride)


def subpel_conv3x3(cin, cout, r=1):
    return nn.Sequential(nn.Conv2d(cin, cout * r * r, 3, padding=1), nn.PixelShuffle(r))


class ResidualBlockWithStride(nn.Module):
    def __init__(self, cin, cout, stride=2):
        super().__init__()
        self.conv1 = conv3x3(cin, cout, stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(cout, cout)
        self.gdn = GDN(cout)
        self.skip = conv1x1(cin, cout, stride) if (stride != 1 or cin != cout) else None

    def forward(self, x):
        y = self.gdn(self.conv2(self.leaky_relu(self.conv1(x))))
        return y + (x if self.skip is None else self.skip(x))


class ResidualBlockUpsample(nn.Module):
    def __init__(self, cin, cout, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(cin, cout, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(cout, cout)
        self.igdn = GDN(cout, inverse=True)
        self.upsample = subpel_conv3x3(cin, cout, upsample)

    def forward(self, x):
        return self.igdn(self.conv(self.leaky_relu(self.subpel_conv(x)))) + self.upsample(x)


class ResidualBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = conv3x3(cin, cout)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(cout, cout)
        self.skip = conv1x1(cin, cout) if cin != cout else None

    def forward(self, x):
        y = self.leaky_relu(self.conv2(self.leaky_relu(self.conv1(x))))
        return y + (x if self.skip is None else self.skip(x))


class _ResidualUnit(nn.Module):
    """1x1 (N -> N/2) - ReLU - 3x3 - ReLU - 1x1 (N/2 -> N), identity add, ReLU: the unit the attention block stacks."""

    def __init__(self, N):
        super().__init__()
        self.conv = nn.Sequential(conv1x1(N, N // 2), nn.ReLU(inplace=True), conv3x3(N // 2, N // 2), nn.ReLU(inplace=True),
                                  conv1x1(N // 2, N))
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.conv(x) + x)


class AttentionBlock(nn.Module):
    """x + trunk(x) * sigmoid(mask(x)) of Cheng2020-attn (CompressAI attribute names: conv_a = trunk, conv_b = mask)."""

    ResidualUnit = _ResidualUnit

    def __init__(self, N):
        super().__init__()
        self.conv_a = nn.Sequential(*[_ResidualUnit(N) for _ in range(3)])
        self.conv_b = nn.Sequential(*[_ResidualUnit(N) for _ in range(3)], conv1x1(N, N))

    def forward(self, x):
        return x + self.conv_a(x) * torch.sigmoid(self.conv_b(x))


class MaskedConv2d(nn.Conv2d):
    """Causal (type-A) masked convolution of the autoregressive context model."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        mask = torch.ones_like(self.weight.data)
        kh, kw = mask.shape[2:]
        mask[:, :, kh // 2, kw // 2 + (mask_type == "B"):] = 0
        mask[:, :, kh // 2 + 1:] = 0
        self.register_buffer("mask", mask)

    def forward(self, x):
        self.weight.data *= self.mask
        return super().forward(x)
