"""Minnen2018 mean-scale hyperprior ('mbt2018-mean'): 5x5 stride-2 convs / transposed convs with GDN, CompressAI module names
and child order (entropy_bottleneck, g_a, g_s, h_a, h_s, gaussian_conditional).  [3P-unverified topology.]"""
import torch
import torch.nn as nn

from .entropy import EntropyBottleneck, GaussianConditional
from .layers import GDN, MaskedConv2d


def conv(cin, cout, kernel_size=5, stride=2):
    return nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(cin, cout, kernel_size=5, stride=2):
    return nn.ConvTranspose2d(cin, cout, kernel_size, stride=stride, output_padding=stride - 1, padding=kernel_size // 2)


def _lrelu():
    return nn.LeakyReLU(inplace=True)


class MeanScaleHyperprior(nn.Module):
    def __init__(self, N=128, M=192):
        super().__init__()
        self.N, self.M = N, M
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = nn.Sequential(conv(3, N), GDN(N), conv(N, N), GDN(N), conv(N, N), GDN(N), conv(N, M))
        self.g_s = nn.Sequential(deconv(M, N), GDN(N, inverse=True), deconv(N, N), GDN(N, inverse=True), deconv(N, N),
                                 GDN(N, inverse=True), deconv(N, 3))
        self.h_a = nn.Sequential(conv(M, N, kernel_size=3, stride=1), _lrelu(), conv(N, N), _lrelu(), conv(N, N))
        self.h_s = nn.Sequential(deconv(N, M), _lrelu(), deconv(M, M * 3 // 2), _lrelu(),
                                 conv(M * 3 // 2, M * 2, kernel_size=3, stride=1))
        self.gaussian_conditional = GaussianConditional(None)

    def forward(self, x):
        y = self.g_a(x)
        z_hat, z_lik = self.entropy_bottleneck(self.h_a(y))
        scales, means = self.h_s(z_hat).chunk(2, 1)
        y_hat, y_lik = self.gaussian_conditional(y, scales, means=means)
        return {"x_hat": self.g_s(y_hat), "likelihoods": {"y": y_lik, "z": z_lik}}


class JointAutoregressiveHierarchicalPriors(MeanScaleHyperprior):
    """'mbt2018': the mean-scale model plus the 5x5 masked-conv context model and the 1x1 entropy-parameter network
    (BASELINE config 5).  Child order: ..., gaussian_conditional, entropy_parameters, context_prediction.  [3P-unverified.]"""

    def __init__(self, N=192, M=192):
        super().__init__(N=N, M=M)
        self.h_s = nn.Sequential(deconv(N, M), _lrelu(), deconv(M, M * 3 // 2), _lrelu(),
                                 conv(M * 3 // 2, M * 2, kernel_size=3, stride=1))
        self.entropy_parameters = nn.Sequential(nn.Conv2d(M * 12 // 3, M * 10 // 3, 1), _lrelu(),
                                                nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), _lrelu(),
                                                nn.Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)

    def forward(self, x):
        y = self.g_a(x)
        z_hat, z_lik = self.entropy_bottleneck(self.h_a(y))
        hyper = self.h_s(z_hat)
        y_hat = self.gaussian_conditional.quantize(y, "noise" if self.training else "dequantize")
        ctx = self.context_prediction(y_hat)
        scales, means = self.entropy_parameters(torch.cat((hyper, ctx), dim=1)).chunk(2, 1)
        _, y_lik = self.gaussian_conditional(y, scales, means=means)
        return {"x_hat": self.g_s(y_hat), "likelihoods": {"y": y_lik, "z": z_lik}}
