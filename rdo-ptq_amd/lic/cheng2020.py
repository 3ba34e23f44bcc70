"""Cheng2020 'anchor' (no attention) with CompressAI's module names and child registration order
(entropy_bottleneck, g_a, g_s, h_a, h_s, gaussian_conditional, entropy_parameters, context_prediction) -- the order
decides the unit schedule of main2.py:227-253 and which module `disable_network_output_quantization` hits.
[3P-unverified topology.]"""
import torch
import torch.nn as nn

from .entropy import EntropyBottleneck, GaussianConditional
from .layers import (AttentionBlock, MaskedConv2d, ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride, conv3x3,
                     subpel_conv3x3)


def _lrelu():
    return nn.LeakyReLU(inplace=True)


class Cheng2020Anchor(nn.Module):
    def __init__(self, N=192):
        super().__init__()
        self.N = self.M = N
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = nn.Sequential(ResidualBlockWithStride(3, N), ResidualBlock(N, N), ResidualBlockWithStride(N, N),
                                 ResidualBlock(N, N), ResidualBlockWithStride(N, N), ResidualBlock(N, N),
                                 conv3x3(N, N, stride=2))
        self.g_s = nn.Sequential(ResidualBlock(N, N), ResidualBlockUpsample(N, N), ResidualBlock(N, N),
                                 ResidualBlockUpsample(N, N), ResidualBlock(N, N), ResidualBlockUpsample(N, N),
                                 ResidualBlock(N, N), subpel_conv3x3(N, 3, 2))
        self.h_a = nn.Sequential(conv3x3(N, N), _lrelu(), conv3x3(N, N), _lrelu(), conv3x3(N, N, stride=2), _lrelu(),
                                 conv3x3(N, N), _lrelu(), conv3x3(N, N, stride=2))
        self.h_s = nn.Sequential(conv3x3(N, N), _lrelu(), subpel_conv3x3(N, N, 2), _lrelu(), conv3x3(N, N * 3 // 2),
                                 _lrelu(), subpel_conv3x3(N * 3 // 2, N * 3 // 2, 2), _lrelu(),
                                 conv3x3(N * 3 // 2, N * 2))
        self.gaussian_conditional = GaussianConditional(None)
        self.entropy_parameters = nn.Sequential(nn.Conv2d(N * 4, N * 10 // 3, 1), _lrelu(),
                                                nn.Conv2d(N * 10 // 3, N * 8 // 3, 1), _lrelu(),
                                                nn.Conv2d(N * 8 // 3, N * 2, 1))
        self.context_prediction = MaskedConv2d(N, 2 * N, kernel_size=5, padding=2, stride=1)

    def forward(self, x):
        y = self.g_a(x)
        z = self.h_a(y)
        z_hat, z_lik = self.entropy_bottleneck(z)
        hyper = self.h_s(z_hat)
        y_hat = self.gaussian_conditional.quantize(y, "noise" if self.training else "dequantize")
        ctx = self.context_prediction(y_hat)
        scales, means = self.entropy_parameters(torch.cat((hyper, ctx), dim=1)).chunk(2, 1)
        _, y_lik = self.gaussian_conditional(y, scales, means=means)
        return {"x_hat": self.g_s(y_hat), "likelihoods": {"y": y_lik, "z": z_lik}}


class Cheng2020Attention(Cheng2020Anchor):
    """`cheng2020_attn`: the anchor with attention blocks at 1/4 and 1/16 resolution in g_a, mirrored in g_s."""

    def __init__(self, N=192):
        super().__init__(N)
        self.g_a = nn.Sequential(ResidualBlockWithStride(3, N), ResidualBlock(N, N), ResidualBlockWithStride(N, N),
                                 AttentionBlock(N), ResidualBlock(N, N), ResidualBlockWithStride(N, N), ResidualBlock(N, N),
                                 conv3x3(N, N, stride=2), AttentionBlock(N))
        self.g_s = nn.Sequential(AttentionBlock(N), ResidualBlock(N, N), ResidualBlockUpsample(N, N), ResidualBlock(N, N),
                                 ResidualBlockUpsample(N, N), AttentionBlock(N), ResidualBlock(N, N),
                                 ResidualBlockUpsample(N, N), ResidualBlock(N, N), subpel_conv3x3(N, 3, 2))
