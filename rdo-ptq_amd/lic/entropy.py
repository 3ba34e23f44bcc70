"""Entropy models -- forward likelihoods only (no range coder): factorised prior and Gaussian conditional.
[3P-unverified restatement of compressai.entropy_models; eval/cache-building path, not the calibration hot loop.]"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from hipops import ops


class EntropyBottleneck(nn.Module):
    def __init__(self, channels, init_scale=10.0, filters=(3, 3, 3, 3), tail_mass=1e-9, likelihood_bound=1e-9):
        super().__init__()
        self.channels, self.filters = int(channels), tuple(filters)
        self.likelihood_bound = float(likelihood_bound)
        dims = (1,) + self.filters + (1,)
        scale = init_scale ** (1.0 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            fill = math.log(math.expm1(1.0 / scale / dims[i + 1]))
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.full((channels, dims[i + 1], dims[i]), fill)))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.empty(channels, dims[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(channels, dims[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.tensor([-init_scale, 0.0, init_scale]).repeat(channels, 1, 1))

    def _cdf_logits(self, v):
        for i in range(len(self.filters) + 1):
            v = torch.matmul(F.softplus(getattr(self, f"_matrix{i}")), v) + getattr(self, f"_bias{i}")
            if i < len(self.filters):
                v = v + torch.tanh(getattr(self, f"_factor{i}")) * torch.tanh(v)
        return v

    def kernel_params(self):
        """[C, 58] = per channel [33 softplus(matrix) | 13 bias | 12 tanh(factor)] for rdo_factorized_likelihood_fwd."""
        n = len(self.filters) + 1
        c = self.channels
        mats = [F.softplus(getattr(self, f"_matrix{i}")).reshape(c, -1) for i in range(n)]
        bias = [getattr(self, f"_bias{i}").reshape(c, -1) for i in range(n)]
        fac = [torch.tanh(getattr(self, f"_factor{i}")).reshape(c, -1) for i in range(n - 1)]
        return torch.cat(mats + bias + fac, dim=1).contiguous()

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad and not self.training and x.is_cuda and self.filters == (3, 3, 3, 3) and x.dim() == 4:
            # differentiable evaluation path (R + lambda*D task loss) on the HIP kernels: rounding with a straight-through gradient,
            # likelihood forward / backward element-wise (rdo_factorized_likelihood_fwd / _bwd)
            from hipops.autograd import FactorizedLikelihoodFn
            return FactorizedLikelihoodFn.apply(x, self.kernel_params().detach(), self.quantiles[:, 0, 1].detach().contiguous())
        if torch.is_grad_enabled() and x.requires_grad and not self.training:
            # the same in torch (other filter shapes, CPU tensors)
            order = [1, 0] + list(range(2, x.dim()))
            xt = x.permute(*order).contiguous()
            flat = xt.reshape(xt.shape[0], 1, -1)
            med = self.quantiles[:, :, 1:2].detach()
            d = flat - med
            q = d + (torch.round(d) - d).detach() + med
            lo, hi = self._cdf_logits(q - 0.5), self._cdf_logits(q + 0.5)
            sgn = -torch.sign(lo + hi).detach()
            lik = (torch.sigmoid(sgn * hi) - torch.sigmoid(sgn * lo)).abs().clamp_min(self.likelihood_bound)
            back = lambda t: t.reshape(xt.shape).permute(*order).contiguous()
            return back(q), back(lik)
        if x.is_cuda and not self.training and self.filters == (3, 3, 3, 3) and x.dim() == 4:
            # eval path on the HIP kernel (NHWC element-wise); training-time noise stays in torch (not on the PTQ path)
            with torch.no_grad():
                xc = x.permute(0, 2, 3, 1).contiguous()
                zhat, lik = ops.factorized_likelihood(xc, self.kernel_params().detach(),
                                                      self.quantiles[:, 0, 1].detach().contiguous())
            return zhat.permute(0, 3, 1, 2), lik.permute(0, 3, 1, 2)
        order = [1, 0] + list(range(2, x.dim()))
        xt = x.permute(*order).contiguous()
        flat = xt.reshape(xt.shape[0], 1, -1)
        med = self.quantiles[:, :, 1:2]
        q = flat + torch.empty_like(flat).uniform_(-0.5, 0.5) if self.training else torch.round(flat - med) + med
        lo, hi = self._cdf_logits(q - 0.5), self._cdf_logits(q + 0.5)
        sgn = -torch.sign(lo + hi).detach()
        lik = (torch.sigmoid(sgn * hi) - torch.sigmoid(sgn * lo)).abs().clamp_min(self.likelihood_bound)
        back = lambda t: t.reshape(xt.shape).permute(*order).contiguous()
        return back(q), back(lik)


class GaussianConditional(nn.Module):
    def __init__(self, scale_table=None, scale_bound=0.11, likelihood_bound=1e-9):
        super().__init__()
        self.scale_bound, self.likelihood_bound = float(scale_bound), float(likelihood_bound)

    def quantize(self, inputs, mode, means=None):
        if mode == "noise":
            return inputs + torch.empty_like(inputs).uniform_(-0.5, 0.5)
        if torch.is_grad_enabled() and inputs.requires_grad:      # straight-through rounding on the differentiable evaluation path
            from hipops.autograd import round_ste
            return round_ste(inputs) if means is None else round_ste(inputs - means) + means
        if means is None:
            return torch.round(inputs)
        return torch.round(inputs - means) + means

    def forward(self, inputs, scales, means=None):
        tracked = torch.is_grad_enabled() and (inputs.requires_grad or scales.requires_grad or (means is not None and means.requires_grad))
        if tracked and inputs.is_cuda and not self.training and inputs.dim() == 4:
            # differentiable evaluation path on the HIP likelihood kernels (forward + rdo_gaussian_likelihood_bwd), rounding with a
            # straight-through gradient
            from hipops.autograd import GaussianLikelihoodFn, round_ste
            m = torch.zeros_like(inputs) if means is None else means
            yhat = round_ste(inputs - m) + m
            return yhat, GaussianLikelihoodFn.apply(yhat, scales, m, self.scale_bound)
        if inputs.is_cuda and not self.training and inputs.dim() == 4 and not tracked:
            cl = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous()
            yhat, lik = ops.gaussian_likelihood(cl(inputs), cl(scales), cl(means), self.scale_bound)
            return yhat.permute(0, 3, 1, 2), lik.permute(0, 3, 1, 2)
        y = self.quantize(inputs, "noise" if self.training else "dequantize", means)
        v = (y if means is None else y - means).abs()
        s = scales.clamp_min(self.scale_bound)
        phi = lambda t: 0.5 * torch.erfc(-t * (2 ** -0.5))
        lik = (phi((0.5 - v) / s) - phi((-0.5 - v) / s)).clamp_min(self.likelihood_bound)
        return y, lik
