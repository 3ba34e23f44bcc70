"""Rate-distortion loss and metrics (reference surface: losses/losses.py:8-84) on the HIP reduction kernels.

    bpp = sum_over_likelihood_tensors( -log2(p) ) / (N*H*W),   loss = lambda * 255^2 * MSE + bpp      (metric 'mse')

MS-SSIM (pytorch_msssim in the reference) is not built yet: the 'ms-ssim' metric raises, and `ms_ssim_loss` is reported
as NaN for the 'mse' metric (the reference computes it there for logging only, losses.py:24-28)."""
import math

import torch
import torch.nn as nn

from hipops import ops


def _flat(t):
    return t.detach().contiguous().reshape(-1)


def bpp_of(likelihoods: dict, num_pixels: int) -> torch.Tensor:
    out = None
    for lik in likelihoods.values():
        out = ops.neg_log2_sum(_flat(lik), 1.0 / num_pixels, out)
    return out.reshape(())


def mse_of(a, b, clamp01=False) -> torch.Tensor:
    # both tensors must share one memory layout for the element-wise kernel: use the logical NCHW order
    a, b = a.detach().contiguous(), b.detach().contiguous()
    return ops.sq_diff_sum(a.reshape(-1), b.reshape(-1), 1.0 / a.numel(), clamp01).reshape(())


class RateDistortionLoss(nn.Module):
    def __init__(self, lmbda=1e-2, metric="mse"):
        super().__init__()
        self.lmbda, self.metric = lmbda, metric

    def forward(self, output, target):
        N, _, H, W = target.size()
        out = {"bpp_loss": bpp_of(output["likelihoods"], N * H * W), "mse_loss": mse_of(output["x_hat"], target)}
        if self.metric == "mse":
            out["ms_ssim_loss"] = torch.tensor(float("nan"), device=target.device)
            out["loss"] = self.lmbda * 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        elif self.metric == "ms-ssim":
            raise NotImplementedError("MS-SSIM is not built yet (SURVEY 8f row 2)")
        else:
            raise ValueError(self.metric)
        return out


class Metrics(nn.Module):
    def MSE(self, x, y):
        return torch.stack([mse_of(x[i:i + 1], y[i:i + 1]) for i in range(x.shape[0])])

    def PSNR(self, x, y):
        return torch.mean(10 * torch.log10(1.0 / self.MSE(x, y)))

    def forward(self, output, target):
        N, _, H, W = target.size()
        bpp = bpp_of(output["likelihoods"], N * H * W)
        return bpp, self.PSNR(output["x_hat"], target), torch.tensor(float("nan"), device=target.device)


def compute_psnr(a, b):
    """test_datasets.py:21-23"""
    return -10 * math.log10(float(mse_of(a, b)))


def compute_bpp(out_net):
    """test_datasets.py:29-33 (num_pixels of the padded reconstruction)"""
    size = out_net["x_hat"].size()
    return float(bpp_of(out_net["likelihoods"], size[0] * size[2] * size[3]))
