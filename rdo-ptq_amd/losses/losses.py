"""Rate-distortion loss and metrics (reference surface: losses/losses.py:8-84) on the HIP reduction kernels.

    bpp = sum_over_likelihood_tensors( -log2(p) ) / (N*H*W),   loss = lambda * 255^2 * MSE + bpp      (metric 'mse')

MS-SSIM (`pytorch_msssim.ms_ssim` in the reference, requirements.txt:6) runs on `rdo_ssim_level` / `rdo_avg_pool2`: five scales,
11-tap Gaussian window (sigma 1.5), weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333) -- restated from the package's published
algorithm [3P-unverified]."""
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


_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
_MS_WINDOW = [math.exp(-((k - 5) ** 2) / (2 * 1.5 ** 2)) for k in range(11)]
_MS_WINDOW = [v / sum(_MS_WINDOW) for v in _MS_WINDOW]


def ms_ssim(x, y, data_range=1.0, size_average=True):
    """MS-SSIM of NCHW image batches in [0, data_range]; sides must exceed 160 pixels (five scales of an 11-tap window)."""
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("ms_ssim expects two NCHW tensors of the same shape")
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("image side must exceed 160 for the 5-scale MS-SSIM")
    B, Cc, H, W = x.shape
    a = x.detach().contiguous().reshape(B * Cc, H, W)
    b = y.detach().contiguous().reshape(B * Cc, H, W)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    terms = []
    for s in range(5):
        ssim, cs = ops.ssim_level(a, b, _MS_WINDOW, c1, c2)
        if s < 4:
            terms.append(torch.relu(cs))
            a, b = ops.avg_pool2(a), ops.avg_pool2(b)
    terms.append(torch.relu(ssim))
    w = torch.tensor(_MS_WEIGHTS, device=x.device).view(-1, 1)
    val = torch.prod(torch.stack(terms) ** w, dim=0).view(B, Cc)          # 5 x (B*C) scalars: host-side glue
    return val.mean() if size_average else val.mean(1)


class RateDistortionLoss(nn.Module):
    def __init__(self, lmbda=1e-2, metric="mse"):
        super().__init__()
        self.lmbda, self.metric = lmbda, metric

    def forward(self, output, target):
        N, _, H, W = target.size()
        tracked = output["x_hat"].requires_grad or any(v.requires_grad for v in output["likelihoods"].values())
        if torch.is_grad_enabled() and tracked and self.metric == "mse":
            # differentiable form (the opt-in R + lambda*D task loss of the calibration loop): same reduction kernels, gradients
            # through hipops.autograd; no MS-SSIM side value
            from hipops.autograd import NegLog2SumFn, SqDiffSumFn
            bpp = sum(NegLog2SumFn.apply(lik, 1.0 / (N * H * W)) for lik in output["likelihoods"].values())
            mse = SqDiffSumFn.apply(output["x_hat"], target, 1.0 / target.numel())
            return {"bpp_loss": bpp, "mse_loss": mse, "loss": self.lmbda * 255 ** 2 * mse + bpp}
        out = {"bpp_loss": bpp_of(output["likelihoods"], N * H * W), "mse_loss": mse_of(output["x_hat"], target)}
        big = min(target.shape[-2:]) > 160
        if self.metric == "mse":
            out["ms_ssim_loss"] = 1 - ms_ssim(output["x_hat"], target, data_range=1.0) if big else \
                torch.tensor(float("nan"), device=target.device)
            out["loss"] = self.lmbda * 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        elif self.metric == "ms-ssim":
            out["ms_ssim_loss"] = 1 - ms_ssim(output["x_hat"], target, data_range=1.0)
            out["loss"] = self.lmbda * out["ms_ssim_loss"] + out["bpp_loss"]
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
        return bpp, self.PSNR(output["x_hat"], target), self.MS_SSIM(output["x_hat"], target)

    def MS_SSIM(self, x, y):
        if min(x.shape[-2:]) <= 160:           # pytorch_msssim asserts here; report "not available" for small crops instead
            return torch.tensor(float("nan"), device=x.device)
        return torch.mean(ms_ssim(x, y, data_range=1.0, size_average=True))


def compute_psnr(a, b):
    """test_datasets.py:21-23"""
    return -10 * math.log10(float(mse_of(a, b)))


def compute_msssim(a, b):
    """test_datasets.py:25-27: MS-SSIM in dB."""
    return -10 * math.log10(1 - float(ms_ssim(a, b, data_range=1.0)))


def compute_bpp(out_net):
    """test_datasets.py:29-33 (num_pixels of the padded reconstruction)"""
    size = out_net["x_hat"].size()
    return float(bpp_of(out_net["likelihoods"], size[0] * size[2] * size[3]))
