"""ORACLE (test infrastructure): MS-SSIM as the reference consumes it (`from pytorch_msssim import ms_ssim`, test_datasets.py:16,
25-27; losses/losses.py:5,27,54).  pytorch_msssim==1.0.0 (requirements.txt:6) is not vendored: restated from its published
algorithm -- **parity unpinned [3P]**:
  * 11-tap Gaussian window, sigma 1.5, normalised; applied separably per channel with VALID padding;
  * SSIM / contrast-structure maps with K = (0.01, 0.03), data_range L: C1 = (K1 L)^2, C2 = (K2 L)^2, averaged over space;
  * 5 scales, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333); between scales 2x2 average pooling with padding (H%2, W%2);
  * per channel  prod_{s<4} relu(cs_s)^w_s * relu(ssim_4)^w_4, then the mean over channels (and over the batch if asked)."""
import torch
import torch.nn.functional as F

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gaussian_window(size=11, sigma=1.5):
    c = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, win):
    C = x.shape[1]
    k = win.numel()
    x = F.conv2d(x, win.view(1, 1, k, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, win.view(1, 1, 1, k).repeat(C, 1, 1, 1), groups=C)


def ssim_level(x, y, win, data_range=1.0, K=(0.01, 0.03)):
    """-> (ssim, cs) per (batch, channel)."""
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filter(x, win), _filter(y, win)
    s11 = _filter(x * x, win) - mu1 * mu1
    s22 = _filter(y * y, win) - mu2 * mu2
    s12 = _filter(x * y, win) - mu1 * mu2
    cs_map = (2 * s12 + C2) / (s11 + s22 + C2)
    ssim_map = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(x, y, data_range=1.0, size_average=True):
    if min(x.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("image side must exceed 160 for the 5-scale MS-SSIM")
    win = gaussian_window()
    w = torch.tensor(WEIGHTS)
    mcs = []
    for s in range(5):
        ssim_c, cs = ssim_level(x, y, win, data_range)
        if s < 4:
            mcs.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim_c)], dim=0)               # [5, B, C]
    out = torch.prod(vals ** w.view(-1, 1, 1), dim=0)
    return out.mean() if size_average else out.mean(1)
