"""Evaluation harness `Test_kodak` (reference surface: test_datasets.py:76-117): pad to a multiple of 256, full-model
forward, crop, clamp, PSNR / MS-SSIM / bpp."""
import logging
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from losses.losses import compute_bpp, compute_msssim, compute_psnr


def pad(x, p=2 ** 6):
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode="constant", value=0)


def crop(x, size):
    H, W = x.size(2), x.size(3)
    h, w = size
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (-left, -(W - w - left), -top, -(H - h - top)), mode="constant", value=0)


def evaluate_images(model, images, p=256, with_msssim=False, distributed=None):
    """images: iterable of [1,3,h,w] tensors in [0,1] -> (mean PSNR dB, mean bpp[, mean MS-SSIM dB]).

    Image-parallel over the ranks of an initialised process group (BASELINE config 5: Kodak / Tecnick over 8 GPUs): rank r
    evaluates images r, r + world, ... and the four sums are all-reduced; `distributed=False` forces a local evaluation."""
    device = next(model.parameters()).device
    use_dist = torch.distributed.is_available() and torch.distributed.is_initialized() if distributed is None else distributed
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if use_dist else (0, 1)
    psnr = bpp = msssim = 0.0
    n = 0
    for i, x in enumerate(images):
        if i % world != rank:
            continue
        x = x.to(device)
        h, w = x.size(2), x.size(3)
        with torch.no_grad():
            out = model.forward(pad(x, p))
        rec = crop(out["x_hat"], (h, w)).clamp(0, 1)
        psnr += compute_psnr(x, rec)
        bpp += compute_bpp(out)
        if with_msssim:
            msssim += compute_msssim(x, rec)
        n += 1
    if use_dist:
        tot = torch.tensor([psnr, bpp, msssim, float(n)], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tot)
        psnr, bpp, msssim, n = (float(v) for v in tot)
    return (psnr / n, bpp / n, msssim / n) if with_msssim else (psnr / n, bpp / n)


def Test_kodak(model=None, testset_path="./datasets/kodak24"):
    from PIL import Image
    files = sorted(f for f in os.listdir(testset_path) if f.lower().endswith(".png"))

    def load():
        for f in files:
            img = np.asarray(Image.open(os.path.join(testset_path, f)).convert("RGB"), dtype=np.float32) / 255.0
            yield torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0)
    psnr, bpp, msssim = evaluate_images(model, load(), with_msssim=True)
    logging.info("Test Data: Kodak24 with 512x768 ")
    logging.info(f"AVG PSNR: {psnr:.2f}dB")
    logging.info(f"AVG MS-SSIM: {msssim:.2f}dB")
    logging.info(f"AVG Bit-rate: {bpp:.4f} bpp")
    return psnr, bpp
