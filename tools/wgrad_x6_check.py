#!/usr/bin/env python3
"""A/B of the bf16x6 weight-gradient variants (RDO_WGX6_W8 in the environment): dw against an fp64 reference on a few output
channels, and event timing.  usage: RDO_WGX6_W8=1 python tools/wgrad_x6_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

SHAPES = [(4, 128, 192, 192, 3, 1, 1), (4, 64, 192, 768, 3, 1, 1), (4, 64, 192, 192, 3, 1, 1), (4, 128, 192, 192, 3, 2, 1),
          (4, 32, 192, 192, 3, 1, 1), (2, 64, 320, 192, 5, 2, 2)]
if os.environ.get("WG_SMALL"):
    SHAPES = [(4, 32, 192, 320, 5, 2, 2), (4, 32, 192, 320, 3, 2, 1), (4, 16, 320, 320, 3, 1, 1), (4, 32, 192, 192, 3, 1, 1), (4, 16, 192, 768, 3, 1, 1)]
for (B, H, Cin, Cout, K, s, p) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda")
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda") * 0.1
    wshape = (Cout, K, K, Cin)
    x6 = ops.wgrad_uses_bf16x6(tuple(x.shape), wshape, s, p) if hasattr(ops, "wgrad_uses_bf16x6") else None
    slabs = ops.conv2d_wgrad(x, dy, wshape, s, p)
    dw = ops.reduce_slabs(slabs)
    torch.cuda.synchronize()
    # fp64 reference for 8 output channels: dw[co] = sum_b,ho,wo dy[.., co] * patch(x)
    cos = [0, 1, 47, 48, 95, 96, 143, Cout - 1]
    xd = x.permute(0, 3, 1, 2).double().cpu()
    dyd = dy.permute(0, 3, 1, 2).double().cpu()[:, cos]
    w0 = torch.zeros(len(cos), Cin, K, K, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, w0, stride=s, padding=p) * dyd).sum().backward()
    ref = w0.grad.permute(0, 2, 3, 1)
    err = float((dw[cos].double().cpu() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        ops.conv2d_wgrad(x, dy, wshape, s, p, slabs=slabs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_wgrad(x, dy, wshape, s, p, slabs=slabs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * B * Ho * Ho * Cin * Cout * K * K / 1e9
    print(f"w8={os.environ.get('RDO_WGX6_W8', '0')} B={B} H={H} {Cin}->{Cout} k{K} s{s}: nsplit {slabs.shape[0]:3d}  {us:7.1f} us  "
          f"{gf / us * 1e-3 * 1e3:6.1f} TF  err {err:.2e}")
