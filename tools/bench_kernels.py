#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels at the Cheng2020 (N=192) unit shapes.  GPU box only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    shapes = [  # H, Cin, Cout, K, stride, pad
        (128, 192, 192, 3, 1, 1), (64, 192, 192, 3, 1, 1), (32, 192, 192, 3, 1, 1), (16, 192, 192, 3, 1, 1),
        (256, 3, 192, 3, 2, 1), (128, 192, 192, 3, 2, 1), (128, 192, 192, 1, 1, 0), (64, 192, 768, 3, 1, 1),
        (128, 192, 12, 3, 1, 1), (16, 192, 384, 5, 1, 2), (8, 288, 1152, 3, 1, 1), (4, 192, 192, 3, 1, 1),
    ]
    print(f"B={B}")
    for H, Cin, Cout, K, s, p in shapes:
        x = torch.randn(B, H, H, Cin, device="cuda")
        w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
        Ho = (H + 2 * p - K) // s + 1
        dy = torch.randn(B, Ho, Ho, Cout, device="cuda")
        out = torch.empty(B, Ho, Ho, Cout, device="cuda")
        flops = 2.0 * B * Ho * Ho * Cout * Cin * K * K
        t_f = timeit(lambda: ops.conv2d_fwd(x, w, None, s, p, out=out))
        ns = ops.wgrad_nsplit(x.shape, w.shape, s, p)
        slabs = torch.empty((ns,) + tuple(w.shape), device="cuda")
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, tuple(w.shape), s, p, slabs=slabs))
        xt = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        wt = w.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        try:
            t_m = timeit(lambda: torch.nn.functional.conv2d(xt, wt, None, s, p))
        except Exception:
            t_m = float("nan")
        print(f"H={H:4d} Cin={Cin:4d} Cout={Cout:4d} K={K} s={s}: fwd {t_f:8.3f} ms {flops/t_f/1e9:7.1f} TF | "
              f"wgrad(ns={ns:3d}) {t_w:8.3f} ms {flops/t_w/1e9:7.1f} TF | miopen fwd {t_m:8.3f} ms {flops/t_m/1e9:7.1f} TF")


if __name__ == "__main__":
    main()
