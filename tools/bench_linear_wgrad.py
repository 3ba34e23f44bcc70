#!/usr/bin/env python3
"""Token-matrix weight gradients (1 x 1, fp32 operands) on the Lu2022 / GDN shapes: linear_wgrad_h2_kernel (default) against the split-bf16
kernel it replaces (run again with RDO_LIN_WGRAD_H2=0).  usage: python tools/bench_linear_wgrad.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for rows, cin, cout, sq in [(65536, 192, 576, False), (65536, 192, 192, False), (65536, 192, 384, False), (65536, 384, 192, False),
                            (16384, 192, 576, False), (16384, 384, 192, False), (65536, 192, 192, True), (16384, 192, 192, True), (4096, 192, 192, True),
                            (4096, 192, 576, False), (4096, 192, 192, False), (4096, 384, 192, False), (4096, 192, 384, False)]:
    x = torch.randn(1, 1, rows, cin, device="cuda", generator=g)
    dy = torch.randn(1, 1, rows, cout, device="cuda", generator=g)
    ws = (cout, 1, 1, cin)
    slabs = ops.conv2d_wgrad(x, dy, ws, 1, 0, square_input=sq)
    t = timed(lambda: ops.conv2d_wgrad(x, dy, ws, 1, 0, square_input=sq, slabs=slabs))
    fl = 2.0 * rows * cin * cout
    by = 4.0 * rows * (cin + cout)
    print(f"{rows:6d} tokens  {cin:3d} x {cout:3d}{' sq' if sq else '   '}: {t:7.1f} us  {fl / t * 1e-6:6.1f} TFLOP/s  {by / t * 1e-3:6.0f} GB/s  ({slabs.shape[0]} slabs)", flush=True)
