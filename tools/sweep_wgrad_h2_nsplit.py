#!/usr/bin/env python3
"""Row weight-gradient kernel at different pixel-split counts (slabs): kernel time, and kernel time + the time the AdaRound step needs to
read the slabs at its measured ~2.9 TB/s -- the quantity that matters for the step.  usage: python tools/sweep_wgrad_h2_nsplit.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, Cin, Cout) in [(4, 32, 192, 192), (4, 64, 192, 192), (4, 128, 192, 192), (4, 32, 192, 768), (4, 64, 192, 768)]:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda")
    dy = torch.randn(B, H, H, Cout, device="cuda") * 0.1
    wshape = (Cout, 3, 3, Cin)
    xp, dyp = ops.split_h2(x), ops.split_h2(dy)
    dflt = ops.wgrad_nsplit(tuple(x.shape), wshape, 1, 1)
    row = []
    for ns in sorted({2, 4, 7, 8, 14, 16, 28, dflt}):
        if ns * 32 > B * H * H:
            continue
        slabs = torch.empty((ns,) + wshape, device="cuda")
        t = timeit(lambda: ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, wshape, 1, 1, slabs=slabs))
        read = ns * slabs[0].numel() * 4 / 2.9e12 * 1e6
        row.append(f"ns {ns:3d}{'*' if ns == dflt else ' '}: {t:6.1f} + {read:5.1f} = {t + read:6.1f}")
    print(f"B={B} H={H} {Cin}->{Cout}: " + " | ".join(row), flush=True)
