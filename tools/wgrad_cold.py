#!/usr/bin/env python3
"""Why does the row weight gradient take longer in the calibration loop than back to back?  Times it (HIP events around the launch
alone) in a stream of identical launches, behind a kernel that has just re-written its two inputs, and behind the forward conv of
the same shape (the loop's order).   usage: python tools/wgrad_cold.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

B, H, Cc = 4, 128, 192
torch.manual_seed(0)
x = torch.randn(B, H, H, Cc, device="cuda")
dy = torch.randn(B, H, H, Cc, device="cuda") * 0.1
w = torch.randn(Cc, 3, 3, Cc, device="cuda") / (Cc * 9) ** 0.5
xp, dyp, wp = ops.split_h2(x), ops.split_h2(dy), ops.split_h2_conv(w)
o1 = ops.h2_empty(x.shape, "cuda", 16.0)
slabs = ops.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1)
xs, ws = tuple(x.shape), tuple(w.shape)
wg = lambda: ops.conv2d_wgrad_h2(xp, xs, dyp, ws, 1, 1, slabs=slabs)
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")


def timed(before, n=30):
    ts = []
    for _ in range(n):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wg()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for _ in range(20):
    wg()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    wg()
e1.record()
torch.cuda.synchronize()
print(f"back to back                                  {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
print(f"alone, inputs untouched since the last launch {timed(lambda: None):7.1f} us")
print(f"alone, behind split_h2 of both inputs         {timed(lambda: (ops.split_h2(x, xp), ops.split_h2(dy, dyp))):7.1f} us")
print(f"alone, behind the forward conv (writes planes){timed(lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, None, 1, 1, out_planes=o1)):7.1f} us")
print(f"alone, behind a 256 MiB memset (caches cold)  {timed(lambda: big.zero_()):7.1f} us")
