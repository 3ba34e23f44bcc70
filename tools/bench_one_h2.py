#!/usr/bin/env python3
"""Run one 3x3/s1/p1 conv on the split-precision plane path a few times (for rocprofv3 --pmc passes).
usage: bench_one_h2.py fwd|wgrad H Cin Cout [B] [reps] [key=value tuning ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

kind = sys.argv[1]
H, Cin, Cout = (int(v) for v in sys.argv[2:5])
pos = [a for a in sys.argv[5:] if "=" not in a]
B = int(pos[0]) if pos else 4
reps = int(pos[1]) if len(pos) > 1 else 5
for a in sys.argv[5:]:
    if "=" in a:
        k, v = a.split("=")
        ops.set_tuning(k, int(v))
x = torch.randn(B, H, H, Cin, device="cuda")
w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
dy = torch.randn(B, H, H, Cout, device="cuda") * 0.1
xp, dyp = ops.split_h2(x), ops.split_h2(dy)
if kind == "wgrad":
    slabs = ops.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1)
    for _ in range(reps):
        ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, tuple(w.shape), 1, 1, slabs=slabs)
else:
    wp = ops.split_h2_conv(w)
    out = torch.empty(B, H, H, Cout, device="cuda")
    for _ in range(reps):
        ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wp, None, 1, 1, out=out)
torch.cuda.synchronize()
