#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc passes).  usage: bench_one.py fwd|wgrad H Cin Cout K stride pad [B] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

kind = sys.argv[1]
H, Cin, Cout, K, s, p = (int(v) for v in sys.argv[2:8])
B = int(sys.argv[8]) if len(sys.argv) > 8 else 4
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 5
x = torch.randn(B, H, H, Cin, device="cuda")
w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
Ho = (H + 2 * p - K) // s + 1
dy = torch.randn(B, Ho, Ho, Cout, device="cuda")
out = torch.empty(B, Ho, Ho, Cout, device="cuda")
ns = ops.wgrad_nsplit(x.shape, w.shape, s, p)
slabs = torch.empty((ns,) + tuple(w.shape), device="cuda")
for _ in range(reps):
    if kind == "fwd":
        ops.conv2d_fwd(x, w, None, s, p, out=out)
    else:
        ops.conv2d_wgrad(x, dy, tuple(w.shape), s, p, slabs=slabs)
torch.cuda.synchronize()
