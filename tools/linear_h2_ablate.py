#!/usr/bin/env python3
"""Where rdo_linear_h2 spends its time (needs a `make DIAG=1` library): the kernel with its output stores, its MFMAs or its panel loads
switched off (RDO_LIN_DIAG bits 1 / 2 / 4), on the Lu2022 shapes.  usage: python tools/linear_h2_ablate.py"""
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
names = {0: "full", 1: "no stores", 2: "no MFMA", 4: "no loads", 3: "loads only", 5: "MFMA only", 6: "stores only", 7: "nothing"}
for rows, K, N in [(65536, 192, 576), (65536, 192, 192), (65536, 576, 192), (16384, 192, 576), (16384, 192, 192)]:
    x = torch.randn(rows, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    planes = ops.split_h2_linear(w)
    y = torch.empty(rows, N, device="cuda")
    line = []
    for d in (0, 1, 2, 4, 3, 5, 6, 7):
        os.environ["RDO_LIN_DIAG"] = str(d)
        line.append(f"{names[d]} {timed(lambda: ops.linear_h2(x, planes, None, out=y)):6.1f}")
    os.environ["RDO_LIN_DIAG"] = "0"
    print(f"{rows:6d} x {K:3d} -> {N:3d}: " + " | ".join(line), flush=True)
