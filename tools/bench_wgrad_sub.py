#!/usr/bin/env python3
"""Same-process A/B of a tuning key on the row weight-gradient kernel (default: wgrad_sub = 1 / 2), interleaved rounds, random data; checks that
both settings give the same slabs bit for bit.
usage: python tools/bench_wgrad_sub.py [key] [v0] [v1] [rounds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

KEY = sys.argv[1] if len(sys.argv) > 1 else "wgrad_sub"
V0, V1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 2)
ROUNDS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
SHAPES = [(4, 128, 192, 192), (4, 64, 192, 192), (4, 64, 192, 768), (4, 32, 192, 192)]


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


for (B, H, Cin, Cout) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda")
    dy = torch.randn(B, H, H, Cout, device="cuda") * 0.1
    wshape = (Cout, 3, 3, Cin)
    if not ops.wgrad_h2_supported(tuple(x.shape), wshape, 1, 1):
        print(f"B={B} H={H} {Cin}->{Cout}: not on the plane path")
        continue
    slabs = ops.conv2d_wgrad(x, dy, wshape, 1, 1)
    xp, dyp = ops.split_h2(x), ops.split_h2(dy)
    gf = 2.0 * dy.numel() * Cin * 9 / 1e9
    res, outs = {V0: [], V1: []}, {}
    for _ in range(ROUNDS):
        for v in (V0, V1):
            ops.set_tuning(KEY, v)
            res[v].append(timeit(lambda: ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, wshape, 1, 1, slabs=slabs)))
            outs[v] = slabs.clone()
    ops.set_tuning(KEY, V0)
    med = lambda l: sorted(l)[len(l) // 2]
    same = bool(torch.equal(outs[V0], outs[V1]))
    print(f"B={B} H={H} {Cin}->{Cout} nsplit {slabs.shape[0]:3d}: {KEY}={V0} {med(res[V0]):6.1f} us ({gf / med(res[V0]) * 1e3:5.0f} TF)  {KEY}={V1} {med(res[V1]):6.1f} us "
          f"({gf / med(res[V1]) * 1e3:5.0f} TF)  same bits: {same}", flush=True)
