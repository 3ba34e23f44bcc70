#!/usr/bin/env python3
"""Diagnostic builds: event time of the 4 x 128^2 halo conv under x6p_ablate values (compile-time variants of the K32 kernel).
usage: python tools/ab_ablate.py 0 101 102 103"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rdo-ptq_amd"))
from hipops import ops
vals = [int(v) for v in sys.argv[1:]] or [0]
def timeit(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, Cin, Cout = 4, 128, 192, 192
torch.manual_seed(1)
x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
opl = ops.h2_empty((B, H, H, Cout), "cuda", 16.0)
r, o = {v: [] for v in vals}, {}
for _ in range(5):
    for v in vals:
        ops.set_tuning("x6p_ablate", v)
        r[v].append(timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, None, 1, 1, out_planes=opl)))
        o[v] = opl.t.clone()
ops.set_tuning("x6p_ablate", 0)
print({v: round(sorted(l)[len(l) // 2], 1) for v, l in r.items()}, "same bits as first:", {v: bool(torch.equal(o[v], o[vals[0]])) for v in vals})
