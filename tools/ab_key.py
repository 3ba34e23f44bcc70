#!/usr/bin/env python3
"""Same-process A/B of any tuning key on the plane-input forward conv: python tools/ab_key.py KEY V0 V1 [B H Cin Cout]..."""
import sys, torch
sys.path.insert(0, "/root/repo/rdo-ptq_amd")
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rdo-ptq_amd"))
from hipops import ops
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
shapes = [(4, 64, 192, 192), (4, 32, 192, 768), (4, 128, 192, 192)]
def timeit(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, Cin, Cout) in shapes:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda")
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    opl = ops.h2_empty((B, H, H, Cout), "cuda", 16.0)
    out = torch.empty(B, H, H, Cout, device="cuda")
    r, o = {v0: [], v1: []}, {}
    for _ in range(5):
        for v in (v0, v1):
            ops.set_tuning(key, v)
            r[v].append(timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, out=out, out_planes=opl)))
            o[v] = (out.clone(), opl.t.clone())
    ops.set_tuning(key, v0)
    print(B, H, Cin, Cout, {v: round(sorted(l)[len(l) // 2], 1) for v, l in r.items()}, "same bits:", bool(torch.equal(o[v0][0], o[v1][0]) and torch.equal(o[v0][1], o[v1][1])))
