#!/usr/bin/env python3
"""Is the halo conv slower on inputs that are not cache-resident?  Times the 4 x 128^2 conv (planes in, planes out) back to back on ONE
input (Infinity-Cache resident after the first launch) and on inputs that were just written by another kernel / evicted by a 512 MiB fill."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rdo-ptq_amd"))
from hipops import ops
B, H, Cin, Cout = 4, 128, 192, 192
torch.manual_seed(1)
w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
wpl = ops.split_h2_conv(w)
xs = [torch.randn(B, H, H, Cin, device="cuda") for _ in range(4)]
xps = [ops.split_h2(x) for x in xs]
opl = ops.h2_empty((B, H, H, Cout), "cuda", 16.0)
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def conv(xp): ops.conv2d_fwd_h2(xp, (B, H, H, Cin), tuple(w.shape), wpl, None, 1, 1, out_planes=opl)
def ev(): return torch.cuda.Event(enable_timing=True)
def timed(pre):
    ts = []
    for i in range(24):
        pre(i)
        e0, e1 = ev(), ev()
        e0.record(); conv(xps[i % 4]); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[4:])
    return ts[len(ts) // 2]
for _ in range(10): conv(xps[0])
print("same input, back to back      :", round(timed(lambda i: None), 1), "us")
print("input re-split just before    :", round(timed(lambda i: ops.split_h2(xs[i % 4], xps[i % 4])), 1), "us")
print("512 MiB fill before the launch:", round(timed(lambda i: big.fill_(i & 1)), 1), "us")
