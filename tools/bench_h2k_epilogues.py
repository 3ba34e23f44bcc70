#!/usr/bin/env python3
"""Back-to-back launch times of the K32 halo kernel with the epilogues the calibration loop uses: planes only (conv1 of a residual block),
conv + unit tail (conv2), dgrad with the LeakyReLU mask off the planes of h1, fp32 out only (the conv of a GDN block).
usage: python tools/bench_h2k_epilogues.py [B H C]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import _lib as L  # noqa: E402
from hipops import ops  # noqa: E402

B, H, Cc = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 128, 192)


def timeit(fn, n=40, rounds=3):
    best = []
    for _ in range(rounds):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(best)[len(best) // 2]


torch.manual_seed(0)
n_img = 16
x = torch.randn(B, H, H, Cc, device="cuda")
w = torch.randn(Cc, 3, 3, Cc, device="cuda") / (Cc * 9) ** 0.5
bias = torch.randn(Cc, device="cuda") * 0.1
tgt = torch.randn(n_img, H, H, Cc, device="cuda")
idx = torch.randint(0, n_img, (4, B), dtype=torch.int32, device="cuda")
it = torch.zeros(1, dtype=torch.int32, device="cuda")
log = torch.zeros(4, 32, device="cuda")
xp, wp = ops.split_h2(x), ops.split_h2_conv(w)
resp = ops.split_h2(torch.randn_like(x))
auxp = ops.split_h2(torch.randn_like(x))
o1, o2, o3 = (ops.h2_empty(x.shape, "cuda", 16.0) for _ in range(3))
out = torch.empty_like(x)
xs, ws = tuple(x.shape), tuple(w.shape)
rows = [
    ("planes only, LeakyReLU (conv1)", lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, bias, 1, 1, epilogue=L.EPI_LRELU, out_planes=o1)),
    ("conv + unit tail (conv2)", lambda: ops.conv2d_fwd_h2_tail(xp, xs, ws, wp, bias, 1, 1, resp, tgt, idx, it, 2.0, ops.ACT_LRELU, o2, log)),
    ("dgrad, LeakyReLU mask off planes", lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, None, 1, 1, epilogue=L.EPI_LRELU_BWD, aux_planes=auxp, out_planes=o3)),
    ("fp32 out only (GDN block conv)", lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, bias, 1, 1, out=out)),
    ("fp32 out, LeakyReLU", lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, bias, 1, 1, epilogue=L.EPI_LRELU, out=out)),
    ("planes only, no activation", lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, bias, 1, 1, out_planes=o1)),
    ("fp32 out + planes", lambda: ops.conv2d_fwd_h2(xp, xs, ws, wp, bias, 1, 1, out=out, out_planes=o1)),
]
for name, fn in rows + rows[:2]:
    print(f"B={B} H={H} C={Cc}  {name:36s} {timeit(fn):7.1f} us", flush=True)
