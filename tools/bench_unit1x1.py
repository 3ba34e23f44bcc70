#!/usr/bin/env python3
"""Micro-benchmark of rdo_unit1x1 (forward + tail + weight-gradient slabs of a 1 x 1 layer unit in one launch) at the shapes of Cheng2020-attn's
attention blocks, next to the three launches it replaces (conv forward, fused tail, weight gradient)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timeit(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, K, N in [(4, 64, 192, 96), (4, 64, 96, 192), (4, 64, 192, 192), (4, 16, 192, 96), (4, 16, 96, 192), (4, 16, 192, 192)]:
    x = torch.randn(B, H, H, K, device="cuda")
    w = torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    tgt = torch.randn(B + 4, H, H, N, device="cuda")
    idx = torch.stack([torch.randperm(B + 4)[:B] for _ in range(2)]).to(torch.int32).cuda()
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    log = torch.zeros(2, 32, device="cuda")
    ts = {}
    for form in (0, 1):
        was = ops.unit1x1_form(form)
        slabs = torch.empty(ops.unit1x1_nslab(B * H * H, N), N, 1, 1, K, device="cuda")
        ts[form] = timeit(lambda: ops.unit1x1(x, w, b, tgt, idx, it, 2.0, 2, log, slabs))
        ops.unit1x1_form(was)
    t1 = ts[0]
    y, dpre = torch.empty(B, H, H, N, device="cuda"), torch.empty(B, H, H, N, device="cuda")

    def three():
        ops.conv2d_fwd(x, w, b, 1, 0, out=y)
        ops.loss_act_bwd(y, None, tgt, idx, it, 2.0, ops.ACT_RELU, log, dpre=dpre)
        ops.conv2d_wgrad(x, dpre, tuple(w.shape), 1, 0)
    t3 = timeit(three)
    print(f"B {B} {H}x{H} {K}->{N}: unit1x1 fp32 {t1:6.1f} us, split-fp16 {ts[1]:6.1f} us ({slabs.shape[0]} slabs) | conv + tail + wgrad {t3:6.1f} us (back to back, eager)")
