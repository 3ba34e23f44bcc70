#!/usr/bin/env python3
"""Same-process A/B of the two halo forward kernels (tuning key h2_k32: 0 = 16-channel stages on v_mfma_f32_32x32x16_f16,
1 = 32-channel stages on v_mfma_f32_16x16x32_f16) on the Cheng2020 N=192 shapes that run them, interleaved rounds, random data.
usage: python tools/bench_h2k.py [rounds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

SHAPES = [(4, 128, 192, 192), (4, 64, 192, 192), (4, 64, 192, 768), (4, 32, 192, 768), (4, 32, 192, 192)]
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
VA, VB = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)


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
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda")
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    out = torch.empty(B, H, H, Cout, device="cuda")
    opl = ops.h2_empty(out.shape, "cuda", 16.0)
    gf = 2.0 * out.numel() * Cin * 9 / 1e9
    res = {VA: [], VB: []}
    resp = {VA: [], VB: []}
    for _ in range(ROUNDS):
        for v in (VA, VB):
            ops.set_tuning("h2_k32", v)
            res[v].append(timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, out=out)))
            resp[v].append(timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, out_planes=opl)))
    ops.set_tuning("h2_k32", 1)
    med = lambda l: sorted(l)[len(l) // 2]
    print(f"B={B} H={H} {Cin}->{Cout}: h2_k32={VA} out {med(res[VA]):6.1f} us (min {min(res[VA]):6.1f}, {gf / med(res[VA]) * 1e3:5.0f} TF)  h2_k32={VB} out {med(res[VB]):6.1f} us "
          f"(min {min(res[VB]):6.1f}, {gf / med(res[VB]) * 1e3:5.0f} TF) | planes only: {med(resp[VA]):6.1f}  {med(resp[VB]):6.1f}", flush=True)
