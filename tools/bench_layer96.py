#!/usr/bin/env python3
"""The layer-unit convs of Cheng2020-attn's attention blocks (1 x 1 192 <-> 96, 3 x 3 96 -> 96; BASELINE config 3) and the small 3 x 3 units
of the hyper path on the fp32 MFMA kernels they run on against the plane-input (H2) kernels -- would a layer-unit H2 plan pay?
usage: python tools/bench_layer96.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib as L  # noqa: E402

lib = L.lib()
lib.rdo_debug_force_wgrad_choice.argtypes = [C.c_int, C.c_int]
lib.rdo_debug_force_wgrad_choice.restype = None


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, Cin, Cout, K) in [(4, 64, 192, 96, 1), (4, 64, 96, 96, 3), (4, 64, 96, 192, 1), (4, 16, 192, 96, 1), (4, 16, 96, 96, 3), (4, 16, 192, 192, 3),
                             (4, 32, 192, 192, 3)]:
    torch.manual_seed(1)
    p = K // 2
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda")
    dy = torch.randn(B, H, H, Cout, device="cuda")
    xs, ws = tuple(x.shape), tuple(w.shape)
    out = ops.conv2d_fwd(x, w, b, 1, p)
    t_f = timeit(lambda: ops.conv2d_fwd(x, w, b, 1, p, out=out))
    slabs = ops.conv2d_wgrad(x, dy, ws, 1, p)
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, ws, 1, p, slabs=slabs))
    line = f"B={B} {H}^2 {Cin}->{Cout} k{K}: fp32 fwd {t_f:6.1f} us  wgrad {t_w:6.1f} us ({slabs.shape[0]} slabs)"
    if ops.conv_h2_supported(xs, ws, 1, p):
        wpl, xp, dyp = ops.split_h2_conv(w), ops.split_h2(x), ops.split_h2(dy)
        o2 = torch.empty_like(out)
        t_f2 = timeit(lambda: ops.conv2d_fwd_h2(xp, xs, ws, wpl, b, 1, p, out=o2))
        err = float((o2 - out).abs().max() / out.abs().max())
        line += f"   |  H2 fwd {t_f2:6.1f} us (rel diff {err:.1e})"
        lib.rdo_debug_force_wgrad_choice(1, -1)
        try:
            if ops.wgrad_h2_supported(xs, ws, 1, p):
                s2 = ops.conv2d_wgrad_h2(xp, xs, dyp, ws, 1, p)
                t_w2 = timeit(lambda: ops.conv2d_wgrad_h2(xp, xs, dyp, ws, 1, p, slabs=s2))
                errw = float((s2.sum(0) - slabs.sum(0)).abs().max() / slabs.sum(0).abs().max())
                line += f"  wgrad {t_w2:6.1f} us ({s2.shape[0]} slabs, rel diff {errw:.1e})"
        finally:
            lib.rdo_debug_force_wgrad_choice(-1, -1)
    print(line, flush=True)
