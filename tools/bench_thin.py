#!/usr/bin/env python3
"""The four thin convolutions of Cheng2020-anchor at batch 4 (stem 3 -> 192 3x3 s2 and its 1x1 skip; last layer 192 -> 12 3x3 at
128^2): forward and weight gradient, event timing, weight-gradient pixel-split sweep.   usage: python tools/bench_thin.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import _lib as L  # noqa: E402
from hipops import ops  # noqa: E402


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


lib = L.lib()
lib.rdo_debug_force_wgrad_choice.argtypes = [C.c_int, C.c_int]
lib.rdo_debug_force_wgrad_choice.restype = None
for name, (B, H, Cin, Cout, K, s, p) in (("stem 3->192 3x3 s2", (4, 256, 3, 192, 3, 2, 1)), ("stem skip 3->192 1x1 s2", (4, 256, 3, 192, 1, 2, 0)),
                                         ("last 192->12 3x3", (4, 128, 192, 12, 3, 1, 1))):
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, K, K, Cin, device="cuda") / (K * K * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda")
    out = ops.conv2d_fwd(x, w, b, s, p)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: ops.conv2d_fwd(x, w, b, s, p, out=out))
    mb = (x.numel() + out.numel()) * 4 / 1e6
    print(f"{name}: fwd {t:7.1f} us ({mb / t:5.2f} TB/s of x + out), rel err {err:.1e}")
    dy = torch.randn_like(out) * 0.1
    refw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, K, K), dy.permute(0, 3, 1, 2).double(), stride=s, padding=p).permute(0, 2, 3, 1)
    for ns in (0, 64, 128, 256, 512):
        lib.rdo_debug_force_wgrad_choice(-1, ns if ns else -1)
        try:
            slabs = ops.conv2d_wgrad(x, dy, tuple(w.shape), s, p)
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, tuple(w.shape), s, p, slabs=slabs))
            g = ops.reduce_slabs(slabs)
            tr = timeit(lambda: ops.reduce_slabs(slabs, g))
            err = float((g.double() - refw).abs().max() / refw.abs().max())
            print(f"    wgrad nsplit {slabs.shape[0]:4d}{' (default)' if not ns else ''}: {t:7.1f} us, slab reduce {tr:5.1f} us, rel err {err:.1e}")
        finally:
            lib.rdo_debug_force_wgrad_choice(-1, -1)
