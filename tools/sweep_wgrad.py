#!/usr/bin/env python3
"""Sweep (tile, nsplit) of the wgrad kernel over the workload's shapes."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib  # noqa: E402

lib = _lib.lib()
lib.rdo_debug_force_wgrad_choice.argtypes = [C.c_int, C.c_int]
lib.rdo_debug_force_wgrad_choice.restype = None


def timeit(fn, n=6, warm=2):
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


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(128, 192, 192, 3, 1, 1), (64, 192, 192, 3, 1, 1), (32, 192, 192, 3, 1, 1), (16, 192, 192, 3, 1, 1),
          (8, 192, 192, 3, 1, 1), (4, 192, 192, 3, 1, 1), (128, 192, 192, 3, 2, 1), (128, 192, 192, 1, 1, 0),
          (64, 192, 768, 3, 1, 1), (16, 192, 768, 3, 1, 1), (16, 192, 384, 5, 1, 2), (8, 288, 1152, 3, 1, 1),
          (16, 768, 640, 1, 1, 0), (256, 3, 192, 3, 2, 1), (128, 192, 12, 3, 1, 1)]
for H, Cin, Cout, K, s, p in shapes:
    x = torch.randn(B, H, H, Cin, device="cuda")
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda")
    wshape = (Cout, K, K, Cin)
    fl = 2.0 * B * Ho * Ho * Cout * Cin * K * K
    lib.rdo_debug_force_wgrad_choice(-1, -1)
    ns0 = ops.wgrad_nsplit(x.shape, wshape, s, p)
    slabs = torch.empty((256,) + wshape, device="cuda") if Cout * K * K * Cin * 256 * 4 < 6e9 else None
    def run(ns):
        sl = slabs[:ns] if slabs is not None else torch.empty((ns,) + wshape, device="cuda")
        return timeit(lambda: ops.conv2d_wgrad(x, dy, wshape, s, p, slabs=sl))
    t_model = run(ns0)
    res = []
    M = B * Ho * Ho
    for big in (0, 1):
        for ns in (1, 2, 4, 7, 8, 14, 16, 28, 32, 56, 64, 128):
            if ns > max(1, M // 64) or (slabs is None and ns > 32):
                continue
            lib.rdo_debug_force_wgrad_choice(big, ns)
            res.append((run(ns), big, ns))
    res.sort()
    print(f"H={H:4d} Cin={Cin:4d} Cout={Cout:4d} K={K} s={s} M={M:6d}: model ns={ns0:3d} {t_model:7.1f} us ({fl/t_model/1e6:6.1f} TF) | best "
          f"{res[0][0]:7.1f} us big{res[0][1]} ns{res[0][2]} ({fl/res[0][0]/1e6:6.1f} TF) | " + " ".join(f"b{b}n{n}:{us:.0f}" for us, b, n in res[:6]))
lib.rdo_debug_force_wgrad_choice(-1, -1)
