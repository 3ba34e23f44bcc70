#!/usr/bin/env python3
"""The GDN gamma gradient (1x1 weight gradient of a 192 x 192 matrix over all pixels, squared input): fp32 small-tile kernel against
the split-bf16 192 x 192 kernel at several pixel-split counts, slab reduction (rdo_reduce_slabs) timed beside it.
usage: python tools/bench_wgrad_1x1.py"""
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
for H in (128, 64, 32):
    torch.manual_seed(0)
    x = torch.randn(4, H, H, 192, device="cuda")
    dy = torch.randn(4, H, H, 192, device="cuda") * 0.1
    ref = torch.einsum("bhwo,bhwi->oi", dy.double(), (x * x).double())
    for big, ns in ((0, 0), (1, 64), (1, 128), (1, 256), (1, 512)):
        if big and ns * 32 > 4 * H * H:
            continue
        lib.rdo_debug_force_wgrad_choice(big if big else -1, ns if ns else -1)
        try:
            slabs = ops.conv2d_wgrad(x, dy, (192, 1, 1, 192), 1, 0, square_input=True)
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, (192, 1, 1, 192), 1, 0, square_input=True, slabs=slabs))
            out = torch.empty(192, 1, 1, 192, device="cuda")
            tr = timeit(lambda: ops.reduce_slabs(slabs, out))
            err = float((out.reshape(192, 192).double() - ref).abs().max() / ref.abs().max())
            print(f"H={H} {'x6 192x192' if big else 'fp32 64x64 '} nsplit {slabs.shape[0]:4d}: wgrad {t:7.1f} us, slab reduce {tr:6.1f} us, rel err {err:.2e}")
        finally:
            lib.rdo_debug_force_wgrad_choice(-1, -1)
