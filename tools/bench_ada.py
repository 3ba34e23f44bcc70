#!/usr/bin/env python3
"""Micro-benchmark of rdo_adaround_step at the Cheng2020 weight sizes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib  # noqa: E402


def timeit(fn, n=20, warm=3):
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


for shape, ns, dgrad in [((192, 3, 3, 192), 8, False), ((192, 3, 3, 192), 28, True), ((768, 3, 3, 192), 7, False), ((192, 1, 1, 192), 64, True)]:
    w = torch.randn(shape, device="cuda") * 0.1
    delta, zp = ops.uaq_init_minmax(w.reshape(shape[0], -1), 256)
    d = ops.ada_desc(w)
    alpha = ops.adaround_init_alpha(d, w, delta)
    m, v, wq = torch.zeros_like(w), torch.zeros_like(w), torch.empty_like(w)
    wd = torch.empty_like(w) if dgrad else None
    slabs = torch.randn((ns,) + shape, device="cuda") * 1e-3
    for on in (0.0, 1.0):
        sched = torch.tensor([[10.0 * on, on, 1e-3, 1.0]] * 4, device="cuda")
        it = torch.zeros(1, dtype=torch.int32, device="cuda")
        log = torch.zeros(4, 32, device="cuda")
        t = timeit(lambda: ops.adaround_step(d, w, delta, zp, slabs, 1.0, 0.01, sched, it, alpha, m, v, wq, wd, log))
        by = w.numel() * 4 * (ns + 9)
        print(f"shape {shape} ns={ns:2d} dgrad={dgrad} round_on={on}: {t:6.1f} us  {by/t/1e6:7.1f} GB/s")
