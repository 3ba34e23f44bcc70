#!/usr/bin/env python3
"""In-kernel clocks of the row weight-gradient kernel (needs a `make DIAG=1` library): cycles per 32-pixel stage of the K loop for the
complete kernel and every combination of its three ablations (DMA, fragment reads, MFMAs), each run for half a second.
usage: python tools/wgrad_ablate.py [wgrad_sub]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import _lib as L  # noqa: E402
from hipops import ops  # noqa: E402

B, H, Cc = 4, 128, 192
torch.manual_seed(0)
x = torch.randn(B, H, H, Cc, device="cuda")
w = torch.randn(Cc, 3, 3, Cc, device="cuda")
dy = torch.randn(B, H, H, Cc, device="cuda") * 0.1
xp, dyp = ops.split_h2(x), ops.split_h2(dy)
slabs = ops.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1)
if len(sys.argv) > 1:
    ops.set_tuning("wgrad_sub", int(sys.argv[1]))
wg = lambda: ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, tuple(w.shape), 1, 1, slabs=slabs)
fw = L.lib().rdo_diag_wgrad_stamps
fw.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(1024, dtype=np.uint64)
for name, abl in (("complete", 0), ("no fragment reads", 8), ("no DMA", 1), ("MFMAs only", 9), ("no MFMA", 4), ("DMA only", 12), ("reads only", 5), ("skeleton", 13)):
    ops.set_tuning("x6p_ablate", abl)
    t0 = time.perf_counter()
    n = 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(100):
            wg()
        n += 100
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    assert fw(buf.ctypes.data, 1024) == 0
    s = buf.reshape(256, 4).astype(np.int64)
    cyc, ticks, st = s[:252, 0], s[:252, 1], s[:252, 2]
    loop_us = np.median(ticks) / 100.0
    print(f"wgrad rows {name:18s}: launch {us:6.1f} us | K loop {np.median(cyc / np.maximum(st, 1)):6.0f} cycles per stage "
          f"({int(np.median(st))} stages), {loop_us:6.1f} us wall -> {np.median(cyc) / loop_us / 1e3:5.2f} GHz", flush=True)
ops.set_tuning("x6p_ablate", 0)
