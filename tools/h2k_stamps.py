#!/usr/bin/env python3
"""In-kernel clocks of the K32 halo conv (conv_fwd_h2k.hip; needs a `make DIAG=1` library): shader-clock cycles and 100 MHz wall ticks
of the K loop per workgroup for the complete kernel and its compile-time ablations, each run back to back on random data for a second.
usage: python tools/h2k_stamps.py"""
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
w = torch.randn(Cc, 3, 3, Cc, device="cuda") / (Cc * 9) ** 0.5
wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
opl = ops.h2_empty((B, H, H, Cc), "cuda", 8.0)
out = torch.empty(B, H, H, Cc, device="cuda")
fn = L.lib().rdo_diag_h2k_stamps
fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(1024, dtype=np.uint64)
stages = 9 * Cc // 32
ops.set_tuning("h2_k32", 1)
for mode, conv in (("planes", lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, None, 1, 1, out_planes=opl)),
                   ("fp32 out", lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, None, 1, 1, out=out))):
    for name, abl in (("complete", 0), ("no epilogue", 16), ("no fragment reads", 8), ("no DMA", 3), ("MFMAs only", 11), ("no MFMA", 4), ("skeleton", 15),
                      ("skeleton, no epilogue", 31)):
        ops.set_tuning("x6p_ablate", abl)
        t0 = time.perf_counter()
        n = 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(100):
                conv()
            n += 100
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        assert fn(buf.ctypes.data, 1024) == 0
        s = buf.reshape(256, 4).astype(np.int64)
        cyc, ticks = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
        loop_us = np.median(ticks) / 100.0
        print(f"{mode:8s} {name:22s}: launch {us:6.1f} us | K loop median {np.median(cyc):8.0f} cycles = {np.median(cyc) / stages:6.0f} per stage, "
              f"{loop_us:6.1f} us wall -> {np.median(cyc) / max(loop_us, 1e-9) / 1e3:5.2f} GHz", flush=True)
    ops.set_tuning("x6p_ablate", 0)
