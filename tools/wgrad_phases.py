#!/usr/bin/env python3
"""Where a stage of the row weight-gradient kernel spends its cycles (needs a `make DIAG=1` library; tuning key x6p_ablate bit 64 turns
the in-kernel phase stamps on): per 32-pixel segment, for the eight waves (0-3 issue their DMAs early, 4-7 late) of every workgroup -- wait for the DMAs
of this stage, barrier, DMA issue, wait for the first fragments, the nine MFMA slots.  The stamps cost cycles themselves: read the
shares, not the total.   usage: python tools/wgrad_phases.py [wgrad_sub]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import _lib as L  # noqa: E402
from hipops import ops  # noqa: E402

B, H, Cc = 4, 128, 192
SUB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
x = torch.randn(B, H, H, Cc, device="cuda")
dy = torch.randn(B, H, H, Cc, device="cuda") * 0.1
xp, dyp = ops.split_h2(x), ops.split_h2(dy)
slabs = ops.conv2d_wgrad(x, dy, (Cc, 3, 3, Cc), 1, 1)
ops.set_tuning("wgrad_sub", SUB)
wg = lambda: ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, (Cc, 3, 3, Cc), 1, 1, slabs=slabs)
fs = L.lib().rdo_diag_wgrad_stamps
fs.argtypes = [C.c_void_p, C.c_int]
fp = L.lib().rdo_diag_wgrad_phases
fp.argtypes = [C.c_void_p, C.c_int]
sb, pb = np.zeros(1024, dtype=np.uint64), np.zeros(256 * 64, dtype=np.uint64)
for abl in (0, 64):
    ops.set_tuning("x6p_ablate", abl)
    for _ in range(200):
        wg()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        wg()
    e1.record()
    torch.cuda.synchronize()
    assert fs(sb.ctypes.data, 1024) == 0
    s = sb.reshape(256, 4).astype(np.int64)[:252]
    print(f"wgrad_sub={SUB} stamps {'on ' if abl else 'off'}: launch {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us, K loop {np.median(s[:, 0] / np.maximum(s[:, 2], 1)):6.0f} cycles per 32-pixel segment "
          f"({int(np.median(s[:, 2]))} segments) at {np.median(s[:, 0]) / (np.median(s[:, 1]) / 100.0) / 1e3:4.2f} GHz")
assert fp(pb.ctypes.data, 256 * 64) == 0
ph = pb.reshape(256, 8, 8).astype(np.float64)[:252]
segs = np.median(s[:, 2])
names = ["wait DMA (vmcnt)", "barrier", "DMA issue (early)", "wait first fragments", "nine MFMA slots (+ late DMA issue)"]
for w, tag in [(i, f"wave {i} ({'early' if i < 4 else 'late '})") for i in range(8)]:
    print(tag + ": " + " | ".join(f"{n} {np.median(ph[:, w, k]) / segs:6.0f}" for k, n in enumerate(names)) + f" | sum {np.median(ph[:, w, :5].sum(1)) / segs:6.0f}")
ops.set_tuning("x6p_ablate", 0)
