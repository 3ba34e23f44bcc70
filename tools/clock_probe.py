#!/usr/bin/env python3
"""Shader clock and socket power while one kernel runs back to back for a few seconds each: the plane-input conv (MFMA-bound), the
same with its DMA or MFMA ablated, and an HBM-bound element-wise kernel.  Samples `rocm-smi --showclocks --showpower --json` from a
thread every 0.25 s.  usage: python tools/clock_probe.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            samples.append((time.perf_counter(), {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower()}))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), {"error": str(e)}))
        time.sleep(0.25)


def run_for(fn, seconds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return t0, time.perf_counter(), e0.elapsed_time(e1) / n * 1e3


B, H, C = 4, 128, 192
torch.manual_seed(0)
x = torch.randn(B, H, H, C, device="cuda")
w = torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5
wpl6, wpl, xp = ops.split_bf16x3(w), ops.split_h2_conv(w), ops.split_h2(x)
out = torch.empty(B, H, H, C, device="cuda")
y = torch.empty_like(x)
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.5)
phases = [("idle", None)]
conv = lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, None, 1, 1, out=out)
for name, abl in (("plane-input conv (halo kernel)", 0), ("same, no fragment reads", 8), ("same, no DMA (MFMAs + fragment reads)", 3),
                  ("same, no DMA, no fragment reads (MFMAs only)", 11), ("same, no MFMA (DMA + fragment reads)", 4)):
    ops.set_tuning("x6p_ablate", abl)
    phases.append((name, run_for(conv, 4.0)))
ops.set_tuning("x6p_ablate", 0)
phases.append(("fp32-input x6 conv (v6)", run_for(lambda: ops.conv2d_fwd(x, w, None, 1, 1, out=out, wplanes=wpl6), 4.0)))
phases.append(("element-wise add (HBM-bound)", run_for(lambda: ops.add(x, out, out=y), 4.0)))
stop.set()
th.join()
t_first = samples[0][0]
print("idle:", samples[0][1], samples[1][1] if len(samples) > 1 else "")
for name, r in phases[1:]:
    t0, t1, us = r
    mine = [s for (t, s) in samples if t0 + 1.0 < t < t1]
    print(f"{name}: {us:.1f} us/launch; samples: {mine[:3]} ... {mine[-1:] if mine else ''}")
