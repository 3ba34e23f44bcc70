#!/usr/bin/env python3
"""rdo_linear_h2 (per-token-scaled fp16-split Linear) against the split-bf16 1x1-conv path it replaces and float64, on the Lu2022 shapes.
usage: python tools/bench_linear_h2.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for rows, K, N, kind in [(65536, 192, 576, "act"), (65536, 192, 192, "act"), (65536, 192, 384, "act"), (65536, 384, 192, "act"),
                         (65536, 576, 192, "grad"), (65536, 384, 192, "grad"), (16384, 192, 576, "act"), (16384, 576, 192, "grad"), (16384, 384, 192, "act")]:
    x = torch.randn(rows, K, device="cuda", generator=g)
    if kind == "grad":       # gradient-like: tiny, magnitudes varying by orders from token to token
        x = x * 1e-6 * torch.exp(3 * torch.randn(rows, 1, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    planes = ops.split_h2_linear(w)
    y = ops.linear_h2(x, planes, b)
    w4 = w.reshape(N, 1, 1, K).contiguous()
    x4 = x.view(1, 1, rows, K)
    p3 = ops.split_bf16x3(w4) if ops.uses_bf16x6(tuple(x4.shape), tuple(w4.shape), 1, 0) else None
    y6 = ops.conv2d_fwd(x4, w4, b, 1, 0, wplanes=p3).view(rows, N)
    sub = slice(0, 4096)
    want = x[sub].double() @ w.double().t() + b.double()
    # error per token relative to the token's own output scale (what a per-token scale promises), and to the tensor's max
    tok = want.abs().amax(1, keepdim=True).clamp_min(1e-300)
    e2 = float(((y[sub].double() - want).abs() / tok).max())
    e6 = float(((y6[sub].double() - want).abs() / tok).max())
    t2 = timed(lambda: ops.linear_h2(x, planes, b, out=y))
    t6 = timed(lambda: ops.conv2d_fwd(x4, w4, b, 1, 0, wplanes=p3, out=y6.view(1, 1, rows, N)))
    fl = 2.0 * rows * K * N
    print(f"{rows:6d} x {K:3d} -> {N:3d} {kind:4s}: linear_h2 {t2:7.1f} us ({fl / t2 * 1e-6:6.1f} TFLOP/s)  bf16x6 conv {t6:7.1f} us ({fl / t6 * 1e-6:6.1f})   "
          f"max err / token max: h2 {e2:.2e}  x6 {e6:.2e}", flush=True)

# the GELU epilogues (Mlp.fc1 + GELU: 192 -> 384 with the pre-activation kept; fc2's input gradient through the GELU: 192 -> 384 reading the pre-activation)
from hipops import _lib as L  # noqa: E402
for rows in (65536, 16384):
    x = torch.randn(rows, 192, device="cuda", generator=g)
    w = torch.randn(384, 192, device="cuda", generator=g) / 192 ** 0.5
    b = torch.randn(384, device="cuda", generator=g)
    planes = ops.split_h2_linear(w)
    pre, y, aux = torch.empty(rows, 384, device="cuda"), torch.empty(rows, 384, device="cuda"), torch.randn(rows, 384, device="cuda", generator=g)
    t0 = timed(lambda: ops.linear_h2(x, planes, b, out=y))
    t1 = timed(lambda: ops.linear_h2(x, planes, b, out=y, epilogue=L.EPI_GELU, pre=pre))
    t2 = timed(lambda: ops.linear_h2(x, planes, None, out=y, epilogue=L.EPI_GELU_BWD, aux=aux))
    print(f"{rows:6d} x 192 -> 384: plain {t0:6.1f} us   + GELU (pre kept) {t1:6.1f} us   GELU backward (aux read) {t2:6.1f} us", flush=True)
