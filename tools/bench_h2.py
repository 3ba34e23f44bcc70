#!/usr/bin/env python3
"""Event timing of the plane-input conv / weight gradient (rdo_conv2d_fwd_h2 / rdo_conv2d_wgrad_h2: fp16 two-way planes, three MFMA
products) against the fp32-input bf16x6 kernels (six products) on the Cheng2020 N=192 shapes, with the error of both against fp64 on
the first image.
usage: python tools/bench_h2.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402

SHAPES = [(4, 128, 192, 192, 3, 1, 1), (4, 64, 192, 768, 3, 1, 1), (4, 64, 192, 192, 3, 1, 1), (4, 128, 192, 192, 3, 2, 1),
          (4, 32, 192, 192, 3, 1, 1), (4, 128, 192, 192, 1, 1, 0), (4, 32, 192, 768, 3, 1, 1)]


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


for (B, H, Cin, Cout, K, s, p) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda")
    if not ops.conv_h2_supported(tuple(x.shape), tuple(w.shape), s, p):
        print(f"B={B} H={H} {Cin}->{Cout} k{K} s{s}: not on the plane path")
        continue
    wpl6, wpl, xp = ops.split_bf16x3(w), ops.split_h2_conv(w), ops.split_h2(x)
    out6 = ops.conv2d_fwd(x, w, b, s, p, wplanes=wpl6)
    out = torch.empty_like(out6)
    opl = ops.h2_empty(out.shape, "cuda", ops.pow2_scale(out6.abs().max()))
    gf = 2.0 * out.numel() * Cin * K * K / 1e9
    t_old = timeit(lambda: ops.conv2d_fwd(x, w, b, s, p, out=out6, wplanes=wpl6))
    t_h2 = timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out=out))
    t_h2b = timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out=out, out_planes=opl))
    t_h2p = timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out_planes=opl))
    ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), b.double().cpu(), stride=s, padding=p).permute(0, 2, 3, 1).cuda()
    sc = float(ref.abs().max())
    e_h2, e_6 = float((out[:1].double() - ref).abs().max()) / sc, float((out6[:1].double() - ref).abs().max()) / sc
    abl = []
    if os.environ.get("H2_AB"):              # same-process A/B of a tuning key: H2_AB=h2_stagger
        key = os.environ["H2_AB"]
        for v in (0, 1, 0, 1):
            ops.set_tuning(key, v)
            abl.append((f"{key}={v}", timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out=out), 60)))
    if os.environ.get("H2_ABLATE"):          # needs a `make DIAG=1` library
        for m in (3, 4, 8, 12, 15, 16, 31):
            ops.set_tuning("x6p_ablate", m)
            abl.append((f"abl{m}", timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out_planes=opl))))
        ops.set_tuning("x6p_ablate", 0)
    print(f"B={B} H={H} {Cin}->{Cout} k{K} s{s}: fp32-in x6 {t_old:7.1f} us ({gf / t_old * 1e3:6.1f} TF, err {e_6:.1e}) | h2 out {t_h2:7.1f} us ({gf / t_h2 * 1e3:6.1f} TF, err {e_h2:.1e})"
          f" | out+planes {t_h2b:7.1f} | planes only {t_h2p:7.1f}" + "".join(f" | {m}: {t:6.1f}" for m, t in abl))


print("---- weight gradient: fp32-input x6 (eight-wave) vs plane input")
for (B, H, Cin, Cout, K, s, p) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda")
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda") * 0.1
    wshape = (Cout, K, K, Cin)
    if not ops.wgrad_h2_supported(tuple(x.shape), wshape, s, p):
        print(f"B={B} H={H} {Cin}->{Cout} k{K} s{s}: not on the plane path")
        continue
    slabs = ops.conv2d_wgrad(x, dy, wshape, s, p)
    xp, dyp = ops.split_h2(x), ops.split_h2(dy)
    gf = 2.0 * dy.numel() * Cin * K * K / 1e9
    t_old = timeit(lambda: ops.conv2d_wgrad(x, dy, wshape, s, p, slabs=slabs))
    t_new = timeit(lambda: ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, wshape, s, p, slabs=slabs))
    print(f"B={B} H={H} {Cin}->{Cout} k{K} s{s}: nsplit {slabs.shape[0]:3d}  fp32-in {t_old:7.1f} us ({gf / t_old * 1e3:6.1f} TF) | h2 {t_new:7.1f} us ({gf / t_new * 1e3:6.1f} TF)")
