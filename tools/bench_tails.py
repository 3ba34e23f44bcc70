"""Per-kernel timings of the fused tail / P3 producer kernels at the 4 x 128^2 x 192 shape of the P3 units (HIP events over 50
back-to-back launches each).  Usage: python tools/bench_tails.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = "cuda"
    B, H, C, n = 4, 128, 192, 16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    cq, cf, tgt = r(n, H, H, C), r(n, H, H, C), r(n, H, H, C)
    idx = torch.randint(0, n, (4, B), dtype=torch.int32, device=dev)
    it = torch.zeros(1, dtype=torch.int32, device=dev)
    x, nrm, acc, gg = r(B, H, H, C), r(B, H, H, C).abs() + 1, r(B, H, H, C), r(B, H, H, C)
    out, gout, t = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    pl, pl2 = ops.h2_empty(x.shape, dev, 16.0), ops.h2_empty(x.shape, dev, 16.0)
    log = torch.zeros(4, 32, device=dev)
    small = r(B, H // 2, H // 2, 4 * C)
    spl = ops.h2_empty(small.shape, dev, 16.0)
    mb = x.numel() / 1e6
    rows = [
        ("gather_qdrop (fp32)", lambda: ops.gather_qdrop(cq, cf, idx, it, B, 0.5, 1, out), 12),
        ("gather_qdrop_h2 (fp32 + planes)", lambda: ops.gather_qdrop_h2(cq, cf, idx, it, B, 0.5, 1, out, pl), 18),
        ("gather_qdrop_h2 (planes)", lambda: ops.gather_qdrop_h2(cq, cf, idx, it, B, 0.5, 1, None, pl), 14),
        ("split_h2", lambda: ops.split_h2(x, pl), 10),
        ("loss_act_bwd (res, dpre fp32)", lambda: ops.loss_act_bwd(x, acc, tgt, idx, it, 2.0, 1, log, dpre=t), 16),
        ("loss_act_bwd (res, dpre planes)", lambda: ops.loss_act_bwd(x, acc, tgt, idx, it, 2.0, 1, log, dpre_planes=pl), 18),
        ("loss_act_bwd (res planes, dpre planes)",
         lambda: ops.loss_act_bwd(x, None, tgt, idx, it, 2.0, 1, log, dpre_planes=pl, residual_planes=pl2), 20),
        ("loss_gdn_bwd (res, gout, t fp32)", lambda: ops.loss_gdn_bwd(x, nrm, acc, tgt, idx, it, 2.0, False, log, gout, t=t), 24),
        ("loss_gdn_bwd (res, gout, t fp32 + planes)",
         lambda: ops.loss_gdn_bwd(x, nrm, acc, tgt, idx, it, 2.0, False, log, gout, t=t, t_planes=pl), 30),
        ("gdn_bwd_dx (fp32)", lambda: ops.gdn_bwd_dx_h2(gg, x, nrm, acc, False, dx=out), 20),
        ("gdn_bwd_dx (planes)", lambda: ops.gdn_bwd_dx_h2(gg, x, nrm, acc, False, dx_planes=pl), 22),
        ("pixel_shuffle (fp32)", lambda: ops.pixel_shuffle_h2(small, out=out), 8),
        ("pixel_shuffle (fp32 + planes)", lambda: ops.pixel_shuffle_h2(small, out=out, out_planes=pl), 14),
        ("pixel_shuffle (planes)", lambda: ops.pixel_shuffle_h2(small, out_planes=pl), 10),
        ("pixel_unshuffle (fp32)", lambda: ops.pixel_unshuffle2(x, out=small), 8),
        ("pixel_unshuffle (planes)", lambda: ops.pixel_unshuffle2(x, out_planes=spl), 10),
    ]
    print(f"{'kernel':46s} {'us':>8s} {'TB/s':>8s}")
    for name, fn, bytes_per_el in rows:
        us = timeit(fn)
        print(f"{name:46s} {us:8.1f} {bytes_per_el * mb / us:8.2f}")
    for grid in (256, 512, 1024, 2048, 4096, 8192):
        ops.set_tuning("tail_grid", grid)
        print(f"tail_grid {grid:5d}: " + "  ".join(f"{timeit(fn):6.1f}" for name, fn, _ in rows if name.startswith("loss_")))


if __name__ == "__main__":
    main()
