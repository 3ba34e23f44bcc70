#!/usr/bin/env python3
"""Do two independent big kernels overlap usefully on two streams?  Halo conv (dgrad-shaped) + row weight gradient of the 4 x 128^2 and
4 x 64^2 problems: sequential on one stream vs one on each of two streams (events around the pair; steady state, 20 pairs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rdo-ptq_amd"))
from hipops import ops
for (B, H, C) in [(4, 128, 192), (4, 64, 192), (4, 32, 192)]:
    torch.manual_seed(1)
    x = torch.randn(B, H, H, C, device="cuda"); dy = torch.randn(B, H, H, C, device="cuda") * 0.1
    w = torch.randn(C, 3, 3, C, device="cuda") / (C * 9) ** 0.5
    wpl, xp, dyp = ops.split_h2_conv(w), ops.split_h2(x), ops.split_h2(dy)
    opl = ops.h2_empty((B, H, H, C), "cuda", 16.0)
    slabs = ops.conv2d_wgrad(x, dy, (C, 3, 3, C), 1, 1)
    ada_in = torch.empty(40 << 20, device="cuda")            # a memory-bound stand-in (160 MB copy)
    ada_out = torch.empty_like(ada_in)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    conv = lambda: ops.conv2d_fwd_h2(dyp, (B, H, H, C), tuple(w.shape), wpl, None, 1, 1, out_planes=opl)
    wg = lambda: ops.conv2d_wgrad_h2(xp, (B, H, H, C), dyp, (C, 3, 3, C), 1, 1, slabs=slabs)
    def run(par, n=20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            if par:
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(s1):
                    s1.wait_event(ev); conv(); a = torch.cuda.Event(); a.record()
                with torch.cuda.stream(s2):
                    s2.wait_event(ev); wg(); b = torch.cuda.Event(); b.record()
                torch.cuda.current_stream().wait_event(a); torch.cuda.current_stream().wait_event(b)
            else:
                conv(); wg()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    run(False, 5); run(True, 5)
    print(f"B={B} H={H}: sequential {run(False):6.1f} us per pair | two streams {run(True):6.1f} us per pair", flush=True)
