#!/usr/bin/env python3
"""Would parallel graph branches pay?  The weight gradient of a block's second conv is a leaf of the backward pass: it can run beside the dgrad
of that conv and the weight gradient of the first conv.  This probe times the three kernels of a ResidualBlock unit back to back on one
stream against the same three with the leaf on a second stream (fork / join by events), at 128^2, 64^2 and 32^2 (batch 4, 192 channels).
Round 6: measured, not built -- see DESIGN.md section 3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import _lib as L  # noqa: E402
from hipops import ops  # noqa: E402

side = torch.cuda.Stream()
for H in (128, 64, 32):
    B, C = 4, 192
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, C, device="cuda", generator=g)
    dy = torch.randn(B, H, H, C, device="cuda", generator=g) * 0.1
    w = torch.randn(C, 3, 3, C, device="cuda", generator=g) * 0.05
    xp, dyp = ops.split_h2(x), ops.split_h2(dy)
    wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    wdp = ops.split_h2_conv(wd)
    ns = ops.wgrad_nsplit(tuple(x.shape), tuple(w.shape), 1, 1)
    s1, s2 = (torch.empty((ns,) + tuple(w.shape), device="cuda") for _ in range(2))
    dh = ops.h2_empty(x.shape, x.device, 4.0)
    out = torch.empty_like(x)

    def wg(sl):
        ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, tuple(w.shape), 1, 1, slabs=sl)

    def dg():
        ops.conv2d_fwd_h2(dyp, tuple(x.shape), tuple(wd.shape), wdp, None, 1, 1, out=out)

    def seq():
        wg(s1); dg(); wg(s2)

    def par():
        cur = torch.cuda.current_stream()
        e0 = torch.cuda.Event(); e0.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(e0)
            wg(s1)
            e1 = torch.cuda.Event(); e1.record(side)
        dg(); wg(s2)
        cur.wait_event(e1)

    def graph_of(fn):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(8):
                fn()
        return gr

    res = {}
    for name, fn in (("sequential", seq), ("two branches", par)):
        gr = graph_of(fn)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            gr.replay()
        b.record()
        torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / 160 * 1e3
    print(f"{H}x{H}: wgrad + dgrad + wgrad  sequential {res['sequential']:.1f} us, leaf on a second branch {res['two branches']:.1f} us", flush=True)
