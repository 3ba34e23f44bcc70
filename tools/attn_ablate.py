#!/usr/bin/env python3
"""Where the window-attention kernels spend their time (needs a `make DIAG=1` library): forward and backward with the output stores,
the products, the tile loads or the softmax rows switched off (RDO_ATTN_DIAG bits 1 / 2 / 4 / 8), on the Lu2022 shapes.
usage: python tools/attn_ablate.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops  # noqa: E402


def timed(fn, n=20):
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


names = {0: "full", 1: "-stores", 2: "-mfma", 4: "-loads", 8: "-softmax", 10: "-mfma-softmax", 5: "-loads-stores", 15: "nothing"}
g = torch.Generator(device="cuda").manual_seed(0)
for B, H, C, heads, shift in [(4, 128, 192, 4, 0), (4, 128, 192, 4, 4), (4, 64, 192, 8, 4), (4, 32, 192, 8, 4), (4, 16, 192, 16, 4)]:
    d = ops.attn_desc(B, H, H, C, heads, 8, shift)
    qkv = torch.randn(B, H, H, 3 * C, device="cuda", generator=g)
    bias = torch.randn(heads, 64, 64, device="cuda", generator=g)
    dout = torch.randn(B, H, H, C, device="cuda", generator=g)
    out = torch.empty(B, H, H, C, device="cuda")
    dqkv = torch.empty_like(qkv)
    for what, fn in (("fwd", lambda: ops.window_attention(d, qkv, bias, out=out)), ("bwd", lambda: ops.window_attention_bwd(d, qkv, bias, dout, dqkv=dqkv))):
        line = []
        for k in (0, 1, 2, 4, 8, 10, 5, 15):
            os.environ["RDO_ATTN_DIAG"] = str(k)
            line.append(f"{names[k]} {timed(fn):6.1f}")
        os.environ["RDO_ATTN_DIAG"] = "0"
        print(f"{H:3d}^2 heads {heads:2d} shift {shift} {what}: " + " | ".join(line), flush=True)
