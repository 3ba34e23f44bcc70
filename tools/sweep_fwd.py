#!/usr/bin/env python3
"""Sweep (tile, ksplit) of the forward conv kernel over the workload's shapes and compare with the cost model's pick."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib  # noqa: E402

lib = _lib.lib()
lib.rdo_debug_force_fwd_choice.restype = C.c_char_p
lib.rdo_debug_force_fwd_choice.argtypes = [C.c_int, C.c_int]


def timeit(fn, n=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(128, 192, 192, 3, 1, 1), (64, 192, 192, 3, 1, 1), (32, 192, 192, 3, 1, 1), (16, 192, 192, 3, 1, 1),
          (8, 192, 192, 3, 1, 1), (4, 192, 192, 3, 1, 1), (128, 192, 192, 3, 2, 1), (64, 192, 192, 3, 2, 1),
          (128, 192, 192, 1, 1, 0), (64, 192, 192, 1, 1, 0), (32, 192, 192, 1, 1, 0), (64, 192, 768, 3, 1, 1),
          (32, 192, 768, 3, 1, 1), (16, 192, 768, 3, 1, 1), (16, 192, 384, 5, 1, 2), (8, 288, 1152, 3, 1, 1),
          (16, 768, 640, 1, 1, 0), (4, 192, 768, 3, 1, 1)]
for H, Cin, Cout, K, s, p in shapes:
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
    Ho = (H + 2 * p - K) // s + 1
    out = torch.empty(B, Ho, Ho, Cout, device="cuda")
    fl = 2.0 * B * Ho * Ho * Cout * Cin * K * K
    f = lambda: ops.conv2d_fwd(x, w, None, s, p, out=out)
    ops._scratch(x.device, 1 << 26)
    lib.rdo_debug_force_fwd_choice(-1, -1)
    t_model = timeit(f)
    res = []
    for tile in range(5):
        for ks in (1, 2, 3, 4, 6, 9, 12, 18):
            if ks > 1 and (K * K * ((Cin + 31) // 32)) // ks < 4:
                continue
            lib.rdo_debug_force_fwd_choice(tile, ks)
            try:
                res.append((timeit(f, n=4, warm=1), tile, ks))
            except RuntimeError:
                pass
    res.sort()
    best = res[0]
    print(f"H={H:4d} Cin={Cin:4d} Cout={Cout:4d} K={K} s={s} M={B*Ho*Ho:6d}: model {t_model:7.1f} us ({fl/t_model/1e6:6.1f} TF) | "
          f"best {best[0]:7.1f} us tile{best[1]} ks{best[2]} ({fl/best[0]/1e6:6.1f} TF) | " +
          " ".join(f"t{t}k{k}:{us:.0f}" for us, t, k in res[:5]))
lib.rdo_debug_force_fwd_choice(-1, -1)
