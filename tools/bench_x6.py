#!/usr/bin/env python3
"""Accuracy and speed of the experimental bf16x6 forward conv vs the fp32-MFMA kernel."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib  # noqa: E402

lib = _lib.lib()
P = C.c_void_p



def timeit(fn, n=10, warm=3):
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


B = 4
for H, Cin, Cout, K, s, p in [(16, 32, 192, 3, 1, 1), (128, 192, 192, 3, 1, 1), (64, 192, 192, 3, 1, 1), (64, 192, 768, 3, 1, 1),
                              (128, 192, 192, 3, 2, 1), (128, 192, 192, 1, 1, 0)]:
    x = torch.randn(B, H, H, Cin, device="cuda") * 3
    w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
    d = ops.conv_desc(x.shape, w.shape, s, p)
    out32 = torch.empty(d.B, d.Ho, d.Wo, d.Cout, device="cuda")
    out6 = torch.empty_like(out32)
    planes = torch.empty(3 * w.numel(), dtype=torch.int16, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ops.split_bf16x3(w, planes)
    ws = ops._scratch(x.device, 1 << 26)
    f6 = lambda: _lib.check(lib.rdo_conv2d_fwd_bf16x6(C.byref(d), x.data_ptr(), planes.data_ptr(), None, None, None, out6.data_ptr(), None, ws.data_ptr(), ws.numel(), st))
    f32 = lambda: ops.conv2d_fwd(x, w, None, s, p, out=out32)
    f6(); f32(); torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, s, p).permute(0, 2, 3, 1)
    e32 = float((out32.double() - ref).abs().max() / ref.abs().max())
    e6 = float((out6.double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * B * d.Ho * d.Wo * Cout * Cin * K * K
    t6, t32 = timeit(f6), timeit(f32)
    print(f"H={H:4d} Cin={Cin} Cout={Cout} K={K} s={s}: fp32 {t32:7.1f} us {fl/t32/1e6:6.1f} TF err {e32:.2e} | bf16x6 {t6:7.1f} us {fl/t6/1e6:6.1f} TF err {e6:.2e} | x{t32/t6:.2f}")
