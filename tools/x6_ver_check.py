#!/usr/bin/env python3
"""A/B of the bf16x6 forward variants (RDO_X6_VER in the environment): correctness against an fp64 reference on the first and
last image and event timing.  usage: RDO_X6_VER=6 python tools/x6_ver_check.py"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib  # noqa: E402

lib = _lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ops._scratch(torch.device("cuda"), 1 << 27)
SHAPES = [(4, 128, 192, 192), (4, 64, 192, 768), (4, 64, 192, 192), (4, 32, 192, 192), (4, 16, 192, 192), (2, 256, 64, 192)]
if os.environ.get("X6_SLOPE"):
    SHAPES = [(4, 128, 96, 192), (4, 128, 192, 192), (4, 128, 384, 192), (4, 128, 768, 192)]
for (B, H, Cin, Cout) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5
    d = ops.conv_desc(x.shape, w.shape, 1, 1)
    out = torch.empty(d.B, d.Ho, d.Wo, d.Cout, device="cuda")
    planes = ops.split_bf16x3(w)

    def run():
        _lib.check(lib.rdo_conv2d_fwd_bf16x6(C.byref(d), x.data_ptr(), planes.data_ptr(), None, None, None, out.data_ptr(), None,
                                             ws.data_ptr(), ws.numel(), st), "rdo_conv2d_fwd_bf16x6")
    run()
    torch.cuda.synchronize()
    errs = []
    for sl in (slice(0, 1), slice(B - 1, B)):
        ref = F.conv2d(x[sl].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
        errs.append(float((out[sl].double() - ref).abs().max() / ref.abs().max()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * B * H * H * Cin * Cout * 9 / 1e9
    print(f"ver={os.environ.get('RDO_X6_VER', 'default')} B={B} H={H} {Cin}->{Cout}: {us:7.1f} us  {gf / us * 1e-3 * 1e3:6.1f} TF  "
          f"err {errs[0]:.2e} {errs[1]:.2e}")
