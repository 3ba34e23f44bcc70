#!/usr/bin/env python3
"""Run one bf16x6 forward-conv shape a few times (for rocprofv3 --pmc passes).  usage: bench_one_x6.py H Cin Cout K stride pad [B] [reps]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
from hipops import ops, _lib  # noqa: E402

lib = _lib.lib()
H, Cin, Cout, K, s, p = (int(v) for v in sys.argv[1:7])
B = int(sys.argv[7]) if len(sys.argv) > 7 else 4
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
x = torch.randn(B, H, H, Cin, device="cuda")
w = torch.randn(Cout, K, K, Cin, device="cuda") / (Cin * K * K) ** 0.5
d = ops.conv_desc(x.shape, w.shape, s, p)
out = torch.empty(d.B, d.Ho, d.Wo, d.Cout, device="cuda")
planes = ops.split_bf16x3(w)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = ops._scratch(x.device, 1 << 26)
for _ in range(reps):
    _lib.check(lib.rdo_conv2d_fwd_bf16x6(C.byref(d), x.data_ptr(), planes.data_ptr(), None, None, None, out.data_ptr(), None,
                                         ws.data_ptr(), ws.numel(), st))
torch.cuda.synchronize()
