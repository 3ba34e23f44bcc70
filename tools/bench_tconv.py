#!/usr/bin/env python3
"""Transposed conv with and without zero insertion: the `deconv(192, 192, 5, 2)` of the Minnen2018 / Lu2022
decoders at 4 x 64^2 -> 128^2, forward through ops.conv_transpose2d and a whole calibration iteration of that layer as a unit
(RDO_TCONV_PHASE=0 forces the old zero-insertion + dense-conv form).  usage: python tools/bench_tconv.py"""
import os
import subprocess
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))


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


def main():
    from hipops import ops
    from quantization.engine import UnitEngine
    from quantization.quant_layer import QuantModule
    phase = os.environ.get("RDO_TCONV_PHASE", "1") != "0"
    torch.manual_seed(0)
    B, H, C, K, s, p, op = 4, 64, 192, 5, 2, 2, 1
    m = nn.ConvTranspose2d(C, C, K, stride=s, padding=p, output_padding=op).cuda()
    x = torch.randn(B, H, H, C, device="cuda")
    rows = m.weight.detach().permute(1, 2, 3, 0).contiguous()
    if phase:
        fwd = lambda: ops.conv_transpose2d(x, rows, m.bias.detach(), s, p, op)
    else:
        q = K - 1 - p
        Hu = (H - 1) * s + 1 + 2 * q + op
        wf = rows.flip(1, 2).contiguous()
        fwd = lambda: ops.conv2d_fwd(ops.zero_insert(x, s, q, q, Hu, Hu), wf, m.bias.detach(), 1, 0)
    t_fwd = timeit(fwd)
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qm = QuantModule(m, WQ, dict(WQ, leaf_param=False)).cuda()
    n = 16
    cq = torch.randn(n, H, H, C, device="cuda")
    cf = cq + 0.01 * torch.randn_like(cq)
    with torch.no_grad():
        co = torch.cat([m(cf[i:i + 4].permute(0, 3, 1, 2)).permute(0, 2, 3, 1) for i in range(0, n, 4)]).contiguous()
    iters = 60
    eng = UnitEngine("layer", {"layer": qm}, cq, cf, co, batch_size=B, iters=iters, warmup=0.2, input_prob=0.5, seed=1)
    eng.run(10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.run(40)
    e1.record()
    torch.cuda.synchronize()
    t_it = e0.elapsed_time(e1) / 40 * 1e3
    gf = 2.0 * B * H * H * C * C * K * K / 1e9
    print(f"deconv(192,192,5,2) 4x64^2->128^2  {'phase weights + pixel shuffle' if phase else 'zero insertion + dense conv'}: "
          f"forward {t_fwd:7.1f} us ({gf / t_fwd * 1e3:6.1f} TFLOP/s of the {gf:.1f} algorithmic GFLOP) | calibration iteration {t_it:7.1f} us "
          f"({eng.plan_a.num_ops} launches)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "both":
        for v in ("0", "1"):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RDO_TCONV_PHASE=v), check=True)
    else:
        main()
