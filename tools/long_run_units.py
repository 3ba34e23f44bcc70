#!/usr/bin/env python3
"""The reference's full schedule -- 20 000 iterations (main2.py:49 default) -- on the two most expensive unit kinds of Cheng2020-anchor
N=192 at batch 4: g_a.1 (ResidualBlock at 128^2) and g_s.5 (ResidualBlockUpsample 64^2 -> 128^2), one `run()` call each (hipGraph
replays, no host work in between).  Prints wall time per iteration in windows of 2 000, the loss trajectory and how far the soft
rounding targets have converged.   usage: python tools/long_run_units.py [--iters 20000] [--images 64]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
import lic  # noqa: E402
from quantization.engine import UnitEngine  # noqa: E402
from quantization.quant_block import QuantRB, QuantRBU  # noqa: E402
from quantization.recon import _unit_modules  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20000)
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--units", default="rb,rbu", help="comma list out of rb,rbu")
ap.add_argument("--tune", default="", help="comma list of key=value for rdo_set_tuning (kernel ablation / variant switches)")
a = ap.parse_args()
if a.tune:
    from hipops import ops as _ops
    for kv in a.tune.split(","):
        k, v = kv.split("=")
        _ops.set_tuning(k, int(v))
N = 192
WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
for name, mk, qcls, shape in (("g_a.1 RB @128^2", lambda: lic.ResidualBlock(N, N), QuantRB, (a.images, 128, 128, N)),
                              ("g_s.5 RBU 64^2->128^2", lambda: lic.ResidualBlockUpsample(N, N, 2), QuantRBU, (a.images, 64, 64, N))):
    if ("rbu" if qcls is QuantRBU else "rb") not in a.units.split(","):
        continue
    torch.manual_seed(1)
    blk = mk().cuda()
    for m in blk.modules():
        if isinstance(m, lic.GDN):
            c = m.gamma.shape[0]
            with torch.no_grad():
                m.gamma.copy_(torch.sqrt(0.1 * torch.eye(c) + 0.002 * torch.rand(c, c) + 2.0 ** -36))
    unit = qcls(blk, WQ, dict(WQ, leaf_param=False)).cuda()
    kind, mods = _unit_modules(unit)
    cq = torch.randn(*shape, device="cuda")
    cf = cq + 0.01 * torch.randn_like(cq)
    with torch.no_grad():
        co = torch.cat([blk(cf[i:i + 8].permute(0, 3, 1, 2)).permute(0, 2, 3, 1) for i in range(0, shape[0], 8)]).contiguous()
    eng = UnitEngine(kind, mods, cq, cf, co, batch_size=4, iters=a.iters, weight=0.01, b_range=(20, 2), warmup=0.2, input_prob=0.5, seed=1)
    eng.run(20)
    torch.cuda.synchronize()
    win, done, times = max(1, (a.iters - 20) // 10), 20, []
    t0 = time.perf_counter()
    while done < a.iters:
        n = min(win, a.iters - done)
        t1 = time.perf_counter()
        eng.run(n)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t1) / n * 1e3)
        done += n
    dt = time.perf_counter() - t0
    total, rt, rd = eng.logs()
    h = torch.cat([torch.clamp(torch.sigmoid(eng.alpha_of(k)) * 1.2 - 0.1, 0, 1).reshape(-1) for k in eng.ops])
    print(f"{name}: h2_plan={eng.h2_plan}; {a.iters - 20} iterations in {dt:.1f} s = {dt / (a.iters - 20) * 1e3:.3f} ms/iteration "
          f"(windows of {win}: {min(times):.3f} .. {max(times):.3f}); loss {float(total[0]):.4e} -> {float(total[a.iters // 5 - 1]):.4e} (end of warm-up) "
          f"-> {float(total[-1]):.4e}; round term last {float(rd[-1]):.4e}; soft targets in {{0,1}}: {float(((h < 1e-3) | (h > 1 - 1e-3)).float().mean()):.4f}")
    del eng, cq, cf, co
