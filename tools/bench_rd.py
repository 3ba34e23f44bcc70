#!/usr/bin/env python3
"""Time one calibration iteration of the opt-in R + lambda*D task-loss mode (`loss_mode='rd'`, main2.py:125-137 through
layer_opt.py:258-274) on a full-size Cheng2020-anchor (N=192): the captured-graph iteration against the host-driven one
(RDO_RD_GRAPH=0), next to the default MSE-only iteration of the same unit.

    python tools/bench_rd.py [--unit g_s.2] [--batch 8] [--size 256] [--iters 60]

Prints one line per path: ms per iteration (HIP events around `iters` iterations after the engine's own warm-up)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rdo-ptq_amd")):
    sys.path.insert(0, p)
import torch

import lic
from quantization import QuantModel
from quantization.engine import UnitEngine
from quantization.recon import _unit_modules

WQ = dict(n_bits=8, channel_wise=True, scale_method="max")
AQ = dict(n_bits=8, channel_wise=True, scale_method="max", leaf_param=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="g_s.2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--N", type=int, default=192)
    a = ap.parse_args()
    torch.manual_seed(0)
    cali = torch.rand(a.images, 3, a.size, a.size).cuda()
    for mode in ("mse", "rd-graph", "rd-host"):
        torch.manual_seed(1)
        qnn = QuantModel(lic.Cheng2020Anchor(N=a.N).eval().cuda(), WQ, AQ, is_cheng=True).cuda().eval()
        qnn.set_quant_state(False, False)
        unit = qnn.model
        for part in a.unit.split("."):
            unit = unit[int(part)] if part.isdigit() else getattr(unit, part)
        store = {}
        h = unit.register_forward_hook(lambda m, i, o: store.update(inp=i[0].detach().clone(), out=o.detach().clone()))
        with torch.no_grad():
            for i in range(0, a.images, a.batch):
                qnn(cali[i:i + a.batch])
                store.setdefault("inps", []).append(store["inp"])
                store.setdefault("outs", []).append(store["out"])
        h.remove()
        nh = lambda ts: torch.cat(ts).permute(0, 2, 3, 1).contiguous()
        inp, out = nh(store["inps"]), nh(store["outs"])
        os.environ["RDO_RD_GRAPH"] = "0" if mode == "rd-host" else "1"
        k, mods = _unit_modules(unit)
        warm = 8
        rd = None if mode == "mse" else dict(model=qnn, unit=unit, cali=cali, lmbda=0.0483)
        eng = UnitEngine(k, mods, inp, inp, out, batch_size=a.batch, iters=warm + a.iters, seed=1, rd=rd)
        eng.run(warm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.run(a.iters)
        e1.record()
        torch.cuda.synchronize()
        path = eng.rd_path if rd is not None else "graph"
        print(f"{a.unit} batch {a.batch} {a.size}x{a.size}  {mode:9s} ({path}): {e0.elapsed_time(e1) / a.iters:.3f} ms/iteration", flush=True)
        del eng, qnn


if __name__ == "__main__":
    main()
