#!/usr/bin/env python3
"""Where the AdaRound step of every headline unit spends its time.

For each of the 29 units of the bench workload (Cheng2020-anchor N=192, batch 4) the recorded plan's "ada_step" op -- the batched step launch
plus the dgrad-layout launch behind it -- is timed with hipEvents (rdo_plan_profile, median of `--reps` iterations) next to the bytes it has
to move: SURVEY 8d's 13 passes x 4 B x params, the kernel's own count 4 B x numel x (nsplit + 9) per tensor, and what a fused step needs at
least (9 passes).  With a diagnostic library (`make DIAG=1`: the masks are compiled out of the shipped one) every ablation is timed as well:

    1 slab reads off   2 plane writes off   4 rounding term (powf) off   8 Adam state (m, v) off   16 dgrad-layout launch off   32 fp32 wq write off

usage: python tools/ada_step_ablate.py [--images 16] [--reps 15] [--json out.json]
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
import bench  # noqa: E402
from hipops import _lib as L  # noqa: E402
from hipops import ops  # noqa: E402

MASKS = (("complete", 0), ("no slab reads", 1), ("no plane writes", 2), ("no rounding term", 4), ("no Adam state", 8), ("no dgrad layout", 16),
         ("no fp32 wq", 32), ("no planes, no dgrad layout, no wq", 50), ("arithmetic only (all streams but w, alpha off)", 59))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--json", default=None)
    ap.add_argument("--wall-iters", type=int, default=64, help="graph-replayed iterations per mask for the wall-time column (what the loop really pays)")
    ap.add_argument("--w1-min", type=int, default=None, help="tuning key ada_w1_min for this run (slab count from which a tensor is walked one element per thread)")
    a = ap.parse_args()
    if a.w1_min is not None:
        ops.set_tuning("ada_w1_min", a.w1_min)
    from quantization import QuantModel
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    dev = torch.device("cuda", 0)
    diag = "DIAG" in L.lib().rdo_version().decode()
    model = bench.seeded_model(192, 1005, dev)
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=dict(wq, leaf_param=False), is_cheng=True).to(dev).eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    cali = torch.rand(a.images, 3, 256, 256, generator=torch.Generator().manual_seed(1005)).to(dev)
    units = bench.unit_list(qnn)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:4])
    caches = bench.build_caches(qnn, units, cali, bs=min(32, a.images))
    masks = MASKS if diag else MASKS[:1]
    iters = (a.reps + a.wall_iters + 4) * len(masks) + 8
    gi = torch.Generator().manual_seed(7)
    rows = []
    for name, u in units:
        kind, mods = _unit_modules(u)
        cq, cf, co = caches[name]
        idx = torch.stack([torch.randperm(a.images, generator=gi)[:4] for _ in range(iters)])
        e = UnitEngine(kind, mods, cq, cf, co, batch_size=4, iters=iters, weight=0.01, b_range=(20, 2), warmup=0.0, input_prob=0.5, seed=1005,
                       idx_table=idx)
        info = e.plan_a.op_info()
        sel = [i for i, (tag, _, _) in enumerate(info) if tag in ("ada_step", "ada_step_gather")]
        params = sum(op.numel() for op in e.ops.values())
        tensors = {n: dict(numel=op.numel(), nsplit=int(op.slabs.shape[0]), dgrad=op.wd is not None,
                           planes=("h2" if isinstance(op.wq_planes, ops.H2) else ("bf16x3" if op.wq_planes is not None else None)),
                           lin_planes=op.lin_fwd is not None or op.lin_bwd is not None) for n, op in e.ops.items()}
        kbytes = sum(info[i][2] for i in sel)
        row = dict(unit=name, kind=kind, plan=getattr(e, "h2_plan", None), params=params, tensors=tensors,
                   bytes_survey_13_passes=13 * 4 * params, bytes_min_9_passes=9 * 4 * params, bytes_kernel_count=kbytes, us={}, wall_us={})
        e.run(2)
        for label, m in masks:
            if diag:
                L.check(L.lib().rdo_diag_ada_ablate(int(m)), "rdo_diag_ada_ablate")
            ts = []
            for _ in range(a.reps):
                ms = e.plan_a.profile()
                e._done += 1
                ts.append(sum(ms[i] for i in sel) * 1e3)
            row["us"][label] = round(statistics.median(ts), 2)
            # what the loop pays: wall of graph-replayed iterations of the WHOLE unit under this mask (events add ~6 us per op)
            import time
            e.run(4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.run(a.wall_iters)
            torch.cuda.synchronize()
            row["wall_us"][label] = round((time.perf_counter() - t0) / a.wall_iters * 1e6, 2)
        if diag:
            L.check(L.lib().rdo_diag_ada_ablate(0), "rdo_diag_ada_ablate")
        rows.append(row)
        del e
        torch.cuda.empty_cache()
    print(f"library: {L.lib().rdo_version().decode()}; {len(rows)} units, median of {a.reps} event-timed iterations (the op = step launch + dgrad-layout launch)")
    hdr = "| unit | kind | params | slabs (nsplit per tensor) | 13-pass MB | kernel-count MB | " + " | ".join(l for l, _ in masks) + " | GB/s (kernel count) | us at 6.3 TB/s (13-pass) |"
    print(hdr)
    print("|" + "---|" * (hdr.count("|") - 1))
    tot = {l: 0.0 for l, _ in masks}
    for r in rows:
        ns = ",".join(str(t["nsplit"]) for t in r["tensors"].values())
        full = r["us"]["complete"]
        print(f"| {r['unit']} | {r['kind']} | {r['params']} | {ns} | {r['bytes_survey_13_passes'] / 1e6:.1f} | {r['bytes_kernel_count'] / 1e6:.1f} | "
              + " | ".join(f"{r['us'][l]:.1f}" for l, _ in masks)
              + f" | {r['bytes_kernel_count'] / full / 1e3:.0f} | {r['bytes_survey_13_passes'] / 6.3e6:.1f} |")
        for l, _ in masks:
            tot[l] += r["us"][l]
    print("| **sum** | | " + f"{sum(r['params'] for r in rows)} | | {sum(r['bytes_survey_13_passes'] for r in rows) / 1e6:.0f} | "
          f"{sum(r['bytes_kernel_count'] for r in rows) / 1e6:.0f} | " + " | ".join(f"{tot[l]:.0f}" for l, _ in masks) + " | | "
          f"{sum(r['bytes_survey_13_passes'] for r in rows) / 6.3e6:.0f} |")
    print()
    print("wall of one graph-replayed iteration of the whole unit (us) and what each ablation takes off it:")
    print("| unit | complete | " + " | ".join(l for l, _ in masks[1:]) + " |")
    print("|---|---|" + "---|" * (len(masks) - 1))
    wt = {l: 0.0 for l, _ in masks}
    for r in rows:
        print(f"| {r['unit']} | {r['wall_us']['complete']:.1f} | " + " | ".join(f"{r['wall_us'][l] - r['wall_us']['complete']:+.1f}" for l, _ in masks[1:]) + " |")
        for l, _ in masks:
            wt[l] += r["wall_us"][l]
    print(f"| **sum** | {wt['complete']:.0f} | " + " | ".join(f"{wt[l] - wt['complete']:+.0f}" for l, _ in masks[1:]) + " |")
    if a.json:
        json.dump(dict(library=L.lib().rdo_version().decode(), reps=a.reps, units=rows), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
