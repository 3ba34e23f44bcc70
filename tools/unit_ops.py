#!/usr/bin/env python3
"""Per-op event times of one iteration of chosen units of the bench workload (Cheng2020-anchor N=192, batch 4, 8 images), op by op in
plan order: python tools/unit_ops.py g_a.1 g_s.5 ...   (median of 7 profiled iterations)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
import bench  # noqa: E402
from quantization import QuantModel  # noqa: E402
from quantization.engine import UnitEngine  # noqa: E402
from quantization.recon import _unit_modules  # noqa: E402

want = sys.argv[1:] or ["g_a.1"]
dev = torch.device("cuda", 0)
model = bench.seeded_model(192, 1005, dev)
wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=dict(wq, leaf_param=False), is_cheng=True).to(dev).eval()
qnn.set_first_last_layer_to_8bit()
qnn.disable_network_output_quantization()
cali = torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
units = [(n, u) for n, u in bench.unit_list(qnn) if n in want]
caches = bench.build_caches(qnn, units, cali, bs=8)
for name, u in units:
    kind, mods = _unit_modules(u)
    cq, cf, co = caches[name]
    iters = 40
    idx = torch.stack([torch.randperm(8)[:4] for _ in range(iters)])
    e = UnitEngine(kind, mods, cq, cf, co, batch_size=4, iters=iters, idx_table=idx, seed=1)
    e.run(8)
    info = e.plan_a.op_info()
    runs = []
    for _ in range(7):
        runs.append(e.plan_a.profile())
        e._done += 1
    med = [sorted(r[i] for r in runs)[3] for i in range(len(info))]
    print(f"== {name} ({kind}, h2 plan {e.h2_plan}): {sum(med) * 1e3:.0f} us per iteration")
    for (tag, fl, by), m in zip(info, med):
        extra = f"{fl / (m * 1e-3) / 1e12:7.1f} TF" if fl else (f"{by / (m * 1e-3) / 1e9:7.0f} GB/s" if by else "")
        print(f"   {tag:28s} {m * 1e3:8.1f} us  {extra}")
