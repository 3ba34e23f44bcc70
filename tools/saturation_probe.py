#!/usr/bin/env python3
"""would a "low weight plane is zero" skip pay?  With every soft rounding target h(alpha) of a K stage's weights
saturated in {0, 1} the soft weight w~ / delta is an integer <= 255 -- ONE fp16 plane after the per-channel delta is taken out -- and
the (x_hi, w_lo) product of that stage could be skipped (3 -> 2 MFMA products).  This tool MEASURES how often that is the case over the
reference's real schedule (20 000 iterations, warm-up 0.2, b 20 -> 2) for the 3x3 192 -> 192 convs of the four 128^2 units of
Cheng2020-anchor N=192: per 1000 iterations the share of saturated weights and the share of K stages (32 input channels x 1 tap x all
output channels of the tile, the granularity of conv_fwd_h2k.hip's K loop) in which EVERY weight is saturated.

    python tools/saturation_probe.py [--iters 20000] [--images 32] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--every", type=int, default=1000)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import bench
    from quantization import QuantModel
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    dev = torch.device("cuda:0")
    model = bench.seeded_model(192, 1005, dev)
    g = torch.Generator().manual_seed(1005)
    with torch.no_grad():      # variance-preserving conv weights, as tools/full_schedule.py: the signal reaches every unit
        for name, p_ in model.named_parameters():
            if p_.dim() == 4 and "entropy_bottleneck" not in name:
                p_.copy_(((torch.rand(p_.shape, generator=g) - 0.5) * 2 * (3.0 / p_[0].numel()) ** 0.5).to(dev))
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=dict(wq, leaf_param=False), is_cheng=True).to(dev).eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    cali = torch.rand(a.images, 3, 256, 256, generator=g).to(dev)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:4])
    units = [(n, u) for n, u in bench.unit_list(qnn) if n in ("g_a.0", "g_a.1", "g_s.5", "g_s.6")]
    allu = bench.unit_list(qnn)
    caches = bench.build_caches(qnn, allu[:max(i for i, (n, _) in enumerate(allu) if n in ("g_a.0", "g_a.1", "g_s.5", "g_s.6")) + 1], cali, bs=16)
    out = {"iters": a.iters, "units": {}}
    gi = torch.Generator().manual_seed(77)
    for name, u in units:
        kind, mods = _unit_modules(u)
        cq, cf, co = caches[name]
        idx = torch.stack([torch.randperm(a.images, generator=gi)[:4] for _ in range(a.iters)])
        e = UnitEngine(kind, mods, cq, cf, co, batch_size=4, iters=a.iters, weight=0.01, b_range=(20, 2), warmup=0.2, input_prob=0.5,
                       seed=1005, idx_table=idx)
        rows = []
        for done in range(0, a.iters, a.every):
            e.run(min(a.every, a.iters - done))
            torch.cuda.synchronize()
            rec = {"iter": done + min(a.every, a.iters - done)}
            for oname, op in e.ops.items():
                if op.is_gdn or op.K != 3 or op.w4[3] % 32:
                    continue
                h = torch.clamp(torch.sigmoid(op.alpha) * 1.2 - 0.1, 0, 1)           # [Cout][3][3][Cin]
                sat = (h == 0) | (h == 1)
                co_, kh, kw, ci = sat.shape
                stage = sat.reshape(co_, kh * kw, ci // 32, 32).permute(1, 2, 0, 3).reshape(kh * kw * (ci // 32), -1).all(1)
                # ... and at the granularity of one wave's 16-channel A fragment rows x the stage (a finer skip inside a wave)
                frag = sat.reshape(co_ // 16, 16, kh * kw, ci // 32, 32).permute(0, 2, 3, 1, 4).reshape(-1, 16 * 32).all(1)
                rec[oname] = {"weights": round(float(sat.float().mean()), 4), "stages": round(float(stage.float().mean()), 4),
                              "wave_fragments": round(float(frag.float().mean()), 4)}
            rows.append(rec)
            print(name, rec, flush=True)
        out["units"][name] = rows
        del e
        torch.cuda.empty_cache()
    # stage-visits that qualify over the whole schedule (each sample stands for `every` iterations)
    tot = {"stages": 0.0, "wave_fragments": 0.0, "weights": 0.0, "n": 0}
    for rows in out["units"].values():
        for rec in rows:
            for k, v in rec.items():
                if isinstance(v, dict):
                    for kk in ("stages", "wave_fragments", "weights"):
                        tot[kk] += v[kk]
                    tot["n"] += 1
    out["schedule_share"] = {k: round(tot[k] / max(tot["n"], 1), 4) for k in ("weights", "stages", "wave_fragments")}
    print("share of the schedule's stage-visits with every weight saturated:", out["schedule_share"], flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
