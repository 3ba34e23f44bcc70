#!/usr/bin/env python3
"""The reference's REAL schedule, timed (wall time of `recon_model`, main2.py:227-253): Cheng2020-anchor N=192, 256
synthetic 256x256 calibration crops, 29 units x 20 000 iterations (main2.py:54 `--iters_w`), batch 4, through the public
`layer_reconstruction` / `block_reconstruction` API on one MI355X.  Reports the total wall, its split per unit (asymmetric cache
building / plan recording + graph capture / hot loop), calibration images/s by SURVEY 8d's definition (units x B x iters / wall),
the share of soft rounding targets that ended in {0, 1} per unit, and W8 / W8A8 quality next to FP32 and nearest rounding.

    python tools/full_schedule.py [--iters 20000] [--images 256] [--json out.json]"""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
sys.path.insert(0, ROOT)


def schedule_roofline(engines, loop_s, iters, top=8):
    """Algorithmic-FLOP accounting of a finished schedule (missing 2 / next 2a): every recorded op of every unit's
    plans carries its algorithmic FLOPs (2 M N K of its GEMM: unit forward, input and weight gradients, the full-precision tail's
    forward and input gradient, the attention products) and bytes; ONE more eager iteration of every unit with a hipEvent pair around
    every op (rdo_plan_profile) gives the time per kernel family.  -> per-unit GFLOP per iteration against the unit's loop time, the
    schedule's achieved TFLOP/s, and the kernel families by time, each against the peak of ITS pipe or -- below the ridge -- against bytes."""
    import bench
    per_tag, units = {}, []
    for (name, e), t_loop in zip(engines, loop_s):
        e.dp_drain()
        e._it2.fill_(max(0, e.iters - 1))      # the profiled iteration stays inside the unit's index / schedule tables
        fl_u = ms_u = 0.0
        for p in (e.plan_a, getattr(e, "plan_a2", None), getattr(e, "plan_rd", None), getattr(e, "plan_b", None)):
            if p is None:
                continue
            for (tag, fl, by), ms in zip(p.op_info(), p.profile()):
                d = per_tag.setdefault(tag, [0, 0.0, 0.0, 0.0])
                d[0] += 1; d[1] += ms; d[2] += fl; d[3] += by
                fl_u += fl; ms_u += ms
        it_ms = t_loop / iters * 1e3
        units.append(dict(unit=name, kind=e.kind, gflop_per_iteration=round(fl_u / 1e9, 2), loop_ms_per_iteration=round(it_ms, 4),
                          tflops=round(fl_u / (it_ms * 1e-3) / 1e12, 1) if it_ms > 0 else None, profiled_kernel_ms=round(ms_u, 4)))
    step_ms = sum(loop_s) / iters * 1e3
    flops = sum(v[2] for v in per_tag.values())
    rows = {t: bench.kernel_row(t, *v) for t, v in sorted(per_tag.items(), key=lambda kv: -kv[1][1])}
    mm = {t: v for t, v in per_tag.items() if v[2] > 0}
    dom = max(mm, key=lambda t: mm[t][1]) if mm else None
    ach = flops / (step_ms * 1e-3) / 1e12
    split_peak = bench.PEAK_BF16_MFMA_TFLOPS / 3.0
    return {"bound": "mfma", "unit": "TFLOP/s", "achieved": round(ach, 2), "peak": round(split_peak, 1),
            "peak_note": "whole schedule against the fp16-split ceiling 2500 / 3 (the pipe the bulk of its FLOPs run on); per kernel family below",
            "frac": round(ach / split_peak, 4), "frac_of_fp32_mfma_peak": round(ach / bench.PEAK_F32_MFMA_TFLOPS, 3),
            "algorithmic_gflop_per_step": round(flops / 1e9, 1), "ms_per_step": round(step_ms, 3),
            "dominant_kernel": dom, "dominant_kernel_row": rows.get(dom),
            "kernels": dict(list(rows.items())[:top]),
            "units_by_time": sorted(units, key=lambda u: -u["loop_ms_per_iteration"])[:top], "units": units}


def run_schedule(images=256, iters=20000, batch=4, eval_hw=(512, 768), n_eval=2, log=print, quality=True, arch="anchor", w_bits=8,
                 a_bits=8, per_unit_log=True, roofline=False, model=None, cali=None):
    """arch: "anchor" | "attn" (Cheng2020-attn, BASELINE config 3) | "lu2022" (BASELINE config 4: NIC embed 192 / latent 320, the 25
    units of main2.py's recon_model on the tape engine); w_bits / a_bits: weight grid and dynamic activation grid."""
    import math
    import bench
    from quantization import BaseQuantBlock, QuantModel, QuantModule, block_reconstruction, layer_reconstruction
    from test_datasets import evaluate_images
    dev = torch.device("cuda:0")
    lu = arch == "lu2022"
    if lu:
        eval_hw = (256, 256)       # the NIC is built for one resolution (window masks, token counts)
    g = torch.Generator().manual_seed(1005)
    if lu:
        import lic
        torch.manual_seed(1005)
        model = lic.NIC(dict(height=256, width=256, in_chans=3, embed_dim=192, latent_dim=320, window_size=8, mlp_ratio=2.0, qkv_bias=True,
                             qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False)).to(dev).eval()
    elif model is not None:        # the caller's model and calibration images (tests/run_kodak_schedule.py: natural-image statistics)
        model = model.to(dev).eval()
    else:
        model = bench.seeded_model(192, 1005, dev, arch=arch)
        with torch.no_grad():      # variance-preserving conv weights: the signal (and the quantisation error) reaches the output
            for name, p_ in model.named_parameters():
                if p_.dim() == 4 and "entropy_bottleneck" not in name:
                    p_.copy_(((torch.rand(p_.shape, generator=g) - 0.5) * 2 * (3.0 / p_[0].numel()) ** 0.5).to(dev))
    if cali is not None:
        cali = cali.to(dev)
        images = cali.shape[0]
    else:
        cali = torch.rand(images, 3, 256, 256, generator=g).to(dev)
    test_imgs = [torch.rand(1, 3, eval_hw[0], eval_hw[1], generator=g) for _ in range(n_eval)]
    probe = torch.rand(4, 3, 256, 256, generator=g).to(dev)
    res = {}
    if quality:
        res["fp32"] = evaluate_images(model, test_imgs)
        with torch.no_grad():
            ref_out = model(probe)["x_hat"].clone()

    def fidelity(net):
        """PSNR of the quantised reconstruction against the FP32 reconstruction (peak = FP output range)."""
        with torch.no_grad():
            out = net(probe)["x_hat"]
        mse = float(((out - ref_out) ** 2).mean())
        peak = float(ref_out.max() - ref_out.min())
        return 10 * math.log10(peak * peak / max(mse, 1e-30))
    wq = {"n_bits": w_bits, "channel_wise": True, "scale_method": "max"}
    aq = {"n_bits": a_bits, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    if a_bits != 8:
        aq["dynamic_bits"] = a_bits           # the reference's dynamic activation quantiser hard-wires 8 bits (quantizer.py:81)
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq, is_cheng=not lu).to(dev).eval()
    last = (lambda: qnn.model.g_s7) if lu else (lambda: qnn.model.g_s[-1][0])        # main2.py:258-263
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:batch])
    if quality:
        res["w8_rtn"] = evaluate_images(qnn.eval(), test_imgs) + (fidelity(qnn),)
    timing = []
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Lu2022" if lu else "Cheng2020", timing=timing)
    kwargs = dict(cali_data=cali, batch_size=batch, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True, b_range=(20, 2),
                  warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    engines = []

    def recon_model(m: nn.Module, prefix=""):
        for name, module in m.named_children():
            if isinstance(module, QuantModule):
                e = layer_reconstruction(qnn, module, name, **kwargs)
                if e is not None:
                    engines.append((prefix + name, e))
            elif isinstance(module, BaseQuantBlock):
                engines.append((prefix + name, block_reconstruction(qnn, module, name, **kwargs)))
            else:
                recon_model(module, prefix + name + ".")
    qnn.set_quant_state(True, False)
    last().set_quant_state(True, False)
    torch.cuda.synchronize()
    t0 = time.time()
    recon_model(qnn.model)
    torch.cuda.synchronize()
    wall = time.time() - t0
    n_units = len(engines)
    assert len(timing) == n_units          # one entry per unit that reached the engine (PixelShuffle pseudo-units return before)
    cache_s = sum(t["cache_s"] for t in timing)
    record_s = sum(t["record_s"] for t in timing)
    loop_s = sum(t["loop_s"] for t in timing)
    res.update(n_units=n_units, iters=iters, batch=batch, images=images, recon_model_wall_s=wall, cache_s=cache_s, record_s=record_s,
               loop_s=loop_s, images_per_s=n_units * batch * iters / wall, loop_ms_per_step=loop_s / iters * 1e3)
    log(f"recon_model: {n_units} units x {iters} iterations x batch {batch} on {images} images: {wall:.2f} s wall "
        f"(cache building {cache_s:.2f} s, recording + graph capture {record_s:.2f} s, loops {loop_s:.2f} s = {loop_s / iters * 1e3:.3f} ms/step)"
        f" => {res['images_per_s']:.0f} calibration images/s")
    units = []
    for (name, e), t in zip(engines, timing):
        done = tot = 0
        for op in e.ops.values():
            h = torch.clamp(torch.sigmoid(op.alpha) * 1.2 - 0.1, 0, 1)
            done += int(((h == 0) | (h == 1)).sum())
            tot += h.numel()
        rec, task, rnd, b = e.logs_terms()
        units.append(dict(unit=name, kind=e.kind, h2_plan=getattr(e, "h2_plan", None), h2_restarts=e.h2_restarts, use_h2=bool(e.use_h2), loop_ms_per_iter=round(t["loop_s"] / iters * 1e3, 4), loop_s=round(t["loop_s"], 3), cache_s=round(t["cache_s"], 3), record_s=round(t["record_s"], 3),
                          hard_frac=done / tot, loss_first=float(rec[0] + task[0] + rnd[0]), loss_last=float(rec[-1] + task[-1] + rnd[-1]),
                          rec_first=float(rec[0]), rec_last=float(rec[-1]), round_last=float(rnd[-1])))
        if per_unit_log:
            log(f"  {name:24s} {e.kind:5s} loop {t['loop_s']:7.2f} s = {t['loop_s'] / iters * 1e3:7.3f} ms/it  cache {t['cache_s']:5.2f} s  record {t['record_s']:5.2f} s  "
                f"soft targets in {{0,1}}: {100 * done / tot:6.2f} %  rec {float(rec[0]):.4e} -> {float(rec[-1]):.4e}  round {float(rnd[-1]):.3e}"
                f"  plan {getattr(e, 'h2_plan', None)} restarts {e.h2_restarts}")
    res["units"] = units
    if roofline:
        res["roofline"] = schedule_roofline(engines, [t["loop_s"] for t in timing], iters)
        r = res["roofline"]
        log(f"roofline: {r['algorithmic_gflop_per_step']:.0f} GFLOP per step in {r['ms_per_step']:.2f} ms = {r['achieved']:.0f} TFLOP/s = "
            f"{r['frac']:.3f} of {r['peak']:.0f}; dominant kernel {r['dominant_kernel']}: {r['dominant_kernel_row']}")
    if quality:
        qnn.set_quant_state(True, False)
        res["w8"] = evaluate_images(qnn.eval(), test_imgs) + (fidelity(qnn),)
        qnn.set_quant_state(True, True)
        last().set_quant_state(True, False)
        res["w8a8"] = evaluate_images(qnn.eval(), test_imgs) + (fidelity(qnn),)
        log(f"FP32      PSNR {res['fp32'][0]:.3f} dB  bpp {res['fp32'][1]:.4f}")
        for k, lab in (("w8_rtn", "W8 RTN   "), ("w8", "W8 cal.  "), ("w8a8", "W8A8 cal.")):
            p_, b_, f_ = res[k]
            log(f"{lab} PSNR {p_:.3f} dB ({p_ - res['fp32'][0]:+.3f})  bpp {b_:.4f} ({(b_ / res['fp32'][1] - 1) * 100:+.2f} %)  x_hat vs FP32 x_hat: {f_:.2f} dB")
    res["peak_mem_gib"] = torch.cuda.max_memory_allocated() / 2 ** 30
    log(f"peak memory {res['peak_mem_gib']:.1f} GiB")
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--json", default=None)
    ap.add_argument("--arch", default="anchor", choices=["anchor", "attn", "lu2022"])
    ap.add_argument("--no-quality", action="store_true")
    ap.add_argument("--w-bits", type=int, default=8)
    ap.add_argument("--a-bits", type=int, default=8)
    ap.add_argument("--roofline", action="store_true", help="algorithmic-FLOP table per unit and kernel-family fractions (one more profiled iteration per unit)")
    a = ap.parse_args()
    r = run_schedule(a.images, a.iters, a.batch, arch=a.arch, quality=not a.no_quality, w_bits=a.w_bits, a_bits=a.a_bits, roofline=a.roofline)
    if a.roofline:
        print("| unit | kind | GFLOP / iteration | ms / iteration | TFLOP/s |\n|---|---|---|---|---|")
        for u in r["roofline"]["units"]:
            print(f"| {u['unit']} | {u['kind']} | {u['gflop_per_iteration']} | {u['loop_ms_per_iteration']} | {u['tflops']} |")
        print("\n| kernel family | launches | ms | TFLOP/s | GB/s | bound | fraction |\n|---|---|---|---|---|---|---|")
        for t, k in r["roofline"]["kernels"].items():
            print(f"| {t} | {k['launches_per_step']} | {k['ms_per_step']} | {k['tflops']} | {k['gbs']} | {k['bound']} | {k['frac_of_peak']} |")
    if a.json:
        json.dump(r, open(a.json, "w"), indent=1)
