#!/usr/bin/env python3
"""The measurement plan beyond the headline ("GPU runs"): one number each for
BASELINE configs 3, 4 and 5 on one MI355X.  bench.py calls these after its timed region and reports them as EXTRA keys of its JSON
line; standalone:   python tools/bench_configs.py [attn] [lu2022] [mbt2018]"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rdo-ptq_amd"), ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def attn_w10(images=16, iters=320, batch=4, log=print):
    """BASELINE config 3 on one GPU: Cheng2020-attn N=192, W10 channel-wise weights, the whole recon_model schedule (105 units) for a few
    hundred iterations per unit through the public API; ms per step (one iteration of every unit) from the loop share of the wall.
    (320 iterations since round 6: with 60, the first replay of each of the 105 captured graphs -- the upload of the graph -- was 0.25 ms
    of the 8.6 ms "step"; the reference's schedule runs 20 000 per unit, tools/full_schedule.py times that: 8.375.)"""
    from full_schedule import run_schedule
    r = run_schedule(images=images, iters=iters, batch=batch, log=log, quality=False, arch="attn", w_bits=10, a_bits=10, per_unit_log=False, roofline=True)
    rf = r["roofline"]
    rf.pop("units", None)                       # (the full per-unit table: tools/full_schedule.py --roofline, profiles/)
    return {"workload": f"Cheng2020-attn N=192 W10A10 channel-wise, {r['n_units']} units, {images} images 256x256, batch {batch}, {iters} iterations per unit "
                        "(graph capture counted with the set-up, as in the full-length runs)",
            "units": r["n_units"], "ms_per_step": round(r["loop_s"] / iters * 1e3, 3),
            "images_per_s": round(r["n_units"] * batch * iters / r["loop_s"], 1), "wall_s": round(r["recon_model_wall_s"], 2), "roofline": rf}


def lu2022_unit(name="g_a1", log=print):
    """BASELINE config 4: one reconstruction unit of the full-size Lu2022 model (embed 192 / latent 320, 256x256, batch 4) on the tape
    engine -- forward and backward of the unit AND of the full-precision rest of its sub-coder (the task loss): ms per iteration as the
    difference of two runs (4 and 24 iterations: cache building and plan recording cancel) behind a throw-away run."""
    import lic
    from quantization import BaseQuantBlock, QuantModel, QuantModule, block_reconstruction, layer_reconstruction
    cfg = dict(height=256, width=256, in_chans=3, embed_dim=192, latent_dim=320, window_size=8, mlp_ratio=2.0, qkv_bias=True,
               qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False)
    torch.manual_seed(0)
    model = lic.NIC(cfg).cuda().eval()
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    aq = dict(wq, leaf_param=False)
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    B = 4
    cali = torch.rand(8, 3, 256, 256, device="cuda")
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Lu2022")
    unit = getattr(qnn.model, name)
    fn = layer_reconstruction if isinstance(unit, QuantModule) else block_reconstruction
    order = [n for n, m in qnn.model.named_children() if isinstance(m, (QuantModule, BaseQuantBlock))]
    for n in order:                                   # units are calibrated in order: everything before `name` counts as trained
        for m in getattr(qnn.model, n).modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = order.index(n) < order.index(name)
    ts = []
    # the first call pays one-time costs (kernel attributes, allocator growth, lazy tables) that a later call does not: a throw-away
    # call first, or the difference of the two timed calls under-reports the iteration (round 4 did: 6.75 instead of ~9 ms)
    for iters in (4, 4, 24):
        from quantization.quantizer import UniformAffineQuantizer
        for m in unit.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = False
            if isinstance(m, QuantModule) and hasattr(m.weight_quantizer, "alpha"):
                u = UniformAffineQuantizer(**wq, tconv=m.if_tconv)
                u.delta, u.zero_point, u.inited = m.weight_quantizer.delta, m.weight_quantizer.zero_point, True
                m.weight_quantizer = u
        torch.cuda.synchronize()
        t0 = time.time()
        fn(qnn, unit, name, cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True, b_range=(20, 2),
           warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    ms = (ts[2] - ts[1]) / 20 * 1e3
    log(f"Lu2022 {name}: {ms:.2f} ms/iteration")
    return {"workload": f"Lu2022 (embed 192, latent 320) unit {name} + FP tail of its sub-coder, 256x256, batch {B}", "ms_per_iteration": round(ms, 3),
            "images_per_s": round(B / ms * 1e3, 1)}


def lu2022_schedule(images=8, iters=30, batch=4, log=print):
    """BASELINE config 4 as a SCHEDULE (missing 2): all 28 reconstruction units of the full-size Lu2022 model (g_a0..g_s7,
    the context conv and the three entropy-parameter convs: main2.py:227-253 on models/nic_cvt.py) through the public API for a few
    iterations each; ms per step (one iteration of every unit) from the loop share of the wall, the slowest units by name."""
    from full_schedule import run_schedule
    r = run_schedule(images=images, iters=iters, batch=batch, log=log, quality=False, arch="lu2022", per_unit_log=False, roofline=True)
    rf = r["roofline"]
    rf.pop("units", None)
    slow = sorted(r["units"], key=lambda u: -u["loop_ms_per_iter"])[:6]
    return {"workload": f"Lu2022 (embed 192, latent 320), {r['n_units']} units, {images} images 256x256, batch {batch}, {iters} iterations per unit "
                        "(graph capture counted with the set-up, as in the full-length runs)",
            "units": r["n_units"], "ms_per_step": round(r["loop_s"] / iters * 1e3, 3),
            "images_per_s": round(r["n_units"] * batch * iters / r["loop_s"], 1), "wall_s": round(r["recon_model_wall_s"], 2),
            "slowest_units_ms_per_iteration": {u["unit"]: u["loop_ms_per_iter"] for u in slow}, "roofline": rf}


def rd_mode(units=("g_a.0", "g_a.6", "g_s.5"), batch=4, images=16, iters=24, log=print):
    """The opt-in R + lambda*D task loss (`loss_mode='rd'`; north_star names the R + lambda*D loss / backward as part of the hot path): ms
    per calibration iteration of three units of the headline model -- the first analysis block, the last analysis conv (whose tail is the
    whole hyper-path + synthesis transform) and the 128^2 synthesis block -- with the model behind the unit evaluated on torch's tape
    (hipops.autograd) inside the captured iteration, next to the same unit's default (lp) iteration."""
    import lic
    from quantization import QuantModel
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    out = {"workload": f"Cheng2020-anchor N=192, batch {batch}, 256x256, {iters} iterations behind 8 warm-up iterations", "units": {}}
    torch.manual_seed(0)
    cali = torch.rand(images, 3, 256, 256).cuda()
    for name in units:
        row = {}
        for mode in ("lp", "rd"):
            torch.manual_seed(1)
            qnn = QuantModel(lic.Cheng2020Anchor(N=192).eval().cuda(), wq, dict(wq, leaf_param=False), is_cheng=True).cuda().eval()
            qnn.set_quant_state(False, False)
            unit = qnn.model
            for part in name.split("."):
                unit = unit[int(part)] if part.isdigit() else getattr(unit, part)
            store = {"inps": [], "outs": []}
            def keep(m, i, o, store=store):
                store["inps"].append(i[0].detach().clone())
                store["outs"].append(o.detach().clone())
            h = unit.register_forward_hook(keep)
            with torch.no_grad():
                for i in range(0, images, batch):
                    qnn(cali[i:i + batch])
            h.remove()
            nh = lambda ts: torch.cat(ts).permute(0, 2, 3, 1).contiguous()
            inp, tgt = nh(store["inps"]), nh(store["outs"])
            k, mods = _unit_modules(unit)
            rd = None if mode == "lp" else dict(model=qnn, unit=unit, cali=cali, lmbda=0.0483)
            eng = UnitEngine(k, mods, inp, inp, tgt, batch_size=batch, iters=8 + iters, seed=1, rd=rd)
            eng.run(8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(iters)
            torch.cuda.synchronize()
            row[f"{mode}_ms_per_iteration"] = round((time.perf_counter() - t0) / iters * 1e3, 3)
            if rd is not None:
                row["rd_loop"] = eng.rd_path
            del eng, qnn
            torch.cuda.empty_cache()
        log(f"rd mode {name}: {row}")
        out["units"][name] = row
    return out


def mbt2018_eval(hw=(512, 768), n=6, log=print):
    """BASELINE config 5 on one GPU: W8A8 evaluation (pad, forward through the wrapped model incl. the masked context conv and the
    entropy models, crop, PSNR / bpp) of Minnen2018 with the autoregressive context model at full width on Kodak-sized images."""
    import lic
    from quantization import BaseQuantBlock, QuantModel, QuantModule
    from test_datasets import evaluate_images
    torch.manual_seed(31)
    model = lic.JointAutoregressiveHierarchicalPriors(N=192, M=192).cuda().eval()
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(model, wq, dict(wq, leaf_param=False)).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    for m in qnn.modules():
        if isinstance(m, (QuantModule, BaseQuantBlock)):
            m.trained = True
    qnn.set_quant_state(True, True)
    qnn.model.g_s[-1].set_quant_state(True, False)
    g = torch.Generator().manual_seed(3)
    imgs = [torch.rand(1, 3, hw[0], hw[1], generator=g) for _ in range(n)]
    evaluate_images(qnn, imgs[:2], p=64, distributed=False)          # warm-up (scale init, lazy allocations)
    torch.cuda.synchronize()
    t0 = time.time()
    psnr, bpp = evaluate_images(qnn, imgs, p=64, distributed=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    log(f"mbt2018 W8A8 eval: {n / dt:.2f} images/s at {hw[1]}x{hw[0]}")
    return {"workload": f"Minnen2018 (autoregressive context) N=M=192 W8A8 evaluation, {n} images {hw[1]}x{hw[0]}", "images_per_s": round(n / dt, 2),
            "ms_per_image": round(dt / n * 1e3, 2)}


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "lu2022", "lu2022_schedule", "mbt2018"]
    fns = {"attn": attn_w10, "lu2022": lu2022_unit, "lu2022_schedule": lu2022_schedule, "mbt2018": mbt2018_eval, "rd": rd_mode}
    for w in which:
        print(w, fns[w]())
