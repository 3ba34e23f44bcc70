#!/usr/bin/env python3
"""Full-size main2.py flow through the drop-in package only: Cheng2020-anchor N=192 (seeded random init), `n` synthetic 256x256
calibration crops, QuantModel surgery, scale init, recon_model over all 29 units with layer_/block_reconstruction (`iters`
iterations each, batch 4), then W8 and W8A8 evaluation on synthetic 512x768 images.  Prints the wall time of recon_model (cache
building + plan recording + hot loops) and the metric of SURVEY 8d: image-iterations per second over the whole unit schedule.
Quality numbers on seeded random weights only show that the flow is numerically sane (the rounding of random latents dominates
every difference); BD-rate claims need the trained checkpoints, which are not available offline.

    python tools/e2e_fullsize.py [--images 64] [--iters 200]"""
import argparse
import os
import sys
import time
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (seeded_model)
from quantization import BaseQuantBlock, QuantModel, QuantModule, block_reconstruction, layer_reconstruction  # noqa: E402
from test_datasets import evaluate_images  # noqa: E402


def run_flow(images=64, iters=200, batch=4, eval_hw=(512, 768), n_eval=2, log=print, arch="anchor", w_bits=8, a_bits=8):
    """The whole flow once; returns the numbers it prints (dict).  arch: "anchor" | "attn" (Cheng2020-attn, BASELINE config 3);
    w_bits / a_bits: weight grid and (dynamic) activation grid (10 / 10 = the W10A10 configuration)."""
    import math
    dev = torch.device("cuda:0")
    model = bench.seeded_model(192, 1005, dev, arch=arch)
    g = torch.Generator().manual_seed(1005)
    with torch.no_grad():      # variance-preserving conv weights, so that the signal (and quantisation error) reaches the output
        for name, p_ in model.named_parameters():
            if p_.dim() == 4 and "entropy_bottleneck" not in name:
                p_.copy_(((torch.rand(p_.shape, generator=g) - 0.5) * 2 * (3.0 / p_[0].numel()) ** 0.5).to(dev))
    cali = torch.rand(images, 3, 256, 256, generator=g).to(dev)
    test_imgs = [torch.rand(1, 3, eval_hw[0], eval_hw[1], generator=g) for _ in range(n_eval)]
    psnr_fp, bpp_fp = evaluate_images(model, test_imgs)
    probe = torch.rand(4, 3, 256, 256, generator=g).to(dev)
    with torch.no_grad():
        ref_out = model(probe)["x_hat"].clone()

    def fidelity(net):
        """PSNR of the quantised reconstruction against the FP32 reconstruction (peak = FP output range): without trained
        checkpoints the distortion against the image itself says nothing, the distance to the FP model does."""
        with torch.no_grad():
            out = net(probe)["x_hat"]
        mse = float(((out - ref_out) ** 2).mean())
        peak = float(ref_out.max() - ref_out.min())
        return 10 * math.log10(peak * peak / max(mse, 1e-30))
    wq = {"n_bits": w_bits, "channel_wise": True, "scale_method": "max"}
    aq = {"n_bits": a_bits, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    if a_bits != 8:
        aq["dynamic_bits"] = a_bits           # the reference's dynamic activation quantiser hard-wires 8 bits (quantizer.py:81)
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq, is_cheng=True).to(dev).eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:batch])
    psnr_w8_rtn, bpp_w8_rtn = evaluate_images(qnn.eval(), test_imgs)          # nearest rounding, before calibration
    fid_rtn = fidelity(qnn)
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Cheng2020")
    kwargs = dict(cali_data=cali, batch_size=batch, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True, b_range=(20, 2),
                  warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    units = []

    def recon_model(m: nn.Module):
        for name, module in m.named_children():
            if isinstance(module, QuantModule):
                if module.org_weight is not None:
                    units.append(name)
                layer_reconstruction(qnn, module, name, **kwargs)
            elif isinstance(module, BaseQuantBlock):
                units.append(name)
                block_reconstruction(qnn, module, name, **kwargs)
            else:
                recon_model(module)

    qnn.set_quant_state(True, False)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    torch.cuda.synchronize()
    t0 = time.time()
    recon_model(qnn)
    torch.cuda.synchronize()
    dt = time.time() - t0
    n_units = len(units)
    qnn.set_quant_state(True, False)
    psnr_w8, bpp_w8 = evaluate_images(qnn.eval(), test_imgs)
    fid_w8 = fidelity(qnn)
    qnn.set_quant_state(True, True)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    psnr_w8a8, bpp_w8a8 = evaluate_images(qnn.eval(), test_imgs)
    fid_w8a8 = fidelity(qnn)
    log(f"recon_model: {n_units} units x {iters} iterations x batch {batch} on {images} images: {dt:.2f} s wall "
        f"=> {n_units * iters * batch / dt:.0f} image-iterations/s (cache building and plan recording included)")
    log(f"FP32      PSNR {psnr_fp:.3f} dB  bpp {bpp_fp:.4f}")
    log(f"W8 RTN    PSNR {psnr_w8_rtn:.3f} dB  bpp {bpp_w8_rtn:.4f}   x_hat vs FP32 x_hat: {fid_rtn:.2f} dB")
    log(f"W8 cal.   PSNR {psnr_w8:.3f} dB  bpp {bpp_w8:.4f}   x_hat vs FP32 x_hat: {fid_w8:.2f} dB")
    log(f"W8A8 cal. PSNR {psnr_w8a8:.3f} dB  bpp {bpp_w8a8:.4f}   x_hat vs FP32 x_hat: {fid_w8a8:.2f} dB")
    log(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    return dict(n_units=n_units, wall_s=dt, psnr_fp=psnr_fp, bpp_fp=bpp_fp, psnr_w8_rtn=psnr_w8_rtn, bpp_w8_rtn=bpp_w8_rtn,
                psnr_w8=psnr_w8, bpp_w8=bpp_w8, psnr_w8a8=psnr_w8a8, bpp_w8a8=bpp_w8a8, fid_rtn=fid_rtn, fid_w8=fid_w8,
                fid_w8a8=fid_w8a8, qnn=qnn)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    run_flow(a.images, a.iters, a.batch)
