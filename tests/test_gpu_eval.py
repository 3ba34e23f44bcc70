"""Evaluation-side parity: entropy-model likelihood kernels, rate/distortion sums and the
whole Cheng2020 forward through the product (HIP) against the oracle's restatement on the CPU."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sync_state(dst, src):
    """copy every parameter/buffer of the oracle model into the product model (same CompressAI names)"""
    sd = src.state_dict()
    own = dst.state_dict()
    missing = [k for k in own if k not in sd]
    assert not missing, missing
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(sd[k])


def test_entropy_bottleneck_kernel_matches_oracle():
    import lic
    from oracle import lic_oracle as L
    torch.manual_seed(3)
    ref = L.EntropyBottleneck(24).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "_matrix" in n or "_factor" in n:
                p.add_(0.3 * torch.randn_like(p))
            if "quantiles" in n:
                p[:, 0, 1] = torch.randn(24) * 0.7
    prod = lic.EntropyBottleneck(24).eval()
    _sync_state(prod, ref)
    prod = prod.cuda()
    z = torch.randn(3, 24, 5, 7) * 4
    with torch.no_grad():
        zr, lr = ref(z)
        zp, lp = prod(z.cuda())
    torch.testing.assert_close(zp.cpu(), zr, rtol=0, atol=0)
    torch.testing.assert_close(lp.cpu(), lr, rtol=2e-5, atol=2e-8)


def test_gaussian_conditional_kernel_matches_oracle_and_sums_to_one():
    import lic
    from hipops import ops
    from oracle import lic_oracle as L
    g = torch.Generator().manual_seed(4)
    y = torch.randn(2, 16, 6, 5, generator=g) * 5
    scales = torch.rand(2, 16, 6, 5, generator=g) * 3
    scales[0, 0] = 0.01                                   # below the 0.11 scale floor
    means = torch.randn(2, 16, 6, 5, generator=g)
    ref = L.GaussianConditional(None).eval()
    with torch.no_grad():
        yr, lr = ref(y, scales, means=means)
        yp, lp = lic.GaussianConditional(None).eval()(y.cuda(), scales.cuda(), means=means.cuda())
    torch.testing.assert_close(yp.cpu(), yr, rtol=0, atol=0)
    torch.testing.assert_close(lp.cpu(), lr, rtol=3e-5, atol=3e-8)
    # bin masses of one Gaussian over all integer offsets sum to 1
    k = torch.arange(-60, 61, dtype=torch.float32).cuda()
    s = torch.full_like(k, 2.3)
    _, lik = ops.gaussian_likelihood(k + 0.3, s, torch.full_like(k, 0.3))
    assert abs(float(lik.sum()) - 1.0) < 1e-5


def test_gaussian_likelihood_backward_matches_autograd():
    from hipops import ops
    g = torch.Generator().manual_seed(5)
    yhat = torch.round(torch.randn(4000, generator=g) * 4)
    scales = (torch.rand(4000, generator=g) * 2 + 0.05).requires_grad_(True)
    means = (torch.randn(4000, generator=g) * 0.5).requires_grad_(True)
    s = torch.clamp(scales, min=0.11)
    v = (yhat - means).abs()
    phi = lambda t: 0.5 * torch.erfc(-t * (2 ** -0.5))
    lik = torch.clamp(phi((0.5 - v) / s) - phi((-0.5 - v) / s), min=1e-9)
    (-torch.log2(lik)).sum().backward()
    ds, dm = ops.gaussian_likelihood_bwd(yhat.cuda(), scales.detach().cuda(), means.detach().cuda())
    torch.testing.assert_close(ds.cpu(), scales.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(dm.cpu(), means.grad, rtol=2e-4, atol=2e-5)


def test_rd_loss_and_metrics_match_reference_formulas():
    from losses.losses import Metrics, RateDistortionLoss, compute_bpp, compute_psnr
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 64, 48, generator=g)
    xh = (x + 0.05 * torch.randn(2, 3, 64, 48, generator=g))
    ly = torch.rand(2, 8, 4, 3, generator=g) * 0.9 + 1e-3
    lz = torch.rand(2, 8, 1, 1, generator=g) * 0.9 + 1e-3
    out = {"x_hat": xh.cuda(), "likelihoods": {"y": ly.cuda(), "z": lz.cuda()}}
    npx = 2 * 64 * 48
    bpp = float((-torch.log2(ly)).sum() / npx + (-torch.log2(lz)).sum() / npx)
    mse = float(torch.mean((xh - x) ** 2))
    r = RateDistortionLoss(lmbda=0.0483, metric="mse")(out, x.cuda())
    assert abs(float(r["bpp_loss"]) - bpp) < 1e-5 * bpp
    assert abs(float(r["mse_loss"]) - mse) < 1e-5 * mse
    assert abs(float(r["loss"]) - (0.0483 * 255 ** 2 * mse + bpp)) < 1e-5 * (0.0483 * 255 ** 2 * mse + bpp)
    assert abs(compute_bpp(out) - bpp) < 1e-5 * bpp
    assert abs(compute_psnr(xh.cuda(), x.cuda()) - (-10 * math.log10(mse))) < 1e-4
    b2, psnr, _ = Metrics()(out, x.cuda())
    per_img = torch.mean((xh - x) ** 2, dim=[1, 2, 3])
    assert abs(float(psnr) - float(torch.mean(10 * torch.log10(1. / per_img)))) < 1e-4
    with pytest.raises(ValueError):                      # 64 x 48 is too small for the five MS-SSIM scales
        RateDistortionLoss(metric="ms-ssim")(out, x.cuda())
    # the 'ms-ssim' objective on a large enough image: lambda * (1 - MS-SSIM) + bpp (losses.py:30-33)
    from oracle import msssim_oracle as MO
    xb = torch.rand(1, 3, 192, 192, generator=g)
    xbh = (xb + 0.05 * torch.randn(1, 3, 192, 192, generator=g)).clamp(0, 1)
    outb = {"x_hat": xbh.cuda(), "likelihoods": {"y": ly[:1].cuda()}}
    rb = RateDistortionLoss(lmbda=4.58, metric="ms-ssim")(outb, xb.cuda())
    want = 4.58 * (1 - float(MO.ms_ssim(xbh, xb))) + float((-torch.log2(ly[:1])).sum() / (192 * 192))
    assert abs(float(rb["loss"]) - want) < 1e-4 * want


@pytest.mark.parametrize("quant", [False, True])
def test_full_model_forward_matches_oracle(quant):
    """x_hat, likelihoods, PSNR and bpp of the whole (wrapped) Cheng2020 forward: product on HIP vs oracle on CPU, with
    full-precision weights and with W8 nearest-rounded weights (the 'Weight quantization model w/o opt' line of main2.py)."""
    import lic
    from oracle import lic_oracle as L
    from oracle import rdo_oracle as O
    from oracle.cheng_units import schedule
    from quantization import QuantModel
    from test_datasets import evaluate_images
    torch.manual_seed(21)
    N = 16
    ref = L.Cheng2020Anchor(N=N).eval()
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("gamma"):
                c = p.shape[0]
                p.copy_(torch.sqrt(0.1 * torch.eye(c) + 0.01 * torch.rand(c, c, generator=g) + 2.0 ** -36))
    prod = lic.Cheng2020Anchor(N=N).eval()
    _sync_state(prod, ref)
    x = torch.rand(2, 3, 128, 64, generator=g)
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(prod.cuda(), wq, dict(wq, leaf_param=False), is_cheng=True).eval()
    qnn.set_quant_state(quant, False)
    if quant:
        # oracle side: nearest-rounded weights for every conv / GDN gamma, masked context conv left as the wrapper sees it
        with torch.no_grad():
            for mod in ref.modules():
                if isinstance(mod, torch.nn.Conv2d):
                    d, z = O.uaq_init(mod.weight.data, 8, True, "max")
                    mod.weight.data = O.uaq_fakequant(mod.weight.data, d, z, 256)
                elif isinstance(mod, L.GDN):
                    d, z = O.uaq_init(mod.gamma.data, 8, True, "max")
                    mod.gamma.data = O.uaq_fakequant(mod.gamma.data, d, z, 256)
    # reference quirk: the wrapped PixelShuffle applies LeakyReLU to the decoder output and QuantModule bypasses
    # MaskedConv2d's mask; mirror both in the oracle model for this comparison
    ref.context_prediction.mask.fill_(1.0)
    with torch.no_grad():
        out_r = ref(x)
        out_r["x_hat"] = torch.nn.functional.leaky_relu(out_r["x_hat"], 0.01)
        out_p = qnn(x.cuda())
    scale = float(out_r["x_hat"].abs().max())
    assert float((out_p["x_hat"].cpu() - out_r["x_hat"]).abs().max()) < 2e-4 * scale
    for k in ("y", "z"):
        lr_, lp_ = out_r["likelihoods"][k], out_p["likelihoods"][k].cpu()
        # a latent that lands within float noise of a rounding boundary may round differently: allow a handful
        bad = ((lp_ - lr_).abs() > 1e-3 * lr_ + 1e-7).float().mean()
        assert float(bad) < 5e-3, (k, float(bad))
    psnr, bpp = evaluate_images(qnn, [x[i:i + 1] for i in range(2)], p=64)
    assert math.isfinite(psnr) and bpp > 0


@pytest.mark.parametrize("shape", [(1, 3, 192, 176), (2, 3, 256, 256), (1, 3, 337, 263)])
def test_ms_ssim_matches_oracle(shape):
    """MS-SSIM (rdo_ssim_level + rdo_avg_pool2, five scales) against the oracle's restatement of pytorch_msssim.ms_ssim, incl.
    odd sizes (padded 2x2 pooling).  fp32 separable filtering in a different summation order: 2e-5 absolute."""
    from oracle import msssim_oracle as MO
    from losses.losses import compute_msssim, ms_ssim
    g = torch.Generator().manual_seed(shape[2])
    x = torch.rand(shape, generator=g)
    y = (x + 0.08 * torch.randn(shape, generator=g)).clamp(0, 1)
    ref = MO.ms_ssim(x, y, data_range=1.0, size_average=False)
    got = ms_ssim(x.cuda(), y.cuda(), data_range=1.0, size_average=False).cpu()
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-5)
    assert abs(compute_msssim(x[:1].cuda(), y[:1].cuda()) - (-10 * math.log10(1 - float(MO.ms_ssim(x[:1], y[:1]))))) < 1e-3
    with pytest.raises(ValueError):
        ms_ssim(x[..., :160, :160].cuda(), y[..., :160, :160].cuda())


@pytest.mark.parametrize("hw", [(512, 768), (1200, 1200)])
def test_mbt2018_full_width_w8_w8a8_eval_matches_oracle(hw):
    """BASELINE config 5 at full width: Minnen2018 with the autoregressive context model (N = M = 192), nearest-rounded W8 weights
    and dynamic A8 activations, one Kodak-sized (768 x 512) and one Tecnick-sized (1200 x 1200) image through `evaluate_images`
    (pad to 64, forward, crop, PSNR / bpp) on HIP against the oracle's restatement of the same quantised forward on the CPU.
    fp tolerance: W8: bpp 2e-4 relative, PSNR 0.01 dB (fp32 summation order, a rounded latent at a .5 boundary); W8A8: bpp 1e-3
    relative, PSNR 0.03 dB -- a value within float noise of a boundary of one of the ~30 cascaded dynamic 8-bit grids lands one level
    (1/255 of the channel's range) apart and moves a handful of rounded latents downstream (measured: 2.6e-4 on 768 x 512).  The image-parallel path (process
    group of one rank) must reproduce the local numbers."""
    import os
    import lic
    from oracle import lic_oracle as L
    from quantization import BaseQuantBlock, QuantModel, QuantModule
    from test_datasets import crop, evaluate_images, pad
    torch.manual_seed(31)
    ref = L.JointAutoregressiveHierarchicalPriors(N=192, M=192).eval()
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("gamma"):
                c = p.shape[0]
                p.copy_(torch.sqrt(0.1 * torch.eye(c) + 0.002 * torch.rand(c, c, generator=g) + 2.0 ** -36))
            elif p.dim() == 4 and "entropy_bottleneck" not in name:
                fan = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 2 * (3.0 / fan) ** 0.5)
    ref.context_prediction.mask.fill_(1.0)          # the wrapper bypasses the mask: keep both sides unmasked
    prod = lic.JointAutoregressiveHierarchicalPriors(N=192, M=192).eval()
    _sync_state(prod, ref)
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(prod.cuda(), wq, dict(wq, leaf_param=False)).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    for m in qnn.modules():
        if isinstance(m, (QuantModule, BaseQuantBlock)):
            m.trained = True
    x = torch.rand(1, 3, *hw, generator=g)
    xp = pad(x, 64)

    def oracle_metrics(act):
        out = L.mbt2018_forward_w8a8(ref, xp, act_quant=act)
        rec = crop(out["x_hat"], hw).clamp(0, 1)
        psnr = 10 * math.log10(1.0 / float(((x - rec) ** 2).mean()))
        bpp = sum(float((-torch.log2(v)).sum()) for v in out["likelihoods"].values()) / (xp.shape[2] * xp.shape[3])
        return psnr, bpp

    for act in (False, True):
        qnn.set_quant_state(True, act)
        qnn.model.g_s[-1].set_quant_state(True, False)
        psnr, bpp = evaluate_images(qnn, [x], p=64, distributed=False)
        psnr_o, bpp_o = oracle_metrics(act)
        assert abs(bpp - bpp_o) <= (1e-3 if act else 2e-4) * bpp_o, (act, bpp, bpp_o)
        assert abs(psnr - psnr_o) <= (0.03 if act else 0.01), (act, psnr, psnr_o)
    if not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    try:
        psnr_d, bpp_d = evaluate_images(qnn, [x], p=64)
    finally:
        torch.distributed.destroy_process_group()
    assert abs(psnr_d - psnr) < 1e-4 and abs(bpp_d - bpp) < 1e-5 * bpp


def test_factorized_likelihood_backward_matches_autograd():
    """rdo_factorized_likelihood_bwd (the R term of the opt-in R + lambda*D task loss through the factorised prior, straight-through
    rounding) against torch autograd of the same EntropyBottleneck evaluated in float64 on the CPU."""
    import copy
    from lic.entropy import EntropyBottleneck
    torch.manual_seed(4)
    eb = EntropyBottleneck(24).eval()
    with torch.no_grad():
        for n, p in eb.named_parameters():
            if n.startswith("_factor"):
                p.uniform_(-0.8, 0.8)
            elif n.startswith("_matrix"):
                p.add_(0.5 * torch.randn_like(p))
            elif n == "quantiles":
                p[:, 0, 1] = torch.randn(24) * 0.3
    ref = copy.deepcopy(eb).double()
    eb = eb.cuda()
    g = torch.Generator().manual_seed(5)
    z0 = torch.randn(2, 24, 6, 5, generator=g) * 3
    wz = torch.randn(z0.shape, generator=g)
    z = z0.cuda().requires_grad_(True)
    zhat, lik = eb(z)                                             # HIP forward, HIP backward
    ((-torch.log2(lik)).sum() * 0.37 + (zhat * wz.cuda()).sum()).backward()
    zc = z0.double().requires_grad_(True)
    zh_c, lik_c = ref(zc)                                         # torch path (CPU tensors)
    ((-torch.log2(lik_c)).sum() * 0.37 + (zh_c * wz.double()).sum()).backward()
    torch.testing.assert_close(zhat.detach().cpu().double(), zh_c.detach(), rtol=0, atol=1e-6)
    torch.testing.assert_close(lik.detach().cpu().double(), lik_c.detach(), rtol=2e-4, atol=1e-9)
    scale = float(zc.grad.abs().max())
    assert float((z.grad.cpu().double() - zc.grad).abs().max()) <= 2e-4 * scale


@pytest.mark.parametrize("case", ["conv3s1", "conv3s2", "conv1", "tconv5s2", "gdn", "igdn"])
def test_autograd_functions_on_the_split_precision_path(case):
    """hipops.autograd Conv2dFn / ConvTranspose2dFn / GDNFn with a WeightPack at 192 channels and enough pixels that forward and
    input gradient take the bf16x6 MFMA kernels: value and input gradient against torch in float64."""
    import torch.nn.functional as F
    from hipops import _lib as HL
    from hipops import autograd as A
    from hipops import ops
    g = torch.Generator().manual_seed(17)
    C, B, H = 192, 2, 128
    x0 = torch.randn(B, C, H, H, generator=g)
    go = None
    if case.startswith("conv"):
        K, s = (3, 1) if case == "conv3s1" else ((3, 2) if case == "conv3s2" else (1, 1))
        w = torch.randn(C, C, K, K, generator=g) / math.sqrt(C * K * K)
        b = torch.randn(C, generator=g) * 0.1
        pack = ops.WeightPack(w.permute(0, 2, 3, 1).cuda(), b.cuda())
        assert pack.planes((B, H, H, C), s, K // 2) is not None
        run = lambda x: A.Conv2dFn.apply(x, pack, pack.bias, s, K // 2, HL.EPI_LRELU)
        ref = lambda x: F.leaky_relu(F.conv2d(x, w.double(), b.double(), stride=s, padding=K // 2), 0.01)
    elif case == "tconv5s2":
        H = 64
        x0 = x0[:, :, :H, :H].contiguous()
        w = torch.randn(C, C, 5, 5, generator=g) / math.sqrt(C * 25 / 4)          # [Cin, Cout, K, K]
        b = torch.randn(C, generator=g) * 0.1
        pack = ops.WeightPack(w.permute(1, 2, 3, 0).cuda(), b.cuda())              # to_rows(W, tconv=True)
        run = lambda x: A.ConvTranspose2dFn.apply(x, pack, pack.bias, 2, 2, 1, HL.EPI_NONE)
        ref = lambda x: F.conv_transpose2d(x, w.double(), b.double(), stride=2, padding=2, output_padding=1)
    else:
        inverse = case == "igdn"
        gamma = 0.1 * torch.eye(C) + 0.01 * torch.rand(C, C, generator=g)
        beta = 1.0 + torch.rand(C, generator=g)
        pack = ops.WeightPack(gamma.reshape(C, 1, 1, C).cuda(), beta.cuda())
        assert pack.planes((B, H, H, C), 1, 0) is not None
        run = lambda x: A.GDNFn.apply(x, pack, pack.bias, inverse)

        def ref(x):
            n = torch.sqrt(F.conv2d(x * x, gamma.double().reshape(C, C, 1, 1), beta.double()))
            return x * n if inverse else x / n
    xg = x0.cuda().requires_grad_(True)
    y = run(xg)
    xr = x0.double().requires_grad_(True)
    yr = ref(xr)
    if case.startswith("conv"):
        # LeakyReLU's derivative jumps at 0: take the branch the kernel took (pre-activations within rounding of 0 differ in sign)
        slope = torch.where(y.detach().cpu() > 0, 1.0, 0.01).double()
        yr_lin = F.conv2d(xr, w.double(), b.double(), stride=s, padding=K // 2) * slope
    go = torch.randn(yr.shape, generator=g)
    (dx,) = torch.autograd.grad(y, xg, go.cuda())
    (dxr,) = torch.autograd.grad(yr_lin if case.startswith("conv") else yr, xr, go.double())
    for what, got, want in (("value", y, yr), ("input gradient", dx, dxr)):
        err = (got.detach().cpu().double() - want.detach()).abs().max().item()
        assert err <= 2e-6 * want.detach().abs().max().item() + 1e-7, (case, what, err)
    # the pack's derived tensors were built once and are reused by a second pass
    before = {k: (v.w.data_ptr() if isinstance(v, ops.WeightPack) else None) for k, v in pack._d.items()}
    (dx2,) = torch.autograd.grad(run(xg), xg, go.cuda())
    assert torch.equal(dx2, dx)
    assert before == {k: (v.w.data_ptr() if isinstance(v, ops.WeightPack) else None) for k, v in pack._d.items()}


def test_quant_module_weight_pack_follows_the_quant_state():
    """QuantModule.weight_pack(): one pack per value of the effective weight -- reused across forwards, rebuilt when the quant
    state, the quantiser or the weight changes (the forward then matches a fresh module)."""
    import torch.nn as nn
    from helpers import AQ, WQ
    from quantization.quant_layer import QuantModule
    torch.manual_seed(5)
    conv = nn.Conv2d(16, 24, 3, padding=1).cuda()
    qm = QuantModule(conv, WQ, AQ).cuda()
    x = torch.randn(2, 16, 20, 20, device="cuda")
    p_fp = qm.weight_pack()
    assert qm.weight_pack() is p_fp
    y_fp = qm(x).clone()
    qm.set_quant_state(True, False)
    y_q = qm(x).clone()
    p_q = qm.weight_pack()
    assert p_q is not p_fp and qm.weight_pack() is p_q
    assert not torch.equal(y_q, y_fp)
    with torch.no_grad():
        qm.weight.mul_(1.5)
    assert qm.weight_pack() is not p_q
    fresh = QuantModule(conv, WQ, AQ).cuda()
    fresh.weight_quantizer = qm.weight_quantizer
    fresh.set_quant_state(True, False)
    assert torch.equal(qm(x), fresh(x))
    qm.set_quant_state(False, False)
    assert torch.equal(qm(x), y_fp)
