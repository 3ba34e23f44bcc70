"""Per-kernel parity of the HIP library (through the C ABI) against torch fp64/fp32 references and the oracle.
Tolerances are stated per test: fp32 MFMA accumulation is a k-ordered fmaf chain, so conv results differ from an fp64
reference by ~1e-6 relative to sum|a*b|; element-wise kernels match the oracle to a few ulp."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from hipops import ops as o
    return o


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


CONV_CASES = [
    # B, H, W, Cin, Cout, K, stride, pad
    (2, 16, 16, 32, 64, 3, 1, 1),
    (2, 16, 16, 192, 192, 3, 1, 1),
    (1, 32, 32, 192, 192, 3, 2, 1),
    (2, 17, 13, 8, 8, 3, 1, 1),       # ragged, tiny channels (toy goldens)
    (2, 16, 16, 3, 192, 3, 2, 1),     # first layer, Cin=3 (scalar loads)
    (2, 16, 16, 192, 12, 3, 1, 1),    # g_s.7.0
    (2, 8, 8, 768, 640, 1, 1, 0),     # entropy_parameters.0
    (1, 16, 16, 192, 384, 5, 1, 2),   # context_prediction
    (2, 16, 16, 192, 192, 1, 2, 0),   # skip 1x1 s2
    (1, 8, 8, 192, 768, 3, 1, 1),     # subpel conv
    (3, 5, 7, 36, 20, 5, 2, 2),       # everything ragged
    (2, 32, 32, 3, 192, 1, 2, 0),     # RGB stems (thin-conv kernels): 1x1 s2 skip of g_a.0
    (2, 20, 28, 3, 100, 5, 2, 2),     #   5x5 s2 first layer of Minnen2018 / Lu2022, ragged Cout
    (1, 9, 11, 3, 300, 3, 1, 1),      #   more than 256 output channels
    (4, 64, 64, 3, 192, 3, 2, 1),     #   enough pixels for several chunks per slab
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_matches_fp64(ops, case):
    B, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    out = ops.conv2d_fwd(xd, wd, b.cuda(), stride=s, pad=p)
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 5e-6   # ~sqrt(K)*2^-24 for K up to 4800 (fp32 fmaf chain)
    if Cin == 3:                   # fused activations of the thin-conv path
        from hipops import _lib as L
        for epi, fn in ((L.EPI_LRELU, lambda t: F.leaky_relu(t, 0.01)), (L.EPI_RELU, F.relu)):
            y = ops.conv2d_fwd(xd, wd, b.cuda(), stride=s, pad=p, epilogue=epi).cpu().permute(0, 3, 1, 2)
            assert _rel(y, fn(ref)) < 5e-6


def test_conv_fwd_epilogues(ops):
    from hipops import _lib as L
    g = torch.Generator().manual_seed(3)
    B, H, W, Cc = 2, 12, 12, 64
    x = torch.randn(B, H, W, Cc, generator=g).cuda()
    w = (torch.randn(Cc, 3, 3, Cc, generator=g) / 24).cuda()
    b = torch.randn(Cc, generator=g).cuda()
    aux = torch.randn(B, H, W, Cc, generator=g).cuda()
    res = torch.randn(B, H, W, Cc, generator=g).cuda()
    base = ops.conv2d_fwd(x, w, b, 1, 1)
    pre = torch.empty_like(base)
    y = ops.conv2d_fwd(x, w, b, 1, 1, epilogue=L.EPI_LRELU, residual=res, pre=pre)
    torch.testing.assert_close(pre, base, rtol=0, atol=0)
    torch.testing.assert_close(y, F.leaky_relu(base, 0.01) + res, rtol=1e-6, atol=1e-6)
    y = ops.conv2d_fwd(x, w, None, 1, 1, epilogue=L.EPI_LRELU_BWD, aux=aux)
    nb = ops.conv2d_fwd(x, w, None, 1, 1)
    torch.testing.assert_close(y, torch.where(aux > 0, nb, 0.01 * nb), rtol=1e-6, atol=1e-6)
    # GDN-style: square on load, positive weights/bias, aux * rsqrt / sqrt
    wp = w.abs()
    bp = b.abs() + 0.5
    n = ops.conv2d_fwd(x * x, wp, bp, 1, 1)
    for epi, fn in ((L.EPI_GDN, torch.rsqrt), (L.EPI_IGDN, torch.sqrt)):
        y = ops.conv2d_fwd(x, wp, bp, 1, 1, epilogue=epi, aux=aux, square_input=True)
        torch.testing.assert_close(y, aux * fn(n), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad_matches_fp64(ops, case):
    B, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2**31 + 1)
    x = torch.randn(B, Cin, H, W, generator=g)
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, K, K), dy.double(), stride=s, padding=p)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    slabs = ops.conv2d_wgrad(xd, dyd, (Cout, K, K, Cin), stride=s, pad=p)
    dw = ops.reduce_slabs(slabs)
    torch.cuda.synchronize()
    got = dw.cpu().permute(0, 3, 1, 2)
    assert _rel(got, ref) < 3e-6


def test_conv_dgrad_via_flipped_weights(ops):
    """dgrad of a stride-1 'same' conv == forward conv of dy with the wd layout emitted by rdo_adaround_fwd."""
    g = torch.Generator().manual_seed(5)
    B, H, W, Ci, Co, K = 2, 10, 9, 32, 48, 3
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, K, K, generator=g, dtype=torch.float64) / 17
    dy = torch.randn(B, Co, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    w_ohwi = w.float().permute(0, 2, 3, 1).contiguous().cuda()
    delta = torch.full((Co,), 1e-3).cuda()
    zp = torch.full((Co,), 128.0).cuda()
    d = ops.ada_desc(w_ohwi, n_levels=1 << 16)   # fine grid: fake-quant ~ identity is not needed, use wd of the quantised w
    wq = torch.empty_like(w_ohwi)
    wd = torch.empty(Ci, K, K, Co, device="cuda")
    ops.uaq_fakequant(d, w_ohwi, delta, zp, wq, wd)
    # reference dgrad with the same quantised weight
    wq_oihw = wq.cpu().permute(0, 3, 1, 2).double()
    x2 = x.detach().clone().requires_grad_(True)
    F.conv2d(x2, wq_oihw, None, 1, 1).backward(dy)
    dx = ops.conv2d_fwd(dy.float().permute(0, 2, 3, 1).contiguous().cuda(), wd, None, 1, K - 1 - 1)
    torch.cuda.synchronize()
    assert _rel(dx.cpu().permute(0, 3, 1, 2), x2.grad) < 2e-6


def test_pixel_shuffle_roundtrip(ops):
    x = torch.randn(2, 6, 5, 12 * 4).cuda()
    y = ops.pixel_shuffle(x, 2)
    ref = F.pixel_shuffle(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    torch.testing.assert_close(y, ref.contiguous(), rtol=0, atol=0)
    torch.testing.assert_close(ops.pixel_unshuffle(y, 2), x, rtol=0, atol=0)


def test_layout_and_small_ops(ops):
    x = torch.randn(2, 5, 6, 8).cuda()
    torch.testing.assert_close(ops.nchw_to_nhwc(x), x.permute(0, 2, 3, 1).contiguous(), rtol=0, atol=0)
    torch.testing.assert_close(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x)), x, rtol=0, atol=0)
    g, y = torch.randn(4, 64).cuda(), torch.randn(4, 64).cuda()
    torch.testing.assert_close(ops.lrelu_bwd(g, y), torch.where(y > 0, g, 0.01 * g), rtol=0, atol=0)
    torch.testing.assert_close(ops.add(g, y), g + y, rtol=0, atol=0)


def test_actquant_matches_oracle(ops):
    from oracle import rdo_oracle as O
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 24, 7, 9, generator=g) * 2
    x[:, 5] = 0.75
    ref = O.act_quant(x)
    got = ops.actquant_perchannel(x.permute(0, 2, 3, 1).contiguous().cuda()).cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-7)


@pytest.mark.parametrize("shape", [(1, 192, 128, 96), (2, 1280, 5, 7), (1, 3, 33, 17), (4, 64, 64, 64), (1, 6, 1, 1)])
def test_actquant_shapes_match_oracle(ops, shape):
    """the two-level min / max reduction at sizes that use many workgroups, more channel groups than threads, scalar (C % 4 != 0) lanes"""
    from oracle import rdo_oracle as O
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * 3 + 0.5
    ref = O.act_quant(x)
    got = ops.actquant_perchannel(x.permute(0, 2, 3, 1).contiguous().cuda()).cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, rtol=0, atol=5e-7)
    # per-tensor use (one channel = the whole tensor, quantizer.py's non-4-D branch) and a reused workspace
    flat = x.reshape(-1, 1).contiguous().cuda()
    ws = torch.empty(int(ops.L.lib().rdo_actquant_workspace(1)), device="cuda")
    a = ops.actquant_perchannel(flat, ws=ws)
    b = ops.actquant_perchannel(flat, ws=ws)
    assert torch.equal(a, b)
    lo, hi = float(x.min()), float(x.max())
    q = torch.round(((x - lo) / max(hi - lo, 1e-6)).clamp(-1, 1) * 255) / 255 * max(hi - lo, 1e-6) + lo
    torch.testing.assert_close(a.cpu().reshape(x.shape), q, rtol=0, atol=5e-6)


def test_uaq_init_and_fakequant_match_oracle(ops):
    from oracle import rdo_oracle as O
    g = torch.Generator().manual_seed(10)
    w = torch.randn(16, 8, 3, 3, generator=g) * 0.2
    delta, zp = O.uaq_init(w, 8, True, "max")
    wo = w.permute(0, 2, 3, 1).contiguous().cuda()
    d_gpu, z_gpu = ops.uaq_init_minmax(wo, 256)
    torch.testing.assert_close(d_gpu.cpu(), delta.view(-1), rtol=0, atol=0)
    torch.testing.assert_close(z_gpu.cpu(), zp.view(-1), rtol=0, atol=0)
    d = ops.ada_desc(wo)
    wq = ops.uaq_fakequant(d, wo, d_gpu, z_gpu)
    ref = O.uaq_fakequant(w, delta, zp, 256)
    torch.testing.assert_close(wq.cpu().permute(0, 3, 1, 2), ref, rtol=0, atol=0)


def test_adaround_fwd_init_match_oracle(ops):
    from oracle import rdo_oracle as O
    g = torch.Generator().manual_seed(11)
    w = torch.randn(16, 8, 3, 3, generator=g) * 0.2
    delta, zp = O.uaq_init(w, 8, True, "max")
    a0 = O.adaround_init_alpha(w, delta)
    wo = w.permute(0, 2, 3, 1).contiguous().cuda()
    d = ops.ada_desc(wo)
    dg, zg = delta.view(-1).cuda(), zp.view(-1).cuda()
    alpha = ops.adaround_init_alpha(d, wo, dg)
    torch.testing.assert_close(alpha.cpu().permute(0, 3, 1, 2), a0, rtol=2e-6, atol=2e-6)
    al = (a0 + torch.randn(a0.shape, generator=g)).contiguous()
    alg = al.permute(0, 2, 3, 1).contiguous().cuda()
    for soft in (True, False):
        ref = O.adaround_forward(w, al, delta, zp, 256, soft)
        wd = torch.empty(8, 3, 3, 16, device="cuda")
        got = ops.adaround_fwd(d, wo, alg, dg, zg, soft, wd=wd)
        torch.testing.assert_close(got.cpu().permute(0, 3, 1, 2), ref, rtol=0, atol=float(delta.max()) * 1e-4)  # (floor + h + zp) rounds on the 2^-16 grid at |x_int| ~ 128
        # wd[ci][kh'][kw'][co] = wq[co][K-1-kh'][K-1-kw'][ci]
        torch.testing.assert_close(wd, got.flip(1, 2).permute(3, 1, 2, 0).contiguous(), rtol=0, atol=0)


def test_gather_qdrop_and_lp2_match_oracle(ops):
    from oracle import rdo_oracle as O
    g = torch.Generator().manual_seed(12)
    n, Cc, H, W, B, iters = 6, 8, 6, 5, 3, 4
    cq = torch.randn(n, Cc, H, W, generator=g)
    cf = torch.randn(n, Cc, H, W, generator=g)
    tgt = torch.randn(n, Cc, H, W, generator=g)
    idx = torch.stack([torch.randperm(n, generator=g)[:B] for _ in range(iters)]).int()
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.empty(B, H, W, Cc, device="cuda")
    grad = torch.empty_like(out)
    log = torch.zeros(iters, 32, device="cuda")
    for i in range(iters):
        ops.gather_qdrop(nh(cq), nh(cf), idx.cuda(), it, B, 0.5, 1005, out)
        keep = O.qdrop_keep_mask_nhwc(1005, i, (B, Cc, H, W), 0.5)
        ref = torch.where(keep, cq[idx[i].long()], cf[idx[i].long()])
        torch.testing.assert_close(out.cpu().permute(0, 3, 1, 2), ref, rtol=0, atol=0)
        ops.lp2_loss_grad(out, nh(tgt), idx.cuda(), it, 2.0, grad, log)
        pr = ref.clone().requires_grad_(True)
        t = tgt[idx[i].long()]
        loss = O.lp_loss(pr, t, p=2.0) + O.lp_loss(pr, t, p=2.0)
        loss.backward()
        torch.testing.assert_close(grad.cpu().permute(0, 3, 1, 2), pr.grad, rtol=1e-6, atol=1e-8)
        assert abs(float(log[i].sum()) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
        ops.iter_advance(it)
    assert int(it.item()) == iters


def test_gdn_forward_backward_match_oracle(ops):
    """f_gdn (quant_layer.py:142-154) forward and its input/gamma gradients, composed from the conv kernels."""
    from hipops import _lib as L
    from oracle import rdo_oracle as O
    g = torch.Generator().manual_seed(13)
    B, Cc, H, W = 2, 32, 9, 7
    for inverse in (False, True):
        x = torch.randn(B, Cc, H, W, generator=g, requires_grad=True)
        gamma = (0.1 * torch.eye(Cc) + 0.02 * torch.rand(Cc, Cc, generator=g)).sqrt().requires_grad_(True)
        beta = (0.5 + torch.rand(Cc, generator=g)).sqrt()
        gy = torch.randn(B, Cc, H, W, generator=g)
        y = O.f_gdn(x, gamma, beta, inverse)
        y.backward(gy)
        gp = O._GAMMA_REPARAM(gamma.detach())
        bp = O._BETA_REPARAM(beta)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
        xg, gyg = nh(x), nh(gy)
        w1 = gp.reshape(Cc, 1, 1, Cc).contiguous().cuda()
        w1t = gp.t().reshape(Cc, 1, 1, Cc).contiguous().cuda()
        norm = torch.empty_like(xg)
        yg = ops.conv2d_fwd(xg, w1, bp.cuda(), 1, 0, epilogue=L.EPI_IGDN if inverse else L.EPI_GDN, aux=xg, square_input=True,
                            pre=norm)
        torch.testing.assert_close(yg.cpu().permute(0, 3, 1, 2), y.detach(), rtol=3e-6, atol=1e-6)
        t = ops.gdn_bwd_t(gyg, xg, norm, inverse)
        acc = ops.conv2d_fwd(t, w1t, None, 1, 0)
        dx = ops.gdn_bwd_dx(gyg, xg, norm, acc, inverse)
        torch.testing.assert_close(dx.cpu().permute(0, 3, 1, 2), x.grad, rtol=2e-5, atol=2e-6)
        dgp = ops.reduce_slabs(ops.conv2d_wgrad(xg, t, (Cc, 1, 1, Cc), 1, 0, square_input=True)).reshape(Cc, Cc).cpu()
        # chain through gamma' = max(gamma, bound)^2 - pedestal on the host for the comparison
        bound = float(O._GAMMA_REPARAM.lower_bound.bound)
        lb = torch.clamp(gamma.detach(), min=bound)
        go = dgp * 2 * lb
        dg = torch.where((gamma.detach() >= bound) | (go < 0), go, torch.zeros_like(go))
        torch.testing.assert_close(dg, gamma.grad, rtol=2e-5, atol=2e-5)
        x.grad = None


@pytest.mark.parametrize("tag,tconv", [("conv", False), ("tconv", True), ("gdn", False), ("vec", False)])
@pytest.mark.parametrize("method", ["max", "mse", "l1", "l2"])
@pytest.mark.parametrize("cw", [True, False])
@pytest.mark.parametrize("bits", [8, 4])
def test_product_uaq_matches_reference_goldens(golden_dir, tag, tconv, method, cw, bits):
    """The drop-in UniformAffineQuantizer (HIP min/max + fake-quant kernels, vectorised search inits) against the vectors the
    reference's own UniformAffineQuantizer produced (tests/golden/quantizers.npz)."""
    import os
    from quantization.quantizer import UniformAffineQuantizer
    qz = np.load(os.path.join(golden_dir, "quantizers.npz"))
    w = torch.from_numpy(qz[f"w_{tag}"]).cuda()
    key = f"uaq_{tag}_{method}_{'cw' if cw else 'lw'}_{bits}"
    q = UniformAffineQuantizer(n_bits=bits, channel_wise=cw, scale_method=method, tconv=tconv)
    out = q(w)
    torch.cuda.synchronize()
    d_ref, z_ref = qz[key + "_delta"], qz[key + "_zp"]
    np.testing.assert_allclose(q.delta.cpu().numpy().reshape(-1), d_ref.reshape(-1), rtol=2e-7, atol=0)
    np.testing.assert_array_equal(q.zero_point.cpu().numpy().reshape(-1), z_ref.reshape(-1))
    if cw and w.dim() > 1:
        assert tuple(q.delta.shape) == tuple(d_ref.shape)
    assert tuple(out.shape) == tuple(w.shape)
    np.testing.assert_allclose(out.cpu().numpy(), qz[key + "_out"], rtol=3e-7, atol=float(d_ref.max()) * 1e-6)  # 1 ulp of delta scales every level


@pytest.mark.parametrize("tag", ["conv", "lin"])
@pytest.mark.parametrize("method,sym", [("max_scale", False), ("max", True), ("max_scale", True)])
@pytest.mark.parametrize("cw", [True, False])
@pytest.mark.parametrize("bits", [8, 6, 4])
def test_product_uaq_scaled_and_symmetric_init(golden_dir, tag, method, sym, cw, bits):
    """'max_scale' / symmetric inits of the product quantiser vs the reference's vectors: the range is scaled and mirrored in
    double and rounded to fp32 once, so delta and the (tie-prone) zero points agree bit for bit."""
    import os
    from quantization.quantizer import UniformAffineQuantizer
    fx = np.load(os.path.join(golden_dir, "quantizer_ties.npz"))
    w = torch.from_numpy(fx[f"w_{tag}"]).cuda()
    key = f"uaq_{tag}_{method}{'_sym' if sym else ''}_{'cw' if cw else 'lw'}_{bits}"
    q = UniformAffineQuantizer(n_bits=bits, symmetric=sym, channel_wise=cw, scale_method=method)
    out = q(w)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(q.delta.cpu().numpy().reshape(-1), fx[key + "_delta"].reshape(-1))
    np.testing.assert_array_equal(q.zero_point.cpu().numpy().reshape(-1), fx[key + "_zp"].reshape(-1))
    d = float(fx[key + "_delta"].max())
    diff = np.abs(out.cpu().numpy() - fx[key + "_out"])
    assert float((diff > 1e-7).mean()) < 2e-3 and float(diff.max()) <= d * (1 + 1e-5)


@pytest.mark.parametrize("tag,tconv", [("conv", False), ("tconv", True), ("lin", False)])
@pytest.mark.parametrize("cw", [True, False])
@pytest.mark.parametrize("bits", [8, 6, 4])
def test_product_uaq_zero_points_on_symmetric_ranges(golden_dir, tag, tconv, cw, bits):
    """The HIP min/max init on channels whose range is exactly symmetric (-min/delta on x.5) reproduces the reference's zero
    points bit for bit (tests/golden/quantizer_ties.npz, made by the reference's UniformAffineQuantizer)."""
    import os
    from quantization.quantizer import UniformAffineQuantizer
    fx = np.load(os.path.join(golden_dir, "quantizer_ties.npz"))
    w = torch.from_numpy(fx[f"w_{tag}"]).cuda()
    key = f"uaq_{tag}_{'cw' if cw else 'lw'}_{bits}"
    q = UniformAffineQuantizer(n_bits=bits, channel_wise=cw, scale_method="max", tconv=tconv)
    out = q(w)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(q.delta.cpu().numpy().reshape(-1), fx[key + "_delta"].reshape(-1))
    np.testing.assert_array_equal(q.zero_point.cpu().numpy().reshape(-1), fx[key + "_zp"].reshape(-1))
    d = float(fx[key + "_delta"].max())
    # values within an ulp of a rounding boundary may land one level apart; everything else is exact
    diff = np.abs(out.cpu().numpy() - fx[key + "_out"])
    assert float((diff > 1e-7).mean()) < 2e-3 and float(diff.max()) <= d * (1 + 1e-5)


@pytest.mark.parametrize("bits", [10, 6, 16])
def test_actquant_other_widths_match_oracle(bits):
    """Activation grids other than the reference's hard-wired 8 bits (BASELINE config "W10A10"): the kernel with n_bits against
    the oracle's Handle_Parameter(b_w) (quantizer.py:81-97 with its `b_w` argument lifted), 4-D and 3-D layouts, and through
    UniformAffineQuantizer(dynamic_bits=...)."""
    from oracle import rdo_oracle as O
    from quantization.quantizer import ActQuantizer, UniformAffineQuantizer
    g = torch.Generator().manual_seed(bits)
    a4 = torch.randn(2, 5, 6, 7, generator=g) * 3
    a4[:, 2] = 0.25
    a3 = torch.randn(2, 9, 6, generator=g)
    for a in (a4, a3):
        ref = O.act_quant(a, b_w=bits)
        got = ActQuantizer(a.cuda(), bits).cpu()
        step = float((a.max() - a.min())) / (2 ** bits - 1)
        diff = (got - ref).abs()
        # a value within fp32 rounding of a grid boundary may land one level apart
        assert float((diff > 1e-6).float().mean()) < 5e-3 and float(diff.max()) <= step * (1 + 1e-4)
        q = UniformAffineQuantizer(n_bits=bits, channel_wise=True, scale_method="max", leaf_param=False, act=True, dynamic_bits=bits)
        assert torch.equal(q(a.cuda(), True).cpu(), got)
    # default stays the reference's 8 bits whatever n_bits says (quantizer.py:81,158-159)
    q8 = UniformAffineQuantizer(n_bits=4, channel_wise=True, scale_method="max", leaf_param=False, act=True)
    assert torch.equal(q8(a4.cuda(), True).cpu(), ActQuantizer(a4.cuda()).cpu())


@pytest.mark.parametrize("tag", ["a4", "a3", "a2"])
def test_product_actquantizer_matches_reference_goldens(golden_dir, tag):
    import os
    from quantization.quantizer import ActQuantizer
    qz = np.load(os.path.join(golden_dir, "quantizers.npz"))
    out = ActQuantizer(torch.from_numpy(qz[f"act_{tag}_in"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), qz[f"act_{tag}_out"], rtol=0, atol=3e-7)


@pytest.mark.parametrize("tag,tconv", [("conv", False), ("tconv", True), ("gdn", False)])
def test_product_adaround_quantizer_matches_reference_goldens(golden_dir, tag, tconv):
    import os
    from quantization.quantizer import AdaRoundQuantizer, UniformAffineQuantizer
    qz = np.load(os.path.join(golden_dir, "quantizers.npz"))
    w = torch.from_numpy(qz[f"w_{tag}"]).cuda()
    uaq = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max", tconv=tconv)
    uaq(w)
    ada = AdaRoundQuantizer(uaq=uaq, round_mode="learned_hard_sigmoid", weight_tensor=w)
    assert tuple(ada.alpha.shape) == tuple(w.shape)
    np.testing.assert_allclose(ada.alpha.detach().cpu().numpy(), qz[f"ada_{tag}_alpha0"], rtol=3e-6, atol=3e-6)
    with torch.no_grad():
        ada.alpha.copy_(torch.from_numpy(qz[f"ada_{tag}_alpha"]).cuda())
    tol = float(uaq.delta.max()) * 1e-4
    ada.soft_targets = True
    np.testing.assert_allclose(ada(w).cpu().numpy(), qz[f"ada_{tag}_soft"], rtol=0, atol=tol)
    ada.soft_targets = False
    np.testing.assert_allclose(ada(w).cpu().numpy(), qz[f"ada_{tag}_hard"], rtol=0, atol=tol)


@pytest.mark.parametrize("cfg", [(2, 9, 7, 12, 20, 5, 2, 2, 1), (1, 8, 8, 32, 16, 3, 2, 1, 1), (2, 6, 5, 8, 8, 4, 2, 1, 0),
                                 (1, 7, 7, 16, 24, 3, 1, 1, 0)])
def test_quantmodule_conv_transpose_matches_torch(cfg):
    """QuantModule(nn.ConvTranspose2d) forward (zero-insertion + HIP conv) vs F.conv_transpose2d in fp64, FP and W8."""
    from oracle import rdo_oracle as O
    from quantization.quant_layer import QuantModule
    B, H, W, Cin, Cout, K, s, p, op = cfg
    torch.manual_seed(sum(cfg))
    m = torch.nn.ConvTranspose2d(Cin, Cout, K, stride=s, padding=p, output_padding=op)
    x = torch.randn(B, Cin, H, W)
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    w_cpu, b_cpu = m.weight.detach().clone(), m.bias.detach().clone()
    qm = QuantModule(m, wq, dict(wq, leaf_param=False)).cuda()
    ref = F.conv_transpose2d(x.double(), w_cpu.double(), b_cpu.double(), stride=s, padding=p, output_padding=op)
    y = qm(x.cuda())
    assert tuple(y.shape) == tuple(ref.shape)
    assert _rel(y.cpu(), ref) < 5e-6
    qm.set_quant_state(True, False)
    yq = qm(x.cuda())
    d, z = O.uaq_init(w_cpu, 8, True, "max", tconv=True)
    wq_ref = O.uaq_fakequant(w_cpu, d, z, 256)
    refq = F.conv_transpose2d(x.double(), wq_ref.double(), b_cpu.double(), stride=s, padding=p, output_padding=op)
    assert _rel(yq.cpu(), refq) < 5e-6


def test_quantmodule_layernorm_and_linear_match_torch():
    from quantization.quant_layer import QuantModule
    torch.manual_seed(5)
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    x = torch.randn(3, 50, 96)
    ln = torch.nn.LayerNorm(96)
    with torch.no_grad():
        ln.weight.add_(0.3 * torch.randn(96)); ln.bias.add_(0.1 * torch.randn(96))
    ref_ln = F.layer_norm(x, (96,), ln.weight.detach().clone(), ln.bias.detach().clone())
    q = QuantModule(ln, wq, dict(wq, leaf_param=False)).cuda()
    torch.testing.assert_close(q(x.cuda()).cpu(), ref_ln, rtol=2e-5, atol=2e-5)
    lin = torch.nn.Linear(96, 40)
    ref = F.linear(x.double(), lin.weight.detach().double(), lin.bias.detach().double())
    q = QuantModule(lin, wq, dict(wq, leaf_param=False)).cuda()
    assert _rel(q(x.cuda()).cpu(), ref) < 5e-6


def test_integration_md_ctypes_stub_runs_as_written():
    """The ctypes binding shown in INTEGRATION.md section B (what a reference maintainer would add) is executed verbatim
    against the built library and compared with the oracle's conv (torch CPU fp32)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# quantization/_hip\.py.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("librdoptq_hip.so")', f'C.CDLL("{os.path.join(root, "rdo-ptq_amd", "lib", "librdoptq_hip.so")}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 12, 10, 8, generator=g)
    w = torch.randn(16, 3, 3, 8, generator=g) * 0.2
    b = torch.randn(16, generator=g)
    y = ns["conv2d_nhwc"](x.cuda(), w.cuda(), b.cuda(), 2, 1, lrelu=True)
    torch.cuda.synchronize()
    ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=2, padding=1), 0.01).permute(0, 2, 3, 1)
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(32, 3, 3, 48), (16, 1, 1, 16), (24, 5, 5, 32)])
def test_adaround_step_writes_fragment_ordered_planes(shape):
    """rdo_adaround_step's optional bf16 planes of the new wq / wd equal rdo_split_bf16x3_conv of those tensors (wd is the
    weight [Cin][KH][KW][Cout] of the dgrad conv)."""
    from hipops import ops
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(shape, generator=g) * 0.1).cuda()
    co, kh, kw, ci = shape
    delta, zp = ops.uaq_init_minmax(w.reshape(co, -1), 256)
    d = ops.ada_desc(w)
    alpha = ops.adaround_init_alpha(d, w, delta)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    wq, wd = torch.empty_like(w), torch.empty_like(w)
    wqp = torch.zeros((3,) + shape, dtype=torch.int16, device="cuda")
    wdp = torch.zeros((3,) + shape, dtype=torch.int16, device="cuda")
    slabs = (torch.randn((2,) + shape, generator=g) * 1e-2).cuda()
    sched = torch.tensor([[10.0, 1.0, 1e-3, 1.0]], device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    log = torch.zeros(1, 32, device="cuda")
    ops.adaround_step(d, w, delta, zp, slabs, 1.0, 0.01, sched, it, alpha, m, v, wq, wd, log, wqp, wdp)
    torch.cuda.synchronize()
    assert torch.equal(wqp, ops.split_bf16x3(wq))
    wd4 = wd.reshape(ci, kh, kw, co)
    torch.testing.assert_close(wd4, wq.flip(1, 2).permute(3, 1, 2, 0).contiguous(), rtol=0, atol=0)
    assert torch.equal(wdp.reshape(3, -1), ops.split_bf16x3(wd4).reshape(3, -1))


# ---- transposed conv without zero insertion (phase weights + pixel shuffle) ----------------------------------------------------------
@pytest.mark.parametrize("K,s,p,op,Cin,Cout,H,W", [(5, 2, 2, 1, 16, 24, 9, 12), (3, 2, 1, 1, 8, 8, 7, 5), (4, 2, 1, 0, 12, 4, 6, 6),
                                                  (6, 3, 2, 1, 4, 8, 5, 4), (5, 2, 2, 0, 8, 8, 6, 6), (5, 2, 2, 1, 192, 192, 16, 16)])
def test_conv_transpose2d_phase_form_matches_torch(K, s, p, op, Cin, Cout, H, W):
    """ops.conv_transpose2d (stride-1 conv with the phase weight + pixel shuffle where output = stride x input, zero insertion
    otherwise -- the (5, 2, 2, 0) case) against F.conv_transpose2d, with bias and a fused LeakyReLU; rdo_tconv_fold inverts
    rdo_tconv_expand on the taps."""
    import torch.nn.functional as F
    from hipops import _lib as L
    from hipops import ops
    g = torch.Generator(device="cuda").manual_seed(K * 100 + s * 10 + p)
    x = torch.randn(2, H, W, Cin, device="cuda", generator=g)
    wt = torch.randn(Cin, Cout, K, K, device="cuda", generator=g) / (Cin * K * K / (s * s)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    rows = wt.permute(1, 2, 3, 0).contiguous()                                   # to_rows(W, tconv=True)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=s, padding=p, output_padding=op)
    for epi in (L.EPI_NONE, L.EPI_LRELU):
        want = (F.leaky_relu(ref, 0.01) if epi == L.EPI_LRELU else ref).permute(0, 2, 3, 1)
        got = ops.conv_transpose2d(x, rows, b, s, p, op, epilogue=epi)
        assert got.shape == want.shape
        assert float((got.double() - want).abs().max()) <= 3e-6 * float(want.abs().max())
    ph = ops.TconvPhase.get(K, s, p, op, False, x.device)
    assert (ph is None) == (op != s + 2 * p - K)
    if ph is not None:
        wp = ops.tconv_expand(rows, ph)
        assert wp.shape == (Cout * s * s, ph.Kp, ph.Kp, Cin)
        assert int((wp != 0).sum()) <= rows.numel()                            # every tap once, the rest structural zeros
        back = ops.tconv_fold(wp.unsqueeze(0).contiguous(), ph, Cout, Cin)
        assert torch.equal(back[0], rows)


def test_plan_run_then_equals_two_runs(ops):
    """rdo_plan_run_then: the ops of two recorded plans in ONE graph launch (what the data-parallel host loop enqueues between two
    collectives) leave the same values as the two plans run one after the other, eagerly and as graphs; re-recording the partner
    re-captures the chained graph."""
    from hipops.plan import Plan
    g = torch.Generator().manual_seed(21)
    a = torch.randn(4, 64, generator=g).cuda()
    b = torch.randn(4, 64, generator=g).cuda()
    t1, t2, t3 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    p, q = Plan(), Plan()
    with p.record():
        ops.add(a, b, out=t1)              # t1 = a + b
        ops.lrelu(t1, out=t2)              # t2 = lrelu(t1)
    with q.record():
        ops.add(t2, a, out=t3)             # t3 = t2 + a
        ops.add(t3, t3, out=a)             # a  = 2 t3   (feeds the next p)
    assert p.num_ops == 2 and q.num_ops == 2

    def reference(a0, n):
        x = a0.clone()
        for _ in range(n):
            t = torch.nn.functional.leaky_relu(x + b, 0.01)
            x = 2 * (t + x)
        return x
    a0 = a.clone()
    for graph in (False, True):
        a.copy_(a0)
        for _ in range(3):
            p.run_then(q, graph=graph)
        torch.cuda.synchronize()
        torch.testing.assert_close(a, reference(a0, 3), rtol=0, atol=0)
    with q.record():                        # a different partner op list: the chained graph must follow it
        ops.add(t2, a, out=t3)
        ops.add(t3, b, out=a)               # a = t3 + b
    a.copy_(a0)
    p.run_then(q, graph=True)
    torch.cuda.synchronize()
    t = torch.nn.functional.leaky_relu(a0 + b, 0.01)
    torch.testing.assert_close(a, (t + a0) + b, rtol=0, atol=0)


@pytest.mark.parametrize("planes", [False, True])
def test_step_launch_assembles_the_next_mini_batch(planes):
    """rdo_adaround_step_batch_gather (round 6): the step launch of iteration i leaves in `out` / `out_planes` exactly what a stand-alone
    rdo_gather_qdrop(_h2) produces for iteration i + 1, steps the weights exactly as rdo_adaround_step_batch does, moves the counter by
    the hand-over (reads the published word, stores it + 1 into the real counter), and skips the gather behind the last row of the
    index table."""
    from hipops import ops
    g = torch.Generator().manual_seed(5)
    n_img, B, H, C, iters = 6, 2, 8, 32, 3
    cq = torch.randn(n_img, H, H, C, generator=g).cuda()
    cf = torch.randn(n_img, H, H, C, generator=g).cuda()
    idx = torch.stack([torch.randperm(n_img, generator=g)[:B] for _ in range(iters)]).to(torch.int32).cuda()
    shape = (32, 3, 3, 32)
    w = (torch.randn(shape, generator=g) * 0.1).cuda()
    delta, zp = ops.uaq_init_minmax(w.reshape(shape[0], -1), 256)
    d = ops.ada_desc(w)
    slabs = (torch.randn((3,) + shape, generator=g) * 1e-2).cuda()
    sched = torch.tensor([[10.0, 1.0, 1e-3, 1.0]] * iters, device="cuda")

    def fresh():
        alpha = ops.adaround_init_alpha(d, w, delta)
        return dict(d=d, w=w, delta=delta, zp=zp, slabs=slabs, alpha=alpha, m=torch.zeros_like(w), v=torch.zeros_like(w), wq=torch.empty_like(w),
                    wd=torch.empty_like(w))
    for it0 in (0, iters - 1):
        a, b = fresh(), fresh()
        word = torch.full((2,), it0, dtype=torch.int32, device="cuda")                 # [real counter, published copy]
        log_a, log_b = torch.zeros(iters, 32, device="cuda"), torch.zeros(iters, 32, device="cuda")
        out = torch.full((B, H, H, C), 7.0, device="cuda")
        xp = ops.h2_empty(out.shape, out.device, 4.0) if planes else None
        if planes:
            xp.t.fill_(77)
        ops.adaround_step_batch([a], 1.0, 0.01, sched, word[1:2], log_a, iter_shadow=word[0:1],
                                gather=dict(cache_q=cq, cache_fp=cf, idx_table=idx, B=B, batch_offset=0, prob=0.5, seed=99, out=out, out_planes=xp))
        ref_word = torch.full((1,), it0, dtype=torch.int32, device="cuda")
        ops.adaround_step_batch([b], 1.0, 0.01, sched, ref_word, log_b)
        torch.cuda.synchronize()
        for k in ("alpha", "m", "v", "wq", "wd"):
            assert torch.equal(a[k], b[k]), k
        assert torch.equal(log_a, log_b)
        assert word.tolist() == [it0 + 1, it0]
        if it0 + 1 < iters:
            nxt = torch.full((1,), it0 + 1, dtype=torch.int32, device="cuda")
            want = torch.empty_like(out)
            if planes:
                wp = ops.h2_empty(out.shape, out.device, 4.0)
                ops.gather_qdrop_h2(cq, cf, idx, nxt, B, 0.5, 99, want, wp)
                torch.cuda.synchronize()
                assert torch.equal(xp.t, wp.t)
            else:
                ops.gather_qdrop(cq, cf, idx, nxt, B, 0.5, 99, want)
            torch.cuda.synchronize()
            assert torch.equal(out, want)
        else:
            assert bool((out == 7.0).all()) and (xp is None or bool((xp.t == 77).all()))     # behind the last iteration: nothing is gathered


@pytest.mark.parametrize("form", [1, 0])
@pytest.mark.parametrize("B,H,K,N,act,spread", [(4, 16, 192, 96, 2, 0), (4, 16, 96, 192, 0, 0), (2, 8, 96, 96, 1, 1), (4, 64, 192, 96, 2, 1),
                                                (4, 64, 192, 192, 0, 0), (4, 64, 96, 192, 1, 1), (3, 16, 64, 32, 1, 0), (4, 32, 160, 288, 2, 1)])
def test_unit1x1_matches_float64(B, H, K, N, act, spread, form):
    """rdo_unit1x1 (round 6): forward of a 1 x 1 conv with bias, activation, 2 x lp_loss against the cached target rows picked by the
    device index table, activation backward and the weight-gradient slabs in one launch, against float64 -- loss to 1e-6 relative, the
    summed slabs to 2e-6 of the largest gradient entry.  Both forms: split-fp16 with on-the-fly scales (Cout in blocks of 96) and exact
    fp32 MFMA (an fmaf chain per accumulator).  `spread`: token magnitudes over several decades, channel scales over two, targets 1e-2 from
    the outputs (small residuals) -- what per-token / per-channel / per-tile scales are for."""
    from hipops import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + H + K + N)
    n_img = B + 2
    x = torch.randn(B, H, H, K, generator=g)
    w = torch.randn(N, 1, 1, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    tgt = torch.randn(n_img, H, H, N, generator=g)
    idx = torch.stack([torch.randperm(n_img, generator=g)[:B] for _ in range(3)]).to(torch.int32)
    if spread:
        x = x * torch.exp(2.5 * torch.randn(B, H, H, 1, generator=g))
        w = w * torch.exp(1.5 * torch.randn(N, 1, 1, 1, generator=g))
        with torch.no_grad():
            pre = (x.double().reshape(-1, K) @ w.double().reshape(N, K).t() + b.double()).reshape(B, H, H, N)
            out = {0: pre, 1: F.leaky_relu(pre, 0.01), 2: F.relu(pre)}[act]
            tgt[idx[1].long()] = (out * (1 + 1e-2 * torch.randn(out.shape, generator=g, dtype=torch.float64))).float()
    it = torch.tensor([1], dtype=torch.int32)
    assert ops.unit1x1_supported(B * H * H, K, N)
    was = ops.unit1x1_form(form)
    try:
        ns = ops.unit1x1_nslab(B * H * H, N)
        slabs = torch.full((ns, N, 1, 1, K), 7.0, device="cuda")
        log = torch.zeros(3, 32, device="cuda")
        ops.unit1x1(x.cuda(), w.cuda(), b.cuda(), tgt.cuda(), idx.cuda(), it.cuda(), 2.0, act, log, slabs)
        torch.cuda.synchronize()
    finally:
        ops.unit1x1_form(was)
    x64 = x.double().reshape(-1, K)
    w64 = w.double().reshape(N, K).requires_grad_(True)
    pre = x64 @ w64.t() + b.double()
    out = {0: pre, 1: F.leaky_relu(pre, 0.01), 2: F.relu(pre)}[act]
    y = tgt[idx[1].long()].double().reshape(-1, N)
    loss = 2.0 * ((out - y) ** 2).sum(1).mean()
    loss.backward()
    got_loss = float(log[1].sum())
    assert float(log[0].abs().sum()) == 0.0 and float(log[2].abs().sum()) == 0.0
    # (with targets 1e-2 from the outputs the residual itself carries the fp32 rounding of `out`: the loss is a sum of squares of
    # differences of nearly equal numbers, so its bar is looser there -- for both forms alike)
    assert abs(got_loss - float(loss)) <= (2e-4 if spread else 1e-6) * abs(float(loss)) + 1e-9, (got_loss, float(loss))
    gw = slabs.double().sum(0).reshape(N, K).cpu()
    err = float((gw - w64.grad).abs().max()) / float(w64.grad.abs().max())
    print(f"form {form} B {B} {H}x{H} {K}->{N} act {act} spread {spread}: loss rel {abs(got_loss - float(loss)) / abs(float(loss)):.2e}, grad {err:.2e}")
    assert err <= (2e-4 if spread else 2e-6)


@pytest.mark.parametrize("form", [1, 0])
def test_unit1x1_ragged_last_chunk(form):
    """290 token tiles over chunks of 3 (97 slabs, the last chunk holds two tiles) and a second block of output channels: the workgroups of
    the last chunk stop behind their tokens (the split-fp16 form re-requests its last tile instead of reading past the mini-batch), every
    slab is written, and the summed gradient is float64's."""
    from hipops import ops
    B, Hh, Ww, K, N = 5, 32, 58, 192, 192
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Hh, Ww, K, generator=g)
    w = torch.randn(N, 1, 1, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    tgt = torch.randn(B + 1, Hh, Ww, N, generator=g)
    idx = torch.stack([torch.randperm(B + 1, generator=g)[:B] for _ in range(2)]).to(torch.int32)
    it = torch.tensor([1], dtype=torch.int32)
    M = B * Hh * Ww
    was = ops.unit1x1_form(form)
    try:
        ns = ops.unit1x1_nslab(M, N)
        assert (M // 32) % ((M // 32 + ns - 1) // ns) != 0           # the last chunk really is short
        slabs = torch.full((ns, N, 1, 1, K), float("nan"), device="cuda")
        guard = torch.full((4096,), 3.0, device="cuda")              # (allocated right behind x on the device: not a proof, a tripwire)
        xg = x.cuda()
        log = torch.zeros(2, 32, device="cuda")
        ops.unit1x1(xg, w.cuda(), b.cuda(), tgt.cuda(), idx.cuda(), it.cuda(), 2.0, 1, log, slabs)
        again = torch.empty_like(slabs)
        for _ in range(3):                                           # fixed summation order: the slabs are bit-reproducible
            ops.unit1x1(xg, w.cuda(), b.cuda(), tgt.cuda(), idx.cuda(), it.cuda(), 2.0, 1, torch.zeros_like(log), again)
            assert torch.equal(again, slabs)
        torch.cuda.synchronize()
    finally:
        ops.unit1x1_form(was)
    assert bool(torch.isfinite(slabs).all()) and bool((guard == 3.0).all())
    x64, w64 = x.double().reshape(-1, K), w.double().reshape(N, K).requires_grad_(True)
    pre = x64 @ w64.t() + b.double()
    out = torch.nn.functional.leaky_relu(pre, 0.01)
    loss = 2.0 * ((out - tgt[idx[1].long()].double().reshape(-1, N)) ** 2).sum(1).mean()
    loss.backward()
    assert abs(float(log[1].sum()) - float(loss)) <= 1e-6 * float(loss)
    gw = slabs.double().sum(0).reshape(N, K).cpu()
    assert float((gw - w64.grad).abs().max()) <= 2e-6 * float(w64.grad.abs().max())
