"""H2 tensors (two-way fp16 splits of the power-of-two-scaled values, the operand format of the LDS-DMA conv kernels) and the fused
unit-tail kernels: each entry point against fp64 and against the chain of separate kernels it replaces (which the rest of the suite
pins to fp64 / torch / the oracle).  Accuracy bar of the three-product fp16 GEMMs: the error against fp64 of the fp32-input kernels
(exact bf16 split, six products) -- i.e. fp32-chain accuracy.  The fused tails execute the same fp32 operations element by element."""
import pytest
import torch

pytestmark = pytest.mark.gpu
N = 192


@pytest.fixture(scope="module")
def ops():
    from hipops import ops as o
    return o


@pytest.fixture(scope="module")
def L():
    from hipops import _lib
    return _lib


def _close_planes(ops, planes, ref, tol=2.0 ** -22):
    """the planes hold `ref` to the two-way split's accuracy (2^-24 relative per element; an absolute floor of 2^-25 / scale)"""
    got = ops.h2_to_float(planes, ref.shape)
    err = (got.double() - ref.double()).abs()
    bound = tol * ref.double().abs() + 2.0 ** -24 / planes.scale
    assert bool((err <= bound).all()), float((err - bound).max())


def test_split_h2_round_trip_and_overflow_flag(ops):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(4, 16, 16, 64, device="cuda", generator=g) * torch.logspace(-6, 2, 64, device="cuda")
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3e-20, 300.0, 1e-30, -7.5], device="cuda")
    ops.h2_overflow(reset=True)
    pl = ops.split_h2(x)
    assert pl.t.shape == (2, 4, 4 * 16 * 16, 16) and pl.t.dtype == torch.int16
    amax = float(x.abs().max())
    assert 64.0 < amax * pl.scale <= 128.0                      # the power-of-two scale puts the largest magnitude in (2^6, 2^7]
    _close_planes(ops, pl, x)
    # plane 0 is the RNE fp16 of x * s, stored slice-major [C/16][pixel][16]; plane 1 the RNE fp16 of the exact remainder
    xs = x * pl.scale
    p0 = xs.to(torch.float16)
    assert torch.equal(pl.t[0], p0.view(torch.int16).reshape(-1, 4, 16).permute(1, 0, 2))
    p1 = (xs - p0.float()).to(torch.float16)
    assert torch.equal(pl.t[1], p1.view(torch.int16).reshape(-1, 4, 16).permute(1, 0, 2))
    assert not ops.h2_overflow(reset=True)
    # a scale that pushes values past 65504 raises the sticky flag (and only then)
    ops.split_h2(x, scale=2.0 ** 12)
    assert ops.h2_overflow(reset=True)
    assert not ops.h2_overflow(reset=True)
    with pytest.raises(ValueError):
        ops.split_h2(x, scale=3.0)


def conv_fp64(x, w, b, s, p):
    import torch.nn.functional as F
    y = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), None if b is None else b.double().cpu(), stride=s, padding=p)
    return y.permute(0, 2, 3, 1).cuda()


CONV_SHAPES = [(128, N, N, 3, 1, 1), (64, N, 4 * N, 3, 1, 1), (64, N, N, 3, 1, 1), (128, N, N, 3, 2, 1), (128, N, N, 1, 1, 0),
               (64, 32, N, 5, 1, 2), (128, N, 320, 5, 2, 2), (32, N, 4 * N, 3, 1, 1), (70, N, N, 3, 1, 1)]


@pytest.mark.parametrize("H,Cin,Cout,K,s,p", CONV_SHAPES)
def test_conv_h2_matches_fp64_like_the_fp32_input_kernel(ops, L, H, Cin, Cout, K, s, p):
    g = torch.Generator(device="cuda").manual_seed(H + Cout + K)
    x = torch.randn(4, H, H, Cin, device="cuda", generator=g) * 3
    w = torch.randn(Cout, K, K, Cin, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    assert ops.conv_h2_supported(tuple(x.shape), tuple(w.shape), s, p)
    ops.h2_overflow(reset=True)
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    ref6 = ops.conv2d_fwd(x, w, b, s, p, wplanes=ops.split_bf16x3(w))           # fp32-input kernel: exact bf16 split, six products
    out = torch.empty_like(ref6)
    opl = ops.h2_empty(ref6.shape, "cuda", ops.pow2_scale(ref6.abs().max()))
    ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out=out, out_planes=opl)
    ref = conv_fp64(x[:1], w, b, s, p)                                          # first image in fp64 (CPU)
    scale = float(ref.abs().max())
    e = float((out[:1].double() - ref).abs().max()) / scale
    e6 = float((ref6[:1].double() - ref).abs().max()) / scale
    assert e < 4e-6 and e < 2.0 * e6 + 1e-7, (e, e6)
    assert float((out - ref6).abs().max()) <= 6e-6 * float(ref6.abs().max())
    _close_planes(ops, opl, out)
    # planes only (no fp32 output at all): the same planes
    opl2 = ops.h2_empty(ref6.shape, "cuda", opl.scale)
    ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, s, p, out_planes=opl2)
    assert torch.equal(opl2.t, opl.t)
    assert not ops.h2_overflow(reset=True)


def test_conv_h2_scales_do_not_change_the_result(ops):
    """Power-of-two scales only move exponents: wherever max |x s| lies between 1 and 2^13 the result stays at fp32-chain accuracy
    (the elements whose second plane is a fp16 denormal change with the scale, far below that level)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4, 128, 128, N, device="cuda", generator=g)
    w = torch.randn(N, 3, 3, N, device="cuda", generator=g) / 41.6
    outs = []
    for sx, sw in ((16.0, 128.0), (1024.0, 16384.0), (0.25, 4.0)):
        out = torch.empty(4, 128, 128, N, device="cuda")
        ops.conv2d_fwd_h2(ops.split_h2(x, scale=sx), tuple(x.shape), tuple(w.shape), ops.split_h2_conv(w, scale=sw), None, 1, 1, out=out)
        outs.append(out)
    ref = conv_fp64(x[:1], w, None, 1, 1)
    for o in outs:
        assert float((o[:1].double() - ref).abs().max()) <= 2.5e-6 * float(ref.abs().max())
    assert float((outs[1] - outs[0]).abs().max()) <= 5e-7 * float(outs[0].abs().max())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 128, 128, N, N), (4, 64, 64, N, 4 * N), (2, 128, 256, 64, N), (1, 256, 256, 32, 2 * N),
                                            (4, 64, 64, N, N), (4, 32, 32, N, 4 * N), (2, 64, 96, 64, 320), (3, 64, 64, 48, 256),
                                            (1, 256, 256, 32, 48), (8, 96, 96, 48, 208), (2, 176, 208, 32, 16), (3, 112, 240, 80, 320)])
def test_conv_h2_halo_kernel_equals_per_tap_kernel(ops, L, B, H, W, Cin, Cout):
    """3x3 / stride 1 / pad 1 on 16 x 16 patches with the halo tile resident in LDS (the 16-channel-stage form, h2_k32 = 0: the kernel
    of the shapes whose Cin is not a multiple of 32): same K order and accumulation as the per-tap kernel, so the same bits -- borders
    (zero halo), all epilogue outputs and the H2 planes included."""
    saved_k32 = ops.set_tuning("h2_k32", 0)
    try:
        _halo_vs_per_tap(ops, L, B, H, W, Cin, Cout)
    finally:
        ops.set_tuning("h2_k32", saved_k32)


def _halo_vs_per_tap(ops, L, B, H, W, Cin, Cout):
    g = torch.Generator(device="cuda").manual_seed(H + Cin)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    got = {}
    for halo in (0, 1):
        ops.set_tuning("x6p_halo", halo)
        try:
            out, pre = torch.empty(B, H, W, Cout, device="cuda"), torch.empty(B, H, W, Cout, device="cuda")
            opl = ops.h2_empty(out.shape, "cuda", 16.0)
            ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, epilogue=L.EPI_LRELU, residual=res, out=out, pre=pre,
                              out_planes=opl)
            got[halo] = (out, pre, opl.t)
        finally:
            ops.set_tuning("x6p_halo", 1)
    if Cout * (B * H * W // 256) >= 192 * 192 or Cout * (B * H * W // 256) < 160 * 64:
        # both runs walk K in one piece (halo kernel with 256 x 192 tiles / per-tap kernel without a K split): the same bits
        for u, v in zip(got[0], got[1]):
            assert torch.equal(u, v)
    else:
        # the halo kernel runs 256 x 64 tiles over the whole K, the per-tap kernel splits K and sums partial tiles: summation order
        for u, v in zip(got[0][:2], got[1][:2]):
            assert float((u - v).abs().max()) <= 2e-6 * float(v.abs().max())
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert (got[1][1].double() - ref).abs().max() <= 2e-6 * ref.abs().max()


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 128, 128, N, N), (4, 64, 64, N, 4 * N), (4, 64, 64, N, N), (4, 32, 32, N, 4 * N), (2, 64, 96, 64, 320),
                                            (1, 256, 256, 32, 2 * N), (2, 128, 256, 64, N), (3, 112, 240, 96, 320)])
def test_conv_h2_k32_halo_kernel_equals_k16_halo_kernel(ops, L, B, H, W, Cin, Cout):
    _k32_vs_k16(ops, L, B, H, W, Cin, Cout, 1)


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 32, 32, N, N), (8, 32, 32, N, N), (6, 32, 32, N, N), (4, 16, 16, N, 4 * N)])
def test_conv_h2_k32_halo_kernel_with_k_split(ops, L, B, H, W, Cin, Cout, mode):
    """Few tiles (the 32^2 convs): the K32 halo kernel split over slice pairs + the second pass -- h2_k32 = 2: 256 x 64 tiles, the smallest
    split that reaches 192 workgroups; 3 (default): 256 x 48 tiles and the largest even split that keeps all workgroups resident at once
    (4 x 32^2 192 -> 192: 64 tiles x 3 = 192 workgroups of 18 stages, 35.8 -> 21.1 us) -- against the path these shapes take otherwise
    (12-way K split of the 16-channel per-tap kernel)."""
    if not ops.conv_h2_supported((B, H, W, Cin), (Cout, 3, 3, Cin), 1, 1):
        pytest.skip("shape not on the plane path")
    _k32_vs_k16(ops, L, B, H, W, Cin, Cout, mode)


def _k32_vs_k16(ops, L, B, H, W, Cin, Cout, k32_value):
    """The halo kernel with 32-channel stages on v_mfma_f32_16x16x32_f16 (conv_fwd_h2k.hip, tuning key h2_k32) against the 16-channel
    kernel on 32x32x16: per output element the same three products per stage pair in the same order, only the MFMA's internal
    summation width differs -- agreement to 1e-6 of the output range, against fp64 within the same bound as the 16-channel kernel,
    bit-identical planes given equal fp32 results, borders / epilogue outputs / both tile shapes (256 x 192, 256 x 64) included."""
    g = torch.Generator(device="cuda").manual_seed(H + Cin + 1)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    got = {}
    saved = ops.set_tuning("h2_k32", 0)
    try:
        for k32 in (0, 1):
            ops.set_tuning("h2_k32", k32 * k32_value)
            out, pre = torch.zeros(B, H, W, Cout, device="cuda"), torch.zeros(B, H, W, Cout, device="cuda")
            opl = ops.h2_empty(out.shape, "cuda", 16.0)
            opl.t.zero_()
            ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, epilogue=L.EPI_LRELU, residual=res, out=out, pre=pre,
                              out_planes=opl)
            got[k32] = (out, pre, opl)
    finally:
        ops.set_tuning("h2_k32", saved)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    for k32 in (0, 1):
        assert float((got[k32][1].double() - ref).abs().max()) <= 2e-6 * sc, k32
    assert float((got[1][1] - got[0][1]).abs().max()) <= 1e-6 * sc
    # finished output: LeakyReLU is continuous, the residual is added exactly
    assert float((got[1][0] - got[0][0]).abs().max()) <= 1e-6 * float(got[0][0].abs().max())
    _close_planes(ops, got[1][2], got[1][0])


@pytest.mark.parametrize("H,Cout", [(128, N), (64, N)])
def test_conv_h2_epilogues(ops, L, H, Cout):
    g = torch.Generator(device="cuda").manual_seed(H)
    x = torch.randn(4, H, H, N, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, N, device="cuda", generator=g) / 41.6
    b = torch.randn(Cout, device="cuda", generator=g)
    aux = torch.randn(4, H, H, Cout, device="cuda", generator=g)
    res = torch.randn(4, H, H, Cout, device="cuda", generator=g)
    wpl6, wpl, xp = ops.split_bf16x3(w), ops.split_h2_conv(w), ops.split_h2(x)
    for epi, a, r, want_pre in ((L.EPI_LRELU, None, res, True), (L.EPI_LRELU_BWD, aux, None, False), (L.EPI_RELU, None, None, False),
                                (L.EPI_RELU_BWD, aux, res, False), (L.EPI_NONE, None, res, False)):
        pre_ref = torch.empty(4, H, H, Cout, device="cuda") if want_pre else None
        ref = ops.conv2d_fwd(x, w, b, 1, 1, epilogue=epi, aux=a, residual=r, pre=pre_ref, wplanes=wpl6)     # fp32-input kernel
        out = torch.empty_like(ref)
        pre = torch.empty_like(ref) if want_pre else None
        opl = ops.h2_empty(ref.shape, "cuda", 16.0)
        ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, epilogue=epi, aux=a, residual=r, out=out, pre=pre, out_planes=opl)
        # the activation epilogues are discontinuous at pre = 0 (LeakyReLU slope / ReLU): compare where the reference is not at a kink
        tol = 6e-6 * float(ref.abs().max())
        bad = (out - ref).abs() > tol
        assert float(bad.float().mean()) < 1e-5, epi
        _close_planes(ops, opl, out)
        if want_pre:
            assert float((pre - pre_ref).abs().max()) <= tol
        if epi in (L.EPI_LRELU_BWD, L.EPI_RELU_BWD):
            # activation-backward masks from plane 0 of the aux tensor (same sign as the fp32 value), planes-only output
            opl2 = ops.h2_empty(ref.shape, "cuda", 16.0)
            ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, epilogue=epi, aux_planes=ops.split_h2(aux), residual=r,
                              out_planes=opl2)
            assert torch.equal(opl2.t, opl.t), epi


def test_gather_qdrop_h2(ops):
    n, B, shape = 8, 4, (32, 32, N)
    g = torch.Generator(device="cuda").manual_seed(2)
    cq = torch.randn(n, *shape, device="cuda", generator=g)
    cf = torch.randn(n, *shape, device="cuda", generator=g)
    idx = torch.tensor([[3, 7, 1, 0], [2, 2, 5, 6]], dtype=torch.int32, device="cuda")
    it = torch.ones(1, dtype=torch.int32, device="cuda")
    ref = torch.empty(B, *shape, device="cuda")
    ops.gather_qdrop(cq, cf, idx, it, B, 0.5, 77, ref, batch_offset=4)
    out = torch.empty_like(ref)
    pl = ops.h2_empty(ref.shape, "cuda", 16.0)
    ops.gather_qdrop_h2(cq, cf, idx, it, B, 0.5, 77, out, pl, batch_offset=4)
    assert torch.equal(out, ref)
    _close_planes(ops, pl, ref)
    pl2 = ops.h2_empty(ref.shape, "cuda", 16.0)
    ops.gather_qdrop_h2(cq, cf, idx, it, B, 0.5, 77, None, pl2, batch_offset=4)
    assert torch.equal(pl2.t, pl.t)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("with_res", [True, False])
def test_loss_act_bwd_equals_unfused_chain(ops, act, with_res):
    g = torch.Generator(device="cuda").manual_seed(act * 2 + with_res)
    n, B, shape = 6, 4, (16, 16, 64)
    pre = torch.randn(B, *shape, device="cuda", generator=g)
    res = torch.randn(B, *shape, device="cuda", generator=g) if with_res else None
    tgt = torch.randn(n, *shape, device="cuda", generator=g)
    idx = torch.tensor([[5, 0, 3, 3]], dtype=torch.int32, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    # unfused: activation (+ residual), lp2 loss / gradient, activation backward
    o = {0: lambda t: t.clone(), 1: ops.lrelu, 2: ops.relu}[act](pre)
    if with_res:
        o = ops.add(o, res)
    gref = torch.empty_like(pre)
    log_ref = torch.zeros(1, 32, device="cuda")
    ops.lp2_loss_grad(o, tgt, idx, it, 2.0, gref, log_ref)
    dref = {0: lambda a, b: a.clone(), 1: ops.lrelu_bwd, 2: ops.relu_bwd}[act](gref, pre)
    out, gout, dpre = torch.empty_like(pre), torch.empty_like(pre), torch.empty_like(pre)
    psc = ops.pow2_scale(dref.abs().max())
    pl = ops.h2_empty(pre.shape, "cuda", psc)
    log = torch.zeros(1, 32, device="cuda")
    ops.loss_act_bwd(pre, res, tgt, idx, it, 2.0, act, log, out=out, grad_out=gout, dpre=dpre, dpre_planes=pl)
    assert torch.equal(out, o) and torch.equal(gout, gref) and torch.equal(dpre, dref)
    _close_planes(ops, pl, dref)
    torch.testing.assert_close(log.sum(), log_ref.sum(), rtol=1e-5, atol=0)
    # planes as the only gradient output
    pl2 = ops.h2_empty(pre.shape, "cuda", psc)
    ops.loss_act_bwd(pre, res, tgt, idx, it, 2.0, act, torch.zeros(1, 32, device="cuda"), dpre_planes=pl2)
    assert torch.equal(pl2.t, pl.t)
    if with_res:
        # the residual handed over as planes only: (h1 + h2) / s is the fp32 value to 2^-24
        pl3, out3 = ops.h2_empty(pre.shape, "cuda", psc), torch.empty_like(pre)
        ops.loss_act_bwd(pre, None, tgt, idx, it, 2.0, act, torch.zeros(1, 32, device="cuda"), out=out3, dpre_planes=pl3,
                         residual_planes=ops.split_h2(res))
        torch.testing.assert_close(out3, o, rtol=0, atol=2.0 ** -22 * float(res.abs().max()))
        torch.testing.assert_close(ops.h2_to_float(pl3, dref.shape), dref, rtol=0, atol=1e-6 * float(dref.abs().max()))


@pytest.mark.parametrize("inverse", [False, True])
def test_loss_gdn_bwd_equals_unfused_chain(ops, L, inverse):
    g = torch.Generator(device="cuda").manual_seed(5 + inverse)
    n, B, H, C = 6, 4, 16, 64
    x = torch.randn(B, H, H, C, device="cuda", generator=g)
    gam = (0.1 * torch.eye(C, device="cuda") + 0.002 * torch.rand(C, C, device="cuda", generator=g)).reshape(C, 1, 1, C).contiguous()
    beta = 0.5 + torch.rand(C, device="cuda", generator=g)
    res = torch.randn(B, H, H, C, device="cuda", generator=g)
    tgt = torch.randn(n, H, H, C, device="cuda", generator=g)
    idx = torch.tensor([[1, 4, 2, 0]], dtype=torch.int32, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    norm = torch.empty_like(x)
    o = ops.conv2d_fwd(x, gam, beta, 1, 0, epilogue=L.EPI_IGDN if inverse else L.EPI_GDN, aux=x, residual=res, square_input=True, pre=norm)
    gref = torch.empty_like(x)
    log_ref = torch.zeros(1, 32, device="cuda")
    ops.lp2_loss_grad(o, tgt, idx, it, 2.0, gref, log_ref)
    tref = ops.gdn_bwd_t(gref, x, norm, inverse)
    out, gout, t = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    tpl = ops.h2_empty(x.shape, "cuda", ops.pow2_scale(tref.abs().max()))
    log = torch.zeros(1, 32, device="cuda")
    ops.loss_gdn_bwd(x, norm, res, tgt, idx, it, 2.0, inverse, log, gout, t=t, t_planes=tpl, out=out)
    assert torch.equal(out, o) and torch.equal(gout, gref) and torch.equal(t, tref)
    _close_planes(ops, tpl, tref)
    torch.testing.assert_close(log.sum(), log_ref.sum(), rtol=1e-5, atol=0)
    acc = torch.randn(x.shape, device="cuda", generator=g)
    dref = ops.gdn_bwd_dx(gref, x, norm, acc, inverse)
    dx = torch.empty_like(x)
    dpl = ops.h2_empty(x.shape, "cuda", ops.pow2_scale(dref.abs().max()))
    ops.gdn_bwd_dx_h2(gref, x, norm, acc, inverse, dx=dx, dx_planes=dpl)
    assert torch.equal(dx, dref)
    _close_planes(ops, dpl, dref)


def test_pixel_shuffle_h2(ops):
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(2, 8, 12, 4 * 48, device="cuda", generator=g)
    ref = ops.pixel_shuffle(x, 2)
    out = torch.empty_like(ref)
    pl = ops.h2_empty(ref.shape, "cuda", 16.0)
    ops.pixel_shuffle_h2(x, out=out, out_planes=pl)
    assert torch.equal(out, ref)
    _close_planes(ops, pl, ref)
    assert torch.equal(ref, torch.nn.functional.pixel_shuffle(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))
    # and its gradient (pixel unshuffle), fp32 and planes
    back = ops.pixel_unshuffle2(ref)
    assert torch.equal(back, x)
    bpl = ops.h2_empty(x.shape, "cuda", 16.0)
    ops.pixel_unshuffle2(ref, out_planes=bpl)
    _close_planes(ops, bpl, x)


WGRAD_SHAPES = [(4, 128, 192, 192, 3, 1, 1), (4, 64, 192, 768, 3, 1, 1), (4, 64, 192, 192, 3, 1, 1), (4, 128, 192, 192, 3, 2, 1),
                (4, 32, 192, 192, 3, 1, 1), (2, 64, 320, 192, 5, 2, 2), (4, 32, 192, 320, 5, 2, 2), (4, 16, 192, 768, 3, 1, 1)]


@pytest.mark.parametrize("B,H,Cin,Cout,K,s,p", WGRAD_SHAPES)
def test_wgrad_h2_matches_fp64_like_the_fp32_input_kernel(ops, B, H, Cin, Cout, K, s, p):
    """Plane-input weight gradient (LDS-DMA + transposed LDS reads) against fp64 over all output channels, and against the
    fp32-input bf16x6 kernel's error (exact split, six products)."""
    from test_gpu_x6_parity import wgrad_fp64
    g = torch.Generator(device="cuda").manual_seed(H * 31 + Cout + K)
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g)
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda", generator=g) * 0.1
    wshape = (Cout, K, K, Cin)
    assert ops.wgrad_h2_supported(tuple(x.shape), wshape, s, p)
    ref6 = ops.reduce_slabs(ops.conv2d_wgrad(x, dy, wshape, s, p))
    slabs = ops.conv2d_wgrad_h2(ops.split_h2(x), tuple(x.shape), ops.split_h2(dy), wshape, s, p)
    dw = ops.reduce_slabs(slabs)
    ref = wgrad_fp64(x, dy, K, s, p)
    scale = float(ref.abs().max())
    e = float((dw.double() - ref).abs().max()) / scale
    e6 = float((ref6.double() - ref).abs().max()) / scale
    assert e < 1e-5 and e < 2.0 * e6 + 2e-7, (e, e6)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 128, 128, 192, 192), (4, 64, 64, 192, 768), (1, 32, 64, 64, 48), (2, 32, 32, 128, 320), (4, 64, 64, 192, 192),
                                            (2, 96, 64, 256, 160), (1, 48, 160, 192, 208), (3, 16, 32, 320, 192)])
def test_wgrad_h2_row_kernel_equals_per_tap_kernel(ops, B, H, W, Cin, Cout):
    """3x3 / stride 1 / pad 1: three kw taps on one 34-pixel input row image.  Same pixel chunks, same order of sums per output element
    as the per-tap kernel, so the same slabs bit for bit (row ends and image top / bottom included)."""
    g = torch.Generator(device="cuda").manual_seed(H + W + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    dy = torch.randn(B, H, W, Cout, device="cuda", generator=g) * 0.1
    wshape = (Cout, 3, 3, Cin)
    if not ops.wgrad_h2_supported(tuple(x.shape), wshape, 1, 1):
        pytest.skip("shape not on the plane path")
    xp, dyp = ops.split_h2(x), ops.split_h2(dy)
    got = {}
    for row in (0, 1):
        ops.set_tuning("wgrad_p3_row", row)
        try:
            got[row] = ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, wshape, 1, 1)
        finally:
            ops.set_tuning("wgrad_p3_row", 1)
    assert torch.equal(got[0], got[1])


@pytest.mark.parametrize("variant", [2, 3, 9, 13])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 128, 128, 192, 192), (1, 32, 64, 64, 48), (3, 16, 32, 320, 192), (2, 96, 64, 256, 160)])
def test_wgrad_h2_row_kernel_stage_variants_give_the_same_slabs(ops, B, H, W, Cin, Cout, variant):
    """The row kernel's stage shapes (tuning key wgrad_sub: 1 = one 32-pixel segment per barrier, ring of two -- the round-3 loop --, 2 = two
    segments per barrier -- odd segment counts multiply zeros for the missing one --, 3 = ring of three, waves 4-7 issue their DMAs at the end
    of the stage, 9 = ring of three with the next stage's first fragments read one stage ahead, 13 = the same with the reads interleaved
    into the MFMAs -- the shipped one) walk the same pixels in the same order per output element: the same bits."""
    g = torch.Generator(device="cuda").manual_seed(H + W + Cout + 1)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    dy = torch.randn(B, H, W, Cout, device="cuda", generator=g) * 0.1
    wshape = (Cout, 3, 3, Cin)
    if not ops.wgrad_h2_supported(tuple(x.shape), wshape, 1, 1):
        pytest.skip("shape not on the plane path")
    xp, dyp = ops.split_h2(x), ops.split_h2(dy)
    got = {}
    saved = ops.set_tuning("wgrad_sub", 1)
    try:
        for v in (1, variant):
            ops.set_tuning("wgrad_sub", v)
            got[v] = ops.conv2d_wgrad_h2(xp, tuple(x.shape), dyp, wshape, 1, 1)
    finally:
        ops.set_tuning("wgrad_sub", saved)
    assert torch.equal(got[1], got[variant])


@pytest.mark.parametrize("B,H,Cin,Cout,K,s,p,x6", [(4, 64, N, N, 3, 1, 1, True), (4, 32, N, N, 3, 1, 1, True), (4, 16, N, N, 3, 1, 1, False),
                                                   (4, 16, 320, N, 5, 2, 2, False), (4, 32, N, N, 3, 2, 1, False)])
@pytest.mark.parametrize("act", [0, 1])
def test_splitk_conv_with_tail_doing_its_second_pass(ops, L, B, H, Cin, Cout, K, s, p, x6, act):
    """rdo_conv2d_fwd_partials + rdo_loss_act_bwd_splitk (the conv's slab sum and bias inside the tail's first load) against
    rdo_conv2d_fwd + rdo_loss_act_bwd: the same bits for the pre-activation-derived outputs, the same loss up to summation order."""
    g = torch.Generator(device="cuda").manual_seed(H * K + Cin + act)
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, K, K, Cin, device="cuda", generator=g) / (K * K * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    wpl = ops.split_bf16x3(w) if x6 else None
    ks, _ = ops.conv_fwd_ksplit(tuple(x.shape), tuple(w.shape), s, p, x6, "cuda")
    assert ks >= 2
    Ho = (H + 2 * p - K) // s + 1
    n = 6
    res = torch.randn(B, Ho, Ho, Cout, device="cuda", generator=g)
    tgt = torch.randn(n, Ho, Ho, Cout, device="cuda", generator=g)
    idx = torch.tensor([[5, 0, 3, 3]], dtype=torch.int32, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    pre = ops.conv2d_fwd(x, w, b, s, p, wplanes=wpl)
    ref = [torch.empty_like(pre) for _ in range(3)]
    log_ref = torch.zeros(1, 32, device="cuda")
    ops.loss_act_bwd(pre, res, tgt, idx, it, 2.0, act, log_ref, out=ref[0], grad_out=ref[1], dpre=ref[2])
    ws, ks2 = ops.conv2d_fwd_partials(x, w, s, p, wplanes=wpl)
    assert ks2 == ks
    got = [torch.empty_like(pre) for _ in range(3)]
    log = torch.zeros(1, 32, device="cuda")
    ops.loss_act_bwd_splitk(ws, ks, b, tuple(pre.shape), res, tgt, idx, it, 2.0, act, log, out=got[0], grad_out=got[1], dpre=got[2])
    for u, v in zip(got, ref):
        assert torch.equal(u, v)
    torch.testing.assert_close(log.sum(), log_ref.sum(), rtol=1e-5, atol=0)


@pytest.mark.parametrize("B,H,W,Cin,Cout,act,with_res", [(4, 128, 128, N, N, 1, True), (2, 128, 256, 64, N, 0, True), (4, 128, 128, N, N, 2, False)])
def test_conv_h2_with_tail_in_its_epilogue(ops, L, B, H, W, Cin, Cout, act, with_res):
    """rdo_conv2d_fwd_h2_tail (halo kernel whose epilogue forms the loss and dL/dpre) against rdo_conv2d_fwd_h2 + rdo_loss_act_bwd:
    identical dL/dpre planes, the same loss up to summation order."""
    g = torch.Generator(device="cuda").manual_seed(H + W + Cin + act)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    n = B + 2
    res = torch.randn(B, H, W, Cout, device="cuda", generator=g) if with_res else None
    tgt = torch.randn(n, H, W, Cout, device="cuda", generator=g)
    idx = torch.tensor([[(3 * i + 1) % n for i in range(B)], [(5 * i) % n for i in range(B)]], dtype=torch.int32, device="cuda")
    it = torch.ones(1, dtype=torch.int32, device="cuda")
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    resp = ops.split_h2(res) if with_res else None
    assert ops.conv_h2_tail_supported(tuple(x.shape), tuple(w.shape), 1, 1)
    pre = torch.empty(B, H, W, Cout, device="cuda")
    ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, out=pre)
    gsc = 2.0 ** 20                                              # dL/dpre ~ 2 d / npix ~ 1e-4
    ref = ops.h2_empty(pre.shape, "cuda", gsc)
    log_ref = torch.zeros(2, 32, device="cuda")
    ops.loss_act_bwd(pre, None, tgt, idx, it, 2.0, act, log_ref, dpre_planes=ref, residual_planes=resp)
    got = ops.h2_empty(pre.shape, "cuda", gsc)
    log = torch.zeros(2, 32, device="cuda")
    ops.h2_overflow(reset=True)
    ops.conv2d_fwd_h2_tail(xp, tuple(x.shape), tuple(w.shape), wpl, b, 1, 1, resp, tgt, idx, it, 2.0, act, got, log)
    assert torch.equal(got.t, ref.t)
    assert float(log[0].abs().sum()) == 0.0
    torch.testing.assert_close(log[1].sum(), log_ref[1].sum(), rtol=1e-5, atol=0)
    assert not ops.h2_overflow(reset=True)


# ---- engine level: probe scales, overflow flag ----------------------------------------------------------------------------------------
def _rb_engine(cq, cf, co, idx, iters, blk):
    from quantization.engine import UnitEngine
    from quantization.quant_block import QuantRB
    from quantization.recon import _unit_modules
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    unit = QuantRB(blk, WQ, dict(WQ, leaf_param=False)).cuda()
    kind, mods = _unit_modules(unit)
    return UnitEngine(kind, mods, cq, cf, co, batch_size=idx.shape[1], iters=iters, warmup=0.0, input_prob=0.5, seed=3, idx_table=idx)


@pytest.mark.parametrize("mag", [1e-4, 1.0, 3e3])
def test_engine_scales_follow_the_magnitude_of_the_caches(mag):
    """The same ResidualBlock unit (64^2, N = 192, H2 path) on caches scaled by 1e-4 / 1 / 3e3: the probe iteration picks power-of-two
    scales that put every plane tensor near 2^7, no overflow, and the losses scale with mag^2 to fp32 accuracy (the unit is
    positively homogeneous in its input only up to the bias terms, so the check is made with zero biases)."""
    import lic
    torch.manual_seed(5)
    blk = lic.ResidualBlock(N, N)
    with torch.no_grad():
        for m in (blk.conv1, blk.conv2):
            m.bias.zero_()
    blk = blk.cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randn(6, 64, 64, N, device="cuda", generator=g)
    noise = 0.01 * torch.randn(base.shape, device="cuda", generator=g)
    idx = torch.stack([torch.tensor([0, 1, 2, 3], dtype=torch.int32), torch.tensor([2, 3, 4, 5], dtype=torch.int32), torch.tensor([5, 0, 1, 4], dtype=torch.int32)])
    losses = {}
    for mg in (1.0, mag):
        cq, cf = base * mg, (base + noise) * mg
        with torch.no_grad():
            co = blk(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
        eng = _rb_engine(cq, cf, co, idx, 3, blk)
        assert eng.h2_plan == "rb"
        for name, s in eng.scales.items():
            m, e = __import__("math").frexp(s)
            assert m == 0.5, (name, s)                          # a power of two
        amax_x = float(torch.maximum(cq.abs().amax(), cf.abs().amax()))
        assert 16.0 <= eng.scales["x"] * amax_x <= 256.0        # the probed mini-batch puts the largest |x s| in (2^6, 2^7]; other images stay close
        eng.run()
        losses[mg] = eng.logs()[1]                               # logs() raises if a value left fp16's range
    torch.testing.assert_close(losses[mag] / mag ** 2, losses[1.0], rtol=2e-4, atol=0)


def _jump_caches(factor):
    import lic
    torch.manual_seed(6)
    blk = lic.ResidualBlock(N, N).cuda()
    g = torch.Generator(device="cuda").manual_seed(6)
    cf = torch.randn(8, 64, 64, N, device="cuda", generator=g)
    cf[4:] *= factor
    cq = cf + 0.01 * cf.abs().mean(dim=(1, 2, 3), keepdim=True) * torch.randn(cf.shape, device="cuda", generator=g)
    with torch.no_grad():
        co = blk(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    idx = torch.tensor([[0, 1, 2, 3], [4, 5, 6, 7], [1, 6, 2, 7], [0, 3, 5, 4]], dtype=torch.int32)
    return blk, cq, cf, co, idx


def _fp32_engine(cq, cf, co, idx, blk):
    from quantization.engine import UnitEngine
    from quantization.quant_block import QuantRB
    from quantization.recon import _unit_modules
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    kind, mods = _unit_modules(QuantRB(blk, WQ, dict(WQ, leaf_param=False)).cuda())
    return UnitEngine(kind, mods, cq, cf, co, batch_size=idx.shape[1], iters=idx.shape[0], warmup=0.0, input_prob=0.5, seed=3, idx_table=idx,
                      use_h2=False)


def test_engine_restarts_with_reprobed_scales_when_the_caches_outgrow_them(caplog):
    """The probes see iteration 0's mini-batch only; images 2000 x larger in a later iteration leave the fp16 range of the planes
    (missing 4 / ADVICE round 3).  The unit's own overflow word is raised, the engine restarts the unit from its
    initial state with scales re-derived from the magnitudes the overflow words recorded and finishes on H2 planes with the alphas of the fp32-activation
    path -- a warning, never a lost run; the per-device default flag and a neighbouring engine are untouched."""
    import logging
    from hipops import ops
    blk, cq, cf, co, idx = _jump_caches(2000.0)
    ops.h2_overflow(reset=True)
    eng = _rb_engine(cq, cf, co, idx, 4, blk)
    other = _rb_engine(cq[:4], cf[:4], co[:4], idx[:1], 1, blk)
    assert eng.h2_plan == "rb" and other.h2_plan == "rb"
    s0 = dict(eng.scales)
    eng.run()
    other.run()
    with caplog.at_level(logging.WARNING, logger="rdo_ptq.engine"):
        _, rt_o, _ = other.logs()                                # reads ITS word: clean
        assert other.h2_restarts == 0
        tot, rt, rd = eng.logs()
    assert 1 <= eng.h2_restarts <= eng.H2_RESTARTS and eng.h2_plan == "rb" and eng.use_h2
    assert any("restarting it with scales" in r.getMessage() for r in caplog.records)
    assert eng.scales["x"] < s0["x"] / 256                       # the planes now hold the large images
    assert not ops.h2_overflow(reset=True)                       # nobody used the shared default flag
    ref = _fp32_engine(cq, cf, co, idx, blk)
    assert ref.h2_plan is None
    ref.run()
    tot_r, rt_r, rd_r = ref.logs()
    torch.testing.assert_close(rt, rt_r, rtol=1e-3, atol=0)
    torch.testing.assert_close(rd, rd_r, rtol=1e-3, atol=1e-7)
    for n_ in ref.ops:
        a, b = eng.alpha_of(n_), ref.alpha_of(n_)
        assert float(((a >= 0) != (b >= 0)).float().mean()) < 1e-3, n_
        assert float(((a - b).abs() > 2e-3).float().mean()) < 2e-3, n_
    eng.finish()                                                 # hands the rounding back without raising
    assert blk is not None


def test_engine_falls_back_to_fp32_activations_when_rescaled_planes_overflow_too():
    """Second line of defence: a unit that overflows again after its re-scaled restarts (here: forced by marking them as used)
    re-runs on fp32 activations -- the alphas bit-identical to an engine built with use_h2=False."""
    blk, cq, cf, co, idx = _jump_caches(1e6)
    eng = _rb_engine(cq, cf, co, idx, 4, blk)
    assert eng.h2_plan == "rb"
    eng.h2_restarts = eng.H2_RESTARTS
    eng.run()
    tot = eng.logs()[0]
    assert eng.h2_restarts == eng.H2_RESTARTS + 1 and eng.h2_plan is None and not eng.use_h2
    ref = _fp32_engine(cq, cf, co, idx, blk)
    ref.run()
    torch.testing.assert_close(tot, ref.logs()[0], rtol=1e-6, atol=0)      # (the loss log is summed by float atomics: order varies)
    for n_ in ref.ops:
        assert torch.equal(eng.alpha_of(n_), ref.alpha_of(n_)), n_          # the gradients are not: deterministic slabs


def test_long_run_polls_the_overflow_word_and_restarts_early():
    """run() of more than H2_POLL iterations reads the unit's word between chunks: the overflow of iteration 1 is met after the first
    chunk, not after the whole schedule."""
    from quantization.engine import UnitEngine
    blk, cq, cf, co, idx4 = _jump_caches(2000.0)
    iters = 40
    idx = idx4.repeat(10, 1)
    eng = _rb_engine(cq, cf, co, idx, iters, blk)
    eng.H2_POLL = 8
    eng.run(24)
    assert 1 <= eng.h2_restarts <= eng.H2_RESTARTS and eng._done == 24     # restarted inside run(), then ran on to the requested iteration
    n_restarts = eng.h2_restarts
    eng.run()
    eng.logs()
    assert eng.h2_restarts == n_restarts and eng._done == iters and eng.h2_plan == "rb"


def test_zero_probe_tensor_runs_the_unit_on_fp32_activations():
    """a tensor that is exactly zero in the probes must not get scale 1.0 -- the unit leaves the H2 path."""
    import lic
    torch.manual_seed(7)
    blk = lic.ResidualBlock(N, N).cuda()
    z = torch.zeros(4, 64, 64, N, device="cuda")
    with torch.no_grad():
        co = blk(z.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    idx = torch.tensor([[0, 1, 2, 3]], dtype=torch.int32)
    eng = _rb_engine(z, z.clone(), co, idx, 1, blk)
    assert eng.h2_plan is None and not eng.use_h2                # all-zero input planes: no scale
    eng.run()
    assert torch.isfinite(eng.logs()[0]).all()
