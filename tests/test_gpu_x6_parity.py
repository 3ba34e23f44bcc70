"""Element-wise parity of the split-bf16 ("bf16x6") kernels that carry the bench's hot path, at the Cheng2020 N=192 shapes that
actually select them: weight gradient of both kernel variants against an fp64 reference over ALL
output channels, dgrad through the fragment-ordered `wd` planes the AdaRound step writes, and one full-size Adam step of a
ResidualBlockWithStride (g_a.2) and a ResidualBlockUpsample (g_s.5) unit against torch autograd on the same GPU tensors.

fp tolerance: the bf16x6 products are exact and accumulate in fp32, so the only error is the fp32 accumulation order; bounds
are stated relative to the largest reference element and against the fp32-MFMA kernel's own error on the same input."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
N = 192

# (B, H, Cin, Cout, K, stride, pad): every wgrad shape of the Cheng2020-anchor N=192 schedule that takes the x6 path at B=4,
# plus the Minnen2018 / Lu2022 5x5 stride-2 shapes (tools/wgrad_x6_check.py SHAPES + WG_SMALL)
WGRAD_X6_SHAPES = [(4, 128, 192, 192, 3, 1, 1), (4, 64, 192, 768, 3, 1, 1), (4, 64, 192, 192, 3, 1, 1), (4, 128, 192, 192, 3, 2, 1),
                   (4, 32, 192, 192, 3, 1, 1), (2, 64, 320, 192, 5, 2, 2), (4, 32, 192, 320, 5, 2, 2),
                   (4, 16, 320, 320, 3, 1, 1), (4, 16, 192, 768, 3, 1, 1)]


@pytest.fixture(scope="module")
def ops():
    from hipops import ops as o
    return o


def wgrad_fp64(x, dy, K, s, p):
    """dw[co][kh][kw][ci] = sum_m dy[m][co] x[pix(m, kh, kw)][ci] in fp64 on the GPU: one [Cout x M] @ [M x Cin] product per tap."""
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    xp = F.pad(x.double(), (0, 0, p, p, p, p))
    dy2 = dy.double().reshape(-1, Cout).t().contiguous()
    dw = torch.empty(Cout, K, K, Cin, dtype=torch.float64, device=x.device)
    for kh in range(K):
        for kw in range(K):
            xs = xp[:, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s, :].reshape(-1, Cin)
            dw[:, kh, kw, :] = dy2 @ xs
    return dw


@pytest.mark.parametrize("w8", [1])          # the four-wave predecessor ("wgrad_x6_w8" = 0) is compiled only into `make DIAG=1` builds
@pytest.mark.parametrize("B,H,Cin,Cout,K,s,p", WGRAD_X6_SHAPES)
def test_wgrad_x6_matches_fp64_all_channels(ops, w8, B, H, Cin, Cout, K, s, p):
    g = torch.Generator(device="cuda").manual_seed(H * 31 + Cout + K)
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g)
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda", generator=g) * 0.1
    wshape = (Cout, K, K, Cin)
    prev = ops.set_tuning("wgrad_x6_w8", w8)
    try:
        assert ops.wgrad_uses_bf16x6(tuple(x.shape), wshape, s, p), "shape no longer selects the bf16x6 weight-gradient kernel"
        dw6 = ops.reduce_slabs(ops.conv2d_wgrad(x, dy, wshape, s, p))
        prev_x6 = ops.set_tuning("conv_x6", 0)
        try:
            assert not ops.wgrad_uses_bf16x6(tuple(x.shape), wshape, s, p)
            dw32 = ops.reduce_slabs(ops.conv2d_wgrad(x, dy, wshape, s, p))
        finally:
            ops.set_tuning("conv_x6", prev_x6)
    finally:
        ops.set_tuning("wgrad_x6_w8", prev)
    ref = wgrad_fp64(x, dy, K, s, p)
    scale = float(ref.abs().max())
    e6 = float((dw6.double() - ref).abs().max()) / scale
    e32 = float((dw32.double() - ref).abs().max()) / scale
    # both are fp32 accumulations over M = B*Ho*Wo pixels in different orders: same error class, and small
    assert e6 < 1e-5 and e6 < 2.0 * e32 + 2e-7, (e6, e32)


@pytest.mark.parametrize("H,Cin,Cout,K", [(128, N, N, 3), (64, N, N, 3), (128, N, N, 1), (64, N, 4 * N, 3)])
def test_dgrad_x6_through_adaround_planes(ops, H, Cin, Cout, K):
    """dgrad of a stride-1 'same' conv = the forward kernel on dY with the flipped `wd` layout; on the x6 path it reads the
    fragment-ordered bf16 planes that rdo_adaround_step writes.  Planes vs rdo_split_bf16x3_conv(wd) bit for bit, then the conv
    against an fp64 conv_transpose2d of one image."""
    g = torch.Generator(device="cuda").manual_seed(H + Cout)
    w = (torch.randn(Cout, K, K, Cin, device="cuda", generator=g) / (Cin * K * K) ** 0.5).contiguous()
    delta, zp = ops.uaq_init_minmax(w.reshape(Cout, -1), 256)
    desc = ops.ada_desc(w, 256)
    alpha = ops.adaround_init_alpha(desc, w, delta)
    alpha += 0.3 * torch.randn(alpha.shape, device="cuda", generator=g)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    wq, wd = torch.empty_like(w), torch.empty_like(w)
    wq_pl = torch.empty((3,) + tuple(w.shape), device="cuda", dtype=torch.int16)
    wd_pl = torch.empty((3, Cin, K, K, Cout), device="cuda", dtype=torch.int16)
    sched = ops.make_sched(1, 0.0, (20, 2), 1e-3, "cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    rlog = torch.zeros(1, 32, device="cuda")
    slabs = 1e-3 * torch.randn((2,) + tuple(w.shape), device="cuda", generator=g)
    ops.adaround_step(desc, w, delta, zp, slabs, 1.0, 0.01, sched, it, alpha, m, v, wq, wd, rlog, wq_pl, wd_pl)
    wd4 = wd.reshape(Cin, K, K, Cout)
    # wd is the tap-flipped transpose of the new soft weights
    assert torch.equal(wd4, wq.flip(1, 2).permute(3, 1, 2, 0))
    assert torch.equal(wd_pl.reshape(3, -1), ops.split_bf16x3(wd4).reshape(3, -1))
    assert torch.equal(wq_pl.reshape(3, -1), ops.split_bf16x3(wq).reshape(3, -1))
    dy = torch.randn(4, H, H, Cout, device="cuda", generator=g)
    pad = K - 1 - K // 2
    assert ops.uses_bf16x6(tuple(dy.shape), tuple(wd4.shape), 1, pad)
    dx6 = ops.conv2d_fwd(dy, wd4, None, 1, pad, wplanes=wd_pl)
    dx32 = ops.conv2d_fwd(dy, wd4, None, 1, pad)
    ref = F.conv_transpose2d(dy[:1].permute(0, 3, 1, 2).double().cpu(), wq.permute(0, 3, 1, 2).double().cpu(), padding=K // 2)
    ref = ref.permute(0, 2, 3, 1).cuda()
    scale = float(ref.abs().max())
    e6 = float((dx6[:1].double() - ref).abs().max()) / scale
    e32 = float((dx32[:1].double() - ref).abs().max()) / scale
    assert e6 < 4e-6 and e6 < 2.0 * e32 + 1e-7, (e6, e32)
    assert float((dx6 - dx32).abs().max()) / scale < 6e-6


# ---- full-size unit steps vs torch autograd ---------------------------------------------------------------------------------------
def _soft_w(op, alpha):
    """Soft AdaRound weight of an engine op in the logical (OIHW / [C, C]) shape, differentiable in alpha (quantizer.py:437-449)."""
    if op.w.dim() == 4:
        w, shape = op.w.permute(0, 3, 1, 2), (-1, 1, 1, 1)
    else:
        w, shape = op.w, (-1, 1)
    d, z = op.delta.view(shape), op.zp.view(shape)
    h = torch.clamp(torch.sigmoid(alpha) * 1.2 - 0.1, 0, 1)
    return (torch.clamp(torch.floor(w / d) + h + z, 0, op.n_levels - 1) - z) * d


def _gdn_ref(x, op, alpha, inverse):
    """GDN / IGDN with the soft-quantised gamma pushed through the non-negative re-parametrisation (quant_layer.py:142-154)."""
    from oracle.lic_oracle import _LowerBoundFn as _LowerBound      # the checker takes nothing from the product
    c = x.shape[1]
    bound, ped = float(op.desc.reparam_bound), float(op.desc.reparam_pedestal)
    gq = _soft_w(op, alpha)
    gp = _LowerBound.apply(gq, torch.tensor([bound], device=x.device)) ** 2 - ped
    pool = F.conv2d(x * x, gp.view(c, c, 1, 1), op.beta)
    return x * (pool.sqrt() if inverse else pool.rsqrt())


def _step_vs_autograd(eng, forward, cq, co, names):
    a0 = {k: eng.alpha_of(k).clone() for k in eng.ops}
    alphas = {k: a0[k].clone().requires_grad_(True) for k in a0}
    out = forward(cq.permute(0, 3, 1, 2), alphas)
    tgt = co.permute(0, 3, 1, 2)
    rec = (out - tgt).abs().pow(2).sum(1).mean()
    # iters = 1, warmup = 0: LinearTempDecay gives b = end_b = 2
    rl = sum(0.01 * (1 - ((torch.clamp(torch.sigmoid(a) * 1.2 - 0.1, 0, 1) - .5).abs() * 2).pow(2.0)).sum() for a in alphas.values())
    (rl + rec + rec).backward()
    torch.optim.Adam(list(alphas.values()), lr=1e-3).step()
    eng.run()
    torch.cuda.synchronize()
    _, rt, rd = eng.logs()
    assert abs(float(rt[0]) - 2 * float(rec)) <= 2e-4 * 2 * float(rec)
    assert abs(float(rd[0]) - float(rl)) <= 2e-4 * float(rl)
    assert set(names) == set(alphas)
    for k in alphas:
        got, ref = eng.alpha_of(k), alphas[k].detach()
        # Adam's first step is lr * sign(g) for |g| >> eps: disagreements can only come from gradients at the noise level
        bad = (got - ref).abs() > 2e-4
        assert float(bad.float().mean()) < 2e-3, (k, float(bad.float().mean()))


def _make_engine(blk, qcls, cq_shape, seed):
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    unit = qcls(blk, WQ, dict(WQ, leaf_param=False)).cuda()
    kind, mods = _unit_modules(unit)
    n = cq_shape[0]
    g = torch.Generator(device="cuda").manual_seed(seed)
    cq = torch.randn(*cq_shape, device="cuda", generator=g)
    cf = cq + 0.01 * torch.randn(cq.shape, device="cuda", generator=g)
    with torch.no_grad():
        co = blk(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    idx = torch.arange(n, dtype=torch.int32).reshape(1, n)
    eng = UnitEngine(kind, mods, cq, cf, co, batch_size=n, iters=1, warmup=0.0, input_prob=1.0, seed=1, idx_table=idx)
    return eng, cq, co


def _seed_gdn(gdn, gen):
    c = gdn.gamma.shape[0]
    with torch.no_grad():
        gdn.gamma.copy_(torch.sqrt(0.1 * torch.eye(c) + 0.002 * torch.rand(c, c, generator=gen) + 2.0 ** -36))
        gdn.beta.copy_(torch.sqrt(0.5 + torch.rand(c, generator=gen) + 2.0 ** -36))


def test_full_size_rbws_unit_step_matches_torch_autograd():
    """g_a.2 of Cheng2020-anchor N=192 at B=4: 192 -> 192, 128^2 -> 64^2 (stride-2 3x3, 3x3, GDN, stride-2 1x1 skip).  Exercises
    the stride-2 x6 forward / wgrad, the 64^2 x6 dgrad, the GDN backward (t, gamma'^T GEMM, dx, dgamma wgrad with squared input)."""
    import lic
    from quantization.quant_block import QuantRBWS
    torch.manual_seed(12)
    blk = lic.ResidualBlockWithStride(N, N, stride=2)
    _seed_gdn(blk.gdn, torch.Generator().manual_seed(12))
    blk = blk.cuda()
    eng, cq, co = _make_engine(blk, QuantRBWS, (4, 128, 128, N), 21)
    o = eng.ops

    def forward(x, al):
        h1 = F.leaky_relu(F.conv2d(x, _soft_w(o["conv1"], al["conv1"]), o["conv1"].bias, stride=2, padding=1), 0.01)
        c2 = F.conv2d(h1, _soft_w(o["conv2"], al["conv2"]), o["conv2"].bias, padding=1)
        y = _gdn_ref(c2, o["gdn"], al["gdn"], inverse=False)
        return y + F.conv2d(x, _soft_w(o["skip"], al["skip"]), o["skip"].bias, stride=2)
    _step_vs_autograd(eng, forward, cq, co, ["conv1", "conv2", "gdn", "skip"])


def test_full_size_rbu_unit_step_matches_torch_autograd():
    """g_s.5 of Cheng2020-anchor N=192 at B=4 -- the most expensive unit of the schedule: 64^2 -> 128^2 through two 192 -> 768
    sub-pixel convs + pixel shuffle, a 128^2 3x3 conv and an IGDN."""
    import lic
    from quantization.quant_block import QuantRBU
    torch.manual_seed(13)
    blk = lic.ResidualBlockUpsample(N, N, 2)
    _seed_gdn(blk.igdn, torch.Generator().manual_seed(13))
    blk = blk.cuda()
    eng, cq, co = _make_engine(blk, QuantRBU, (4, 64, 64, N), 22)
    o = eng.ops

    def forward(x, al):
        sp = F.pixel_shuffle(F.conv2d(x, _soft_w(o["subpel_conv"], al["subpel_conv"]), o["subpel_conv"].bias, padding=1), 2)
        c = F.conv2d(F.leaky_relu(sp, 0.01), _soft_w(o["conv"], al["conv"]), o["conv"].bias, padding=1)
        y = _gdn_ref(c, o["igdn"], al["igdn"], inverse=True)
        return y + F.pixel_shuffle(F.conv2d(x, _soft_w(o["upsample"], al["upsample"]), o["upsample"].bias, padding=1), 2)
    _step_vs_autograd(eng, forward, cq, co, ["subpel_conv", "conv", "igdn", "upsample"])


# ---- the same units on H2 tensors (plane-input kernels) and on fp32 activations ---------------------------------------------------
def _h2_vs_fp32(make_blk, qcls, cq_shape, seed, plan):
    """Three iterations of one full-size unit with use_h2 on / off: same losses (summation order aside), same alphas except where
    Adam amplifies a gradient at the fp32 noise level."""
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    res = []
    for use_h2 in (True, False):
        torch.manual_seed(seed)
        blk = make_blk()
        unit = qcls(blk, WQ, dict(WQ, leaf_param=False)).cuda()
        kind, mods = _unit_modules(unit)
        g = torch.Generator(device="cuda").manual_seed(seed)
        n = cq_shape[0]
        cq = torch.randn(*cq_shape, device="cuda", generator=g)
        cf = cq + 0.01 * torch.randn(cq.shape, device="cuda", generator=g)
        with torch.no_grad():
            co = blk(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
        idx = torch.stack([torch.arange(n, dtype=torch.int32)] * 3)
        eng = UnitEngine(kind, mods, cq, cf, co, batch_size=n, iters=3, warmup=0.0, input_prob=0.5, seed=3, idx_table=idx, use_h2=use_h2)
        assert eng.h2_plan == (plan if use_h2 else None)
        eng.run()
        torch.cuda.synchronize()
        res.append(({k: eng.alpha_of(k).clone() for k in eng.ops}, eng.logs()[0]))
        del eng
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-5, atol=1e-9)
    for k in res[0][0]:
        bad = (res[0][0][k] - res[1][0][k]).abs() > 2e-4
        assert float(bad.float().mean()) < 2e-3, (k, float(bad.float().mean()))


def test_rb_unit_on_h2_tensors_matches_fp32_activations():
    import lic
    from quantization.quant_block import QuantRB
    _h2_vs_fp32(lambda: lic.ResidualBlock(N, N).cuda(), QuantRB, (4, 128, 128, N), 31, "rb")


def test_rbu_unit_on_h2_tensors_matches_fp32_activations():
    import lic
    from quantization.quant_block import QuantRBU

    def mk():
        blk = lic.ResidualBlockUpsample(N, N, 2)
        _seed_gdn(blk.igdn, torch.Generator().manual_seed(5))
        return blk.cuda()
    _h2_vs_fp32(mk, QuantRBU, (4, 64, 64, N), 32, "rbu")


def test_rbws_stem_unit_on_h2_tensors_matches_fp32_activations():
    """g_a.0 of Cheng2020-anchor: 3 -> 192 at 256^2 -> 128^2 (thin RGB stem kernels feeding a P3 second conv and GDN backward)."""
    import lic
    from quantization.quant_block import QuantRBWS

    def mk():
        blk = lic.ResidualBlockWithStride(3, N, stride=2)
        _seed_gdn(blk.gdn, torch.Generator().manual_seed(6))
        return blk.cuda()
    _h2_vs_fp32(mk, QuantRBWS, (4, 256, 256, 3), 33, "rbws")


@pytest.mark.parametrize("kind", ["rb", "rbu"])
def test_full_size_h2_units_data_parallel_sequence_equals_fused_step(kind):
    """The 128^2 units on H2 tensors (halo / row kernels, conv2 + tail in one launch) through the data-parallel op sequence on one rank
    (gradient -> bucket -> apply, all-reduce split in two around the last weight gradient) against the fused single-launch step:
    bit-identical alphas."""
    import lic
    from quantization.engine import UnitEngine
    from quantization.quant_block import QuantRB, QuantRBU
    from quantization.recon import _unit_modules
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    res = []
    for split in (False, True):
        torch.manual_seed(41)
        if kind == "rb":
            blk, qcls, shape = lic.ResidualBlock(N, N).cuda(), QuantRB, (4, 128, 128, N)
        else:
            blk = lic.ResidualBlockUpsample(N, N, 2)
            _seed_gdn(blk.igdn, torch.Generator().manual_seed(5))
            blk, qcls, shape = blk.cuda(), QuantRBU, (4, 64, 64, N)
        unit = qcls(blk, WQ, dict(WQ, leaf_param=False)).cuda()
        k, mods = _unit_modules(unit)
        g = torch.Generator(device="cuda").manual_seed(41)
        cq = torch.randn(*shape, device="cuda", generator=g)
        cf = cq + 0.01 * torch.randn(cq.shape, device="cuda", generator=g)
        with torch.no_grad():
            co = blk(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
        idx = torch.stack([torch.arange(4, dtype=torch.int32)] * 4)
        eng = UnitEngine(k, mods, cq, cf, co, batch_size=4, iters=4, warmup=0.0, input_prob=0.5, seed=3, idx_table=idx, force_dp_split=split)
        assert eng.h2_plan == kind and (eng.plan_a2 is not None) == split          # 43.5-GFLOP last wgrad: the all-reduce is split
        eng.run()
        torch.cuda.synchronize()
        res.append(({n: eng.alpha_of(n).clone() for n in eng.ops}, eng.logs()[0]))
        del eng
    for n in res[0][0]:
        assert torch.equal(res[0][0][n], res[1][0][n]), n
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-6, atol=0)
