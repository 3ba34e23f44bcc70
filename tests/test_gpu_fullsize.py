"""Size-independent properties at BASELINE.json's full sizes (Cheng2020 N=192, 256x256 crops, batch 4), where the CPU oracle
is too slow to be the checker: adjoint identities of the conv triple (forward / dgrad / wgrad), linearity, QDrop limits,
GDN/IGDN inversion, and one full-size unit step against torch autograd on the same GPU tensors."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
N = 192


@pytest.fixture(scope="module")
def ops():
    from hipops import ops as o
    return o


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("H,Cin,Cout,K,s,p", [(128, N, N, 3, 1, 1), (256, 3, N, 3, 2, 1), (64, N, 4 * N, 3, 1, 1),
                                              (128, N, N, 1, 2, 0), (16, N, 2 * N, 5, 1, 2), (128, N, 12, 3, 1, 1)])
def test_conv_adjoint_identities(ops, H, Cin, Cout, K, s, p):
    """<dy, conv(x, w)> == <w, wgrad(x, dy)>  and (stride 1)  == <x, dgrad(dy, w)>."""
    g = torch.Generator(device="cuda").manual_seed(H * 7 + Cout)
    B = 4
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, K, K, Cin, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    y = ops.conv2d_fwd(x, w, None, s, p)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    lhs = _dot(dy, y)
    dw = ops.reduce_slabs(ops.conv2d_wgrad(x, dy, tuple(w.shape), s, p))
    assert abs(_dot(w, dw) - lhs) <= 2e-5 * (abs(lhs) + dy.norm().item() * y.norm().item() * 1e-3)
    if s == 1 and 2 * p == K - 1:
        wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        dx = ops.conv2d_fwd(dy, wd, None, 1, K - 1 - p)
        assert abs(_dot(x, dx) - lhs) <= 2e-5 * (abs(lhs) + dy.norm().item() * y.norm().item() * 1e-3)


def test_conv_linearity_full_size(ops):
    g = torch.Generator(device="cuda").manual_seed(3)
    x1 = torch.randn(4, 128, 128, N, device="cuda", generator=g)
    x2 = torch.randn(4, 128, 128, N, device="cuda", generator=g)
    w = torch.randn(N, 3, 3, N, device="cuda", generator=g) / 41.6
    y = ops.conv2d_fwd(x1 + x2, w, None, 1, 1)
    y12 = ops.conv2d_fwd(x1, w, None, 1, 1) + ops.conv2d_fwd(x2, w, None, 1, 1)
    assert float((y - y12).abs().max()) <= 2e-5 * float(y.abs().max())


def test_qdrop_limits_and_rate(ops):
    n, B, shape = 16, 4, (128, 128, N)
    cq = torch.randn(n, *shape, device="cuda")
    cf = torch.randn(n, *shape, device="cuda")
    idx = torch.tensor([[3, 7, 11, 0]], dtype=torch.int32, device="cuda")
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.empty(B, *shape, device="cuda")
    ops.gather_qdrop(cq, cf, idx, it, B, 1.0, 5, out)
    assert torch.equal(out, cq[idx[0].long()])
    ops.gather_qdrop(cq, cf, idx, it, B, 0.0, 5, out)
    assert torch.equal(out, cf[idx[0].long()])
    ops.gather_qdrop(cq, cf, idx, it, B, 0.5, 5, out)
    took_q = (out == cq[idx[0].long()])
    assert abs(float(took_q.float().mean()) - 0.5) < 2e-3
    assert bool((took_q | (out == cf[idx[0].long()])).all())


def test_gdn_then_igdn_is_identity(ops):
    """IGDN(GDN(x)) with the same (gamma', beta') returns x up to fp32 rounding only when the norm pool of the second
    stage is fed the ORIGINAL x -- so check the algebraic form: y = x rsqrt(n(x)),  y sqrt(n(x)) == x."""
    from hipops import _lib as L
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(4, 128, 128, N, device="cuda", generator=g)
    gp = (0.1 * torch.eye(N, device="cuda") + 0.002 * torch.rand(N, N, device="cuda", generator=g)).reshape(N, 1, 1, N).contiguous()
    bp = 0.5 + torch.rand(N, device="cuda", generator=g)
    norm = torch.empty_like(x)
    y = ops.conv2d_fwd(x, gp, bp, 1, 0, epilogue=L.EPI_GDN, aux=x, square_input=True, pre=norm)
    back = y * torch.sqrt(norm)
    assert float((back - x).abs().max()) <= 1e-5 * float(x.abs().max())
    y2 = ops.conv2d_fwd(x, gp, bp, 1, 0, epilogue=L.EPI_IGDN, aux=y, square_input=True)
    assert float((y2 - x).abs().max()) <= 1e-5 * float(x.abs().max())


def test_full_size_rb_unit_step_matches_torch_autograd():
    """One AdaRound iteration of a full-size ResidualBlock unit (N=192, 128x128, B=4): alpha after the step vs. the same
    step computed with torch autograd + torch.optim.Adam on the GPU (torch/MIOpen fp32 is the checker here, not a fallback)."""
    import lic
    from quantization.engine import UnitEngine
    from quantization.quant_block import QuantRB
    from quantization.recon import _unit_modules
    torch.manual_seed(11)
    WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    blk = lic.ResidualBlock(N, N).cuda()
    unit = QuantRB(blk, WQ, dict(WQ, leaf_param=False)).cuda()
    kind, mods = _unit_modules(unit)
    n, B = 4, 4
    cq = torch.randn(n, 128, 128, N, device="cuda")
    cf = cq + 0.01 * torch.randn_like(cq)
    with torch.no_grad():
        co = blk(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    idx = torch.arange(n, dtype=torch.int32).reshape(1, n)
    eng = UnitEngine(kind, mods, cq, cf, co, batch_size=B, iters=1, warmup=0.0, input_prob=1.0, seed=1, idx_table=idx)
    a0 = {k: eng.alpha_of(k).clone() for k in eng.ops}
    # ---- torch reference of the same step (soft AdaRound weights, rec+task+round loss, Adam)
    alphas = {k: a0[k].clone().requires_grad_(True) for k in a0}

    def soft_w(op, alpha):
        w = op.w.permute(0, 3, 1, 2)
        d, z = op.delta.view(-1, 1, 1, 1), op.zp.view(-1, 1, 1, 1)
        h = torch.clamp(torch.sigmoid(alpha) * 1.2 - 0.1, 0, 1)
        return (torch.clamp(torch.floor(w / d) + h + z, 0, 255) - z) * d
    x = cq.permute(0, 3, 1, 2)
    h1 = F.leaky_relu(F.conv2d(x, soft_w(eng.ops["conv1"], alphas["conv1"]), eng.ops["conv1"].bias, padding=1), 0.01)
    out = F.leaky_relu(F.conv2d(h1, soft_w(eng.ops["conv2"], alphas["conv2"]), eng.ops["conv2"].bias, padding=1), 0.01) + x
    tgt = co.permute(0, 3, 1, 2)
    rec = (out - tgt).abs().pow(2).sum(1).mean()
    rl = sum(0.01 * (1 - ((torch.clamp(torch.sigmoid(a) * 1.2 - 0.1, 0, 1) - .5).abs() * 2).pow(2.0)).sum() for a in alphas.values())   # iters=1, warmup=0 -> LinearTempDecay gives b = end_b = 2
    (rl + rec + rec).backward()
    opt = torch.optim.Adam(list(alphas.values()), lr=1e-3)
    opt.step()
    eng.run()
    torch.cuda.synchronize()
    total, rt, rd = eng.logs()
    assert abs(float(rt[0]) - 2 * float(rec)) <= 2e-4 * 2 * float(rec)
    assert abs(float(rd[0]) - float(rl)) <= 2e-4 * float(rl)
    for k in alphas:
        got, ref = eng.alpha_of(k), alphas[k].detach()
        # Adam's first step is lr * sign(g) for |g| >> eps: disagreements can only come from gradients at the noise level
        bad = (got - ref).abs() > 2e-4
        assert float(bad.float().mean()) < 2e-3, (k, float(bad.float().mean()))


@pytest.mark.parametrize("H,Cin,Cout,K,s,p", [(128, N, N, 3, 1, 1), (64, N, 4 * N, 3, 1, 1), (128, N, N, 1, 1, 0),
                                              (64, N, N, 3, 1, 1), (128, N, N, 3, 2, 1), (32, N, 4 * N, 3, 1, 1),
                                              (64, 32, N, 5, 1, 2), (128, N, 320, 5, 2, 2), (32, N, N, 3, 1, 1)])
def test_split_bf16_conv_path_has_fp32_accuracy(ops, H, Cin, Cout, K, s, p):
    """The bf16x6 path rdo_conv2d_fwd takes for large problems: error vs an fp64 reference no larger than the fp32-MFMA
    kernel's, and the three bf16 planes re-sum to the fp32 weights exactly."""
    g = torch.Generator(device="cuda").manual_seed(H + Cout)
    x = torch.randn(4, H, H, Cin, device="cuda", generator=g) * 3
    w = torch.randn(Cout, K, K, Cin, device="cuda", generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    assert ops.uses_bf16x6(tuple(x.shape), tuple(w.shape), s, p)
    lin = ops.split_bf16x3_linear(w)
    resum = sum((lin[i].to(torch.int32) << 16).view(torch.float32) for i in range(3))
    assert torch.equal(resum, w)
    # the conv operand is the same split in fragment order [Cin/16][KH][KW][Cout][16] (include/rdo_ptq_hip.h)
    planes = ops.split_bf16x3(w)
    expect = lin.reshape(3, Cout, K, K, Cin // 16, 16).permute(0, 4, 2, 3, 1, 5).reshape(3, -1)
    assert torch.equal(planes.reshape(3, -1), expect)
    y32 = ops.conv2d_fwd(x, w, b, s, p)
    y6 = ops.conv2d_fwd(x, w, b, s, p, wplanes=planes)
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), s, p)
    ref = ref.permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    e32 = float((y32[:1].double() - ref).abs().max()) / scale
    e6 = float((y6[:1].double() - ref).abs().max()) / scale
    assert e6 < 4e-6 and e6 < 2.0 * e32 + 1e-7, (e6, e32)
    assert float((y6 - y32).abs().max()) / scale < 6e-6
