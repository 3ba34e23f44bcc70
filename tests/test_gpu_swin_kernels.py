"""Kernels of the Lu2022 transformer path against the oracle (oracle/swin_oracle.py, itself pinned to the reference's
models/layers.py by tests/golden/recon_nic.npz) and torch autograd on the CPU: window attention forward / backward with the
cyclic shift and mask folded in, the split probability path, LayerNorm backward, GELU, round.  Tolerance: fp32 with different
summation orders, 2e-5 relative to the tensor's max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _ref_attention(x_qkv, table, heads, ws, shift):
    """qkv [B,H,W,3C] natural order -> out [B,H,W,C] via the oracle's roll / partition / attention core (no linears)."""
    from oracle import swin_oracle as S
    B, H, W, C3 = x_qkv.shape
    C = C3 // 3
    hd = C // heads
    x = torch.roll(x_qkv, shifts=(-shift, -shift), dims=(1, 2)) if shift else x_qkv
    xw = x.view(B, H // ws, ws, W // ws, ws, C3).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C3)
    B_, N, _ = xw.shape
    t = xw.reshape(B_, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = t[0] * hd ** -0.5, t[1], t[2]
    attn = q @ k.transpose(-2, -1)
    bias = table[S.relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    mask = S.shifted_window_mask(H, W, ws, shift)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    p = torch.softmax(attn, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B_, N, C)
    o = o.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o, p, bias.contiguous()


@pytest.mark.parametrize("B,H,W,C,heads,ws,shift", [
    (2, 16, 16, 16, 4, 8, 0), (2, 16, 16, 16, 4, 8, 4), (1, 32, 16, 24, 8, 8, 4), (3, 4, 4, 32, 16, 4, 0),
    (2, 8, 8, 16, 8, 4, 2), (2, 2, 2, 16, 16, 2, 0), (2, 1, 1, 16, 16, 1, 0), (1, 16, 16, 192, 4, 8, 4), (1, 8, 8, 320, 16, 8, 0)])
def test_window_attention_forward_backward(B, H, W, C, heads, ws, shift):
    from hipops import ops
    g = torch.Generator().manual_seed(H * 131 + C + shift)
    qkv = (torch.randn(B, H, W, 3 * C, generator=g) * 1.5).requires_grad_(True)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    out_ref, p_ref, bias = _ref_attention(qkv, table, heads, ws, shift)
    dout = torch.randn(out_ref.shape, generator=g)
    (out_ref * dout).sum().backward()
    d = ops.attn_desc(B, H, W, C, heads, ws, shift)
    N = ws * ws
    windows = B * (H // ws) * (W // ws)
    probs = torch.empty(windows, N, N, heads, device="cuda")
    qc, bc = qkv.detach().cuda().contiguous(), bias.cuda().contiguous()
    out = ops.window_attention(d, qc, bc, probs=probs)
    assert _rel(out.cpu(), out_ref.detach()) < 2e-5
    assert _rel(probs.cpu(), p_ref.detach().permute(0, 2, 3, 1)) < 2e-5
    out2 = ops.window_attention_pv(d, qc, probs)                    # split path: probs -> (quantiser) -> probs @ v
    assert _rel(out2.cpu(), out_ref.detach()) < 2e-5
    probs_only = torch.zeros_like(probs)
    ops.window_attention(d, qc, bc, probs=probs_only, compute_out=False)
    torch.testing.assert_close(probs_only, probs, rtol=0, atol=0)
    dqkv = ops.window_attention_bwd(d, qc, bc, dout.cuda().contiguous())
    assert _rel(dqkv.cpu(), qkv.grad) < 5e-5


@pytest.mark.parametrize("rows,C", [(7, 16), (513, 192), (64, 320), (1030, 32)])
def test_layer_norm_backward(rows, C):
    from hipops import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).requires_grad_(True)
    w = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    b = torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    y = F.layer_norm(x, (C,), w, b)
    (y * dy).sum().backward()
    y_gpu = ops.layer_norm(x.detach().cuda(), w.detach().cuda(), b.cuda())
    assert _rel(y_gpu.cpu(), y.detach()) < 2e-6
    slabs = torch.empty(8, C, device="cuda")
    dx = torch.empty(rows, C, device="cuda")
    ops.layer_norm_bwd(x.detach().cuda(), w.detach().cuda(), dy.cuda(), dx=dx, dgamma_slabs=slabs)
    assert _rel(dx.cpu(), x.grad) < 2e-5
    assert _rel(slabs.sum(0).cpu(), w.grad) < 2e-5
    dx2 = torch.empty(rows, C, device="cuda")
    ops.layer_norm_bwd(x.detach().cuda(), w.detach().cuda(), dy.cuda(), dx=dx2)          # dx only (FP tail)
    torch.testing.assert_close(dx2, dx, rtol=0, atol=0)


def test_gelu_and_round():
    from hipops import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4099, generator=g) * 3).requires_grad_(True)
    dy = torch.randn(4099, generator=g)
    y = F.gelu(x)
    (y * dy).sum().backward()
    xc = x.detach().cuda()
    assert _rel(ops.gelu(xc).cpu(), y.detach()) < 2e-6
    assert _rel(ops.gelu_bwd(dy.cuda(), xc).cpu(), x.grad) < 2e-6
    v = torch.cat([torch.arange(-6, 7).float() / 2, torch.randn(100, generator=g) * 4])
    torch.testing.assert_close(ops.round_(v.cuda()).cpu(), torch.round(v), rtol=0, atol=0)


@pytest.mark.parametrize("rows,C", [(7, 16), (513, 192), (65, 320), (1030, 32), (40, 512), (4 * 128 * 128, 192)])
def test_residual_add_layer_norm_fused_forward_and_backward(rows, C):
    """rdo_add_layer_norm / rdo_layer_norm_bwd_add / rdo_add3 (round 5): the residual add of a Swin block folded into the LayerNorm that
    reads it, and the residual-path gradient folded into the LayerNorm backward, against torch autograd in float64 -- with zero, one
    and two addends, with and without the gamma partial sums; the sum tensor bit-equal to the separate add kernel."""
    from hipops import ops
    g = torch.Generator().manual_seed(rows + C)
    a = torch.randn(rows, C, generator=g) * 2 + 0.3
    b = torch.randn(rows, C, generator=g)
    w = 1 + 0.3 * torch.randn(C, generator=g)
    bias = torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    e1, e2 = torch.randn(rows, C, generator=g), torch.randn(rows, C, generator=g)
    s64 = (a.double() + b.double()).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.layer_norm(s64, (C,), w64, bias.double())
    (y64 * dy.double()).sum().backward()
    ac, bc, wc, biasc = a.cuda(), b.cuda(), w.cuda(), bias.cuda()
    s = torch.empty_like(ac)
    y = ops.add_layer_norm(ac, bc, wc, biasc, sum_out=s)
    torch.testing.assert_close(s, ops.add(ac, bc), rtol=0, atol=0)
    assert _rel(y.cpu().double(), y64.detach()) < 3e-6
    y1 = ops.add_layer_norm(s, None, wc, biasc)                       # no second addend: plain LayerNorm of s
    torch.testing.assert_close(y1, y, rtol=0, atol=0)
    assert _rel(ops.layer_norm(s, wc, biasc).cpu().double(), y64.detach()) < 3e-6
    nsl = 8 if rows < 4096 else 1024
    slabs = torch.empty(nsl, C, device="cuda")
    dx0 = torch.empty_like(ac)
    ops.layer_norm_bwd_add(s, wc, dy.cuda(), dx=dx0, dgamma_slabs=slabs)
    scale = float(s64.grad.abs().max())
    assert float((dx0.cpu().double() - s64.grad).abs().max()) < 2e-5 * scale
    assert _rel(slabs.sum(0).cpu().double(), w64.grad) < 2e-5
    dx1, dx2 = torch.empty_like(ac), torch.empty_like(ac)
    ops.layer_norm_bwd_add(s, wc, dy.cuda(), e1.cuda(), dx=dx1)
    ops.layer_norm_bwd_add(s, wc, dy.cuda(), e1.cuda(), e2.cuda(), dx=dx2)
    torch.testing.assert_close(dx1, e1.cuda() + dx0, rtol=0, atol=0)                  # the addends enter in this order: add1 + r, add2 + (...)
    torch.testing.assert_close(dx2, e2.cuda() + (e1.cuda() + dx0), rtol=0, atol=0)
    only = torch.empty(nsl, C, device="cuda")
    ops.layer_norm_bwd_add(s, wc, dy.cuda(), dx=None, dgamma_slabs=only)               # trainable LayerNorm whose input needs no gradient
    torch.testing.assert_close(only, slabs, rtol=0, atol=0)
    if (rows * C) % 4 == 0:
        torch.testing.assert_close(ops.add3(ac, bc, e1.cuda()), (ac + bc) + e1.cuda(), rtol=0, atol=0)


@pytest.mark.parametrize("rows,cin,cout,planes", [(4 * 64 * 64, 192, 384, True), (4 * 64 * 64, 384, 192, True), (300, 192, 384, False),
                                                  (1024, 320, 640, False)])
def test_gelu_epilogues_of_the_linear_kernels(rows, cin, cout, planes):
    """RDO_EPI_GELU (fc1 + nn.GELU() in one launch, pre-activation kept for the backward) and RDO_EPI_GELU_BWD (fc2's input gradient
    times gelu'(pre)) of rdo_conv2d_fwd on the fp32-MFMA and the split-bf16 kernels against the separate kernels and float64."""
    from hipops import _lib as L, ops
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(1, 1, rows, cin, generator=g).cuda()
    w = (torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    wp = ops.split_bf16x3(w) if planes else None
    assert (wp is not None and ops.uses_bf16x6(tuple(x.shape), tuple(w.shape), 1, 0)) == planes
    pre = torch.empty(1, 1, rows, cout, device="cuda")
    y = ops.conv2d_fwd(x, w, b, 1, 0, epilogue=L.EPI_GELU, pre=pre, wplanes=wp)
    pre_ref = ops.conv2d_fwd(x, w, b, 1, 0, wplanes=wp)
    torch.testing.assert_close(pre, pre_ref, rtol=0, atol=0)
    want = F.gelu(pre.double().cpu())
    assert float((y.cpu().double() - want).abs().max()) < 2e-6 * float(want.abs().max())
    # backward: d(pre_in) = (dy W) * gelu'(pre_in) for a linear whose input was gelu(pre_in)
    pin = torch.randn(1, 1, rows, cin, generator=g).cuda() * 2          # the GELU's input: this linear's input was gelu(pin)
    wd = w.permute(3, 1, 2, 0).contiguous()                            # [cin][1][1][cout]: maps dy [rows, cout] -> dx [rows, cin]
    dyv = torch.randn(1, 1, rows, cout, generator=g).cuda()
    wdp = ops.split_bf16x3(wd) if ops.uses_bf16x6(tuple(dyv.shape), tuple(wd.shape), 1, 0) else None
    got = ops.conv2d_fwd(dyv, wd, None, 1, 0, epilogue=L.EPI_GELU_BWD, aux=pin, wplanes=wdp)
    lin = ops.conv2d_fwd(dyv, wd, None, 1, 0, wplanes=wdp)
    sep = ops.gelu_bwd(lin.view(-1), pin.view(-1)).view_as(lin)
    p64 = pin.double().cpu().requires_grad_(True)
    F.gelu(p64).sum().backward()
    want = lin.double().cpu() * p64.grad
    assert float((got.cpu().double() - want).abs().max()) < 2e-6 * float(want.abs().max())
    assert float((got - sep).abs().max()) <= 1e-6 * float(sep.abs().max())


@pytest.mark.parametrize("rows,K,N", [(64, 192, 192), (4 * 64 * 64, 192, 576), (4 * 128 * 128, 384, 192), (16384, 576, 192), (640, 192, 384),
                                      # the weight-stationary kernel (>= 4 token tiles per workgroup; a tile count the XCDs do not share evenly)
                                      (4 * 128 * 128 + 192, 192, 192), (24576, 192, 576), (32768 + 64, 192, 384),
                                      # half a panel / half a chunk (the 192 <-> 96 1x1 convs of Cheng2020-attn's attention blocks)
                                      (4 * 64 * 64, 192, 96), (4 * 64 * 64, 96, 192), (4096 + 64, 96, 96)])
@pytest.mark.parametrize("kind", ["act", "grad", "zero_rows"])
def test_linear_h2_per_token_scale_matches_float64(rows, K, N, kind):
    """rdo_linear_h2 (csrc/linear_h2.hip): Y = X W^T + b on fp16 two-way-split MFMA with a per-token dynamic power-of-two scale, against
    float64 -- for activation-like inputs, for gradient-like inputs (1e-6 and below, magnitudes differing by e^(+-9) from token to token:
    a per-TENSOR scale would drop the small tokens into fp16's denormals), and with all-zero tokens.  The error is measured per token
    against that token's own largest output (what a per-token scale promises): 2e-6, the level of the split-bf16 six-product kernel."""
    from hipops import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g)
    if kind == "grad":
        x = x * 1e-6 * torch.exp(3 * torch.randn(rows, 1, generator=g))
    if kind == "zero_rows":
        x[::3] = 0.0
        x[1, :] = 0.0
        x[1, 5] = 1e-30
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    assert ops.linear_h2_supported(rows, K, N) and not ops.linear_h2_supported(rows + 1, K, N) and not ops.linear_h2_supported(rows, K + 64, N)
    planes = ops.split_h2_linear(w.cuda())
    want = x.double() @ w.double().t()
    tok = want.abs().amax(1, keepdim=True)
    live = tok.squeeze(1) > 0
    y0 = ops.linear_h2(x.cuda(), planes, None).cpu().double()                  # the form the input gradients take: no bias
    assert float(((y0 - want).abs()[live] / tok[live]).max()) < 2e-6
    assert float(y0[~live].abs().max() if (~live).any() else 0.0) == 0.0       # a token of zeros gives exact zeros
    if kind == "act":                                                         # the GDN norm pool: beta' + gamma' . x^2 (squared input, positive weights)
        wp, bp = w.abs(), b.abs() + 0.1
        ysq = ops.linear_h2(x.cuda(), ops.split_h2_linear(wp.cuda()), bp.cuda(), square_input=True).cpu().double()
        ref = (x.double() ** 2) @ wp.double().t() + bp.double()
        assert float(((ysq - ref).abs() / ref.abs().amax(1, keepdim=True)).max()) < 2e-6
    if kind != "grad":                                                        # (a bias next to 1e-8 outputs would measure fp32's own rounding)
        y = ops.linear_h2(x.cuda(), planes, b.cuda()).cpu().double()
        ref = want + b.double()
        assert float(((y - ref).abs() / ref.abs().amax(1, keepdim=True)).max()) < 2e-6
        assert bool((y[~live] == b.double()).all())                           # ... and with a bias exactly the bias


@pytest.mark.parametrize("rows,K,N", [(4 * 64 * 64, 192, 384), (64, 192, 192), (4 * 128 * 128, 384, 192), (256, 192, 576),
                                      (4 * 128 * 128, 192, 384), (4 * 128 * 128 + 64, 192, 192)])      # the weight-stationary kernel
def test_linear_h2_gelu_epilogues(rows, K, N):
    """RDO_EPI_GELU (out = gelu(x W^T + b), pre-activation kept) and RDO_EPI_GELU_BWD (out = (dy W) * gelu'(aux)) of rdo_linear_h2 against
    the plain kernel followed by the separate GELU kernels: the linear part bit for bit, the activation to fp32 rounding."""
    from hipops import _lib as L, ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    planes = ops.split_h2_linear(w)
    pre = torch.empty(rows, N, device="cuda")
    y = ops.linear_h2(x, planes, b, epilogue=L.EPI_GELU, pre=pre)
    plain = ops.linear_h2(x, planes, b)
    torch.testing.assert_close(pre, plain, rtol=0, atol=0)
    sep = ops.gelu(plain.view(-1)).view_as(plain)
    assert float((y - sep).abs().max()) <= 1e-6 * float(sep.abs().max())
    aux = (torch.randn(rows, N, generator=g) * 2).cuda()
    got = ops.linear_h2(x, planes, None, epilogue=L.EPI_GELU_BWD, aux=aux)
    lin = ops.linear_h2(x, planes, None)
    sep = ops.gelu_bwd(lin.view(-1), aux.view(-1)).view_as(lin)
    assert float((got - sep).abs().max()) <= 1e-6 * float(sep.abs().max())


@pytest.mark.parametrize("rows,cin,cout", [(4 * 128 * 128, 192, 576), (4 * 64 * 64, 384, 192), (4096, 192, 192), (4 * 64 * 64 + 32 * 5, 576, 192),
                                           (4 * 64 * 64, 192, 96), (4 * 64 * 64, 96, 192), (8192 + 96, 96, 96)])      # half tiles, masked
@pytest.mark.parametrize("kind", ["act", "grad", "zero_blocks", "square"])
def test_linear_weight_gradient_h2_matches_float64(rows, cin, cout, kind):
    """The token-matrix weight gradient dW = dY^T X straight from fp32 operands on split-fp16 MFMA (linear_wgrad_h2_kernel behind
    rdo_conv2d_wgrad: 1 x 1, channel counts in blocks of 192, >= 4096 tokens) against float64: activation-like operands, gradient-like
    ones (1e-6 and below, magnitudes differing by e^(+-9) from token to token -- the per-stage power-of-two scale follows the largest
    stage seen so far), whole stages of zeros in front of and between the data, and the squared input of the GDN gamma gradient.
    Error against the largest gradient entry: 2e-6 (the split-bf16 kernel it replaces: 1e-6; an fp32 accumulation over 64 K terms: 4e-6)."""
    from hipops import ops
    g = torch.Generator().manual_seed(rows + cin + cout)
    x = torch.randn(rows, cin, generator=g)
    dy = torch.randn(rows, cout, generator=g)
    if kind == "grad":
        dy = dy * 1e-6 * torch.exp(3 * torch.randn(rows, 1, generator=g))
    if kind == "zero_blocks":
        dy[:96] = 0.0
        x[:64] = 0.0
        dy[1024:1024 + 320] = 0.0
        dy[2000] = 0.0
    sq = kind == "square"
    slabs = ops.conv2d_wgrad(x.view(1, 1, rows, cin).cuda(), dy.view(1, 1, rows, cout).cuda(), (cout, 1, 1, cin), 1, 0, square_input=sq)
    got = slabs.double().sum(0).view(cout, cin).cpu()
    xin = x.double() ** 2 if sq else x.double()
    want = dy.double().t() @ xin
    assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max())


@pytest.mark.parametrize("rows,N,epi", [(4 * 128 * 128, 192, "none"), (4 * 128 * 128 + 192, 384, "none"), (32768 + 64, 576, "gelu")])
def test_linear_h2_stationary_kernel_equals_streaming_kernel(rows, N, epi, monkeypatch):
    """The weight-stationary kernel (linear_h2w_kernel: inline-asm prefetch waited for with hand-counted `s_waitcnt vmcnt(n)`, n = the
    number of vector-memory instructions the compiler emits for an epilogue) against the streaming kernel on the same inputs, bit for
    bit: a compiler upgrade that changes that count makes the stationary kernel read its registers before the loads have landed --
    silently -- and shows here."""
    from hipops import _lib as L, ops
    g = torch.Generator().manual_seed(rows + N)
    x = (torch.randn(rows, 192, generator=g) * torch.exp(torch.randn(rows, 1, generator=g))).cuda()
    w = (torch.randn(N, 192, generator=g) / 192 ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    planes = ops.split_h2_linear(w)

    def run():
        if epi == "gelu":
            pre = torch.empty(rows, N, device="cuda")
            return ops.linear_h2(x, planes, b, epilogue=L.EPI_GELU, pre=pre), pre
        return ops.linear_h2(x, planes, b), None
    monkeypatch.setenv("RDO_LIN_H2_STATIONARY", "0")
    y0, p0 = run()
    monkeypatch.setenv("RDO_LIN_H2_STATIONARY", "1")
    for _ in range(3):                                     # (a late load is a race: look more than once)
        y1, p1 = run()
        assert torch.equal(y0, y1)
        if p0 is not None:
            assert torch.equal(p0, p1)


def test_round5_kernels_are_run_to_run_deterministic():
    """linear_wgrad_h2_kernel (LDS slot hand-over of the per-stage scale, two register sets in flight), the weight-stationary linear kernel
    (two LDS panels, hand-counted waits) and the attention kernels (persistent workgroups with a register prefetch) give the same bits on
    every run -- a race in one of those hand-overs would show here first (tools/determinism_check.py: the longer version)."""
    from hipops import ops
    g = torch.Generator(device="cuda").manual_seed(0)

    def same(fn, n=4):
        ref = fn().clone()
        return all(torch.equal(fn(), ref) for _ in range(n))

    x = torch.randn(1, 1, 65536, 192, device="cuda", generator=g)
    dy = torch.randn(1, 1, 65536, 576, device="cuda", generator=g) * torch.exp(3 * torch.randn(1, 1, 65536, 1, device="cuda", generator=g))
    assert same(lambda: ops.conv2d_wgrad(x, dy, (576, 1, 1, 192), 1, 0))
    w = torch.randn(576, 192, device="cuda", generator=g) / 192 ** 0.5
    pl = ops.split_h2_linear(w)
    assert same(lambda: ops.linear_h2(x.view(-1, 192), pl, None))
    for B, H, heads, shift in [(4, 128, 4, 4), (4, 64, 8, 0)]:
        d = ops.attn_desc(B, H, H, 192, heads, 8, shift)
        qkv = torch.randn(B, H, H, 576, device="cuda", generator=g)
        bias = torch.randn(heads, 64, 64, device="cuda", generator=g)
        dout = torch.randn(B, H, H, 192, device="cuda", generator=g)
        assert same(lambda: ops.window_attention(d, qkv, bias))
        assert same(lambda: ops.window_attention_bwd(d, qkv, bias, dout))
