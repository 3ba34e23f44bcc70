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
