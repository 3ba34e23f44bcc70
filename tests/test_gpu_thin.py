"""Thin convolutions: few output channels (conv_thincout.hip: the 192 -> 12 last layer) and few input channels (conv_thin.hip: the
RGB stem) against float64 torch, forward and weight gradient, shapes that exercise every template instance and the image borders."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rdo-ptq_amd"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from hipops import ops as o
    return o


@pytest.fixture(scope="module")
def L():
    from hipops import _lib
    return _lib


THINCOUT = [(4, 128, 128, 192, 12), (1, 32, 48, 64, 16), (2, 16, 16, 32, 3), (1, 16, 32, 16, 1), (3, 48, 16, 80, 12), (1, 64, 64, 128, 12)]


@pytest.mark.parametrize("B,H,W,Cin,Cout", THINCOUT)
@pytest.mark.parametrize("epi", ["none", "lrelu"])
def test_thincout_forward_matches_float64(ops, L, B, H, W, Cin, Cout, epi):
    g = torch.Generator(device="cuda").manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    out = ops.conv2d_fwd(x, w, b, 1, 1, epilogue=L.EPI_LRELU if epi == "lrelu" else L.EPI_NONE)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    if epi == "lrelu":
        ref = F.leaky_relu(ref, 0.01)
    assert (out.double() - ref).abs().max() <= 2e-6 * ref.abs().max()      # fp32 MFMA products and sums over K = 9 Cin


@pytest.mark.parametrize("B,H,W,Cin,Cout", THINCOUT)
def test_thincout_weight_gradient_matches_float64(ops, B, H, W, Cin, Cout):
    g = torch.Generator(device="cuda").manual_seed(H + W + Cin * Cout)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    dy = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    w_shape = (Cout, 3, 3, Cin)
    assert ops.wgrad_nsplit(tuple(x.shape), w_shape, 1, 1) == B * (H // 16) * (W // 16)       # one slab per 16 x 16 patch
    slabs = ops.conv2d_wgrad(x, dy, w_shape, 1, 1)
    got = ops.reduce_slabs(slabs)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    assert (got.double() - ref).abs().max() <= 3e-6 * ref.abs().max()
    again = ops.reduce_slabs(ops.conv2d_wgrad(x, dy, w_shape, 1, 1))
    assert torch.equal(again, got)                                                          # no atomics: bit-reproducible


@pytest.mark.parametrize("B,H,Cin,Cout,K,s,p", [(4, 256, 3, 192, 3, 2, 1), (4, 256, 3, 192, 1, 2, 0), (2, 64, 3, 128, 3, 2, 1), (1, 40, 1, 70, 3, 1, 1)])
def test_thin_input_weight_gradient_with_many_pixel_splits(ops, B, H, Cin, Cout, K, s, p):
    g = torch.Generator(device="cuda").manual_seed(H + Cout)
    x = torch.randn(B, H, H, Cin, device="cuda", generator=g)
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda", generator=g)
    got = ops.reduce_slabs(ops.conv2d_wgrad(x, dy, (Cout, K, K, Cin), s, p))
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, K, K), dy.permute(0, 3, 1, 2).double(), stride=s, padding=p).permute(0, 2, 3, 1)
    assert (got.double() - ref).abs().max() <= 3e-6 * ref.abs().max()
