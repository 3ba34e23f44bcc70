"""Input-gradient `torch.autograd.Function`s over the HIP kernels.

The calibration engines hand-record their backward passes; everything else in this package runs without autograd.  These
Functions exist for the two places where torch's tape is the natural driver: the opt-in R + lambda*D task loss of the calibration
loop (`loss_mode='rd'`: the unit's soft-quantised output is pushed through the REST of the wrapped model and the
rate-distortion loss of `losses.RateDistortionLoss` is differentiated back to it -- the term the reference sketches and comments
out, layer_opt.py:146-148), and callers that want `QuantModule.forward` outputs with a `grad_fn` with respect to their INPUT.
Weights are constants here (the FP / hard-quantised weights of the modules behind the unit): only input gradients are produced.

Tensors cross this boundary as NCHW views of NHWC storage, like everywhere in the package; every forward and backward GEMM and
every GDN / likelihood stage below is a librdoptq_hip kernel."""
import math

import torch

from . import _lib as L
from . import ops


LIN_H2_MIN_ROWS = 4096      # token count from which the tape's Linears take rdo_linear_h2 (below it the conv kernels' split-K forms win)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(y):
    return y.permute(0, 3, 1, 2)


def _pack(w):
    return w if isinstance(w, ops.WeightPack) else ops.WeightPack(w)


def _act_bwd(gr, y, epilogue):
    if epilogue == L.EPI_LRELU:
        return ops.lrelu_bwd(gr, y)
    if epilogue == L.EPI_RELU:
        return ops.relu_bwd(gr, y)
    return gr


class Conv2dFn(torch.autograd.Function):
    """y = act(conv2d(x, w) + b); `w_rows` = OHWI weight [Cout,KH,KW,Cin] or an `ops.WeightPack` of it (the caller keeps the pack
    while the weight is unchanged: planes, flipped and transposed forms are then built once); epilogue in {EPI_NONE, EPI_LRELU,
    EPI_RELU}.  Forward and input gradient take the split-precision MFMA kernels wherever rdo_conv2d_fwd does."""

    @staticmethod
    def forward(ctx, x, w_rows, bias, stride, pad, epilogue):
        xr = _nhwc(x)
        pack = _pack(w_rows)
        y = ops.conv2d_fwd(xr, pack.w, bias, stride, pad, epilogue=epilogue, wplanes=pack.planes(xr.shape, stride, pad))
        ctx.geom = (stride, pad, epilogue, tuple(xr.shape), pack)
        ctx.save_for_backward(y if epilogue != L.EPI_NONE else None)
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        stride, pad, epilogue, xs, pack = ctx.geom
        (y,) = ctx.saved_tensors
        gr = _act_bwd(_nhwc(g), y, epilogue)
        K = pack.w.shape[1]
        if stride == 1 and 2 * pad == K - 1:
            dx = ops.conv2d_fwd_pack(gr, pack.flipped(), 1, K - 1 - pad)     # [Cin][KH'][KW'][Cout], taps flipped
        else:
            # dgrad of a strided conv = the transposed conv with the same weight tensor read as [Cin_t = Cout][Cout_t = Cin][K][K]
            out_pad = (xs[1] + 2 * pad - K) % stride
            dx = ops.conv_transpose2d(gr, None, None, stride, pad, out_pad, pack=pack.transposed())
        return _nchw(dx), None, None, None, None, None


class ConvTranspose2dFn(torch.autograd.Function):
    """y = act(conv_transpose2d(x, W) + b); `w_t_rows` = to_rows(W, tconv=True) = [Cout,KH,KW,Cin] (un-flipped taps) or an
    `ops.WeightPack` of it WITH the bias (the phase bias is derived from it)."""

    @staticmethod
    def forward(ctx, x, w_t_rows, bias, stride, pad, output_padding, epilogue):
        xr = _nhwc(x)
        pack = w_t_rows if isinstance(w_t_rows, ops.WeightPack) else ops.WeightPack(w_t_rows, bias)
        y = ops.conv_transpose2d(xr, None, None, stride, pad, output_padding, epilogue=epilogue, pack=pack)
        ctx.geom = (stride, pad, epilogue, pack)
        ctx.save_for_backward(y if epilogue != L.EPI_NONE else None)
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        stride, pad, epilogue, pack = ctx.geom
        (y,) = ctx.saved_tensors
        gr = _act_bwd(_nhwc(g), y, epilogue)
        # dx = conv2d(dy, W read as a conv weight [O = Cin_t][I = Cout_t][K][K]) with the same stride / padding
        dx = ops.conv2d_fwd_pack(gr, pack.transposed(), stride, pad)
        return _nchw(dx), None, None, None, None, None, None


class GDNFn(torch.autograd.Function):
    """y = x * (beta' + sum_j gamma'_ij x_j^2)^(-1/2 | +1/2); gamma_p [C,C] (or an `ops.WeightPack` of it as [C,1,1,C]) and
    beta_p [C] already re-parametrised."""

    @staticmethod
    def forward(ctx, x, gamma_p, beta_p, inverse):
        xr = _nhwc(x)
        c = xr.shape[-1]
        pack = gamma_p if isinstance(gamma_p, ops.WeightPack) else ops.WeightPack(gamma_p.reshape(c, 1, 1, c))
        norm = torch.empty_like(xr)
        y = ops.conv2d_fwd(xr, pack.w, beta_p.contiguous(), 1, 0, epilogue=L.EPI_IGDN if inverse else L.EPI_GDN, aux=xr,
                           square_input=True, pre=norm, wplanes=pack.planes(xr.shape, 1, 0))
        ctx.inverse = bool(inverse)
        ctx.pack = pack
        ctx.save_for_backward(xr, norm)
        return _nchw(y)

    @staticmethod
    def backward(ctx, g):
        xr, norm = ctx.saved_tensors
        gr = _nhwc(g)
        t = ops.gdn_bwd_t(gr, xr, norm, ctx.inverse)
        acc = ops.conv2d_fwd_pack(t, ctx.pack.transposed(), 1, 0)           # t . gamma'
        dx = ops.gdn_bwd_dx(gr, xr, norm, acc, ctx.inverse)
        return _nchw(dx), None, None, None


class GaussianLikelihoodFn(torch.autograd.Function):
    """Likelihood of the (already rounded) latent under N(means, max(scales, bound)) integrated over the unit bin, with gradients
    to y_hat, scales and means through rdo_gaussian_likelihood_bwd (which yields d(-log2 p); dp = -p ln2 d(-log2 p))."""

    @staticmethod
    def forward(ctx, yhat, scales, means, scale_bound):
        yr, sr, mr = _nhwc(yhat), _nhwc(scales), _nhwc(means)
        _, lik = ops.gaussian_likelihood(yr, sr, mr, scale_bound)
        ctx.bound = scale_bound
        ctx.save_for_backward(yr, sr, mr, lik)
        return _nchw(lik)

    @staticmethod
    def backward(ctx, g):
        yr, sr, mr, lik = ctx.saved_tensors
        ds, dm = ops.gaussian_likelihood_bwd(yr, sr, mr, 1.0, ctx.bound)
        f = _nhwc(g) * lik * (-math.log(2.0))
        f = torch.where(lik > 1e-9, f, torch.zeros_like(f))                  # the likelihood floor has no gradient
        ds, dm = ds * f, dm * f
        return _nchw(-dm), _nchw(ds), _nchw(dm), None


class FactorizedLikelihoodFn(torch.autograd.Function):
    """(z^, likelihood) of the factorised prior for a 4-D latent in evaluation mode: rounding about the medians with a straight-through
    gradient, the likelihood through rdo_factorized_likelihood_fwd / _bwd (which yields d(-log2 p)/dz^; dp = -p ln2 d(-log2 p))."""

    @staticmethod
    def forward(ctx, z, params, medians):
        zr = _nhwc(z)
        zhat, lik = ops.factorized_likelihood(zr, params, medians)
        ctx.save_for_backward(zhat, lik, params)
        return _nchw(zhat), _nchw(lik)

    @staticmethod
    def backward(ctx, g_zhat, g_lik):
        zhat, lik, params = ctx.saved_tensors
        dz = ops.factorized_likelihood_bwd(zhat, params, 1.0)
        f = _nhwc(g_lik) * lik * (-math.log(2.0))
        f = torch.where(lik > 1e-9, f, torch.zeros_like(f))                  # the likelihood floor has no gradient
        return g_zhat + _nchw(dz * f), None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b over the last dimension (F.linear of the Swin blocks, quant_layer.py:119); `pack` = ops.WeightPack of W as a 1x1
    conv weight [Cout, 1, 1, Cin] with the bias.  Forward and input gradient on the conv kernels (split precision where they apply)."""

    @staticmethod
    def forward(ctx, x, pack):
        cin, cout = x.shape[-1], pack.w.shape[0]
        x4 = x.reshape(1, 1, -1, cin).contiguous()
        rows = x4.shape[2]
        if rows >= LIN_H2_MIN_ROWS and ops.linear_h2_supported(rows, cin, cout):     # large token matrices: rdo_linear_h2 (three fp16 products)
            y = ops.linear_h2(x4.view(rows, cin), pack.lin_planes(), pack.bias)
        else:
            y = ops.conv2d_fwd_pack(x4, pack, 1, 0)
        ctx.pack, ctx.xshape = pack, tuple(x.shape)
        return y.reshape(*x.shape[:-1], cout)

    @staticmethod
    def backward(ctx, g):
        cout, cin = ctx.pack.w.shape[0], ctx.xshape[-1]
        g4 = g.reshape(1, 1, -1, cout).contiguous()
        rows = g4.shape[2]
        if rows >= LIN_H2_MIN_ROWS and ops.linear_h2_supported(rows, cout, cin):
            dx = ops.linear_h2(g4.view(rows, cout), ctx.pack.lin_planes(transposed=True), None)
        else:
            dx = ops.conv2d_fwd_pack(g4, ctx.pack.flipped(), 1, 0)        # [Cin][1][1][Cout]
        return dx.reshape(ctx.xshape), None


class LayerNormFn(torch.autograd.Function):
    """F.layer_norm over the last dimension with constant weight / bias (quant_layer.py:44-49,121): rdo_layer_norm / rdo_layer_norm_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xc = x.contiguous()
        ctx.save_for_backward(xc, weight)
        ctx.eps = eps
        return ops.layer_norm(xc, weight, bias, eps)

    @staticmethod
    def backward(ctx, g):
        xc, weight = ctx.saved_tensors
        dx = torch.empty_like(xc)
        ops.layer_norm_bwd(xc, weight, g.contiguous(), ctx.eps, dx=dx)
        return dx, None, None, None


class WindowAttentionFn(torch.autograd.Function):
    """The (shifted-)window attention core on qkv [B, H, W, 3C] in natural pixel order (models/layers.py:138-170 with the roll / window
    partition / mask of :271-294 folded in): rdo_window_attention_fwd / _bwd (probabilities recomputed in the backward)."""

    @staticmethod
    def forward(ctx, qkv, desc, bias):
        qc = qkv.contiguous()
        ctx.desc = desc
        ctx.save_for_backward(qc, bias)
        return ops.window_attention(desc, qc, bias)

    @staticmethod
    def backward(ctx, g):
        qc, bias = ctx.saved_tensors
        return ops.window_attention_bwd(ctx.desc, qc, bias, g.contiguous()).reshape(qc.shape), None, None


class GeluFn(torch.autograd.Function):
    """nn.GELU() (exact erf form): rdo_gelu_fwd / rdo_gelu_bwd."""

    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        return ops.gelu(xc)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return ops.gelu_bwd(g.contiguous(), xc)


def round_ste(x):
    return x + (torch.round(x) - x).detach()


class NegLog2SumFn(torch.autograd.Function):
    """scale * sum(-log2 p): the rate term of losses.RateDistortionLoss (rdo_neg_log2_sum) with its gradient."""

    @staticmethod
    def forward(ctx, lik, scale):
        flat = lik.detach().contiguous().reshape(-1)
        ctx.scale = scale
        ctx.save_for_backward(lik)
        return ops.neg_log2_sum(flat, scale).reshape(())

    @staticmethod
    def backward(ctx, g):
        (lik,) = ctx.saved_tensors
        return g * (-ctx.scale / math.log(2.0)) / lik, None


class SqDiffSumFn(torch.autograd.Function):
    """scale * sum((a - b)^2): the distortion term (rdo_sq_diff_sum) with its gradient with respect to a."""

    @staticmethod
    def forward(ctx, a, b, scale):
        ac, bc = a.detach().contiguous(), b.detach().contiguous()
        ctx.scale = scale
        ctx.save_for_backward(ac, bc)
        return ops.sq_diff_sum(ac.reshape(-1), bc.reshape(-1), scale).reshape(())

    @staticmethod
    def backward(ctx, g):
        ac, bc = ctx.saved_tensors
        return g * (2.0 * ctx.scale) * (ac - bc), None, None
