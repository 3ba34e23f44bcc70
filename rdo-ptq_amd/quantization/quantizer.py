"""Quantisers of the RDO-PTQ calibration path with the reference's class surface, computing through librdoptq_hip.

Mirrors /root/reference/task-oriented-PTQ/quantization/quantizer.py: `UniformAffineQuantizer` (:123-393),
`AdaRoundQuantizer` (:397-470), `ActQuantizer` (:81-121), `round_ste` (:64-68), `lp_loss` (:71-79),
`StraightThrough` (:12-17).  Weights are handed to the kernels in row-major "rows x inner" form where a row is one
quantisation channel: OHWI for conv weights (channel = dim 0), the matrix itself for GDN gammas / Linear weights.
There is no CPU implementation: tensors must live on the GPU."""
import logging

import torch
import torch.nn as nn
import torch.nn.functional as F

from hipops import ops

GAMMA, ZETA = -0.1, 1.1
MAX_BITS = 16


class StraightThrough(nn.Module):
    def __init__(self, channel_num: int = 1):
        super().__init__()

    def forward(self, input):
        return input


def round_ste(x: torch.Tensor):
    return (x.round() - x).detach() + x


def lp_loss(pred, tgt, p=2.0, reduction="none"):
    diff = (pred - tgt).abs().pow(p)
    return diff.sum(1).mean() if reduction == "none" else diff.mean()


# ----------------------------------------------------------------------------- layout helpers
def to_rows(w: torch.Tensor, tconv: bool = False):
    """Weight (logical OIHW / [out,in] / 1-D) -> contiguous kernel layout whose dim 0 is the quantisation channel."""
    if w.dim() == 4:
        return (w.permute(1, 2, 3, 0) if tconv else w.permute(0, 2, 3, 1)).contiguous()   # tconv: channel = dim 1
    if w.dim() == 1:
        return w.reshape(1, -1).contiguous()
    return w.contiguous()


def from_rows(wr: torch.Tensor, like: torch.Tensor, tconv: bool = False):
    """Inverse of to_rows as a zero-copy view with the logical shape of `like`."""
    if like.dim() == 4:
        return wr.permute(3, 0, 1, 2) if tconv else wr.permute(0, 3, 1, 2)
    if like.dim() == 1:
        return wr.reshape(-1)
    return wr


def _scale_shape(w: torch.Tensor, tconv: bool):
    if w.dim() == 4:
        return (1, -1, 1, 1) if tconv else (-1, 1, 1, 1)
    if w.dim() == 1:
        return (-1,)
    return (-1, 1)


# ----------------------------------------------------------------------------- activation quantisation
def ActQuant(x: torch.Tensor, n_bits: int = 8):
    """Dynamic per-channel quant-dequant of a detached copy (channel = dim 1 for 4-D, last dim for 3-D, dim 1 for 2-D).
    The reference hard-wires 8 bits (`Handle_Parameter(param, b_w=8)`, quantizer.py:81) whatever --n_bits_a says; `n_bits` is
    the extension BASELINE config "W10A10" needs (UniformAffineQuantizer(dynamic_bits=10))."""
    x = x.detach()
    if x.dim() == 4:
        xr = x.permute(0, 2, 3, 1).contiguous()
        return ops.actquant_perchannel(xr, n_bits=n_bits).permute(0, 3, 1, 2)
    if x.dim() in (2, 3):
        return ops.actquant_perchannel(x.contiguous(), n_bits=n_bits)
    return ops.actquant_perchannel(x.reshape(-1, 1).contiguous(), n_bits=n_bits).reshape(x.shape)


def ActQuantizer(x: torch.Tensor, n_bits: int = 8):
    return ActQuant(x, n_bits)


# ----------------------------------------------------------------------------- uniform affine quantiser
class UniformAffineQuantizer(nn.Module):
    """Asymmetric uniform fake-quantiser; scales are initialised lazily on the first weight it sees."""

    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False, scale_method: str = "max",
                 leaf_param: bool = False, tconv: bool = False, act: bool = False, prob: float = 1.0,
                 dynamic_bits: int = None):
        super().__init__()
        # width of the dynamic activation grid: None = the reference's fixed 8 bits (its ActQuant ignores n_bits_a,
        # quantizer.py:81,158-159); an explicit value is this build's extension (e.g. 10 for W10A10)
        assert dynamic_bits is None or 2 <= dynamic_bits <= MAX_BITS, "bitwidth not supported"
        self.dynamic_bits = 8 if dynamic_bits is None else int(dynamic_bits)
        # the reference stops at 8 bits (quantizer.py:139); BASELINE config 3 (W10A10) needs wider weight grids, which the
        # kernels handle unchanged (levels are fp32-valued integers): accepted up to 16 bits as an extension
        assert 2 <= n_bits <= MAX_BITS, "bitwidth not supported"
        self.sym = symmetric
        self.n_bits = n_bits
        self.n_levels = 2 ** n_bits
        self.delta = None
        self.zero_point = None
        self.inited = False
        self.leaf_param = leaf_param
        self.channel_wise = channel_wise
        self.scale_method = scale_method
        self.tconv = tconv
        self.act = act
        self.prob = prob
        self.is_training = False

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        for name in ("delta", "zero_point"):
            t = getattr(self, name, None)
            if torch.is_tensor(t):
                setattr(self, name, fn(t))
        return self

    # -- scale initialisation -------------------------------------------------------------------------------------
    def _rows(self, x):
        wr = to_rows(x, self.tconv) if (self.channel_wise and x.dim() != 1) else x.reshape(1, -1).contiguous()
        return wr

    def init_quantization_scale(self, x: torch.Tensor, channel_wise: bool = False):
        rows2d = (to_rows(x, self.tconv) if (channel_wise and x.dim() != 1) else x.reshape(1, -1)).contiguous()
        flat = rows2d.reshape(rows2d.shape[0], -1)
        m = self.scale_method
        if "max" in m and "scale" not in m and not self.sym:
            delta, zp = ops.uaq_init_minmax(flat, self.n_levels)
        else:
            delta, zp = self._init_search(flat)
        shape = _scale_shape(x, self.tconv) if channel_wise else ()
        return delta.reshape(shape), zp.reshape(shape)

    def _init_search(self, flat):
        """'max' variants with scaling/symmetry, 'gaussian', and the 10-step shrink searches ('mse' = L3.5, 'l1', 'l2'),
        vectorised over channels (the reference loops over channels in Python, quantizer.py:260-265)."""
        m, L = self.scale_method, self.n_levels
        eps = torch.tensor(1e-8, device=flat.device)
        if m == "gaussian":
            # tensor arithmetic in fp32 throughout, as in the reference (quantizer.py:318-335; its 'scale' branch is dead there)
            mu, var = flat.mean(1), flat.var(1)
            lo, hi = torch.clamp(mu - 6 * var, max=0), torch.clamp(mu + 6 * var, min=0)
            if self.sym:
                amax = torch.maximum(lo.abs(), hi)
                lo, hi = torch.where(lo < 0, -amax, torch.zeros_like(lo)), amax
            delta = torch.maximum((hi - lo) / (L - 1), eps)
            return delta, (-lo / delta).round()
        if "max" in m:
            # the reference holds x_min / x_max as Python floats (double) from here on: scaling, symmetrisation and the range
            # are evaluated in double and rounded to fp32 ONCE (quantizer.py:282-293); doing the scaling in fp32 can move delta
            # by an ulp and flip a zero point that sits on x.5
            lo, hi = torch.clamp(flat.amin(1), max=0).double(), torch.clamp(flat.amax(1), min=0).double()
            if "scale" in m:
                lo, hi = lo * (self.n_bits + 2) / 8, hi * (self.n_bits + 2) / 8
            if self.sym:
                amax = torch.maximum(lo.abs(), hi)
                lo, hi = torch.where(lo < 0, -amax, torch.zeros_like(lo)), amax
            delta = torch.maximum(((hi - lo) / (L - 1)).float(), eps)
            # the reference divides a Python float by the tensor, which torch evaluates as delta.reciprocal() * (-x_min)
            # (quantizer.py:296) -- one more rounding, it decides ties at x.5
            return delta, (delta.reciprocal() * (-lo).float()).round()
        if m not in ("mse", "l1", "l2"):
            raise NotImplementedError(m)
        hi0, lo0 = flat.amax(1, keepdim=True), flat.amin(1, keepdim=True)
        best = torch.full((flat.shape[0],), 1e10, device=flat.device)
        delta = torch.zeros_like(best)
        zp = torch.zeros_like(best)
        for i in range(10):
            hi, lo = hi0 * (1.0 - i * 0.05), lo0 * (1.0 - i * 0.05)
            d = torch.maximum((hi - lo) / (L - 1), eps)
            z = (-lo / d).round()
            xq = (torch.clamp(torch.round(flat / d) + z, 0, L - 1) - z) * d
            err = (flat - xq).abs()
            score = err.pow(3.5).mean(1) if m == "mse" else (err.mean(1) if m == "l1" else err.pow(2).mean(1))
            better = score < best
            best = torch.where(better, score, best)
            delta = torch.where(better, d.squeeze(1), delta)
            zp = torch.where(better, z.squeeze(1), zp)
        return delta, zp

    # -- forward ---------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, act: bool = False):
        if act:
            return ActQuantizer(x, getattr(self, "dynamic_bits", 8))
        if not self.inited:
            if self.leaf_param:
                return x
            self.delta, self.zero_point = self.init_quantization_scale(x.detach(), self.channel_wise)
            self.inited = True
        wr = self._rows(x.detach())
        desc = ops.ada_desc(wr.reshape(wr.shape[0], -1), self.n_levels, conv_layout=False)
        wq = ops.uaq_fakequant(desc, wr, self.delta.reshape(-1).contiguous().expand(wr.shape[0]).contiguous(),
                               self.zero_point.reshape(-1).contiguous().expand(wr.shape[0]).contiguous())
        if self.channel_wise and x.dim() != 1:
            return from_rows(wq, x, self.tconv)
        return wq.reshape(x.shape)

    def bitwidth_refactor(self, refactored_bit: int):
        assert 2 <= refactored_bit <= MAX_BITS, "bitwidth not supported"
        self.n_bits = refactored_bit
        self.n_levels = 2 ** refactored_bit

    def extra_repr(self):
        return (f"bit={self.n_bits}, scale_method={self.scale_method}, symmetric={self.sym}, "
                f"channel_wise={self.channel_wise}, leaf_param={self.leaf_param}")


# ----------------------------------------------------------------------------- AdaRound
class AdaRoundQuantizer(nn.Module):
    """Learned rounding.  `alpha` is an nn.Parameter with the logical shape of the weight; its storage is the kernel
    layout (OHWI for conv weights), shared with the calibration engine that trains it."""

    def __init__(self, uaq: UniformAffineQuantizer, weight_tensor: torch.Tensor, round_mode="learned_round_sigmoid",
                 alpha_rows: torch.Tensor = None):
        super().__init__()
        self.n_bits, self.sym, self.n_levels = uaq.n_bits, uaq.sym, uaq.n_levels
        self.delta, self.zero_point = uaq.delta, uaq.zero_point
        self.tconv, self.channel_wise = uaq.tconv, uaq.channel_wise
        self.round_mode = round_mode
        self.soft_targets = False
        self.gamma, self.zeta, self.beta = GAMMA, ZETA, 2 / 3
        self._like = weight_tensor
        if round_mode != "learned_hard_sigmoid":
            raise NotImplementedError(round_mode)
        wr = self._rows(weight_tensor.detach())
        if alpha_rows is None:
            logging.info("Init alpha to be FP32")
            alpha_rows = ops.adaround_init_alpha(self._desc(wr), wr, self._row_scales(wr)[0])
        self.alpha = nn.Parameter(self._unrows(alpha_rows, weight_tensor))

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        for name in ("delta", "zero_point"):
            t = getattr(self, name, None)
            if torch.is_tensor(t):
                setattr(self, name, fn(t))
        return self

    def _rows(self, x):
        return to_rows(x, self.tconv) if (self.channel_wise and x.dim() != 1) else x.reshape(1, -1).contiguous()

    def _unrows(self, r, like):
        return from_rows(r, like, self.tconv) if (self.channel_wise and like.dim() != 1) else r.reshape(like.shape)

    def _desc(self, wr):
        return ops.ada_desc(wr.reshape(wr.shape[0], -1), self.n_levels, conv_layout=False)

    def _row_scales(self, wr):
        n = wr.shape[0]
        return (self.delta.reshape(-1).expand(n).contiguous(), self.zero_point.reshape(-1).expand(n).contiguous())

    def forward(self, x):
        wr = self._rows(x.detach())
        ar = self._rows(self.alpha.detach())
        d, z = self._row_scales(wr)
        wq = ops.adaround_fwd(self._desc(wr), wr, ar, d, z, self.soft_targets)
        return self._unrows(wq, x)

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def extra_repr(self):
        return f"bit={self.n_bits}"
