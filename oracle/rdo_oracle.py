"""ORACLE (test infrastructure, not product code).

CPU / torch-fp32 restatement of the RDO-PTQ calibration hot path of the reference
(`/root/reference/task-oriented-PTQ`).  Every function cites the reference lines it
follows.  It is pinned against golden vectors produced by the reference's *own* code
imported in the authoring container (`tools/make_golden.py` -> `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).  The CompressAI arithmetic it leans on
(GDN re-parametrisation) lives in `oracle/lic_oracle.py` and is "parity unpinned".

Only `tests/`, `tools/make_golden.py`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this module; the product (`rdo-ptq_amd/`) never does.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .lic_oracle import NonNegativeParametrizer

GAMMA, ZETA = -0.1, 1.1          # quantizer.py:423


# ----------------------------------------------------------------------------- small pieces
def round_ste(x: torch.Tensor) -> torch.Tensor:
    """quantizer.py:64-68."""
    return (x.round() - x).detach() + x


def lp_loss(pred, tgt, p=2.0, reduction="none"):
    """quantizer.py:71-79 -- sum over dim 1, mean over the rest (or plain mean)."""
    d = (pred - tgt).abs().pow(p)
    return d.sum(1).mean() if reduction == "none" else d.mean()


def linear_temp_decay(t, t_max, rel_start_decay, start_b, end_b):
    """utils.py:37-54 (named 'cosine' in the docstring there, linear in code)."""
    start = rel_start_decay * t_max
    if t < start:
        return start_b
    rel_t = (t - start) / (t_max - start)
    return end_b + (start_b - end_b) * max(0.0, 1 - rel_t)


# ----------------------------------------------------------------------------- activation quantisation
def act_quant(x: torch.Tensor, b_w: int = 8) -> torch.Tensor:
    """quantizer.py:81-117 `Handle_Parameter`/`ActQuant`: dynamic per-channel 8-bit quant-dequant of
    a detached clone.  The reference loops over channels in Python; this is the same arithmetic
    vectorised (channel = dim 1 for 4-D, last dim for 3-D, dim 1 for 2-D, whole tensor otherwise)."""
    x = x.clone().detach()
    if x.dim() == 4:
        dims = (0, 2, 3)
    elif x.dim() == 3:
        dims = (0, 1)
    elif x.dim() == 2:
        dims = (0,)
    else:
        dims = tuple(range(x.dim()))
    bit_range = 2 ** b_w - 1
    zp = x.amin(dim=dims, keepdim=True)
    xn = x - zp
    rng = torch.clamp(xn.abs().amax(dim=dims, keepdim=True), min=1e-6)
    q = torch.round(torch.clamp(xn / rng, -1, 1) * bit_range)
    return (q / bit_range) * rng + zp


# ----------------------------------------------------------------------------- uniform affine quantiser
_EPS = torch.tensor(1e-8, dtype=torch.float32)


def _uaq_quantize(x, mx, mn, n_bits):
    """quantizer.py:376-383."""
    n_levels = 2 ** n_bits
    delta = torch.max((mx - mn) / (2 ** n_bits - 1), _EPS)
    zp = (-mn / delta).round()
    x_q = torch.clamp(torch.round(x / delta) + zp, 0, n_levels - 1)
    return (x_q - zp) * delta


def _uaq_init_tensor(x, n_bits, scale_method, sym=False):
    """quantizer.py:280-372, the non-channel-wise branch."""
    n_levels = 2 ** n_bits
    if "max" in scale_method:
        x_min = min(x.min().item(), 0)
        x_max = max(x.max().item(), 0)
        if "scale" in scale_method:
            x_min = x_min * (n_bits + 2) / 8
            x_max = x_max * (n_bits + 2) / 8
        x_absmax = max(abs(x_min), x_max)
        if sym:
            x_min, x_max = (-x_absmax if x_min < 0 else 0), x_absmax
        delta = torch.max(torch.tensor((x_max - x_min) / (n_levels - 1)), _EPS)
        zp = (-x_min / delta).round()
        return delta.type_as(x), zp.type_as(x)
    if scale_method == "gaussian":
        mu, sigma = torch.mean(x), torch.var(x)           # NB: variance, as in the reference (:320)
        x_min, x_max = min(mu - 6 * sigma, 0), max(mu + 6 * sigma, 0)
        x_absmax = max(abs(x_min), x_max)
        if sym:
            x_min, x_max = (-x_absmax if x_min < 0 else 0), x_absmax
        delta = torch.max(torch.as_tensor((x_max - x_min) / (n_levels - 1), dtype=torch.float32), _EPS)
        zp = (-x_min / delta).round()
        return delta.type_as(x), torch.as_tensor(zp).type_as(x)
    if scale_method in ("mse", "l1", "l2"):
        x_max, x_min = x.max(), x.min()
        best, delta, zp = 1e10, None, None
        for i in range(10):
            new_max = x_max * (1.0 - (i * 0.05))
            new_min = x_min * (1.0 - (i * 0.05))
            x_q = _uaq_quantize(x, new_max, new_min, n_bits)
            if scale_method == "mse":
                score = lp_loss(x, x_q, p=3.5, reduction="all")
            elif scale_method == "l1":
                score = F.l1_loss(x, x_q)
            else:
                score = F.mse_loss(x, x_q)
            if score < best:
                best = score
                delta = torch.max((new_max - new_min) / (2 ** n_bits - 1), _EPS)
                zp = (-new_min / delta).round()
        return delta, zp
    raise NotImplementedError(scale_method)


def uaq_init(x: torch.Tensor, n_bits=8, channel_wise=False, scale_method="max", tconv=False, sym=False):
    """quantizer.py:233-374 `init_quantization_scale` -> (delta, zero_point) with the reference's shapes:
    conv [Co,1,1,1], tconv [1,Co,1,1], 2-D [rows,1], 1-D whole-tensor scalar viewed (-1)."""
    if not channel_wise:
        return _uaq_init_tensor(x, n_bits, scale_method, sym)
    xc = x.clone().detach()
    if x.dim() == 1:
        d, z = _uaq_init_tensor(xc, n_bits, scale_method, sym)
        return d.view(-1), z.view(-1)
    n_ch = xc.shape[1] if tconv else xc.shape[0]
    delta = torch.empty(n_ch, dtype=x.dtype)
    zp = torch.empty(n_ch, dtype=x.dtype)
    for c in range(n_ch):
        sl = xc[:, c] if tconv else xc[c]
        delta[c], zp[c] = _uaq_init_tensor(sl, n_bits, scale_method, sym)
    if x.dim() == 4:
        shape = (1, -1, 1, 1) if tconv else (-1, 1, 1, 1)
    else:
        shape = (-1, 1)
    return delta.view(shape), zp.view(shape)


def uaq_fakequant(x, delta, zp, n_levels):
    """quantizer.py:175-177 (nearest rounding with STE)."""
    x_int = round_ste(x / delta) + zp
    return (torch.clamp(x_int, 0, n_levels - 1) - zp) * delta


# ----------------------------------------------------------------------------- AdaRound
def adaround_init_alpha(w, delta):
    """quantizer.py:454-462: sigmoid-inverse of the rounding residual."""
    x_floor = torch.floor(w / delta)
    rest = (w / delta) - x_floor
    return -torch.log((ZETA - GAMMA) / (rest - GAMMA) - 1)


def adaround_soft_targets(alpha):
    """quantizer.py:451-452 rectified sigmoid h(alpha)."""
    return torch.clamp(torch.sigmoid(alpha) * (ZETA - GAMMA) + GAMMA, 0, 1)


def adaround_forward(w, alpha, delta, zp, n_levels, soft: bool):
    """quantizer.py:437-449 ('learned_hard_sigmoid')."""
    x_floor = torch.floor(w / delta)
    x_int = x_floor + (adaround_soft_targets(alpha) if soft else (alpha >= 0).float())
    x_quant = torch.clamp(x_int + zp, 0, n_levels - 1)
    return (x_quant - zp) * delta


def round_loss_term(alpha, b, weight):
    """layer_opt.py:164-165: weight * sum(1 - |2h-1|^b)."""
    rv = adaround_soft_targets(alpha)
    return weight * (1 - ((rv - .5).abs() * 2).pow(b)).sum()


# ----------------------------------------------------------------------------- GDN
_GAMMA_REPARAM = NonNegativeParametrizer()
_BETA_REPARAM = NonNegativeParametrizer(minimum=1e-6)


def f_gdn(x, gamma, beta, inverse, gamma_reparam=_GAMMA_REPARAM, beta_reparam=_BETA_REPARAM):
    """quant_layer.py:142-154."""
    C = x.size(1)
    g = gamma_reparam(gamma).reshape(C, C, 1, 1)
    b = beta_reparam(beta)
    norm = F.conv2d(x ** 2, g, b)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


# ----------------------------------------------------------------------------- one quantised op (== a reference QuantModule)
@dataclass
class QOp:
    """State of one reference `QuantModule` (quant_layer.py:11-138) reduced to tensors.

    kind: 'conv' | 'tconv' | 'gdn' | 'igdn' | 'linear'.  `mode`: 'fp' (org weight, quant_layer.py:116-118),
    'uaq' (nearest fake-quant, :113-115 with UniformAffineQuantizer), 'ada' (AdaRoundQuantizer)."""
    kind: str
    weight: torch.Tensor
    bias: Optional[torch.Tensor] = None
    stride: int = 1
    padding: int = 0
    output_padding: int = 0
    n_bits: int = 8
    channel_wise: bool = True
    scale_method: str = "max"
    act: Optional[str] = None            # fused activation: None | 'lrelu' | 'relu' (quant_model.py:51-54 fuses either)
    mode: str = "fp"
    delta: Optional[torch.Tensor] = None
    zp: Optional[torch.Tensor] = None
    alpha: Optional[torch.Tensor] = None
    soft: bool = False

    @property
    def n_levels(self):
        return 2 ** self.n_bits

    def init_scale(self):
        if self.delta is None:
            self.delta, self.zp = uaq_init(self.weight, self.n_bits, self.channel_wise, self.scale_method,
                                           tconv=(self.kind == "tconv"))
        return self

    def to_adaround(self):
        """layer_opt.py:248-250."""
        self.init_scale()
        self.alpha = adaround_init_alpha(self.weight.clone(), self.delta).requires_grad_(True)
        self.mode, self.soft = "ada", True
        return self

    def qweight(self):
        if self.mode == "fp":
            return self.weight
        self.init_scale()
        if self.mode == "uaq":
            return uaq_fakequant(self.weight, self.delta, self.zp, self.n_levels)
        return adaround_forward(self.weight, self.alpha, self.delta, self.zp, self.n_levels, self.soft)

    def __call__(self, x):
        w = self.qweight()
        if self.kind == "conv":
            y = F.conv2d(x, w, self.bias, stride=self.stride, padding=self.padding)
        elif self.kind == "tconv":
            y = F.conv_transpose2d(x, w, self.bias, stride=self.stride, padding=self.padding,
                                   output_padding=self.output_padding)
        elif self.kind in ("gdn", "igdn"):
            y = f_gdn(x, w, self.bias, inverse=(self.kind == "igdn"))
        elif self.kind == "linear":
            y = F.linear(x, w, self.bias)
        elif self.kind == "layernorm":                       # quant_layer.py:44-49,121: F.layer_norm over the last dimension
            y = F.layer_norm(x, (w.numel(),), weight=w, bias=self.bias)
        else:
            raise ValueError(self.kind)
        if self.act == "lrelu":
            y = F.leaky_relu(y, 0.01)
        elif self.act == "relu":
            y = F.relu(y)
        return y


# ----------------------------------------------------------------------------- Cheng2020 blocks (quant_block.py:219-313)
def _maybe_aq(x, aq):
    return act_quant(x) if aq else x


def rbws_forward(ops: Dict[str, QOp], x, aq=False, inner_aq=False):
    """QuantRBWS.forward, quant_block.py:235-248.  `aq` == (block.use_act_quant and block.trained).  `inner_aq`: the inner
    QuantModules built WITHOUT disable_act_quant (conv2, gdn, skip: quant_block.py:227-232) quantise their own outputs once they
    are trained and activation quantisation is on (quant_layer.py:130-133) -- the state of a calibrated block in the W8A8 evaluation."""
    out = F.leaky_relu(ops["conv1"](x), 0.01)
    out = _maybe_aq(out, aq)
    out = _maybe_aq(ops["conv2"](out), inner_aq)
    out = _maybe_aq(ops["gdn"](out), inner_aq)
    identity = _maybe_aq(ops["skip"](x), inner_aq) if "skip" in ops else x
    out = out + identity
    return _maybe_aq(out, aq)


def rbu_forward(ops: Dict[str, QOp], x, aq=False, r=2, inner_aq=False):
    """QuantRBU.forward, quant_block.py:270-282; inner quantisers: conv, igdn, upsample[0] (quant_block.py:257-268)."""
    out = F.leaky_relu(F.pixel_shuffle(ops["subpel_conv"](x), r), 0.01)
    out = _maybe_aq(out, aq)
    out = _maybe_aq(ops["conv"](out), inner_aq)
    out = _maybe_aq(ops["igdn"](out), inner_aq)
    out = out + F.pixel_shuffle(_maybe_aq(ops["upsample"](x), inner_aq), r)
    return _maybe_aq(out, aq)


def rb_forward(ops: Dict[str, QOp], x, aq=False, inner_aq=False):
    """QuantRB.forward, quant_block.py:298-313; inner quantiser: skip only (conv1 / conv2 are built with disable_act_quant)."""
    out = F.leaky_relu(ops["conv1"](x), 0.01)
    out = _maybe_aq(out, aq)
    out = F.leaky_relu(ops["conv2"](out), 0.01)
    out = _maybe_aq(out, aq)
    identity = _maybe_aq(ops["skip"](x), inner_aq) if "skip" in ops else x
    out = out + identity
    return _maybe_aq(out, aq)


def layer_forward(ops: Dict[str, QOp], x, aq=False, inner_aq=False):
    """A bare QuantModule unit (quant_layer.py:107-134)."""
    return _maybe_aq(ops["layer"](x), aq)


UNIT_FORWARD = {"rbws": rbws_forward, "rbu": rbu_forward, "rb": rb_forward, "layer": layer_forward}


# ----------------------------------------------------------------------------- counter-based QDrop mask (shared spec with the HIP engine)
def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def qdrop_keep_mask_nhwc(seed: int, it: int, shape_nchw: Sequence[int], prob: float) -> torch.Tensor:
    """Mask for layer_opt.py:291-292 `where(rand < p, x_q, x_fp)` drawn from the engine's counter RNG:
    element (b,c,h,w) -> i = ((b*H+h)*W+w)*C+c ; u = lowbias32(i ^ lowbias32(it + seed*0x9E3779B9)) ;
    keep (take the quantised-prefix input) iff u < floor(p * 2^32).  Returned as a bool NCHW tensor."""
    B, C, H, W = shape_nchw
    with np.errstate(over="ignore"):
        key = _lowbias32(np.array([(it + seed * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint32))[0]
        i = np.arange(B * H * W * C, dtype=np.uint32)
        u = _lowbias32(i ^ key)
    thr = min(int(math.floor(prob * 4294967296.0)), 4294967296)
    keep = (u.astype(np.uint64) < np.uint64(thr)).reshape(B, H, W, C)
    return torch.from_numpy(np.ascontiguousarray(keep.transpose(0, 3, 1, 2)))


# ----------------------------------------------------------------------------- the hot loop
@dataclass
class ReconLog:
    total: List[float] = field(default_factory=list)
    rec: List[float] = field(default_factory=list)
    task: List[float] = field(default_factory=list)
    round: List[float] = field(default_factory=list)
    b: List[float] = field(default_factory=list)


def reconstruct_unit(kind: str, ops: Dict[str, QOp], cached_q, cached_fp, cached_out, *,
                     iters: int, batch_size: int, idx_stream: Optional[Sequence[Sequence[int]]] = None,
                     mask_fn: Optional[Callable[[int, Sequence[int]], torch.Tensor]] = None,
                     input_prob: float = 0.5, weight: float = 0.01, b_range=(20, 2), warmup: float = 0.2,
                     p: float = 2.0, task_p: float = 2.0, tail: Optional[Callable] = None,
                     fp_net_out: Optional[torch.Tensor] = None, lr: float = 1e-3,
                     grad_hook: Optional[Callable[[List[torch.Tensor]], None]] = None,
                     task_fn: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None) -> ReconLog:
    """layer_opt.py:236-315 / block_opt.py:228-321 -- AdaRound optimisation of one unit.

    * every op of the unit gets an AdaRoundQuantizer with soft targets (block_opt.py:239-243),
    * Adam(default lr 1e-3) over the alphas (layer_opt.py:253-254 -- the CLI `--lr` is ignored there),
    * per iteration: idx -> (x_q, x_fp) -> QDrop mix -> unit forward -> `tail` (fp_out of the rest of the
      sub-coder; identity for Sequential-indexed CompressAI models) -> round + rec + task -> backward
      -> step (layer_opt.py:287-309),
    * afterwards soft_targets=False (hard rounding) (layer_opt.py:313-315).
    `idx_stream[i]` replaces `torch.randperm(n)[:batch]` (:289) and `mask_fn(i, shape)` replaces
    `torch.rand_like(x) < input_prob` (:292) so runs are reproducible across implementations.
    `grad_hook(list_of_alpha_grads)` is called between backward and step (data-parallel all-reduce point).
    `task_fn(out_quant, idx)` replaces the lp task term: the build's opt-in R + lambda*D mode evaluates the rate-distortion loss of
    the whole model with the unit's output substituted (the call the reference comments out, layer_opt.py:146-148)."""
    fwd = kind if callable(kind) else UNIT_FORWARD[kind]     # a callable (ops, x) -> y serves units outside the table (RSTB)
    for op in ops.values():
        op.to_adaround()
    alphas = [op.alpha for op in ops.values()]
    opt = torch.optim.Adam(alphas, lr=lr)
    loss_start = iters * warmup
    n = cached_q.size(0)
    if fp_net_out is None:
        fp_net_out = cached_out if tail is None else tail(cached_out)
    log = ReconLog()
    for i in range(iters):
        idx = torch.as_tensor(idx_stream[i], dtype=torch.long) if idx_stream is not None \
            else torch.randperm(n)[:batch_size]
        cur_inp, cur_sym = cached_q[idx], cached_fp[idx]
        if input_prob < 1.0:
            keep = mask_fn(i, cur_inp.shape) if mask_fn is not None else (torch.rand_like(cur_inp) < input_prob)
            cur_inp = torch.where(keep, cur_inp, cur_sym)
        cur_out = cached_out[idx]
        opt.zero_grad()
        out_quant = fwd(ops, cur_inp)
        net_out = out_quant if tail is None else tail(out_quant)
        rec = lp_loss(out_quant, cur_out, p=p)
        task = task_fn(out_quant, idx) if task_fn is not None else lp_loss(net_out, fp_net_out[idx], p=task_p)
        count = i + 1
        b = linear_temp_decay(count, iters, warmup, b_range[0], b_range[1])
        if count < loss_start:
            b = rl = 0
        else:
            rl = 0
            for op in ops.values():
                rl = rl + round_loss_term(op.alpha, b, weight)
        total = rl + rec + task
        total.backward()
        if grad_hook is not None:
            grad_hook([a.grad for a in alphas])
        opt.step()
        log.total.append(float(total.detach())); log.rec.append(float(rec.detach())); log.task.append(float(task.detach()))
        log.round.append(float(rl.detach()) if torch.is_tensor(rl) else float(rl)); log.b.append(float(b))
    for op in ops.values():
        op.soft = False
        op.alpha = op.alpha.detach()
    return log
