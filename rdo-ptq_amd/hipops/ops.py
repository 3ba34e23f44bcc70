"""Thin tensor-level wrappers over the C ABI (include/rdo_ptq_hip.h).

Conventions: every tensor is a contiguous fp32 CUDA tensor; activations are NHWC `[B,H,W,C]`, conv weights OHWI
`[Cout,KH,KW,Cin]` (the memory image of torch's channels_last OIHW weight).  Calls are enqueued on torch's current stream.
When a plan is being recorded (hipops.plan.Plan.record()), the same calls are captured instead of launched."""
import ctypes as C
import math
import threading

import torch

from . import _lib as L


class H2:
    """Planes of an H2 tensor (include/rdo_ptq_hip.h): `.t` int16 [2, ...] holding fp16 bit patterns -- the two-way split of the values
    times `.scale` (a power of two) -- for activations in slice-major order [2, C/16, pixels, 16], for conv weights in fragment order."""
    __slots__ = ("t", "scale", "flag")

    def __init__(self, t, scale, flag=None):
        self.t, self.scale = t, float(scale)
        self.flag = flag          # this tensor's own overflow words (2-element int32 device tensor; see h2_flag) or None: the bound default
        m, e = math.frexp(self.scale)
        if not (self.scale > 0 and m == 0.5):
            raise ValueError(f"H2 scale must be a positive power of two, got {scale}")

    @property
    def device(self):
        return self.t.device


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, H2):
        t = t.t
    if not t.is_cuda:
        raise RuntimeError(f"librdoptq_hip has no CPU path: tensor is on {t.device}")
    if t.dtype not in (torch.float32, torch.int32, torch.int16) or not t.is_contiguous():
        raise RuntimeError(f"librdoptq_hip needs contiguous fp32/int32 tensors, got {t.dtype} contiguous={t.is_contiguous()}")
    return C.c_void_p(t.data_ptr())


def _pscale(planes):
    """plane scale argument of the AdaRound entries: 0 = bf16 three-way planes (or none), > 0 = fp16 two-way planes of w * scale"""
    return planes.scale if isinstance(planes, H2) else 0.0


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def conv_desc(x_shape, w_shape, stride, pad, epilogue=L.EPI_NONE, square_input=False, add_residual=False):
    B, H, W, Cin = x_shape
    Cout, KH, KW, Cin2 = w_shape
    if Cin != Cin2:
        raise ValueError(f"conv: input has {Cin} channels, weight expects {Cin2}")
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    return L.ConvDesc(B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, epilogue, int(square_input), int(add_residual))


_SCRATCH = {}       # device -> list of scratch tensors (kept alive: recorded plans hold raw pointers into them)


def _scratch(device, n_floats):
    """Split-K workspace shared by all convolutions of a device (launches are stream-ordered, so one buffer suffices)."""
    pool = _SCRATCH.setdefault(device, [])
    if not pool or pool[-1].numel() < n_floats:
        pool.append(torch.empty(max(n_floats, 1 << 22), device=device, dtype=torch.float32))
    return pool[-1]


def uses_bf16x6(x_shape, w_shape, stride, pad):
    d = conv_desc(x_shape, w_shape, stride, pad)
    return bool(L.lib().rdo_conv2d_fwd_uses_bf16x6(C.byref(d)))


def split_bf16x3(w, planes=None):
    """Conv weight in kernel layout [Cout,KH,KW,Cin] (or [C,C] = 1x1) -> int16 tensor of 3 * w.numel() bf16 values: the exact
    split w = p0 + p1 + p2, each plane in the fragment order the bf16x6 kernels read ([Cin/16][KH][KW][Cout][16])."""
    if w.dim() == 2:
        w = w.reshape(w.shape[0], 1, 1, w.shape[1])
    if w.dim() != 4:
        raise ValueError("split_bf16x3: expected a conv weight [Cout,KH,KW,Cin]")
    planes = torch.empty((3,) + tuple(w.shape), device=w.device, dtype=torch.int16) if planes is None else planes
    co, kh, kw, ci = w.shape
    L.check(L.lib().rdo_split_bf16x3_conv(_ptr(w), co, kh, kw, ci, _ptr(planes), _stream()), "rdo_split_bf16x3_conv")
    return planes


def split_bf16x3_linear(w, planes=None):
    """Element order kept: planes[p].view_as(w) is plane p (generic helper, not a conv operand)."""
    planes = torch.empty((3,) + tuple(w.shape), device=w.device, dtype=torch.int16) if planes is None else planes
    L.check(L.lib().rdo_split_bf16x3(_ptr(w), w.numel(), _ptr(planes), _stream()), "rdo_split_bf16x3")
    return planes


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, epilogue=L.EPI_NONE, aux=None, residual=None, square_input=False,
               out=None, pre=None, wplanes=None):
    d = conv_desc(x.shape, w.shape, stride, pad, epilogue, square_input, residual is not None)
    if out is None:
        out = torch.empty((d.B, d.Ho, d.Wo, d.Cout), device=x.device, dtype=torch.float32)
    need = int(L.lib().rdo_conv2d_fwd_workspace(C.byref(d)))
    ws = _scratch(x.device, need) if need else None
    L.check(L.lib().rdo_conv2d_fwd(C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(aux), _ptr(residual), _ptr(out), _ptr(pre),
                                   _ptr(ws), ws.numel() if ws is not None else 0, _ptr(wplanes), _stream()), "rdo_conv2d_fwd")
    return out


SCRATCH_FLOATS = 1 << 22     # smallest split-K workspace _scratch hands out


def conv_fwd_ksplit(x_shape, w_shape, stride, pad, has_planes, device):
    """Split factor conv2d_fwd uses for this shape with the workspace it would get (1: no split-K pass)."""
    d = conv_desc(x_shape, w_shape, stride, pad)
    need = int(L.lib().rdo_conv2d_fwd_workspace(C.byref(d)))
    return int(L.lib().rdo_conv2d_fwd_ksplit(C.byref(d), int(bool(has_planes)), max(need, SCRATCH_FLOATS))), need


def conv2d_fwd_partials(x, w, stride=1, pad=0, wplanes=None):
    """K-split accumulation only: returns (workspace tensor holding [ksplit][M][Cout] partial sums, ksplit).  The workspace is the
    shared split-K scratch of the device: consume it with the next launch (loss_act_bwd_splitk)."""
    d = conv_desc(x.shape, w.shape, stride, pad)
    ks, need = conv_fwd_ksplit(tuple(x.shape), tuple(w.shape), stride, pad, wplanes is not None, x.device)
    ws = _scratch(x.device, need)
    L.check(L.lib().rdo_conv2d_fwd_partials(C.byref(d), _ptr(x), _ptr(w), _ptr(wplanes), _ptr(ws), ws.numel(), _stream()),
            "rdo_conv2d_fwd_partials")
    return ws, ks


def loss_act_bwd_splitk(partial, ksplit, bias, out_shape, residual, tgt_cache, idx_table, iter_ptr, coef, act, loss_log, out=None, grad_out=None,
                        dpre=None):
    B = out_shape[0]
    per_image = 1
    for v in out_shape[1:]:
        per_image *= int(v)
    L.check(L.lib().rdo_loss_act_bwd_splitk(_ptr(partial), int(ksplit), _ptr(bias), _ptr(residual), _ptr(tgt_cache), _ptr(idx_table),
                                            _ptr(iter_ptr), B, per_image, int(out_shape[-1]), coef, int(act), _ptr(out), _ptr(grad_out),
                                            _ptr(dpre), _ptr(loss_log), _stream()), "rdo_loss_act_bwd_splitk")


def wgrad_nsplit(x_shape, w_shape, stride, pad):
    d = conv_desc(x_shape, w_shape, stride, pad)
    return int(L.lib().rdo_conv2d_wgrad_nsplit(C.byref(d)))


def wgrad_uses_bf16x6(x_shape, w_shape, stride, pad):
    d = conv_desc(x_shape, w_shape, stride, pad)
    return bool(L.lib().rdo_conv2d_wgrad_uses_bf16x6(C.byref(d)))


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0, square_input=False, slabs=None):
    d = conv_desc(x.shape, w_shape, stride, pad, square_input=square_input)
    ns = int(L.lib().rdo_conv2d_wgrad_nsplit(C.byref(d))) if slabs is None else slabs.shape[0]
    if slabs is None:
        slabs = torch.empty((ns,) + tuple(w_shape), device=x.device, dtype=torch.float32)
    L.check(L.lib().rdo_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(slabs), ns, _stream()), "rdo_conv2d_wgrad")
    return slabs


def reduce_slabs(slabs, out=None):
    ns = slabs.shape[0]
    numel = slabs[0].numel()
    if out is None:
        out = torch.empty(slabs.shape[1:], device=slabs.device, dtype=torch.float32)
    L.check(L.lib().rdo_reduce_slabs(_ptr(slabs), ns, numel, _ptr(out), _stream()), "rdo_reduce_slabs")
    return out


def ada_desc(w, n_levels=256, reparam=None, conv_layout=True):
    """w: OHWI conv weight [Cout,KH,KW,Cin] or a GDN gamma [C,C]."""
    if w.dim() == 4:
        rows, KH, KW, Cin = w.shape
    else:
        rows, Cin = w.shape
        KH = KW = 1
    if not conv_layout:
        KH = KW = Cin = 0
    bound, ped = (reparam if reparam is not None else (0.0, 0.0))
    return L.AdaDesc(w.numel(), rows, n_levels, int(reparam is not None), bound, ped, KH, KW, Cin)


def adaround_init_alpha(d, w, delta, alpha=None):
    alpha = torch.empty_like(w) if alpha is None else alpha
    L.check(L.lib().rdo_adaround_init_alpha(C.byref(d), _ptr(w), _ptr(delta), _ptr(alpha), _stream()), "rdo_adaround_init_alpha")
    return alpha


def adaround_fwd(d, w, alpha, delta, zp, soft, wq=None, wd=None):
    wq = torch.empty_like(w) if wq is None else wq
    L.check(L.lib().rdo_adaround_fwd(C.byref(d), _ptr(w), _ptr(alpha), _ptr(delta), _ptr(zp), int(soft), _ptr(wq), _ptr(wd),
                                     _stream()), "rdo_adaround_fwd")
    return wq


def uaq_fakequant(d, w, delta, zp, wq=None, wd=None):
    wq = torch.empty_like(w) if wq is None else wq
    L.check(L.lib().rdo_uaq_fakequant(C.byref(d), _ptr(w), _ptr(delta), _ptr(zp), _ptr(wq), _ptr(wd), _stream()),
            "rdo_uaq_fakequant")
    return wq


def uaq_init_minmax(w, n_levels):
    rows = w.shape[0]
    delta = torch.empty(rows, device=w.device, dtype=torch.float32)
    zp = torch.empty_like(delta)
    L.check(L.lib().rdo_uaq_init_minmax(_ptr(w), rows, w.numel() // rows, n_levels, _ptr(delta), _ptr(zp), _stream()),
            "rdo_uaq_init_minmax")
    return delta, zp


def adaround_step(d, w, delta, zp, slabs, grad_scale, round_weight, sched, iter_ptr, alpha, m, v, wq, wd, round_log,
                  wq_planes=None, wd_planes=None):
    """`wq_planes` / `wd_planes`: optional planes of the new weights in fragment order -- an int16 [3, numel] tensor receives the bf16
    three-way split, an `H2` object the fp16 two-way split of w * scale."""
    _bind_out(None)            # weight planes raise the enclosing block's word
    L.check(L.lib().rdo_adaround_step(C.byref(d), _ptr(w), _ptr(delta), _ptr(zp), _ptr(slabs), slabs.shape[0], grad_scale,
                                      round_weight, _ptr(sched), _ptr(iter_ptr), _ptr(alpha), _ptr(m), _ptr(v), _ptr(wq),
                                      _ptr(wd), _ptr(round_log), _ptr(wq_planes), _ptr(wd_planes), _pscale(wq_planes), _pscale(wd_planes),
                                      _stream()), "rdo_adaround_step")


def unit1x1_supported(M, K, N):
    return bool(L.lib().rdo_unit1x1_supported(int(M), int(K), int(N)))


def unit1x1_nslab(M, N):
    return int(L.lib().rdo_unit1x1_nslab(int(M), int(N)))


def unit1x1_form(form):
    """1: split-fp16 form of rdo_unit1x1 where Cout is a multiple of 96 (default); 0: always the exact fp32 form.  -> previous setting."""
    return int(L.lib().rdo_unit1x1_form(int(form)))


def unit1x1(x, w, bias, tgt_cache, idx_table, iter_ptr, coef, act, loss_log, slabs):
    """One launch for a 1 x 1 layer unit's data path (include/rdo_ptq_hip.h: rdo_unit1x1): x [B, H, W, K] mini-batch, w [N, 1, 1, K] soft
    weights -> loss into `loss_log`, weight-gradient slabs [nslab, N, 1, 1, K]."""
    K, N = int(x.shape[-1]), int(w.shape[0])
    M = x.numel() // K
    L.check(L.lib().rdo_unit1x1(_ptr(x), M, K, N, _ptr(w), _ptr(bias), _ptr(tgt_cache), _ptr(idx_table), _ptr(iter_ptr), int(x.shape[0]),
                                float(coef), int(act), _ptr(slabs), int(slabs.shape[0]), _ptr(loss_log), _stream()), "rdo_unit1x1")


def iter_bind_publish(word):
    """The next loss / tail launch of this thread leaves the iteration number it read in `word` (a one-element int32 tensor; None clears).
    Returns True when an earlier binding was still pending (include/rdo_ptq_hip.h: rdo_iter_bind_publish)."""
    return bool(L.lib().rdo_iter_bind_publish(_ptr(word)))


def adaround_step_batch(items, grad_scale, round_weight, sched, iter_ptr, round_log, advance_iter=None, mode=0, iter_shadow=None, gather=None):
    """items: list of dicts(d, w, delta, zp, slabs, alpha, m, v, wq, wd, wq_planes, wd_planes[, dalpha, lin_fwd, lin_bwd]) -- one launch for every
    weight tensor of a unit (<= 8, numel % 4 == 0): mode 0 the fused AdaRound step, 1 the data gradient into `dalpha`, 2 the update
    from an (all-reduced) `dalpha`; `advance_iter`: the device iteration counter to increment afterwards.
    `gather`: dict(cache_q, cache_fp, idx_table, B, batch_offset, prob, seed, out, out_planes) -- the launch also assembles the NEXT
    iteration's mini-batch (rdo_adaround_step_batch_gather)."""
    _bind_out(None)            # weight planes raise the enclosing block's word
    arr = (L.AdaStepItem * len(items))()
    dp = lambda t: None if t is None else _ptr(t).value
    for k, it in enumerate(items):
        a = arr[k]
        a.d = it["d"]
        a.w, a.delta, a.zp, a.slabs = dp(it["w"]), dp(it["delta"]), dp(it["zp"]), dp(it["slabs"])
        a.nsplit = it["slabs"].shape[0] if it.get("slabs") is not None else 0
        a.alpha, a.adam_m, a.adam_v, a.wq, a.wd = dp(it["alpha"]), dp(it["m"]), dp(it["v"]), dp(it["wq"]), dp(it.get("wd"))
        a.wq_planes, a.wd_planes = dp(it.get("wq_planes")), dp(it.get("wd_planes"))
        a.wq_plane_scale, a.wd_plane_scale = _pscale(it.get("wq_planes")), _pscale(it.get("wd_planes"))
        a.dalpha = dp(it.get("dalpha"))
        lf, lb = it.get("lin_fwd"), it.get("lin_bwd")      # H2 planes in rdo_linear_h2's fragment order, written by the step itself
        a.lin_fwd_planes, a.lin_bwd_planes = dp(lf), dp(lb)
        a.lin_plane_scale = float((lf if lf is not None else lb).scale) if (lf is not None or lb is not None) else 0.0
    if gather is not None:
        if advance_iter is not None:
            raise ValueError("adaround_step_batch: the gather form moves the counter by hand-over (iter_shadow), not advance_iter")
        cq, pl = gather["cache_q"], gather.get("out_planes")
        g = L.GatherDesc()
        g.cache_q, g.cache_fp, g.idx_table = dp(cq), dp(gather["cache_fp"]), dp(gather["idx_table"])
        g.n_iters, g.B, g.batch_offset = int(gather["idx_table"].shape[0]), int(gather["B"]), int(gather.get("batch_offset", 0))
        g.per_image, g.C, g.prob, g.seed = cq[0].numel(), int(cq.shape[-1]), float(gather["prob"]), int(gather["seed"])
        g.out, g.out_planes, g.out_scale = dp(gather.get("out")), dp(pl), (float(pl.scale) if pl is not None else 0.0)
        g.overflow_flag = dp(getattr(pl, "flag", None))      # the gathered planes raise their own overflow word
        L.check(L.lib().rdo_adaround_step_batch_gather(arr, len(items), int(mode), grad_scale, round_weight, _ptr(sched), _ptr(iter_ptr),
                                                       _ptr(round_log), _ptr(iter_shadow), C.byref(g), _stream()), "rdo_adaround_step_batch_gather")
        return
    L.check(L.lib().rdo_adaround_step_batch(arr, len(items), int(mode), grad_scale, round_weight, _ptr(sched), _ptr(iter_ptr), _ptr(round_log),
                                            _ptr(advance_iter), _ptr(iter_shadow), _stream()), "rdo_adaround_step_batch")


def adaround_grad(d, w, alpha, delta, zp, slabs, dalpha):
    L.check(L.lib().rdo_adaround_grad(C.byref(d), _ptr(w), _ptr(alpha), _ptr(delta), _ptr(zp), _ptr(slabs), slabs.shape[0],
                                      _ptr(dalpha), _stream()), "rdo_adaround_grad")


def adaround_apply(d, w, delta, zp, dalpha, grad_scale, round_weight, sched, iter_ptr, alpha, m, v, wq, wd, round_log,
                   wq_planes=None, wd_planes=None):
    _bind_out(None)            # weight planes raise the enclosing block's word
    L.check(L.lib().rdo_adaround_apply(C.byref(d), _ptr(w), _ptr(delta), _ptr(zp), _ptr(dalpha), grad_scale, round_weight,
                                       _ptr(sched), _ptr(iter_ptr), _ptr(alpha), _ptr(m), _ptr(v), _ptr(wq), _ptr(wd),
                                       _ptr(round_log), _ptr(wq_planes), _ptr(wd_planes), _pscale(wq_planes), _pscale(wd_planes), _stream()),
            "rdo_adaround_apply")


def set_tuning(key, value):
    """Process-wide kernel-variant switch (include/rdo_ptq_hip.h: rdo_set_tuning); returns the previous value."""
    prev = int(L.lib().rdo_get_tuning(key.encode()))
    L.check(L.lib().rdo_set_tuning(key.encode(), int(value)), f"rdo_set_tuning({key})")
    return prev


def actquant_perchannel(x, out=None, ws=None, n_bits=8):
    """x: [..., C] channels-last; per-channel dynamic quant-dequant (8 bit = the reference's hard-wired width)."""
    Cc = x.shape[-1]
    npix = x.numel() // Cc
    out = torch.empty_like(x) if out is None else out
    need = int(L.lib().rdo_actquant_workspace(Cc))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=x.device, dtype=torch.float32)
    L.check(L.lib().rdo_actquant_perchannel(_ptr(x), npix, Cc, int(n_bits), _ptr(out), _ptr(ws), _stream()), "rdo_actquant_perchannel")
    return out


def gather_qdrop(cache_q, cache_fp, idx_table, iter_ptr, B, prob, seed, out, batch_offset=0, iter_publish=None):
    """`batch_offset`: row of the global mini-batch this (data-parallel) rank's first row is -- the QDrop counter runs over the
    global batch, so N ranks with one seed draw the mask a single process would."""
    per_image = cache_q[0].numel()
    L.check(L.lib().rdo_gather_qdrop(_ptr(cache_q), _ptr(cache_fp), _ptr(idx_table), _ptr(iter_ptr), B, int(batch_offset), per_image,
                                     prob, seed,
                                     _ptr(out), _ptr(iter_publish), _stream()), "rdo_gather_qdrop")
    return out


def lp2_loss_grad(pred, tgt_cache, idx_table, iter_ptr, coef, grad, loss_log):
    B = pred.shape[0]
    per_image = pred[0].numel()
    L.check(L.lib().rdo_lp2_loss_grad(_ptr(pred), _ptr(tgt_cache), _ptr(idx_table), _ptr(iter_ptr), B, per_image,
                                      pred.shape[-1], coef, _ptr(grad), _ptr(loss_log), _stream()), "rdo_lp2_loss_grad")
    return grad


def lp_loss_grad(pred, tgt_cache, idx_table, iter_ptr, coef2, coefp, p, grad, loss_log, loss_log_p=None):
    """coef2 * lp_loss(., p=2) + coefp * lp_loss(., p=p) of the same (pred, tgt) pair, and its gradient.  With `loss_log_p`
    the |d|^p term is logged there and only the p = 2 term in `loss_log`."""
    B = pred.shape[0]
    per_image = pred[0].numel()
    L.check(L.lib().rdo_lp_loss_grad(_ptr(pred), _ptr(tgt_cache), _ptr(idx_table), _ptr(iter_ptr), B, per_image,
                                     pred.shape[-1], coef2, coefp, p, _ptr(grad), _ptr(loss_log), _ptr(loss_log_p), _stream()),
            "rdo_lp_loss_grad")
    return grad


def lrelu(x, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().rdo_lrelu_fwd(_ptr(x), x.numel(), _ptr(out), _stream()), "rdo_lrelu_fwd")
    return out


def lrelu_bwd(g, y, out=None):
    out = torch.empty_like(g) if out is None else out
    L.check(L.lib().rdo_lrelu_bwd(_ptr(g), _ptr(y), g.numel(), _ptr(out), _stream()), "rdo_lrelu_bwd")
    return out


def relu(x, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().rdo_relu_fwd(_ptr(x), x.numel(), _ptr(out), _stream()), "rdo_relu_fwd")
    return out


def relu_bwd(g, y, out=None):
    out = torch.empty_like(g) if out is None else out
    L.check(L.lib().rdo_relu_bwd(_ptr(g), _ptr(y), g.numel(), _ptr(out), _stream()), "rdo_relu_bwd")
    return out


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    L.check(L.lib().rdo_add(_ptr(a), _ptr(b), a.numel(), _ptr(out), _stream()), "rdo_add")
    return out


def pixel_shuffle(x, r, out=None):
    """[B,H,W,C*r*r] -> [B,H*r,W*r,C]"""
    B, H, W, CC = x.shape
    Cc = CC // (r * r)
    out = torch.empty((B, H * r, W * r, Cc), device=x.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_pixel_shuffle(_ptr(x), B, H, W, Cc, r, 0, _ptr(out), _stream()), "rdo_pixel_shuffle")
    return out


def pixel_unshuffle(x, r, out=None):
    """[B,H*r,W*r,C] -> [B,H,W,C*r*r]  (gradient of pixel_shuffle)"""
    B, Hr, Wr, Cc = x.shape
    H, W = Hr // r, Wr // r
    out = torch.empty((B, H, W, Cc * r * r), device=x.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_pixel_shuffle(_ptr(x), B, H, W, Cc, r, 1, _ptr(out), _stream()), "rdo_pixel_shuffle(inverse)")
    return out


def gdn_bwd_t(g, x, norm, inverse, out=None):
    out = torch.empty_like(g) if out is None else out
    L.check(L.lib().rdo_gdn_bwd_t(_ptr(g), _ptr(x), _ptr(norm), g.numel(), int(inverse), _ptr(out), _stream()), "rdo_gdn_bwd_t")
    return out


def gdn_bwd_dx(g, x, norm, acc, inverse, out=None):
    out = torch.empty_like(g) if out is None else out
    L.check(L.lib().rdo_gdn_bwd_dx(_ptr(g), _ptr(x), _ptr(norm), _ptr(acc), g.numel(), int(inverse), _ptr(out), _stream()),
            "rdo_gdn_bwd_dx")
    return out


def nchw_to_nhwc(x, out=None):
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_nchw_to_nhwc(_ptr(x), B, Cc, H, W, 0, _ptr(out), _stream()), "rdo_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, out=None):
    B, H, W, Cc = x.shape
    out = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_nchw_to_nhwc(_ptr(x), B, Cc, H, W, 1, _ptr(out), _stream()), "rdo_nchw_to_nhwc(inverse)")
    return out


def zero_insert(x, stride, pad_top, pad_left, Ho, Wo, out=None):
    """[B,H,W,C] -> [B,Ho,Wo,C] with x placed at (pad + h*stride, pad + w*stride) and zeros elsewhere."""
    B, H, W, Cc = x.shape
    out = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_zero_insert(_ptr(x), B, H, W, Cc, stride, pad_top, pad_left, Ho, Wo, _ptr(out), _stream()), "rdo_zero_insert")
    return out


class TconvPhase:
    """Sub-pixel form of ConvTranspose2d(K, stride, pad) with output = stride x input (include/rdo_ptq_hip.h, rdo_tconv_expand): the
    window size / padding of the equivalent stride-1 conv and the device index tables between the kernel-layout weight
    [Cout][K][K][Cin] (`flipped`: taps stored as [K-1-kh][K-1-kw], the engine's layout; else as they lie in the module's weight) and
    the phase weight [Cout * s^2][Kp][Kp][Cin]."""
    _cache = {}

    def __init__(self, K, stride, pad, flipped, device):
        s = stride
        win = []                                   # per phase a: [(d, kh)] input offset d and tap kh
        for a in range(s):
            r, q = (a + pad) % s, (a + pad) // s
            win.append([(q - t, r + s * t) for t in range((K - r + s - 1) // s) if r + s * t < K])
        dmax = max(max(d for d, _ in w) for w in win if w)
        dmin = min(min(d for d, _ in w) for w in win if w)
        self.pad = max(dmax, -dmin, 0)
        self.Kp = 2 * self.pad + 1
        self.S2, self.K, self.stride = s * s, K, s
        fl = (lambda k: K - 1 - k) if flipped else (lambda k: k)
        mp = [[-1] * (self.Kp * self.Kp) for _ in range(self.S2)]
        inv = [-1] * (K * K)
        for a in range(s):
            for b in range(s):
                for dh, kh in win[a]:
                    for dw, kw in win[b]:
                        u, v = dh + self.pad, dw + self.pad
                        tap = fl(kh) * K + fl(kw)
                        mp[a * s + b][u * self.Kp + v] = tap
                        inv[tap] = (a * s + b) * self.Kp * self.Kp + u * self.Kp + v
        assert all(i >= 0 for i in inv)
        self.map = torch.tensor(mp, dtype=torch.int32, device=device).contiguous()
        self.inv = torch.tensor(inv, dtype=torch.int32, device=device).contiguous()

    @classmethod
    def get(cls, K, stride, pad, output_padding, flipped, device):
        """The phase form, or None when the geometry has none (output must be exactly stride x input)."""
        if output_padding != stride + 2 * pad - K or stride < 2:
            return None
        key = (K, stride, pad, bool(flipped), str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(K, stride, pad, flipped, device)
        return cls._cache[key]

    def w_shape(self, Cout, Cin):
        return (Cout * self.S2, self.Kp, self.Kp, Cin)


def tconv_expand(w_rows, ph, out=None):
    """kernel-layout weight [Cout][K][K][Cin] -> phase weight [Cout * s^2][Kp][Kp][Cin] (structural zeros included)."""
    Cout, K, _, Cin = w_rows.shape
    out = torch.empty(ph.w_shape(Cout, Cin), device=w_rows.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_tconv_expand(_ptr(w_rows), _ptr(ph.map), Cout, K * K, Cin, ph.S2, ph.Kp * ph.Kp, _ptr(out), _stream()), "rdo_tconv_expand")
    return out


def tconv_fold(slabs_phase, ph, Cout, Cin, out=None):
    """gradient slabs [ns][Cout * s^2][Kp][Kp][Cin] of the phase weight -> slabs [ns][Cout][K][K][Cin] of the kernel weight"""
    ns = slabs_phase.shape[0]
    out = torch.empty((ns, Cout, ph.K, ph.K, Cin), device=slabs_phase.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_tconv_fold(_ptr(slabs_phase), _ptr(ph.inv), ns, Cout, ph.K * ph.K, Cin, ph.S2, ph.Kp * ph.Kp, _ptr(out), _stream()),
            "rdo_tconv_fold")
    return out


class WeightPack:
    """Constant tensors derived from ONE weight in kernel layout (`w` = [Cout,KH,KW,Cin] rows), each built on first use and kept:
    the bf16x3 planes the split-precision MFMA forward reads, the flipped / transposed forms the input-gradient convolutions read
    (with their planes), the phase weight of a transposed conv.  The callers that hold a pack (QuantModule, hipops.autograd) own
    its validity: a pack is made for one value of the weight and dropped when that value changes."""

    def __init__(self, w_rows, bias=None):
        self.w = w_rows.detach().contiguous()
        self.bias = None if bias is None else bias.detach().contiguous()
        self._d = {}

    def get(self, key, make):
        v = self._d.get(key)
        if v is None:
            v = self._d[key] = make()
        return v

    def planes(self, x_shape, stride, pad):
        """bf16x3 planes of `w` when rdo_conv2d_fwd takes the split-precision path for this problem, else None."""
        if not uses_bf16x6(tuple(x_shape), tuple(self.w.shape), stride, pad):
            return None
        return self.get("planes", lambda: split_bf16x3(self.w))

    def flipped(self):
        """Pack of the stride-1 input-gradient weight [Cin][KH'][KW'][Cout] (taps flipped)."""
        return self.get("flipped", lambda: WeightPack(self.w.flip(1, 2).permute(3, 1, 2, 0)))

    def transposed(self):
        """Pack of `w` read with input and output channels exchanged, [Cin][KH][KW][Cout] (taps as they lie)."""
        return self.get("transposed", lambda: WeightPack(self.w.permute(3, 1, 2, 0)))

    def lin_planes(self, transposed=False):
        """Fragment-ordered fp16 planes of a Linear's weight [N, 1, 1, K] (or of its transpose: the input-gradient Linear) for
        rdo_linear_h2; the power-of-two scale comes from the weight's own magnitude (a synchronising reduction, once per pack)."""
        def make():
            w2 = self.w.reshape(self.w.shape[0], -1)
            return split_h2_linear(w2.t().contiguous() if transposed else w2)
        return self.get(("lin", bool(transposed)), make)

    def phase(self, ph):
        """(pack of the phase weight, phase bias) of the transposed conv with this weight as `to_rows(W, tconv=True)`."""
        def make():
            wp = tconv_expand(self.w, ph)
            return WeightPack(wp, None if self.bias is None else self.bias.repeat_interleave(ph.S2))
        return self.get(("phase", id(ph)), make)


def conv2d_fwd_pack(x, pack, stride=1, pad=0, bias=True, **kw):
    """conv2d_fwd with the weight, bias and (where the split-precision path applies) the weight planes of a WeightPack."""
    return conv2d_fwd(x, pack.w, pack.bias if bias else None, stride, pad, wplanes=pack.planes(x.shape, stride, pad), **kw)


def conv_transpose2d(x, w_iohw_rows, bias, stride, pad, output_padding, epilogue=L.EPI_NONE, pack=None):
    """x [B,H,W,Cin]; w_iohw_rows = to_rows(W, tconv=True) = [Cout,KH,KW,Cin] (un-flipped taps of the [Cin,Cout,KH,KW] weight), or
    `pack` = a WeightPack of it (weight-derived tensors are then built once).  Output = stride x input (the deconv of the LIC
    decoders): stride-1 conv with the phase weight + pixel shuffle, no zero insertion; other geometries: zero insertion + dense conv."""
    if pack is None:
        pack = WeightPack(w_iohw_rows, bias)
    Cout, KH, KW, Cin = pack.w.shape
    B, H, W, _ = x.shape
    ph = TconvPhase.get(KH, stride, pad, output_padding, False, x.device) if KH == KW else None
    if ph is not None:
        # LeakyReLU / ReLU commute with the pixel shuffle; large problems on the split-precision path
        yp = conv2d_fwd_pack(x, pack.phase(ph), 1, ph.pad, epilogue=epilogue)
        return pixel_shuffle(yp, stride)
    q = KH - 1 - pad
    if q < 0:
        raise ValueError("conv_transpose2d: padding larger than kernel_size - 1 is not supported")
    Hup, Wup = (H - 1) * stride + 1 + 2 * q + output_padding, (W - 1) * stride + 1 + 2 * q + output_padding
    xu = zero_insert(x, stride, q, q, Hup, Wup)
    return conv2d_fwd_pack(xu, pack.get("zi", lambda: WeightPack(pack.w.flip(1, 2), pack.bias)), 1, 0, epilogue=epilogue)


def layer_norm(x, weight, bias, eps=1e-5, out=None):
    Cc = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().rdo_layer_norm(_ptr(x), _ptr(weight), _ptr(bias), x.numel() // Cc, Cc, eps, _ptr(out), _stream()),
            "rdo_layer_norm")
    return out


def linear_h2_supported(rows, cin, cout):
    return bool(L.lib().rdo_linear_h2_supported(int(rows), int(cin), int(cout)))


def split_h2_linear(w, scale=None, planes=None):
    """W [N][K] (any shape whose first dim is N, rest flattened to K) -> H2 planes in the fragment order of rdo_linear_h2; the scale
    defaults to a power of two that puts max |W| at 2^7 (a device synchronisation: set-up only -- pass `scale` inside recorded plans)."""
    N = w.shape[0]
    K = w.numel() // N
    if scale is None:
        scale = pow2_scale(w.abs().max())
    if planes is None:
        planes = H2(torch.empty((2, K // 32, N // 16, 64, 8), device=w.device, dtype=torch.int16), scale)
    L.check(L.lib().rdo_split_h2_linear(_ptr(w), N, K, float(planes.scale), _ptr(planes), _stream()), "rdo_split_h2_linear")
    return planes


def linear_h2(x, planes, bias=None, out=None, square_input=False, epilogue=L.EPI_NONE, pre=None, aux=None):
    """out [rows, N] = x [rows, K] W^T + bias (x^2 instead of x with `square_input`: the GDN norm pool) on the per-token-scaled
    fp16-split kernel; `planes` from split_h2_linear.  epilogue: EPI_GELU (out = gelu(y), `pre` receives y) or EPI_GELU_BWD
    (out = y * gelu'(aux))."""
    K = x.shape[-1]
    rows = x.numel() // K
    N = planes.t.shape[2] * 16
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (N,), device=x.device, dtype=torch.float32)
    if epilogue == L.EPI_NONE:
        L.check(L.lib().rdo_linear_h2(_ptr(x), rows, K, N, _ptr(planes), float(planes.scale), _ptr(bias), int(bool(square_input)), _ptr(out), _stream()),
                "rdo_linear_h2")
    else:
        L.check(L.lib().rdo_linear_h2_epi(_ptr(x), rows, K, N, _ptr(planes), float(planes.scale), _ptr(bias), int(bool(square_input)), int(epilogue),
                                          _ptr(pre), _ptr(aux), _ptr(out), _stream()), "rdo_linear_h2_epi")
    return out


def add_layer_norm(a, b, weight, bias, eps=1e-5, sum_out=None, out=None):
    """s = a + b (b may be None) -> `sum_out` (written when given), LayerNorm(s) -> out: the residual add of a Swin block and the
    LayerNorm that reads it, in one pass."""
    Cc = a.shape[-1]
    out = torch.empty_like(a) if out is None else out
    L.check(L.lib().rdo_add_layer_norm(_ptr(a), _ptr(b), _ptr(weight), _ptr(bias), a.numel() // Cc, Cc, eps, _ptr(sum_out), _ptr(out),
                                       _stream()), "rdo_add_layer_norm")
    return out


def layer_norm_bwd_add(x, weight, dy, add1=None, add2=None, eps=1e-5, dx=None, dgamma_slabs=None):
    """dx = (add1) (+ add2) + LayerNorm-backward(dy); dgamma partial sums as in `layer_norm_bwd`."""
    Cc = x.shape[-1]
    ns = 0 if dgamma_slabs is None else dgamma_slabs.shape[0]
    L.check(L.lib().rdo_layer_norm_bwd_add(_ptr(x), _ptr(weight), _ptr(dy), _ptr(add1), _ptr(add2), x.numel() // Cc, Cc, eps, _ptr(dx),
                                           _ptr(dgamma_slabs), ns, _stream()), "rdo_layer_norm_bwd_add")
    return dx


def add3(a, b, c, out=None):
    """(a + b) + c"""
    out = torch.empty_like(a) if out is None else out
    L.check(L.lib().rdo_add3(_ptr(a), _ptr(b), _ptr(c), a.numel(), _ptr(out), _stream()), "rdo_add3")
    return out


def layer_norm_bwd(x, weight, dy, eps=1e-5, dx=None, dgamma_slabs=None):
    """dx (returned) and, when `dgamma_slabs` [nslabs, C] is given, the partial sums of dy * xhat."""
    Cc = x.shape[-1]
    ns = 0 if dgamma_slabs is None else dgamma_slabs.shape[0]
    L.check(L.lib().rdo_layer_norm_bwd(_ptr(x), _ptr(weight), _ptr(dy), x.numel() // Cc, Cc, eps, _ptr(dx), _ptr(dgamma_slabs), ns,
                                       _stream()), "rdo_layer_norm_bwd")
    return dx


def attn_desc(B, H, W, Cc, heads, window, shift, scale=None):
    return L.AttnDesc(B, H, W, Cc, heads, window, shift, float((Cc // heads) ** -0.5 if scale is None else scale))


def window_attention(d, qkv, bias, out=None, probs=None, compute_out=True):
    """Fused (S)W-MSA core on [B, H, W, 3C] -> [B, H, W, C]; `probs` [windows, N, N, heads] is filled when given."""
    if compute_out and out is None:
        out = torch.empty((d.B, d.H, d.W, d.C), device=qkv.device, dtype=torch.float32)
    L.check(L.lib().rdo_window_attention_fwd(C.byref(d), _ptr(qkv), _ptr(bias), _ptr(out) if compute_out else None, _ptr(probs),
                                             _stream()), "rdo_window_attention_fwd")
    return out


def window_attention_pv(d, qkv, probs, out=None):
    out = torch.empty((d.B, d.H, d.W, d.C), device=qkv.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_window_attention_pv(C.byref(d), _ptr(qkv), _ptr(probs), _ptr(out), _stream()), "rdo_window_attention_pv")
    return out


def window_attention_bwd(d, qkv, bias, dout, dqkv=None):
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    L.check(L.lib().rdo_window_attention_bwd(C.byref(d), _ptr(qkv), _ptr(bias), _ptr(dout), _ptr(dqkv), _stream()),
            "rdo_window_attention_bwd")
    return dqkv


def gelu(x, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().rdo_gelu_fwd(_ptr(x), x.numel(), _ptr(out), _stream()), "rdo_gelu_fwd")
    return out


def gelu_bwd(dy, x, dx=None):
    dx = torch.empty_like(x) if dx is None else dx
    L.check(L.lib().rdo_gelu_bwd(_ptr(dy), _ptr(x), x.numel(), _ptr(dx), _stream()), "rdo_gelu_bwd")
    return dx


def round_(x, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().rdo_round(_ptr(x), x.numel(), _ptr(out), _stream()), "rdo_round")
    return out


def ssim_level(x, y, window, c1, c2):
    """x, y: [planes, H, W]; window: 11 Python floats -> (ssim_mean[planes], cs_mean[planes])."""
    planes, H, W = x.shape
    ssim = torch.empty(planes, device=x.device, dtype=torch.float32)
    cs = torch.empty(planes, device=x.device, dtype=torch.float32)
    win = (C.c_float * 11)(*[float(v) for v in window])
    L.check(L.lib().rdo_ssim_level(_ptr(x), _ptr(y), planes, H, W, win, c1, c2, _ptr(ssim), _ptr(cs), _stream()), "rdo_ssim_level")
    return ssim, cs


def avg_pool2(x):
    planes, H, W = x.shape
    ph, pw = H % 2, W % 2
    out = torch.empty((planes, (H + 2 * ph - 2) // 2 + 1, (W + 2 * pw - 2) // 2 + 1), device=x.device, dtype=torch.float32)
    L.check(L.lib().rdo_avg_pool2(_ptr(x), planes, H, W, _ptr(out), _stream()), "rdo_avg_pool2")
    return out


def iter_advance(iter_ptr):
    L.check(L.lib().rdo_iter_advance(_ptr(iter_ptr), _stream()), "rdo_iter_advance")


def factorized_likelihood(z_cl, params, medians):
    """z_cl: [..., C] channels-last.  -> (z_hat, likelihood)"""
    Cc = z_cl.shape[-1]
    zhat, lik = torch.empty_like(z_cl), torch.empty_like(z_cl)
    L.check(L.lib().rdo_factorized_likelihood_fwd(_ptr(z_cl), _ptr(params), _ptr(medians), z_cl.numel(), Cc, _ptr(zhat),
                                                  _ptr(lik), _stream()), "rdo_factorized_likelihood_fwd")
    return zhat, lik


def factorized_likelihood_bwd(zhat_cl, params, grad_scale=1.0):
    """d(grad_scale * sum(-log2 p)) / dz^ of the factorised prior, element-wise (z^ channels-last)"""
    dz = torch.empty_like(zhat_cl)
    L.check(L.lib().rdo_factorized_likelihood_bwd(_ptr(zhat_cl), _ptr(params), zhat_cl.numel(), zhat_cl.shape[-1], grad_scale, _ptr(dz), _stream()),
            "rdo_factorized_likelihood_bwd")
    return dz


def gaussian_likelihood(y, scales, means=None, scale_bound=0.11):
    yhat, lik = torch.empty_like(y), torch.empty_like(y)
    L.check(L.lib().rdo_gaussian_likelihood_fwd(_ptr(y), _ptr(scales), _ptr(means), y.numel(), scale_bound, _ptr(yhat), _ptr(lik),
                                                _stream()), "rdo_gaussian_likelihood_fwd")
    return yhat, lik


def gaussian_likelihood_bwd(yhat, scales, means, grad_scale=1.0, scale_bound=0.11):
    ds, dm = torch.empty_like(scales), torch.empty_like(scales)
    L.check(L.lib().rdo_gaussian_likelihood_bwd(_ptr(yhat), _ptr(scales), _ptr(means), yhat.numel(), scale_bound, grad_scale,
                                                _ptr(ds), _ptr(dm), _stream()), "rdo_gaussian_likelihood_bwd")
    return ds, dm


def neg_log2_sum(lik, scale=1.0, out=None):
    out = torch.zeros(1, device=lik.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_neg_log2_sum(_ptr(lik), lik.numel(), scale, _ptr(out), _stream()), "rdo_neg_log2_sum")
    return out


def sq_diff_sum(a, b, scale=1.0, clamp01=False, out=None):
    out = torch.zeros(1, device=a.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().rdo_sq_diff_sum(_ptr(a), _ptr(b), a.numel(), scale, int(clamp01), _ptr(out), _stream()), "rdo_sq_diff_sum")
    return out


_SCHED_MEMO = {}


def make_sched(iters, warmup, b_range, lr=1e-3, device="cuda"):
    """Per-iteration schedule table (rdo_sched_row): LinearTempDecay (utils.py:37-54), round-loss gate
    (layer_opt.py:159-161), Adam bias corrections.  Computed on the host in double precision, once per distinct schedule: every unit
    of a calibration run shares one (the Python loop over 20 000 iterations took 0.135 s per unit, 2.8 % of the full schedule)."""
    key = (int(iters), float(warmup), float(b_range[0]), float(b_range[1]), float(lr))
    rows = _SCHED_MEMO.get(key)
    if rows is None:
        if len(_SCHED_MEMO) >= 8:
            _SCHED_MEMO.clear()
        rows = _SCHED_MEMO[key] = _make_sched_rows(iters, warmup, b_range, lr)
    return rows.to(device)


def _make_sched_rows(iters, warmup, b_range, lr):
    import math
    rows = torch.empty((iters, 4), dtype=torch.float32)
    t_max, start = iters, warmup * iters
    loss_start = iters * warmup
    for i in range(iters):
        count = i + 1
        if count < start:
            b = float(b_range[0])
        else:
            rel_t = (count - start) / (t_max - start)
            b = b_range[1] + (b_range[0] - b_range[1]) * max(0.0, 1 - rel_t)
        on = 0.0 if count < loss_start else 1.0
        if not on:
            b = 0.0
        rows[i, 0] = b
        rows[i, 1] = on
        rows[i, 2] = lr / (1 - 0.9 ** count)
        rows[i, 3] = math.sqrt(1 - 0.999 ** count)
    return rows


# ----------------------------------------------------------------------------- H2 tensors and fused unit tails
H2_TARGET = 128.0      # a tensor's scale puts its largest magnitude (as probed) near 2^7: x512 head-room to fp16's 65504, fp32-chain
                       # accuracy down to max |x s| = 2^-2 (tools/f16_probe.hip)


def pow2_scale(amax, target=H2_TARGET):
    """The power of two s with target / 2 < amax * s <= target (1.0 for an all-zero tensor)."""
    amax = float(amax)
    if not math.isfinite(amax):
        raise ValueError("pow2_scale: non-finite magnitude")
    if amax <= 0.0:
        return 1.0
    return 2.0 ** min(60, max(-60, math.floor(math.log2(target / amax))))


def h2_empty(shape, device, scale=1.0, flag=None):
    """Planes of an H2 tensor for an fp32 NHWC tensor of `shape` [..., C] (C % 16 == 0): int16 [2, C/16, pixels, 16] -- slice-major
    planes, include/rdo_ptq_hip.h."""
    Cc = shape[-1]
    if Cc % 16:
        raise ValueError(f"H2 tensors need a channel count that is a multiple of 16, got {Cc}")
    npix = 1
    for d in shape[:-1]:
        npix *= int(d)
    return H2(torch.empty((2, Cc // 16, npix, 16), device=device, dtype=torch.int16), scale, flag)


def h2_to_float(planes, shape):
    """(h1 + h2) / scale as an fp32 tensor of `shape` (the original values to 2^-24 relative)."""
    x = (planes.t[0].view(torch.float16).float() + planes.t[1].view(torch.float16).float()) / planes.scale      # [C/16, pixels, 16]
    return x.permute(1, 0, 2).reshape(shape).contiguous()


def h2_overflow(reset=True):
    """True when a producer of H2 planes met a value outside fp16's range since the last reset (synchronises the device)."""
    rc = int(L.lib().rdo_h2_overflow(int(bool(reset))))
    if rc < 0:
        raise RuntimeError("rdo_h2_overflow failed")
    return bool(rc)


class h2_flag:
    """with h2_flag(word): the H2 producers launched -- or recorded into a plan -- by this thread inside the block raise `word` (a
    zero-initialised int32 device tensor the caller keeps alive as long as the recorded plans) instead of the per-device default flag
    (rdo_h2_bind_flag).  An output `H2` that carries its own `.flag` raises that one instead (`_bind_out`).  A raised word holds the
    fp32 bit pattern of the largest FINITE |x * scale| that did not fit in word 0 (`overflow_magnitude`) and a non-finite mark in word 1.  Blocks nest; leaving restores the previous
    binding."""
    _tls = threading.local()          # rdo_h2_bind_flag is thread_local in the library (csrc/runtime.hip): so is this cache of it

    def __init__(self, word):
        if word is not None and not (word.is_cuda and word.dtype == torch.int32 and word.numel() >= 2):
            raise ValueError("h2_flag: expected an int32 CUDA tensor of two words (finite magnitude, non-finite mark)")
        self.word = word

    @staticmethod
    def _state():
        st = h2_flag._tls
        if not hasattr(st, "stack"):
            st.stack, st.bound = [], None
        return st

    @staticmethod
    def bind(word):
        st = h2_flag._state()
        if word is not st.bound:
            L.check(L.lib().rdo_h2_bind_flag(None if word is None else _ptr(word)), "rdo_h2_bind_flag")
            st.bound = word

    def __enter__(self):
        h2_flag._state().stack.append(self.word)
        h2_flag.bind(self.word)
        return self.word

    def __exit__(self, *exc):
        st = h2_flag._state()
        st.stack.pop()
        h2_flag.bind(st.stack[-1] if st.stack else None)
        return False


def _bind_out(planes):
    """the overflow word the next producer raises: the output tensor's own, else the enclosing h2_flag block's, else the default"""
    own = getattr(planes, "flag", None)
    st = h2_flag._state()
    h2_flag.bind(own if own is not None else (st.stack[-1] if st.stack else None))


def overflow_magnitude(word_value):
    """Largest |x * scale| a raised overflow word recorded (its int32 value is the fp32 bit pattern; inf for NaN / inf inputs)."""
    import struct
    v = struct.unpack("<f", struct.pack("<i", int(word_value)))[0]
    return float("inf") if not math.isfinite(v) else v


def split_h2(x, planes=None, scale=None):
    """fp32 NHWC tensor -> H2 planes; without `planes` / `scale` the scale comes from the tensor's own largest magnitude (a device
    synchronisation: set-up and tests only)."""
    if planes is None:
        planes = h2_empty(x.shape, x.device, pow2_scale(x.abs().max()) if scale is None else scale)
    Cc = x.shape[-1]
    _bind_out(planes)
    L.check(L.lib().rdo_split_h2(_ptr(x), x.numel() // Cc, Cc, planes.scale, _ptr(planes), _stream()), "rdo_split_h2")
    return planes


def split_h2_conv(w, planes=None, scale=None):
    """Conv weight in kernel layout [Cout,KH,KW,Cin] (or [C,C] = 1x1) -> H2 weight planes int16 [2, Cout,KH,KW,Cin] in the fragment
    order the split-precision kernels read ([Cin/16][KH][KW][Cout][16])."""
    if w.dim() == 2:
        w = w.reshape(w.shape[0], 1, 1, w.shape[1])
    if w.dim() != 4:
        raise ValueError("split_h2_conv: expected a conv weight [Cout,KH,KW,Cin]")
    if planes is None:
        planes = H2(torch.empty((2,) + tuple(w.shape), device=w.device, dtype=torch.int16), pow2_scale(w.abs().max()) if scale is None else scale)
    co, kh, kw, ci = w.shape
    _bind_out(planes)
    L.check(L.lib().rdo_split_h2_conv(_ptr(w), co, kh, kw, ci, planes.scale, _ptr(planes), _stream()), "rdo_split_h2_conv")
    return planes


def conv_h2_supported(x_shape, w_shape, stride, pad, square_input=False):
    d = conv_desc(x_shape, w_shape, stride, pad, square_input=square_input)
    return bool(L.lib().rdo_conv2d_fwd_h2_supported(C.byref(d)))


def _oscale(p):
    return p.scale if p is not None else 1.0


def conv2d_fwd_h2(xp, x_shape, w_shape, wplanes, bias=None, stride=1, pad=0, epilogue=L.EPI_NONE, aux=None, residual=None, out=None, pre=None,
                  out_planes=None, aux_planes=None):
    """Conv on an H2 input (`xp` = planes of the NHWC tensor of shape `x_shape`) with fragment-ordered H2 weight planes; writes whichever
    of out / pre / out_planes is given."""
    d = conv_desc(x_shape, w_shape, stride, pad, epilogue, False, residual is not None)
    need = int(L.lib().rdo_conv2d_fwd_h2_workspace(C.byref(d)))
    ws = _scratch(xp.device, need) if need else None
    _bind_out(out_planes)
    L.check(L.lib().rdo_conv2d_fwd_h2(C.byref(d), _ptr(xp), xp.scale, _ptr(wplanes), wplanes.scale, _ptr(bias), _ptr(aux), _ptr(aux_planes),
                                      _ptr(residual), _ptr(out), _ptr(pre), _ptr(out_planes), _oscale(out_planes), _ptr(ws),
                                      ws.numel() if ws is not None else 0, _stream()), "rdo_conv2d_fwd_h2")
    return out


def conv_h2_tail_supported(x_shape, w_shape, stride, pad):
    d = conv_desc(x_shape, w_shape, stride, pad)
    return bool(L.lib().rdo_conv2d_fwd_h2_tail_supported(C.byref(d)))


def conv2d_fwd_h2_tail(xp, x_shape, w_shape, wplanes, bias, stride, pad, residual_planes, tgt_cache, idx_table, iter_ptr, coef, act, dpre_planes,
                       loss_log):
    """Plane-input conv + unit tail in one launch: dpre_planes <- dL/dpre of out = act(conv + bias) + residual against tgt_cache[idx]."""
    d = conv_desc(x_shape, w_shape, stride, pad)
    _bind_out(dpre_planes)
    L.check(L.lib().rdo_conv2d_fwd_h2_tail(C.byref(d), _ptr(xp), xp.scale, _ptr(wplanes), wplanes.scale, _ptr(bias), _ptr(residual_planes),
                                           _oscale(residual_planes), _ptr(tgt_cache), _ptr(idx_table), _ptr(iter_ptr), x_shape[0], coef, int(act),
                                           _ptr(dpre_planes), dpre_planes.scale, _ptr(loss_log), _stream()), "rdo_conv2d_fwd_h2_tail")


def gather_qdrop_h2(cache_q, cache_fp, idx_table, iter_ptr, B, prob, seed, out, out_planes, batch_offset=0, iter_publish=None):
    per_image = cache_q[0].numel()
    _bind_out(out_planes)
    L.check(L.lib().rdo_gather_qdrop_h2(_ptr(cache_q), _ptr(cache_fp), _ptr(idx_table), _ptr(iter_ptr), B, int(batch_offset), per_image,
                                        cache_q.shape[-1], prob, seed, _ptr(out), _ptr(out_planes), out_planes.scale, _ptr(iter_publish),
                                        _stream()), "rdo_gather_qdrop_h2")


ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2


def loss_act_bwd(pre, residual, tgt_cache, idx_table, iter_ptr, coef, act, loss_log, out=None, grad_out=None, dpre=None, dpre_planes=None,
                 residual_planes=None):
    B, per_image = pre.shape[0], pre[0].numel()
    _bind_out(dpre_planes)
    L.check(L.lib().rdo_loss_act_bwd(_ptr(pre), _ptr(residual), _ptr(residual_planes), _oscale(residual_planes), _ptr(tgt_cache), _ptr(idx_table),
                                     _ptr(iter_ptr), B, per_image, pre.shape[-1], coef, int(act), _ptr(out), _ptr(grad_out), _ptr(dpre),
                                     _ptr(dpre_planes), _oscale(dpre_planes), _ptr(loss_log), _stream()), "rdo_loss_act_bwd")


def loss_gdn_bwd(x, norm, residual, tgt_cache, idx_table, iter_ptr, coef, inverse, loss_log, grad_out, t=None, t_planes=None, out=None):
    B, per_image = x.shape[0], x[0].numel()
    _bind_out(t_planes)
    L.check(L.lib().rdo_loss_gdn_bwd(_ptr(x), _ptr(norm), _ptr(residual), _ptr(tgt_cache), _ptr(idx_table), _ptr(iter_ptr), B, per_image,
                                     x.shape[-1], coef, int(inverse), _ptr(out), _ptr(grad_out), _ptr(t), _ptr(t_planes), _oscale(t_planes),
                                     _ptr(loss_log), _stream()), "rdo_loss_gdn_bwd")


def gdn_bwd_dx_h2(g, x, norm, acc, inverse, dx=None, dx_planes=None):
    _bind_out(dx_planes)
    L.check(L.lib().rdo_gdn_bwd_dx_h2(_ptr(g), _ptr(x), _ptr(norm), _ptr(acc), g.numel(), g.shape[-1], int(inverse), _ptr(dx),
                                      _ptr(dx_planes), _oscale(dx_planes), _stream()), "rdo_gdn_bwd_dx_h2")


def pixel_shuffle_h2(x, out=None, out_planes=None):
    """[B,H,W,4C] -> [B,2H,2W,C] (r = 2) as fp32 and / or planes."""
    B, H, W, CC = x.shape
    _bind_out(out_planes)
    L.check(L.lib().rdo_pixel_shuffle_h2(_ptr(x), B, H, W, CC // 4, _ptr(out), _ptr(out_planes), _oscale(out_planes), _stream()),
            "rdo_pixel_shuffle_h2")


def pixel_unshuffle2(x, out=None, out_planes=None):
    """[B,2H,2W,C] -> [B,H,W,4C]: gradient of the r = 2 pixel shuffle (16-byte accesses on both sides), as fp32 and / or H2 planes."""
    B, Hr, Wr, Cc = x.shape
    if out is None and out_planes is None:
        out = torch.empty((B, Hr // 2, Wr // 2, 4 * Cc), device=x.device, dtype=torch.float32)
    _bind_out(out_planes)
    L.check(L.lib().rdo_pixel_unshuffle2(_ptr(x), B, Hr // 2, Wr // 2, Cc, _ptr(out), _ptr(out_planes), _oscale(out_planes), _stream()),
            "rdo_pixel_unshuffle2")
    return out


def wgrad_h2_supported(x_shape, w_shape, stride, pad):
    d = conv_desc(x_shape, w_shape, stride, pad)
    return bool(L.lib().rdo_conv2d_wgrad_h2_supported(C.byref(d)))


def wgrad_h2_layer_supported(x_shape, w_shape, stride, pad):
    """rdo_conv2d_wgrad_h2 takes this shape when it is the weight gradient of a layer unit on planes (narrower channel counts than the
    block units' 192: the general plane kernel masks its 192 x 192 tile)."""
    d = conv_desc(x_shape, w_shape, stride, pad)
    return bool(L.lib().rdo_conv2d_wgrad_h2_layer_supported(C.byref(d)))


def conv2d_wgrad_h2(xp, x_shape, dyp, w_shape, stride=1, pad=0, slabs=None):
    """Weight-gradient slabs from H2 operands (planes of x [B,H,W,Cin] and of dy [B,Ho,Wo,Cout])."""
    d = conv_desc(x_shape, w_shape, stride, pad)
    ns = int(L.lib().rdo_conv2d_wgrad_nsplit(C.byref(d))) if slabs is None else slabs.shape[0]
    if slabs is None:
        slabs = torch.empty((ns,) + tuple(w_shape), device=xp.device, dtype=torch.float32)
    L.check(L.lib().rdo_conv2d_wgrad_h2(C.byref(d), _ptr(xp), xp.scale, _ptr(dyp), dyp.scale, _ptr(slabs), ns, _stream()), "rdo_conv2d_wgrad_h2")
    return slabs
