"""Calibration engine for the Lu2022 path: RSTB block units and the task loss through the FP rest of the sub-coder.

The reference trains a Lu2022 unit against two terms (block_opt.py:299-308, layer_opt.py:296-303):

    rec  = lp_loss(unit(x_q),               cached FP output of the unit)
    task = lp_loss(fp_out(unit(x_q)),       fp_out(cached FP output))       fp_out = remaining untrained stages of g_a / h_a /
                                                                            h_s / g_s in full precision, + round_ste for g_a

so every iteration runs forward AND backward through the rest of the sub-coder (up to 14 Swin blocks and 3 resamplers for
g_a0).  `TapeEngine` records that as one static HIP op list per iteration, exactly like `UnitEngine` does for the Cheng2020
units: the forward pass of the trainable unit and of the frozen tail is issued stage by stage, each stage pushing a closure
that issues its backward kernels; the closures run in reverse after the two loss kernels.  Gradients are plain buffers keyed by
the tensor they belong to; a second contribution to the same tensor goes through a temporary and `rdo_add`.

Tokens stay in natural pixel order ([B, H, W, C] == [B, L, C]); Linears run on the conv kernels as 1x1 convolutions (bf16x6
split-precision path for the large ones), the attention core / LayerNorm / GELU on csrc/swin.hip."""
import os
from collections import OrderedDict

import torch

from hipops import _lib as L
from hipops import ops
from hipops.autograd import LIN_H2_MIN_ROWS
from hipops.plan import Plan

from . import dp
from .engine import UnitEngine, _Op
from .quant_layer import QuantModule
from .quantizer import to_rows


class _FpOp:
    """Frozen full-precision parameters of one conv / transposed conv / linear of the tail, in the kernels' layouts.
    Quacks like `_Op` for UnitEngine._conv / ._dgrad (wq4, wd4, bias, stride, pad, K, plane buffers)."""
    is_gdn = False
    slabs = None

    def __init__(self, qm: QuantModule):
        self.qm, self.kind = qm, qm.kind
        w = qm.org_weight.detach()
        self.bias = None if qm.org_bias is None else qm.org_bias.detach().contiguous()
        self.tconv = self.tc_phase = None
        if qm.kind == "linear":
            self.w = w.reshape(w.shape[0], 1, 1, w.shape[1]).contiguous()
            self.stride, self.pad = 1, 0
        elif qm.kind == "conv":
            self.w = to_rows(w)
            self.stride, self.pad = qm.conv_geometry()
        elif qm.kind == "tconv":
            kw = qm.fwd_kwargs
            self.tconv = (int(kw["stride"][0]), int(kw["padding"][0]), int(kw["output_padding"][0]))
            self.w = to_rows(w, tconv=True).flip(1, 2).contiguous()      # forward = stride-1 conv on the zero-inserted input
            self.stride, self.pad = 1, 0
            self.w_bwd = w.permute(0, 2, 3, 1).contiguous()              # dgrad = strided conv of dy with [Cin_t][kh][kw][Cout_t]
            # forward without zero insertion where the geometry allows: stride-1 conv with the phase weight + pixel shuffle
            self.tc_phase = ops.TconvPhase.get(self.w.shape[1], *self.tconv, True, self.w.device)
            if self.tc_phase is not None:
                # (a tail op is built while the engine's plan is being RECORDED: the derived weight must exist before `fill_planes`
                # splits it, i.e. be computed now and not as a recorded op of every iteration)
                with Plan.eager():
                    self.wp = ops.tconv_expand(self.w, self.tc_phase)
                self.bias_p = None if self.bias is None else self.bias.repeat_interleave(self.tc_phase.S2).contiguous()
        else:
            raise NotImplementedError(f"tail stage of kind '{qm.kind}'")
        self.K = self.w.shape[1]
        self.w4 = tuple(self.w.shape)
        self.same = self.tconv is None and self.stride == 1 and 2 * self.pad == self.K - 1
        if self.same:
            self.wd = self.w.flip(1, 2).permute(3, 1, 2, 0).contiguous()         # [ci][kh'][kw'][co], taps flipped
        elif self.tconv is None:
            # dgrad of a strided conv = transposed conv of dy: stride-1 conv of the zero-inserted dy with this layout
            self.w_bwd = self.w.permute(3, 1, 2, 0).flip(1, 2).contiguous()       # [ci][kh'][kw'][co]
            # ... or, where that transposed conv's output is stride x its input (k = 3 / s = 2 / p = 1 of models/nic_cvt.py on even
            # sizes), its phase form: a stride-1 conv of dy itself with the phase weight + pixel shuffle -- no zero insertion, a quarter
            # of the multiply-adds, and a shape the split-precision kernel takes (decided per problem in `_fp_conv`)
            self.bw_phase_of = lambda opad: ops.TconvPhase.get(self.K, self.stride, self.pad, opad, False, self.w.device)
            self.wbp = self.wbp_planes = None
        self.wq_planes = self.wd_planes = None
        self.lin_fwd = self.lin_bwd = None

    def lin_planes(self, fwd: bool):
        """Frozen Linear for rdo_linear_h2: planes of W (fwd) / W^T (input gradient), allocated here (possibly inside a recording),
        filled by `fill_planes` once the recording is over.  Scale: a power of two from the weight's own magnitude (a torch reduction:
        not a recorded op)."""
        co, _, _, ci = self.w4
        if getattr(self, "_lin_scale", None) is None:
            self._lin_scale = ops.pow2_scale(self.w.abs().max())
        if fwd:
            if self.lin_fwd is None:
                self.lin_fwd = ops.H2(torch.empty((2, ci // 32, co // 16, 64, 8), device=self.w.device, dtype=torch.int16), self._lin_scale)
            return self.lin_fwd
        if self.lin_bwd is None:
            self.lin_bwd = ops.H2(torch.empty((2, co // 32, ci // 16, 64, 8), device=self.w.device, dtype=torch.int16), self._lin_scale)
        return self.lin_bwd

    def wq4(self):
        return self.w

    def wd4(self):
        return self.wd

    def enable_planes(self, fwd, dgrad):
        """Called while the plan is being recorded: allocate only; `fill_planes` runs eagerly afterwards (weights are frozen)."""
        if fwd and self.wq_planes is None:
            self.wq_planes = torch.empty((3,) + tuple(self.w.shape), device=self.w.device, dtype=torch.int16)
        if dgrad and self.wd_planes is None and getattr(self, "wd", None) is not None:
            self.wd_planes = torch.empty((3,) + tuple(self.wd.shape), device=self.wd.device, dtype=torch.int16)

    def fill_planes(self):
        if self.lin_fwd is not None:
            ops.split_h2_linear(self.w, planes=self.lin_fwd)
        if self.lin_bwd is not None:
            ops.split_h2_linear(self.wd, planes=self.lin_bwd)
        if getattr(self, "wp_planes", None) is not None:
            ops.split_bf16x3(self.wp, self.wp_planes)
        if getattr(self, "wbp_planes", None) is not None:
            ops.split_bf16x3(self.wbp, self.wbp_planes)
        if self.wq_planes is not None:
            ops.split_bf16x3(self.w, self.wq_planes)
        if self.wd_planes is not None:
            ops.split_bf16x3(self.wd, self.wd_planes)

    def refresh_planes(self):
        pass


class _Ln:
    """LayerNorm parameters: trainable (gamma = soft-rounded AdaRound weight of `op`) or frozen."""

    def __init__(self, qm: QuantModule, op=None):
        self.op = op
        self.gamma = op.wq.view(-1) if op is not None else qm.org_weight.detach().contiguous()
        self.beta = (op.bias if op is not None else (None if qm.org_bias is None else qm.org_bias.detach().contiguous()))


class TapeEngine(UnitEngine):
    """UnitEngine for kind 'rstb' and for any unit that carries a tail (`tail`: list of untrained QuantModule / QuantRSTB stages
    run in full precision; `tail_round`: round_ste after them, i.e. the unit sits in g_a; `task_cache`: fp_out of the cached FP
    outputs, [n, H', W', C'] NHWC)."""

    def __init__(self, kind, modules, cache_q, cache_fp, cache_out, *, tail=(), tail_round=False, task_cache=None, **kw):
        self.tail, self.tail_round, self.task_cache = list(tail), bool(tail_round), task_cache
        if (self.tail or self.tail_round) and task_cache is None:
            raise ValueError("TapeEngine: a tail needs the task target cache")
        self._keep, self._fp = [], []
        super().__init__(kind, modules, cache_q, cache_fp, cache_out, **kw)
        for p in self._fp:
            p.fill_planes()             # frozen tail weights: exact bf16 splits made once, outside the recorded plan

    # ------------------------------------------------------------------------------------------------------------------ ops
    def _build_ops(self):
        if self.kind != "rstb":
            return super()._build_ops()
        o = OrderedDict()
        rstb = self.mods["rstb"]
        for i, blk in enumerate(rstb.residual_group.blocks):
            pre = f"residual_group.blocks.{i}."
            o[pre + "norm1"] = _Op(pre + "norm1", blk.norm1, False)
            o[pre + "attn.qkv"] = _Op(pre + "attn.qkv", blk.attn.qkv, True)
            o[pre + "attn.proj"] = _Op(pre + "attn.proj", blk.attn.proj, True)
            o[pre + "norm2"] = _Op(pre + "norm2", blk.norm2, False)
            o[pre + "mlp.fc1"] = _Op(pre + "mlp.fc1", blk.mlp.fc1, True)
            o[pre + "mlp.fc2"] = _Op(pre + "mlp.fc2", blk.mlp.fc2, True)
        self.ops = o
        self._late = None                      # one bucket, one all-reduce (no early/late split for the tape units)
        if self.split:
            self.bucket = dp.GradBucket(OrderedDict((n, op.numel()) for n, op in o.items()), device=self.dev, group=self.group)
            for n, op in o.items():
                op.dalpha = self.bucket.view(n)

    def _alloc(self):
        if self.kind != "rstb":
            return super()._alloc()
        _, H, W, Cin = self.cq.shape
        self.x_in = self._buf(self.B, H, W, Cin)
        self.t = {}

    def _buf(self, *shape):
        t = super()._buf(*shape)
        self._keep.append(t)
        return t

    # ------------------------------------------------------------------------------------------------------------------ tape
    # Gradients are LAZY SUMS: G[id(t)] is the list of buffers whose sum is dL/dt.  A backward closure appends the buffer it wrote
    # (`_gput`); a consumer that needs the sum as one tensor asks `_gget` (one add per extra contribution); the LayerNorm backward
    # takes up to two contributions as addends of its own output (`rdo_layer_norm_bwd_add`), which is where the residual paths of a
    # Swin block meet -- so a block's backward runs without a single add kernel.  Buffers are written once and never accumulated in
    # place, so sharing one buffer between two tensors (both inputs of an add) needs no alias tracking.
    # A/B switches (whole runs).  RDO_SWIN_FUSE_GELU: 1 (default) = the Mlp's GELU / its derivative in the epilogue of fc1 / of fc2's input
    # gradient WHERE that conv is split over K -- its second pass touches every output element anyway; 2 = in every launch; 0 = never.
    # Measured on g_a1 at full size (profiles/r05c_lu2022_g_a1_trace.md): inside the main GEMM kernels the erf / exp polynomials run in
    # the epilogue of a 128 x 192 tile while the matrix pipe idles -- 126 us against 56 + 19 (GEMM + separate GELU kernel) per launch
    # on the split-bf16 kernel, 23 against 15.5 + 5 on the fp32 one -- so fusing everywhere LOSES 0.4 ms per iteration.
    # RDO_SWIN_FUSE_LN=0 -> separate add kernels around the plain LayerNorm kernels
    fuse_gelu = int(os.environ.get("RDO_SWIN_FUSE_GELU", "1"))
    fuse_ln = os.environ.get("RDO_SWIN_FUSE_LN", "1") != "0"
    # RDO_SWIN_LIN_H2=0: the Linears of the large token matrices back on the split-bf16 1x1-conv kernel (six products, csrc/conv_fwd_x6.hip)
    lin_h2 = os.environ.get("RDO_SWIN_LIN_H2", "1") != "0"
    lin_gelu = os.environ.get("RDO_SWIN_LIN_GELU", "1") != "0"      # GELU / its derivative in rdo_linear_h2's epilogue (0: separate kernels)
    fuse_ln_fwd = os.environ.get("RDO_SWIN_FUSE_LN_FWD", "1") != "0"

    def _gput(self, t, buf):
        self.G.setdefault(id(t), []).append(buf)

    def _gparts(self, t):
        return self.G.get(id(t), [])

    def _gget(self, t):
        parts = self.G[id(t)]
        while len(parts) > 1:
            acc = self._buf(*t.shape)
            ops.add(parts[0], parts[1], out=acc)
            parts[:2] = [acc]
        return parts[0]

    def _gnew(self, t):
        buf = self._buf(*t.shape)
        self._gput(t, buf)
        return buf

    # -- linear over the token matrix ------------------------------------------------------------------------------------
    def _linear(self, x, p, need_dx=True, gelu=False, gelu_in=None):
        """y = x W^T + b.  gelu=True: the Mlp's activation in fc1's epilogue -- returns (gelu(y), y); gelu_in: this linear's INPUT is
        gelu(gelu_in), so its input gradient goes through RDO_EPI_GELU_BWD and is stored as the gradient of `gelu_in`."""
        rows, cin = x.numel() // x.shape[-1], x.shape[-1]
        cout = p.w4[0]
        y = self._buf(*x.shape[:-1], cout)
        x4, y4 = x.view(1, 1, rows, cin), y.view(1, 1, rows, cout)
        pre = None
        # large token matrices: per-token-scaled fp16-split kernel -- from the same token count as the tape's Linears (hipops.autograd) and
        # the engine's GDN GEMMs: below it the launch is a few dozen workgroups that each stream the whole weight chunk and the conv
        # kernels' split-K forms win (and the tape path and this one would use different arithmetic for the same layer shape)
        big = rows >= LIN_H2_MIN_ROWS
        lin_f = self.lin_h2 and big and ops.linear_h2_supported(rows, cin, cout)
        lin_b = self.lin_h2 and big and ops.linear_h2_supported(rows, cout, cin)
        bias_of = lambda: p.bias
        fuse_here = lambda xs, w4, has_planes: self.fuse_gelu == 2 or (
            self.fuse_gelu == 1 and ops.conv_fwd_ksplit(tuple(xs), tuple(w4), 1, 0, has_planes, self.dev)[0] >= 2)
        if lin_f:
            if gelu and self.lin_gelu:                         # Mlp.fc1 + GELU in the kernel's epilogue (the pre-activation is kept for the backward)
                pre = self._buf(*y.shape)
                ops.linear_h2(x, p.lin_planes(True), bias_of(), out=y, epilogue=L.EPI_GELU, pre=pre)
            elif gelu:
                pre = self._buf(*y.shape)
                ops.linear_h2(x, p.lin_planes(True), bias_of(), out=pre)
                ops.gelu(pre, out=y)
            else:
                ops.linear_h2(x, p.lin_planes(True), bias_of(), out=y)
        elif gelu and fuse_here(x4.shape, p.w4, ops.uses_bf16x6(tuple(x4.shape), p.w4, 1, 0)):
            pre = self._buf(*y.shape)
            self._conv(p, x4, y4, epilogue=L.EPI_GELU, pre=pre.view(1, 1, rows, cout))
        elif gelu:
            pre = self._buf(*y.shape)
            self._conv(p, x4, pre.view(1, 1, rows, cout))
            ops.gelu(pre, out=y)
        else:
            self._conv(p, x4, y4)

        def bwd():
            dy4 = self._gget(pre if gelu else y).view(1, 1, rows, cout)
            if isinstance(p, _Op):
                self._wgrad(p, x4, dy4)
            if need_dx and lin_b:
                dy = dy4.view(rows, cout)
                if gelu_in is not None and self.lin_gelu:      # fc2's input gradient times gelu'(fc1's pre-activation), in the epilogue
                    ops.linear_h2(dy, p.lin_planes(False), None, out=self._gnew(gelu_in), epilogue=L.EPI_GELU_BWD, aux=gelu_in)
                elif gelu_in is not None:
                    dg = self._buf(*x.shape)
                    ops.linear_h2(dy, p.lin_planes(False), None, out=dg)
                    ops.gelu_bwd(dg, gelu_in, self._gnew(gelu_in))
                else:
                    ops.linear_h2(dy, p.lin_planes(False), None, out=self._gnew(x))
            elif need_dx:
                wd4 = tuple(p.wd4().shape)
                if gelu_in is not None and fuse_here(dy4.shape, wd4, ops.uses_bf16x6(tuple(dy4.shape), wd4, 1, 0)):
                    dx = self._gnew(gelu_in)
                    self._dgrad(p, dy4, dx.view(1, 1, rows, cin), epilogue=L.EPI_GELU_BWD, aux=gelu_in.view(1, 1, rows, cin))
                elif gelu_in is not None:
                    dg = self._buf(*x.shape)
                    self._dgrad(p, dy4, dg.view(1, 1, rows, cin))
                    ops.gelu_bwd(dg, gelu_in, self._gnew(gelu_in))
                else:
                    self._dgrad(p, dy4, self._gnew(x).view(1, 1, rows, cin))
        self.tape.append(bwd)
        return (y, pre) if gelu else y

    def _add_ln(self, a, b, ln: _Ln, need_ds=True):
        """s = a + b (b None: s is a), y = LayerNorm(s) -> (s, y).  Backward: ds = (what reached s along the residual path) +
        LayerNorm-backward(dy); a and b both receive ds."""
        y = self._buf(*a.shape)
        s = a if b is None else self._buf(*a.shape)
        if not (self.fuse_ln and self.fuse_ln_fwd):
            if b is not None:
                ops.add(a, b, out=s)
            ops.layer_norm(s, ln.gamma, ln.beta, out=y)
        elif b is None:
            ops.add_layer_norm(a, None, ln.gamma, ln.beta, out=y)
        else:
            ops.add_layer_norm(a, b, ln.gamma, ln.beta, sum_out=s, out=y)

        def bwd():
            dy = self._gget(y)
            slabs = None
            if ln.op is not None:
                # dgamma partial sums: one 4-wave block per slab.  Many slabs keep the 65k-row maps parallel; they are folded
                # 32 -> 1 by rdo_reduce_slabs so that the AdaRound step (one thread per weight) sums at most 32 of them
                rows, Cc = s.numel() // s.shape[-1], s.shape[-1]
                if ln.op.slabs is None:
                    n1 = max(1, min(32, (rows + 15) // 16))
                    n2 = 32 if rows >= 16 * 32 * 4 else 1
                    ln.op.slabs = self._buf(n1, Cc)
                    ln.op.slabs_wide = self._buf(n2 * n1, Cc) if n2 > 1 else ln.op.slabs
                slabs = ln.op.slabs_wide
            ds = None
            if need_ds:
                parts = self._gparts(s)
                if len(parts) > 2:
                    self._gget(s)
                    parts = self._gparts(s)
                ds = self._buf(*s.shape)
                if self.fuse_ln:
                    ops.layer_norm_bwd_add(s, ln.gamma, dy, *parts, dx=ds, dgamma_slabs=slabs)
                else:
                    ops.layer_norm_bwd(s, ln.gamma, dy, dx=ds, dgamma_slabs=slabs)
                    for extra in parts:
                        nxt = self._buf(*s.shape)
                        ops.add(extra, ds, out=nxt)
                        ds = nxt
                if b is None:
                    self.G[id(s)] = [ds]
                else:
                    self._gput(a, ds)
                    self._gput(b, ds)
            elif self.fuse_ln:
                ops.layer_norm_bwd_add(s, ln.gamma, dy, dx=None, dgamma_slabs=slabs)
            else:
                ops.layer_norm_bwd(s, ln.gamma, dy, dx=None, dgamma_slabs=slabs)
            if ln.op is not None and ln.op.slabs_wide is not ln.op.slabs:
                n1 = ln.op.slabs.shape[0]
                ops.reduce_slabs(slabs.view(slabs.shape[0] // n1, n1 * s.shape[-1]), out=ln.op.slabs.view(-1))
        self.tape.append(bwd)
        return s, y

    def _attention(self, qkv, desc, bias):
        out = self._buf(desc.B, desc.H, desc.W, desc.C)
        ops.window_attention(desc, qkv, bias, out=out)

        def bwd():
            ops.window_attention_bwd(desc, qkv, bias, self._gget(out), self._gnew(qkv))
        self.tape.append(bwd)
        return out

    # -- one RSTB (trainable: AdaRound ops of this engine; frozen: the stage's FP parameters) -----------------------------------
    def _rstb(self, x, rstb, ops_of=None, need_dx=True):
        B, H, W, C = x.shape
        ta, tb = x, None                      # the running token tensor is ta + tb; the sum is formed by the LayerNorm that reads it
        for i, blk in enumerate(rstb.residual_group.blocks):
            pre = f"residual_group.blocks.{i}."
            if ops_of is not None:
                p = {k: ops_of[pre + k] for k in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")}
                ln1, ln2 = _Ln(blk.norm1, ops_of[pre + "norm1"]), _Ln(blk.norm2, ops_of[pre + "norm2"])
            else:
                p = {"attn.qkv": _FpOp(blk.attn.qkv), "attn.proj": _FpOp(blk.attn.proj), "mlp.fc1": _FpOp(blk.mlp.fc1),
                     "mlp.fc2": _FpOp(blk.mlp.fc2)}
                ln1, ln2 = _Ln(blk.norm1), _Ln(blk.norm2)
                self._keep.extend(p.values())
                self._fp.extend(p.values())
            want_dt = need_dx or i > 0        # the gradient of the unit's own (cached) input is never needed
            t, n1 = self._add_ln(ta, tb, ln1, need_ds=want_dt)
            qkv = self._linear(n1, p["attn.qkv"])
            desc = ops.attn_desc(B, H, W, C, blk.num_heads, blk.window_size, blk.shift_size, blk.attn.scale)
            bias = blk.attn.position_bias()
            self._keep.append(bias)
            att = self._attention(qkv, desc, bias)
            pr = self._linear(att, p["attn.proj"])
            t1, n2 = self._add_ln(t, pr, ln2)
            g, f1 = self._linear(n2, p["mlp.fc1"], gelu=True)
            f2 = self._linear(g, p["mlp.fc2"], gelu_in=f1)
            ta, tb = t1, f2
        y = self._buf(*x.shape)
        ops.add3(ta, tb, x, out=y)            # (t1 + f2) + x: last block's sum, then the RSTB's residual (layers.py:300, 433)

        def bwd():
            dy = self._gget(y)
            self._gput(ta, dy)
            self._gput(tb, dy)
            if need_dx:
                self._gput(x, dy)
        self.tape.append(bwd)
        return y

    # -- frozen conv / transposed conv stage of the tail -----------------------------------------------------------------------
    def _fp_conv(self, x, qm):
        p = _FpOp(qm)
        self._keep.append(p)
        self._fp.append(p)
        B, H, W, _ = x.shape
        if p.tconv is None:
            Ho, Wo = (H + 2 * p.pad - p.K) // p.stride + 1, (W + 2 * p.pad - p.K) // p.stride + 1
            y = self._buf(B, Ho, Wo, p.w4[0])
            self._conv(p, x, y)

            def bwd():
                dy = self._gget(y)
                dx = self._gnew(x)
                if p.same:
                    self._dgrad(p, dy, dx)
                else:
                    q = p.K - 1 - p.pad
                    opad = H - ((Ho - 1) * p.stride - 2 * p.pad + p.K)
                    ph = p.bw_phase_of(opad) if H == Ho * p.stride and W == Wo * p.stride else None
                    if ph is not None:
                        if p.wbp is None:                    # [s^2 Cin][K'][K'][Cout] from the weight read as a transposed conv's
                            with Plan.eager():
                                p.wbp = ops.tconv_expand(p.w.permute(3, 1, 2, 0).contiguous(), ph)
                            if ops.uses_bf16x6(tuple(dy.shape), tuple(p.wbp.shape), 1, ph.pad):
                                p.wbp_planes = torch.empty((3,) + tuple(p.wbp.shape), device=self.dev, dtype=torch.int16)
                        dp = self._buf(B, Ho, Wo, p.wbp.shape[0])
                        ops.conv2d_fwd(dy, p.wbp, None, 1, ph.pad, out=dp, wplanes=p.wbp_planes)
                        self._shuffle(dp, p.stride, dx)
                    else:
                        Hu, Wu = (Ho - 1) * p.stride + 1 + 2 * q + opad, (Wo - 1) * p.stride + 1 + 2 * q + opad
                        du = self._buf(B, Hu, Wu, p.w4[0])
                        ops.zero_insert(dy, p.stride, q, q, Hu, Wu, out=du)
                        ops.conv2d_fwd(du, p.w_bwd, None, 1, 0, out=dx)
            self.tape.append(bwd)
            return y
        s_, p_, op_ = p.tconv
        if p.tc_phase is not None:
            ph = p.tc_phase
            yp = self._buf(B, H, W, p.wp.shape[0])
            if ops.uses_bf16x6(tuple(x.shape), tuple(p.wp.shape), 1, ph.pad) and getattr(p, "wp_planes", None) is None:
                p.wp_planes = torch.empty((3,) + tuple(p.wp.shape), device=self.dev, dtype=torch.int16)
            ops.conv2d_fwd(x, p.wp, p.bias_p, 1, ph.pad, out=yp, wplanes=getattr(p, "wp_planes", None))
            y = self._buf(B, H * s_, W * s_, p.w4[0])
            self._shuffle(yp, s_, y)
        else:
            q = p.K - 1 - p_
            Hu, Wu = (H - 1) * s_ + 1 + 2 * q + op_, (W - 1) * s_ + 1 + 2 * q + op_
            xu = self._buf(B, Hu, Wu, x.shape[-1])
            ops.zero_insert(x, s_, q, q, Hu, Wu, out=xu)
            y = self._buf(B, Hu - p.K + 1, Wu - p.K + 1, p.w4[0])
            self._conv(p, xu, y)

        def bwd():
            ops.conv2d_fwd(self._gget(y), p.w_bwd, None, s_, p_, out=self._gnew(x))
        self.tape.append(bwd)
        return y

    # ------------------------------------------------------------------------------------------------------------------
    def _unit_forward(self, x):
        """Trainable unit -> (output tensor, closure list already on the tape)."""
        if self.kind == "rstb":
            return self._rstb(x, self.mods["rstb"], ops_of=self.ops, need_dx=False)
        op = self.ops["layer"]
        if op.is_gdn:
            raise NotImplementedError("TapeEngine: GDN units do not occur in Lu2022 coders")
        t = self.t
        xin = x
        phase = op.tconv is not None and op.tc_phase is not None
        if op.tconv is not None and not phase:
            s_, q_, Hu, Wu = self.tc_geom
            xin = ops.zero_insert(x, s_, q_, q_, Hu, Wu, out=t["xu"])
        epi = op.qm.fused_epilogue() if self.include_act else None
        y = t["y"]
        if phase:
            self._tconv_forward(op, x, y, epilogue=L.EPI_NONE if epi is None else epi)
        else:
            self._conv(op, xin, y, epilogue=L.EPI_NONE if epi is None else epi)

        def bwd():
            dy = self._gget(y)
            if epi is not None:
                (ops.lrelu_bwd if epi == L.EPI_LRELU else ops.relu_bwd)(dy, y, t["dpre"])
                dy = t["dpre"]
            if phase:
                self._tconv_wgrad(op, x, dy)
            else:
                self._wgrad(op, xin, dy)
        self.tape.append(bwd)
        return y

    def _forward_backward(self):
        if self.kind != "rstb" and not self.tail and not self.tail_round:
            return super()._forward_backward()
        from .quant_block import QuantRSTB
        self.tape, self.G = [], {}
        x = self.x_in
        self._gather(x)
        y = self._unit_forward(x)
        n_unit = len(self.tape)
        plain = not self.tail and not self.tail_round                 # fp_out is the identity: task == rec (coef 2)
        dy = self._buf(*y.shape)
        if plain:
            self._loss(y, dy)
        else:
            ops.lp2_loss_grad(y, self.co, self.idx, self.it, 1.0, dy, self.loss_log)
        if not plain:
            z = y
            for st in self.tail:
                z = self._rstb(z, st, ops_of=None, need_dx=True) if isinstance(st, QuantRSTB) else self._fp_conv(z, st)
            zr = z
            if self.tail_round:
                zr = self._buf(*z.shape)
                ops.round_(z, out=zr)                                  # round_ste: identity gradient
                self.z_rounded = zr                                    # (inspection: the latents the task term saw in the last iteration)
            dz = self._buf(*z.shape)
            if self.task_p == 2.0:
                ops.lp2_loss_grad(zr, self.task_cache, self.idx, self.it, 1.0, dz, self.task_log)
            else:
                ops.lp_loss_grad(zr, self.task_cache, self.idx, self.it, 0.0, 1.0, self.task_p, dz, self.task_log)
            if z is y:                                                 # round-only tail: both terms meet at the unit output
                self._gput(y, dz)
            else:
                self._gput(z, dz)
                for bw in reversed(self.tape[n_unit:]):
                    bw()
        self._gput(y, dy)                                              # rec term last: task + rec are summed where the unit's backward starts
        for bw in reversed(self.tape[:n_unit]):
            bw()
        self.G.clear()
        self.tape = []
