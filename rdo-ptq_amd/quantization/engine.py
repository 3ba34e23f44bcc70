"""Calibration engine: the AdaRound hot loop of one reconstruction unit as a recorded HIP op list.

This replaces the Python/autograd loop body of the reference (`for i in range(iters)` at
/root/reference/task-oriented-PTQ/quantization/layer_opt.py:287-309 and block_opt.py:287-311):

    idx -> gather cached (x_q, x_fp) -> QDrop mix -> unit forward with soft-rounded weights -> rec + task + round loss
        -> backward -> Adam on the rounding logits alpha

Forward, the hand-derived backward, the loss and the optimiser are enqueued as one native plan per iteration
(hipops.plan.Plan -> rdo_plan_run, hipGraph replay), with every per-iteration scalar (temperature b, Adam bias corrections,
mini-batch indices) read on the device from tables indexed by a device-side iteration counter: zero host work or syncs
inside the loop.  Units: a single `QuantModule` ('layer': conv, transposed conv, GDN / IGDN, with a fused LeakyReLU / ReLU) and the
Cheng2020 blocks QuantRB / QuantRBWS / QuantRBU; the Lu2022 RSTB blocks and every unit whose task loss runs through the rest of
its sub-coder are handled by `swin_engine.TapeEngine`, a subclass that reuses the ops, buffers and plans defined here.

Loss = round + rec + task exactly as the reference computes it for Sequential-indexed CompressAI models, where the task
term degenerates to a second copy of the reconstruction term: `coef = 2`.

Data parallel (world_size > 1): each rank holds its shard of the caches; per iteration the chained data-gradient of all
alphas of the unit is written into one flat bucket, all-reduced (RCCL over xGMI, SUM) and applied with scale 1/world_size;
the rounding regulariser is data independent and is added locally after the reduction."""
import math
import os
from collections import OrderedDict

import torch

from hipops import _lib as L
from hipops import ops
from hipops.plan import Plan

from . import dp
from .quant_layer import QuantModule
from .quantizer import AdaRoundQuantizer, from_rows, to_rows

UNIT_KINDS = ("layer", "rb", "rbws", "rbu", "rstb")


class DpStallError(RuntimeError):
    """The captured data-parallel loop stopped making progress (`UnitEngine._hb_wait`): a collective inside the graph hangs."""


class IdxStream:
    """Mini-batch index tables drawn from torch's GLOBAL CPU generator exactly as the reference draws them -- one `torch.randperm(n)` per
    iteration (layer_opt.py:289) -- with the NEXT unit's table drawn ahead of time while the GPU runs the current unit's loop (20 000
    draws of randperm(256) cost the host ~0.1 s per unit, 2 % of the full schedule, with the GPU idle).

    Exactness: the look-ahead works on a PRIVATE generator started from a copy of the global generator's state; it is adopted only if the
    next request asks for the same (n, B, iters) AND the global generator's state is still that copy (nobody drew from it or re-seeded it in
    between) -- the global generator is then moved to the private generator's final state, i.e. exactly where the draws would have left
    it.  Anything else discards the look-ahead and draws from the global generator as before."""
    _spec = None

    @classmethod
    def begin(cls, n, B, iters):
        g = torch.Generator()
        state0 = torch.get_rng_state()
        g.set_state(state0)
        cls._spec = dict(key=(int(n), int(B), int(iters)), state0=state0, gen=g, rows=[])

    @classmethod
    def step(cls, k):
        """draw up to k more rows of the look-ahead (called between the enqueues of `UnitEngine.run`)"""
        sp = cls._spec
        if sp is None:
            return
        n, B, iters = sp["key"]
        k = min(int(k), iters - len(sp["rows"]))
        g = sp["gen"]
        sp["rows"].extend(torch.randperm(n, generator=g)[:B] for _ in range(k))

    @classmethod
    def take(cls, n, B, iters):
        sp, cls._spec = cls._spec, None
        if sp is not None and sp["key"] == (int(n), int(B), int(iters)) and torch.equal(torch.get_rng_state(), sp["state0"]):
            g = sp["gen"]
            sp["rows"].extend(torch.randperm(n, generator=g)[:B] for _ in range(iters - len(sp["rows"])))
            torch.set_rng_state(g.get_state())
            return torch.stack(sp["rows"])
        return torch.stack([torch.randperm(n)[:B] for _ in range(iters)])


class _Op:
    """Device state of one trainable QuantModule inside a unit."""

    def __init__(self, name, qm: QuantModule, need_dgrad: bool):
        self.name, self.qm, self.need_dgrad = name, qm, need_dgrad
        self.is_gdn = qm.kind == "gdn"
        self.tconv = None                     # (stride, padding, output_padding) of a ConvTranspose2d unit
        self.is_ln = qm.kind == "layernorm"
        if qm.kind not in ("conv", "gdn", "tconv", "linear", "layernorm"):
            raise NotImplementedError(f"calibration engine: QuantModule kind '{qm.kind}' is not supported yet")
        wq = qm.weight_quantizer
        if not wq.inited:
            wq(qm.weight)                      # lazy scale init, as the reference's first forward would do
        w = qm.org_weight.detach()
        if qm.kind == "linear":
            self.w = w.reshape(w.shape[0], 1, 1, w.shape[1]).contiguous()      # a Linear is a 1x1 conv over the token matrix
        elif qm.kind == "layernorm":
            self.w = w.reshape(1, -1).contiguous()                              # gamma: one quantisation "channel" (per tensor)
        elif qm.kind == "tconv":
            # a transposed conv is the forward conv kernel (stride 1, pad 0) on the zero-inserted input with the taps
            # flipped: keep every per-weight tensor of the engine in that flipped [co][kh'][kw'][ci] layout
            self.w = to_rows(w, tconv=True).flip(1, 2).contiguous()
        else:
            self.w = to_rows(w)                # OHWI or [C,C]
        dev = self.w.device
        self.rows = self.w.shape[0]
        # layer-wise scales (main2.py without --channel_wise): one (delta, zero_point) repeated over the rows
        self.channel_wise = bool(wq.channel_wise)
        self.delta = wq.delta.reshape(-1).to(dev).expand(self.rows).contiguous()
        self.zp = wq.zero_point.reshape(-1).to(dev).expand(self.rows).contiguous()
        self.n_levels = wq.n_levels
        if self.is_ln:
            self.desc = ops.ada_desc(self.w, self.n_levels, conv_layout=False)
            self.stride, self.pad, self.K = 1, 0, 1
            self.w4 = tuple(self.w.shape)
            self.bias = None if qm.bias is None else qm.bias.detach().contiguous()
        elif qm.kind == "linear":
            self.stride, self.pad, self.K = 1, 0, 1
            self.desc = ops.ada_desc(self.w, self.n_levels)
            self.w4 = tuple(self.w.shape)
            self.bias = None if qm.bias is None else qm.bias.detach().contiguous()
        elif self.is_gdn:
            self.inverse = bool(qm.fwd_kwargs["inverse"])
            self.beta, reparam = qm.gdn_constants()
            self.beta = self.beta.to(dev).contiguous()
            self.desc = ops.ada_desc(self.w, self.n_levels, reparam=reparam)
            self.stride, self.pad, self.K = 1, 0, 1
            self.w4 = (self.rows, 1, 1, self.rows)
            self.bias = None
        else:
            if qm.kind == "tconv":
                kw = qm.fwd_kwargs
                if kw["groups"] != 1 or tuple(kw["dilation"]) != (1, 1):
                    raise NotImplementedError("calibration engine: grouped / dilated transposed convolutions")
                self.tconv = (int(kw["stride"][0]), int(kw["padding"][0]), int(kw["output_padding"][0]))
                self.stride, self.pad = 1, 0
            else:
                self.stride, self.pad = qm.conv_geometry()
            self.K = self.w.shape[1]
            self.desc = ops.ada_desc(self.w, self.n_levels)
            self.w4 = tuple(self.w.shape)
            self.bias = None if qm.bias is None else qm.bias.detach().contiguous()
        # bound of |soft-quantised weight| over the whole run (every level of every row's grid; through the GDN re-parametrisation
        # where there is one) -> the power-of-two scale of this op's fp16 weight planes
        lv = torch.maximum(self.zp.abs(), (self.n_levels - 1 - self.zp).abs()) * self.delta
        wb = float(lv.max())
        if self.is_gdn:
            wb = max(wb, float(self.desc.reparam_bound)) ** 2
        self.wscale = ops.pow2_scale(wb)
        self.alpha = torch.empty_like(self.w)
        self.m = torch.zeros_like(self.w)
        self.v = torch.zeros_like(self.w)
        self.wq = torch.empty_like(self.w)
        # wd: dgrad layout for convs, transpose for gamma (always needed: GDN backward uses gamma'^T)
        self.wd = torch.empty_like(self.w) if (need_dgrad or self.is_gdn) else None
        ops.adaround_init_alpha(self.desc, self.w, self.delta, self.alpha)
        ops.adaround_fwd(self.desc, self.w, self.alpha, self.delta, self.zp, True, self.wq, self.wd)
        self.slabs = None
        # transposed conv whose output is stride x the input: run as a stride-1 conv with the "phase weight" [Cout * s^2][K'][K'][Cin]
        # + pixel shuffle (include/rdo_ptq_hip.h, rdo_tconv_expand) instead of zero insertion + dense conv.  The AdaRound state stays in
        # the kernel layout of the transposed conv; the phase weight is re-expanded after every step, its gradient folded back.
        self.tc_phase = self.wp = self.wp_planes = self.wp_h2 = self.slabs_p = self.bias_p = None
        if self.tconv is not None:
            s_, p_, op_ = self.tconv
            self.tc_phase = ops.TconvPhase.get(self.K, s_, p_, op_, True, dev) if os.environ.get("RDO_TCONV_PHASE", "1") != "0" else None
            if self.tc_phase is not None:
                self.wp4 = self.tc_phase.w_shape(self.rows, self.w4[3])
                self.wp = torch.empty(self.wp4, device=dev, dtype=torch.float32)
                self.bias_p = None if self.bias is None else self.bias.repeat_interleave(self.tc_phase.S2).contiguous()
                self.expand_phase()
        # exact bf16 three-way splits of the kernel-layout weights, (re)made after every update for the convs that run on
        # the split-bf16 MFMA path (large problems only; decided by the library from the activation shape)
        self.wq_planes = self.wd_planes = None
        self.lin_fwd = self.lin_bwd = None     # fragment-ordered fp16 planes of a Linear for rdo_linear_h2 (large token matrices)

    def lin_planes(self, fwd: bool):
        """Planes of this Linear's soft weight (fwd) / of its transpose (input gradient) for rdo_linear_h2: allocated on first use -- while
        the plan is being recorded --, filled by `refresh_planes` (eagerly after the recording, then inside the plan behind every step)."""
        if self.qm.kind not in ("linear", "gdn") and not (self.qm.kind == "conv" and self.K == 1):
            raise RuntimeError("lin_planes: a Linear, a 1x1 conv or the 1x1 pool of a GDN")
        co, _, _, ci = self.w4
        if fwd:
            if self.lin_fwd is None:
                self.lin_fwd = ops.H2(torch.empty((2, ci // 32, co // 16, 64, 8), device=self.w.device, dtype=torch.int16), self.wscale)
            return self.lin_fwd
        if self.lin_bwd is None:
            self.lin_bwd = ops.H2(torch.empty((2, co // 32, ci // 16, 64, 8), device=self.w.device, dtype=torch.int16), self.wscale)
        return self.lin_bwd

    def enable_planes(self, fwd: bool, dgrad: bool, h2: bool = False):
        """Allocate the plane buffers (called while the plan is being recorded: no kernel may run here; the engine fills them
        once eagerly after recording and re-fills them inside the plan after every AdaRound step).  h2: fp16 two-way planes of
        w * wscale for the plane-input kernels (conv_fwd_h2.hip); else the bf16 three-way planes of the fp32-input kernels."""
        def alloc(like, cur):
            if cur is not None:
                if isinstance(cur, ops.H2) != h2:
                    raise RuntimeError(f"calibration engine: op '{self.name}' needs its weight planes in two formats")
                return cur
            if h2:
                return ops.H2(torch.empty((2,) + tuple(like.shape), device=like.device, dtype=torch.int16), self.wscale)
            return torch.empty((3,) + tuple(like.shape), device=like.device, dtype=torch.int16)
        if fwd:
            self.wq_planes = alloc(self.wq, self.wq_planes)
        if dgrad and self.wd is not None:
            self.wd_planes = alloc(self.wd, self.wd_planes)

    def expand_phase(self):
        """phase weight of a transposed conv (and its bf16 planes) from the current soft weights"""
        ops.tconv_expand(self.wq4(), self.tc_phase, out=self.wp)
        if self.wp_planes is not None:
            ops.split_bf16x3(self.wp, self.wp_planes)
        if self.wp_h2 is not None:
            ops.split_h2_conv(self.wp, planes=self.wp_h2)

    def refresh_planes(self):
        if self.lin_fwd is not None:
            ops.split_h2_linear(self.wq4(), planes=self.lin_fwd)
        if self.lin_bwd is not None:
            ops.split_h2_linear(self.wd4(), planes=self.lin_bwd)
        if self.wp_planes is not None:
            ops.split_bf16x3(self.wp, self.wp_planes)
        if self.wp_h2 is not None:
            ops.split_h2_conv(self.wp, planes=self.wp_h2)
        for planes, w4 in ((self.wq_planes, self.wq4), (self.wd_planes, self.wd4)):
            if isinstance(planes, ops.H2):
                ops.split_h2_conv(w4(), planes=planes)
            elif planes is not None:
                ops.split_bf16x3(w4(), planes)

    def wq4(self):
        return self.wq.reshape(self.w4)

    def wd4(self):
        # [ci][kh][kw][co] for convs, gamma'^T viewed [C][1][1][C]
        co, kh, kw, ci = self.w4
        return self.wd.reshape(ci, kh, kw, co)

    def numel(self):
        return self.w.numel()


class UnitEngine:
    def __init__(self, kind, modules, cache_q, cache_fp, cache_out, *, batch_size, iters, weight=0.01, b_range=(20, 2),
                 warmup=0.2, input_prob=0.5, lr=1e-3, seed=0, idx_table=None, include_act_func=True, group=None,
                 use_graph=True, force_dp_split=False, task_p=2.0, batch_offset=0, dp_overlap=None, fuse_tail=True, batch_step=True, use_h2=True, rd=None):
        if kind not in UNIT_KINDS:
            raise NotImplementedError(f"calibration engine: unit kind '{kind}'")
        for t in (cache_q, cache_fp, cache_out):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4):
                raise RuntimeError("calibration engine: caches must be contiguous fp32 NHWC CUDA tensors")
        self.kind, self.mods = kind, modules
        self.cq, self.cf, self.co = cache_q, cache_fp, cache_out
        self.B, self.iters = int(batch_size), int(iters)
        self.task_p = float(task_p)            # exponent of the task term (main2.py --task_loss); rec_loss is always p = 2
        self.weight, self.input_prob, self.seed = float(weight), float(input_prob), int(seed) & 0xFFFFFFFF
        self.include_act = include_act_func
        self.use_graph = use_graph
        self.batch_offset = int(batch_offset)  # first row of this rank's share of the global mini-batch (QDrop counter)
        self.dp_overlap = None if dp_overlap is None else bool(dp_overlap)
        if os.environ.get("RDO_USE_H2") is not None:
            use_h2 = os.environ["RDO_USE_H2"] != "0"      # A/B switch for whole runs (bench.py)
        self.use_h2 = bool(use_h2)             # big units on H2 tensors (plane-input LDS-DMA GEMM kernels); False: fp32 activations only
        self.P = {}                            # name -> planes of the H2 form of an activation buffer
        self.batch_step = bool(batch_step)     # one AdaRound-step launch per unit (False: one per weight tensor)
        self.fuse_splitk = os.environ.get("RDO_FUSE_SPLITK", "1") != "0"   # split-K conv + unit tail: the conv's second pass inside the tail
        self.fuse_h2_tail = os.environ.get("RDO_H2_TAIL", "1") != "0"       # last conv of a H2 ResidualBlock unit + its tail in one launch
        self.fold_iter = os.environ.get("RDO_FOLD_ITER", "1") != "0"     # iteration-counter hand-over instead of an increment launch
        # round 6: the NEXT iteration's mini-batch is assembled by the AdaRound-step launch (rdo_adaround_step_batch_gather) -- one launch
        # less per iteration, the gather's stream under the step's latency; iteration 0's mini-batch by a stand-alone gather (`_prime`)
        self.fold_gather = os.environ.get("RDO_GATHER_IN_STEP", "1") != "0"
        self._next_gather = None
        # opt-in R + lambda*D task loss (loss_mode='rd'): dict(model=QuantModel, unit=module, cali=calibration images NCHW on the GPU,
        # lmbda=float).  The unit output of every iteration is pushed through the REST of the wrapped model on torch's tape
        # (hipops.autograd) and losses.RateDistortionLoss is differentiated back to it; rec_loss stays the lp term.
        self.rd = rd
        if rd is not None:
            if not include_act_func:
                raise NotImplementedError("calibration engine: loss_mode='rd' substitutes the unit's module OUTPUT (after its fused "
                                          "activation); include_act_func=False optimises the pre-activation and cannot be combined with it")
            fuse_tail = use_h2 = False
            self.use_h2 = False
        self.fuse_tail = bool(fuse_tail)       # False: the separate epilogue / loss / activation-backward kernels (A/B, tests)
        self.dev = cache_q.device
        n = cache_q.shape[0]
        if idx_table is None and self.B > n:
            # fewer cached samples than the batch: the reference's randperm(n)[:batch_size] simply yields all n (layer_opt.py:289)
            self.B = n
        if idx_table is None:
            # same CPU-generator stream as the reference: one torch.randperm(n) per iteration (layer_opt.py:289)
            idx_table = IdxStream.take(n, self.B, self.iters)
        idx_table = torch.as_tensor(idx_table).to(torch.int32)
        if tuple(idx_table.shape) != (self.iters, self.B):
            raise ValueError(f"idx_table must be [{self.iters},{self.B}], got {tuple(idx_table.shape)}")
        if int(idx_table.max()) >= n or int(idx_table.min()) < 0:
            raise ValueError("idx_table refers to images outside the cache")
        self.idx = idx_table.to(self.dev).contiguous()
        self.sched = ops.make_sched(self.iters, warmup, b_range, lr, self.dev)
        # iteration counter and its shadow (rdo_ptq_hip.h, "iteration-counter hand-over"): when the unit's AdaRound step is one batched
        # launch it leaves it + 1 in the shadow and the next gather publishes it -- no separate increment launch
        self._it2 = torch.zeros(2, dtype=torch.int32, device=self.dev)
        self.it, self.it_shadow = self._it2[0:1], self._it2[1:2]
        self.loss_log = torch.zeros(self.iters, L.LOG_SLOTS, device=self.dev)      # kernels spread atomics over the slots
        self.task_log = torch.zeros(self.iters, L.LOG_SLOTS, device=self.dev)      # task term where it is a separate quantity
        self._task_is_rec = False              # True: task == rec on the same tensors, both logged as 2 * rec in loss_log
        self.round_log = torch.zeros(self.iters, L.LOG_SLOTS, device=self.dev)
        self.group = group
        self.world = 1
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
        # the data-parallel op sequence (grad -> all-reduce -> apply) can be forced on a single rank to test it
        self.split = self.world > 1 or force_dp_split
        self.dp_path = None                    # "graph" | "host" once a data-parallel run has started (_run_dp)
        self._dp_graph, self._dp_graph_failed = None, False
        self.dp_fallbacks = 0                  # captures the ranks agreed to drop (this unit then runs the host loop on every rank)
        self._hb_dev = self._hb_host = None    # heartbeat of the captured loop (`_dp_capture`)
        self._hb_issued = 0
        self.rd_path = None                    # "graph" | "host" once an R + lambda*D run has started (_run_rd)
        self._rd_graph, self._rd_graph_failed = None, False
        # this unit's own fp16-overflow word (rdo_h2_bind_flag): raised by every H2 producer of its plans, polled during long runs
        # (pairs: finite magnitude, non-finite mark; pair 0 for the weight planes, one pair per activation tensor kept as planes)
        self._ovf = torch.zeros(2 * self.OVF_PAIRS, dtype=torch.int32, device=self.dev)
        self._ovf_slot = {}                    # plane tensor name -> its pair
        self.h2_restarts = 0                   # times the unit was restarted after an overflow (0 in every measured run so far)
        self._done = 0
        self._ovf_checked = -1                 # position (`_done`) at which the rank-synchronised overflow verdict was last formed
        with ops.h2_flag(self._ovf[0:2]):
            self._build_ops()
            self._alloc()
            self.scales = {}                   # activation buffer name -> power-of-two scale of its H2 planes
            self._amax = {}                    # ... and the magnitude it was derived from
            self._probing = False
            self._probe_scales()
            self._record()
            for op in self.ops.values():
                op.refresh_planes()      # initial soft weights -> bf16 planes (eager, before the first iteration)
            self._prime()                # folded gather: the mini-batch of iteration 0

    # ------------------------------------------------------------------------------------------------------------------
    def _build_ops(self):
        m, k = self.mods, self.kind
        o = OrderedDict()
        if k == "layer":
            o["layer"] = _Op("layer", m["layer"], False)
        elif k == "rb":
            o["conv1"] = _Op("conv1", m["conv1"], False)
            o["conv2"] = _Op("conv2", m["conv2"], True)
            if m.get("skip") is not None:
                o["skip"] = _Op("skip", m["skip"], False)
        elif k == "rbws":
            o["conv1"] = _Op("conv1", m["conv1"], False)
            o["conv2"] = _Op("conv2", m["conv2"], True)
            o["gdn"] = _Op("gdn", m["gdn"], False)
            if m.get("skip") is not None:
                o["skip"] = _Op("skip", m["skip"], False)
        elif k == "rbu":
            o["subpel_conv"] = _Op("subpel_conv", m["subpel_conv"], False)
            o["conv"] = _Op("conv", m["conv"], True)
            o["igdn"] = _Op("igdn", m["igdn"], False)
            o["upsample"] = _Op("upsample", m["upsample"], False)
        for op in o.values():
            if op.need_dgrad and (op.stride != 1 or 2 * op.pad != op.K - 1):
                raise NotImplementedError("calibration engine: dgrad is built for stride-1 'same' convolutions only")
        self.ops = o
        # Data parallel: one flat bucket of d(rec+task)/d(alpha) per unit.  The op whose weight gradient is the LAST kernel of the
        # backward pass (the block's first conv) sits at the end of the bucket: everything in front of it is complete before that
        # wgrad starts, so its all-reduce overlaps the wgrad (two recorded plans, `_split_point`).
        # Only where that last wgrad is long enough to hide a collective behind (>= 64^2 activations at batch 4): on the small units
        # the second all-reduce and the second gradient launch cost more than the overlap saves.
        late = {"rb": "conv1", "rbws": "conv1", "rbu": "subpel_conv"}.get(k)
        overlap = self.dp_overlap
        if overlap is None and late is not None:   # default: by the size of that last weight gradient; True / False force it (tests, A/B)
            lo = o[late]
            ho, wo = self._out_hw(lo, self.cq.shape[1], self.cq.shape[2])
            overlap = 2.0 * self.B * ho * wo * lo.numel() >= self.DP_OVERLAP_MIN_FLOP
        self._late = late if (self.split and overlap and self.rd is None) else None
        if self.split:
            self.bucket = dp.GradBucket(OrderedDict((n, op.numel()) for n, op in o.items()), late=self._late, device=self.dev, group=self.group)
            for n, op in o.items():
                op.dalpha = self.bucket.view(n)

    def _buf(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.float32)

    def _out_hw(self, op, H, W):
        return (H + 2 * op.pad - op.K) // op.stride + 1, (W + 2 * op.pad - op.K) // op.stride + 1

    def _alloc(self):
        B = self.B
        _, H, W, Cin = self.cq.shape
        self.x_in = self._buf(B, H, W, Cin)
        o, t = self.ops, {}
        if self.kind == "layer":
            op = o["layer"]
            if op.tc_phase is not None:
                s_ = op.tc_phase.stride
                Ho, Wo = H * s_, W * s_
                t["yp"] = self._buf(B, H, W, op.wp4[0])
                t["dyp"] = self._buf(B, H, W, op.wp4[0])
            elif op.tconv is not None:
                s_, p_, op_ = op.tconv
                q_ = op.K - 1 - p_
                if q_ < 0:
                    raise NotImplementedError("calibration engine: transposed conv with padding > kernel_size - 1")
                Hu, Wu = (H - 1) * s_ + 1 + 2 * q_ + op_, (W - 1) * s_ + 1 + 2 * q_ + op_
                t["xu"] = self._buf(B, Hu, Wu, Cin)
                self.tc_geom = (s_, q_, Hu, Wu)
                Ho, Wo = Hu - op.K + 1, Wu - op.K + 1
            elif op.is_gdn:
                Ho, Wo = H, W
                t["norm"] = self._buf(B, H, W, Cin)
                t["t"] = self._buf(B, H, W, Cin)
            else:
                Ho, Wo = self._out_hw(op, H, W)
            t["y"] = self._buf(B, Ho, Wo, op.rows)
            t["dy"] = self._buf(B, Ho, Wo, op.rows)
            t["dpre"] = self._buf(B, Ho, Wo, op.rows)
        elif self.kind == "rb":
            C = o["conv1"].rows
            for n_ in ("h1", "pre2", "out", "dout", "dpre2", "dh1"):
                t[n_] = self._buf(B, H, W, C)
            if "skip" in o:
                t["sk"] = self._buf(B, H, W, C)
        elif self.kind == "rbws":
            c1 = o["conv1"]
            Ho, Wo = self._out_hw(c1, H, W)
            C = c1.rows
            for n_ in ("h1", "c2", "norm", "out", "dout", "t", "acc", "dc2", "dh1"):
                t[n_] = self._buf(B, Ho, Wo, C)
            if "skip" in o:
                t["sk"] = self._buf(B, Ho, Wo, C)
        elif self.kind == "rbu":
            sp = o["subpel_conv"]
            C4 = sp.rows
            r = int(self.mods["upscale"])
            C = C4 // (r * r)
            t["sp"] = self._buf(B, H, W, C4)           # lrelu(subpel conv) before the shuffle
            t["h1"] = self._buf(B, H * r, W * r, C)
            t["up"] = self._buf(B, H, W, C4)
            t["ups"] = self._buf(B, H * r, W * r, C)
            for n_ in ("c", "norm", "out", "dout", "t", "acc", "dc", "dh1"):
                t[n_] = self._buf(B, H * r, W * r, C)
            t["dsp"] = self._buf(B, H, W, C4)
            t["dup"] = self._buf(B, H, W, C4)
            self.r = r
        self.t = t
        for op in self.ops.values():
            op.slabs = None     # allocated at record time when the activation shapes are known

    def _slabs(self, op, x_shape):
        ns = ops.wgrad_nsplit(tuple(x_shape), op.w4, op.stride, op.pad)
        op.slabs = self._buf(ns, *op.w4)
        return op.slabs

    # ------------------------------------------------------------------------------------------------------------------
    lin_conv = os.environ.get("RDO_CONV1X1_LIN_H2", "1") != "0"

    def _lin_conv_ok(self, op, x):
        """A 1 x 1 / stride 1 conv over a large pixel matrix is a token-matrix Linear: rdo_linear_h2 (per-pixel dynamic scale, no probed scale)
        instead of the fp32 / split-bf16 conv kernels -- the 192 <-> 96 convs of Cheng2020-attn's attention blocks (BASELINE config 3)."""
        if not (self.lin_conv and not self._probing and op.qm.kind == "conv" and op.K == 1 and op.stride == 1 and op.pad == 0 and op.tconv is None):
            return False
        C = x.shape[-1]
        rows = x.numel() // C
        return rows >= self.LIN_GDN_MIN_ROWS and ops.linear_h2_supported(rows, C, op.w4[0])

    unit1x1 = int(os.environ.get("RDO_UNIT1X1", "2"))       # 0 off, 1 only 16^2 maps and K = 96, 2 wherever the shape is supported (default)

    def _unit1x1_ok(self, op, x):
        """A plain 1 x 1 / stride-1 conv as a LAYER unit with the default objective: rdo_unit1x1 (forward + tail + weight-gradient slabs in
        one launch, exact fp32) -- the 192 <-> 96 convs of Cheng2020-attn's attention blocks (BASELINE config 3)."""
        if not (self.unit1x1 and self.fused and self.include_act and self.kind == "layer" and op.qm.kind == "conv" and op.K == 1
                and op.stride == 1 and op.pad == 0 and op.tconv is None and not op.is_gdn):
            return False
        M, K = x.numel() // x.shape[-1], x.shape[-1]
        # per unit-iteration inside the config-3 schedule (us, three launches -> one): 16^2 maps 31 -> 29 (192 -> 96), 29 -> 22 (96 -> 192),
        # 32 -> 30 (192 -> 192); 64^2 maps 54 -> 47 at K = 96; at K = 192 on the 64^2 maps the first version (serial load / store loops) lost to
        # rdo_linear_h2 + linear_wgrad_h2 (54 -> 58), with its loads batched it wins there too: schedule 9.13 (off) / 8.81 (K = 96 and 16^2
        # only: RDO_UNIT1X1=1) / 8.76 ms per step (everywhere: the default, 2)
        return (self.unit1x1 == 2 or M <= 4096 or K <= 96) and ops.unit1x1_supported(M, K, op.w4[0])

    def _conv(self, op, x, out, epilogue=L.EPI_NONE, aux=None, residual=None, pre=None, square=False, bias=True):
        b = (op.beta if op.is_gdn else op.bias) if bias else None
        if epilogue == L.EPI_NONE and aux is None and residual is None and pre is None and not square and self._lin_conv_ok(op, x):
            return ops.linear_h2(x.view(-1, x.shape[-1]), op.lin_planes(True), b, out=out.view(-1, out.shape[-1]))
        if ops.uses_bf16x6(tuple(x.shape), op.w4, op.stride, op.pad):
            op.enable_planes(True, False)
        return ops.conv2d_fwd(x, op.wq4(), b, op.stride, op.pad, epilogue=epilogue, aux=aux, residual=residual,
                              square_input=square, out=out, pre=pre, wplanes=op.wq_planes)

    def _wgrad(self, op, x, dy, square=False):
        if op.slabs is None:
            self._slabs(op, x.shape)
        ops.conv2d_wgrad(x, dy, op.w4, op.stride, op.pad, square_input=square, slabs=op.slabs)

    def _dgrad(self, op, dy, out, epilogue=L.EPI_NONE, aux=None):
        if ops.uses_bf16x6(tuple(dy.shape), tuple(op.wd4().shape), 1, op.K - 1 - op.pad):
            op.enable_planes(False, True)
        return ops.conv2d_fwd(dy, op.wd4(), None, 1, op.K - 1 - op.pad, epilogue=epilogue, aux=aux, out=out,
                              wplanes=op.wd_planes)

    # ---- the two 1x1 GEMMs of a GDN / IGDN (norm pool beta' + gamma' . x^2 and, backward, t . gamma'): token-matrix Linears over the
    # channels.  From LIN_GDN_MIN_ROWS pixels on they run on rdo_linear_h2 (per-pixel dynamic scale, three fp16 products) instead of the
    # conv kernels (split-bf16 six products at 128^2, fp32 MFMA tiles below).  RDO_GDN_LIN_H2=0 switches it off.
    LIN_GDN_MIN_ROWS = int(os.environ.get("RDO_GDN_LIN_MIN_ROWS", 4096))
    lin_gdn = os.environ.get("RDO_GDN_LIN_H2", "1") != "0"

    def _gdn_lin_ok(self, g, x):
        C = x.shape[-1]
        rows = x.numel() // C
        return (self.lin_gdn and not self._probing and rows >= self.LIN_GDN_MIN_ROWS and g.w4[0] == C and ops.linear_h2_supported(rows, C, C))

    def _gdn_pool(self, g, x, norm):
        """norm = beta' + gamma' . x^2"""
        if self._gdn_lin_ok(g, x):
            C = x.shape[-1]
            return ops.linear_h2(x.view(-1, C), g.lin_planes(True), g.beta, out=norm.view(-1, C), square_input=True)
        return self._conv(g, x, norm, square=True)

    def _gdn_acc(self, g, tbuf, acc):
        """acc = t . gamma'   (wd = gamma'^T as [C][1][1][C])"""
        if self._gdn_lin_ok(g, tbuf):
            C = tbuf.shape[-1]
            return ops.linear_h2(tbuf.view(-1, C), g.lin_planes(False), None, out=acc.view(-1, C))
        if ops.uses_bf16x6(tuple(tbuf.shape), tuple(g.wd4().shape), 1, 0):
            g.enable_planes(False, True)
        return ops.conv2d_fwd(tbuf, g.wd4(), None, 1, 0, out=acc, wplanes=g.wd_planes)

    def _tconv_forward(self, op, x, y, epilogue=L.EPI_NONE):
        """y [B, sH, sW, Cout] <- (activation of) the transposed conv of x with the soft weights: stride-1 conv with the phase weight,
        then the pixel shuffle (LeakyReLU / ReLU commute with it)."""
        ph = op.tc_phase
        if ops.uses_bf16x6(tuple(x.shape), op.wp4, 1, ph.pad) and op.wp_planes is None:
            op.wp_planes = torch.empty((3,) + tuple(op.wp4), device=self.dev, dtype=torch.int16)
        ops.conv2d_fwd(x, op.wp, op.bias_p, 1, ph.pad, epilogue=epilogue, out=self.t["yp"], wplanes=op.wp_planes)
        self._shuffle(self.t["yp"], ph.stride, y)

    def _tconv_wgrad(self, op, x, dy):
        """slabs of dL/d(kernel-layout weight) from dy [B, sH, sW, Cout]: unshuffle, weight gradient of the phase conv, fold"""
        ph = op.tc_phase
        self._unshuffle(dy, ph.stride, self.t["dyp"])
        if op.slabs is None:
            ns = ops.wgrad_nsplit(tuple(x.shape), op.wp4, 1, ph.pad)
            op.slabs_p = self._buf(ns, *op.wp4)
            op.slabs = self._buf(ns, *op.w4)
        ops.conv2d_wgrad(x, self.t["dyp"], op.wp4, 1, ph.pad, slabs=op.slabs_p)
        ops.tconv_fold(op.slabs_p, ph, op.rows, op.w4[3], out=op.slabs)

    def _after_step(self, lin_done=False):
        """recorded right behind the AdaRound step: phase weights of transposed convs follow the new soft weights (lin_done: the batched
        step has written the rdo_linear_h2 planes itself)"""
        for op in self.ops.values():
            if op.tc_phase is not None:
                op.expand_phase()
            if lin_done:
                continue
            if op.lin_fwd is not None:                     # Linears on rdo_linear_h2: fragment-ordered planes of the new soft weights
                ops.split_h2_linear(op.wq4(), planes=op.lin_fwd)
            if op.lin_bwd is not None:
                ops.split_h2_linear(op.wd4(), planes=op.lin_bwd)

    def _loss(self, pred, grad):
        if self.rd is not None:
            # rec_loss here; the task term is the rate-distortion loss of the whole model, evaluated on the host's tape between the
            # two recorded plans and added to the gradient at the top of the second one
            ops.lp2_loss_grad(pred, self.co, self.idx, self.it, 1.0, grad, self.loss_log)
            self._rd_pred = pred
            self.g_task = torch.zeros_like(pred)
            self._rec_ctx.__exit__(None, None, None)
            self.plan_rd = Plan()
            self._rec_ctx = self.plan_rd.record()
            self._rec_ctx.__enter__()
            ops.add(grad, self.g_task, out=grad)
            return
        # rec_loss + task_loss on the same tensors (fp_out is the identity for these coders)
        if self.task_p == 2.0:
            self._task_is_rec = True
            ops.lp2_loss_grad(pred, self.co, self.idx, self.it, 2.0, grad, self.loss_log)
        else:
            ops.lp_loss_grad(pred, self.co, self.idx, self.it, 1.0, 1.0, self.task_p, grad, self.loss_log, self.task_log)

    @property
    def fused(self):
        """Fused unit tails (csrc/fused_tail.hip) cover the reference's default objective: rec_loss + task_loss with exponent 2."""
        return self.task_p == 2.0 and self.fuse_tail

    def _shuffle(self, x, r, out):
        if r == 2 and x.shape[-1] % 16 == 0:
            return ops.pixel_shuffle_h2(x, out=out)
        return ops.pixel_shuffle(x, r, out)

    def _unshuffle(self, x, r, out):
        if r == 2 and x.shape[-1] % 4 == 0:
            return ops.pixel_unshuffle2(x, out)
        return ops.pixel_unshuffle(x, r, out)

    def _tail_act(self, pre, res, act, dpre, gout=None):
        """out = act(pre) + res, rec + task loss against the cached FP output, dL/dout (if wanted) and dL/dpre in one pass."""
        self._task_is_rec = True
        ops.loss_act_bwd(pre, res, self.co, self.idx, self.it, 2.0, act, self.loss_log, grad_out=gout, dpre=dpre)

    def _conv_tail(self, op, x, pre, res, act, dpre, gout=None):
        """Last conv of a unit + its fused tail.  When the conv is split over K, its second pass (sum the partial slabs, add the
        bias) moves into the tail's first load: one launch and one round trip of the pre-activation tensor less."""
        if self._lin_conv_ok(op, x):                           # a 1 x 1 conv over a large pixel matrix: rdo_linear_h2, then the tail
            self._conv(op, x, pre)
            self._tail_act(pre, res, act, dpre, gout=gout)
            return
        if ops.uses_bf16x6(tuple(x.shape), op.w4, op.stride, op.pad):
            op.enable_planes(True, False)
        ks, _ = ops.conv_fwd_ksplit(tuple(x.shape), op.w4, op.stride, op.pad, op.wq_planes is not None, self.dev)
        if self.fuse_splitk and ks >= 2 and not op.is_gdn and pre.shape[-1] % 4 == 0:      # quads of channels (bias, partial sums)
            ws, ks = ops.conv2d_fwd_partials(x, op.wq4(), op.stride, op.pad, wplanes=op.wq_planes)
            self._task_is_rec = True
            ops.loss_act_bwd_splitk(ws, ks, op.bias, tuple(pre.shape), res, self.co, self.idx, self.it, 2.0, act, self.loss_log,
                                    grad_out=gout, dpre=dpre)
            return
        self._conv(op, x, pre)
        self._tail_act(pre, res, act, dpre, gout=gout)

    def _tail_gdn(self, x, norm, res, inverse, gout, tbuf):
        self._task_is_rec = True
        ops.loss_gdn_bwd(x, norm, res, self.co, self.idx, self.it, 2.0, inverse, self.loss_log, gout, t=tbuf)

    # ------------------------------------------------------------------------------------------------------------------ H2 path
    # FLOPs of the unit's last weight gradient from which the bucket all-reduce is split in two (see _build_ops): 43.5 GFLOP (~200 us)
    # for the 3x3 convs of the 128^2 units hides a collective; 10.9 GFLOP (~45 us, the 64^2 units) is about the latency of the second
    # all-reduce the split adds, and the 3 -> 192 stem of g_a.0 is 26 us
    DP_OVERLAP_MIN_FLOP = 30e9
    # plane-input kernels from this many output elements of the conv (pixels x channels) on
    H2_MIN_OUT = int(os.environ.get("RDO_H2_MIN_OUT", 4096 * 192))
    h2_lean = os.environ.get("RDO_H2_LEAN", "1") != "0"     # tensors whose only readers take planes are not also written as fp32
    layer_h2 = os.environ.get("RDO_LAYER_H2", "1") != "0"   # plain k x k layer units over enough pixels on the plane kernels (plan "layer")

    OVF_PAIRS = 8

    def _h2(self, name, like):
        """Planes of activation buffer `name` (H2 form, scale from the probe iterations, its own overflow words)."""
        if name not in self.P:
            k = self._ovf_slot.setdefault(name, 1 + len(self._ovf_slot))
            self.P[name] = ops.h2_empty(like.shape, self.dev, self.scales[name], flag=self._ovf[2 * k:2 * k + 2])
        return self.P[name]

    def _conv_ok_h2(self, op, x_shape, dgrad=False):
        if dgrad:
            w4, stride, pad = tuple(op.wd4().shape), 1, op.K - 1 - op.pad
        else:
            w4, stride, pad = op.w4, op.stride, op.pad
        if not ops.conv_h2_supported(tuple(x_shape), w4, stride, pad):
            return False
        d = ops.conv_desc(tuple(x_shape), w4, stride, pad)
        return d.B * d.Ho * d.Wo * d.Cout >= self.H2_MIN_OUT

    def _conv_h2(self, op, xp, x_shape, out=None, out_planes=None, epilogue=L.EPI_NONE, aux=None, residual=None, pre=None):
        op.enable_planes(True, False, h2=True)
        ops.conv2d_fwd_h2(xp, tuple(x_shape), op.w4, op.wq_planes, op.beta if op.is_gdn else op.bias, op.stride, op.pad,
                          epilogue=epilogue, aux=aux, residual=residual, out=out, pre=pre, out_planes=out_planes)

    def _dgrad_h2(self, op, dyp, dy_shape, out=None, out_planes=None, epilogue=L.EPI_NONE, aux=None, aux_planes=None):
        op.enable_planes(False, True, h2=True)
        ops.conv2d_fwd_h2(dyp, tuple(dy_shape), tuple(op.wd4().shape), op.wd_planes, None, 1, op.K - 1 - op.pad, epilogue=epilogue,
                          aux=aux, aux_planes=aux_planes, out=out, out_planes=out_planes)

    def _wgrad_h2(self, op, xp, x_shape, dyp):
        if op.slabs is None:
            self._slabs(op, x_shape)
        ops.conv2d_wgrad_h2(xp, tuple(x_shape), dyp, op.w4, op.stride, op.pad, slabs=op.slabs)

    def _gdn_backward_h2(self, g, dout, xin, norm, tp, acc, dxp):
        """GDN / IGDN backward with t given as planes `tp` (written by the fused tail): acc = t . gamma' on the plane kernel, then
        dx as planes only (its consumers are the plane-input weight gradient and dgrad)."""
        self._dgrad_h2(g, tp, xin.shape, out=acc)
        ops.gdn_bwd_dx_h2(dout, xin, norm, acc, g.inverse, dx_planes=dxp)

    def _plan_h2(self):
        """Which big convs of this unit run on H2 tensors (decided once, at record time)."""
        o, k = self.ops, self.kind
        if not (self.use_h2 and self.fused):
            return None
        xs = tuple(self.x_in.shape)
        if k == "layer":
            # a transposed conv in phase form (stride 2): the 3 x 3 phase conv and its weight gradient on planes
            op = o["layer"]
            if (op.tc_phase is not None and op.tc_phase.stride == 2 and xs[-1] % 16 == 0 and ops.conv_h2_supported(xs, op.wp4, 1, op.tc_phase.pad)
                    and ops.wgrad_h2_supported(xs, op.wp4, 1, op.tc_phase.pad) and xs[0] * xs[1] * xs[2] * op.wp4[0] >= self.H2_MIN_OUT):
                return "tconv"
            # a plain k x k conv (k > 1) over enough pixels -- the 3 x 3 96 -> 96 convs of Cheng2020-attn's attention blocks at 64^2 (BASELINE
            # config 3; fp32 MFMA 50 + 65 us forward + weight gradient, planes 40 + 32: tools/bench_layer96.py): input and dL/dpre as planes
            if (self.layer_h2 and op.tconv is None and not op.is_gdn and op.qm.kind == "conv" and op.K > 1 and xs[-1] % 16 == 0
                    and "dpre" in self.t and self._conv_ok_h2(op, xs) and ops.wgrad_h2_layer_supported(xs, op.w4, op.stride, op.pad)):
                return "layer"
            return None
        if k == "rb" and "skip" not in o:
            c1, c2 = o["conv1"], o["conv2"]
            hs = tuple(self.t["h1"].shape)
            if (self._conv_ok_h2(c1, xs) and self._conv_ok_h2(c2, hs) and self._conv_ok_h2(c2, hs, dgrad=True)
                    and ops.wgrad_h2_supported(xs, c1.w4, c1.stride, c1.pad) and ops.wgrad_h2_supported(hs, c2.w4, 1, c2.pad)):
                return "rb"
        if k in ("rbws", "rbu"):
            cv, g = (o["conv2"], o["gdn"]) if k == "rbws" else (o["conv"], o["igdn"])
            hs = tuple(self.t["h1"].shape)
            if (hs[-1] % 16 == 0 and (k != "rbu" or self.r == 2) and self._conv_ok_h2(cv, hs) and self._conv_ok_h2(cv, hs, dgrad=True)
                    and ops.wgrad_h2_supported(hs, cv.w4, 1, cv.pad)):
                return k
        return None

    def _fb_rb_h2(self):
        o, t, x = self.ops, self.t, self.x_in
        c1, c2 = o["conv1"], o["conv2"]
        xp, h1p = self._h2("x", x), self._h2("h1", t["h1"])
        dp2p, dh1p = self._h2("dpre2", t["h1"]), self._h2("dh1", t["h1"])
        lean = self.h2_lean
        # lean: x and h1 exist as planes only -- the residual add of the tail sums the three planes back (exactly), the LeakyReLU
        # mask of the dgrad epilogue reads the sign off plane 0
        self._gather(None if lean else x, xp)
        self._conv_h2(c1, xp, x.shape, out=None if lean else t["h1"], out_planes=h1p, epilogue=L.EPI_LRELU)
        self._task_is_rec = True
        if self.fuse_h2_tail and ops.conv_h2_tail_supported(tuple(t["h1"].shape), c2.w4, c2.stride, c2.pad):
            # conv2 + tail in one launch: the pre-activation never reaches memory
            c2.enable_planes(True, False, h2=True)
            ops.conv2d_fwd_h2_tail(h1p, tuple(t["h1"].shape), c2.w4, c2.wq_planes, c2.bias, c2.stride, c2.pad, xp, self.co, self.idx, self.it,
                                   2.0, ops.ACT_LRELU, dp2p, self.loss_log)
        else:
            self._conv_h2(c2, h1p, t["h1"].shape, out=t["pre2"])
            ops.loss_act_bwd(t["pre2"], None if lean else x, self.co, self.idx, self.it, 2.0, ops.ACT_LRELU, self.loss_log, dpre_planes=dp2p,
                             residual_planes=xp if lean else None)
        self._wgrad_h2(c2, h1p, t["h1"].shape, dp2p)
        self._dgrad_h2(c2, dp2p, t["h1"].shape, out_planes=dh1p, epilogue=L.EPI_LRELU_BWD, aux=None if lean else t["h1"],
                       aux_planes=h1p if lean else None)
        self._split_point()
        self._wgrad_h2(c1, xp, x.shape, dh1p)

    def _fb_gdn_block_h2(self):
        """RBWS / RBU whose second conv runs on H2 tensors: that conv, its weight gradient and dgrad; where the shapes qualify also the
        gamma'^T GEMM of the GDN backward, the first conv of an RBWS with >= 16 input channels and the two sub-pixel convs of an RBU
        (each with its weight gradient).  What does not qualify (thin stems, small 1x1 GEMMs) stays on fp32 activations."""
        o, t, x = self.ops, self.t, self.x_in
        rbu = self.kind == "rbu"
        xs = tuple(x.shape)
        if rbu:
            sp, cv, g, up = o["subpel_conv"], o["conv"], o["igdn"], o["upsample"]
            cname, dname = "c", "dc"
            # the two 192 -> 768 sub-pixel convs and their weight gradients on planes as well when the shapes qualify
            x_h2 = (x.shape[-1] % 16 == 0 and self._conv_ok_h2(sp, xs) and self._conv_ok_h2(up, xs)
                    and ops.wgrad_h2_supported(xs, sp.w4, sp.stride, sp.pad) and ops.wgrad_h2_supported(xs, up.w4, up.stride, up.pad))
        else:
            c1, cv, g = o["conv1"], o["conv2"], o["gdn"]
            cname, dname = "c2", "dc2"
            x_h2 = x.shape[-1] % 16 == 0 and self._conv_ok_h2(c1, xs) and ops.wgrad_h2_supported(xs, c1.w4, c1.stride, c1.pad)
        hs = tuple(t["h1"].shape)
        # gamma'^T GEMM: on rdo_linear_h2 over the fp32 t where that applies (then t is not written as planes at all), else on the plane kernel
        g_h2 = self._conv_ok_h2(g, hs, dgrad=True) and not self._gdn_lin_ok(g, t["h1"])
        h1p, dcp = self._h2("h1", t["h1"]), self._h2(dname, t["h1"])
        tp = self._h2("t", t["h1"]) if g_h2 else None
        xp = self._h2("x", x) if x_h2 else None
        self._gather(x, xp if x_h2 else None)
        lean_h1 = self.h2_lean and (rbu or x_h2)          # h1 exists as planes only (its LeakyReLU mask is read off plane 0)
        if rbu:
            r = self.r
            if x_h2:
                self._conv_h2(sp, xp, xs, out=t["sp"], epilogue=L.EPI_LRELU)
                self._conv_h2(up, xp, xs, out=t["up"])
            else:
                self._conv(sp, x, t["sp"], epilogue=L.EPI_LRELU)
                self._conv(up, x, t["up"])
            ops.pixel_shuffle_h2(t["sp"], out=None if lean_h1 else t["h1"], out_planes=h1p)
            self._shuffle(t["up"], r, t["ups"])
            res = t["ups"]
        else:
            if x_h2:
                self._conv_h2(c1, xp, xs, out=None if lean_h1 else t["h1"], out_planes=h1p, epilogue=L.EPI_LRELU)
            else:
                self._conv(c1, x, t["h1"], epilogue=L.EPI_LRELU)
                ops.split_h2(t["h1"], h1p)
            res = x
            if "skip" in o:
                self._conv(o["skip"], x, t["sk"])
                res = t["sk"]
        self._conv_h2(cv, h1p, hs, out=t[cname])
        self._gdn_pool(g, t[cname], t["norm"])
        self._task_is_rec = True
        ops.loss_gdn_bwd(t[cname], t["norm"], res, self.co, self.idx, self.it, 2.0, rbu, self.loss_log, t["dout"], t=t["t"], t_planes=tp)
        if rbu:
            if x_h2:
                dupp = self._h2("dup", t["dup"])
                ops.pixel_unshuffle2(t["dout"], out_planes=dupp)
                self._wgrad_h2(up, xp, xs, dupp)
            else:
                self._unshuffle(t["dout"], r, t["dup"])
                self._wgrad(up, x, t["dup"])
        elif "skip" in o:
            self._wgrad(o["skip"], x, t["dout"])
        if g_h2:
            self._gdn_backward_h2(g, t["dout"], t[cname], t["norm"], tp, t["acc"], dcp)
        else:                                             # gamma'^T GEMM on the fp32 t, dx still as planes only
            self._gdn_acc(g, t["t"], t["acc"])
            ops.gdn_bwd_dx_h2(t["dout"], t[cname], t["norm"], t["acc"], g.inverse, dx_planes=dcp)
        self._wgrad(g, t[cname], t["t"], square=True)
        self._wgrad_h2(cv, h1p, hs, dcp)
        rbws_h2 = x_h2 and not rbu
        dh1p = self._h2("dh1", t["h1"]) if rbws_h2 else None
        self._dgrad_h2(cv, dcp, hs, out=None if rbws_h2 else t["dh1"], out_planes=dh1p, epilogue=L.EPI_LRELU_BWD,
                       aux=None if lean_h1 else t["h1"], aux_planes=h1p if lean_h1 else None)
        if rbu:
            if x_h2:
                dspp = self._h2("dsp", t["dsp"])
                ops.pixel_unshuffle2(t["dh1"], out_planes=dspp)
                self._split_point()
                self._wgrad_h2(sp, xp, xs, dspp)
            else:
                self._unshuffle(t["dh1"], r, t["dsp"])
                self._split_point()
                self._wgrad(sp, x, t["dsp"])
        else:
            self._split_point()
            if x_h2:
                self._wgrad_h2(c1, xp, xs, dh1p)
            else:
                self._wgrad(c1, x, t["dh1"])

    # activation buffers a plan may keep as planes (all probed: a scale that is never used costs nothing)
    H2_PROBED = {"tconv": ("x", "dpre"), "layer": ("x", "dpre"), "rb": ("x", "h1", "dpre2", "dh1"), "rbws": ("x", "h1", "t", "dc2", "dh1"), "rbu": ("x", "h1", "t", "dc", "dup", "dsp")}

    def _probe_amax(self, names):
        """One eager iteration of the unit on fp32 activations and the plain fp32 kernels (no planes, no AdaRound step) at the CURRENT
        weights and iteration counter -> largest magnitude of every tensor the plan keeps as planes (CPU tensor)."""
        saved = ops.set_tuning("conv_x6", 0)
        self._probing = True
        try:
            self._forward_backward()
        finally:
            self._probing = False
            ops.set_tuning("conv_x6", saved)
        return torch.stack([(self.x_in if n == "x" else self.t[n]).abs().max() for n in names]).cpu()

    def _set_weights(self, soft):
        """soft-rounded (the state the loop runs in) or hard-rounded weights in every op's wq / wd (set-up only)"""
        for op in self.ops.values():
            ops.adaround_fwd(op.desc, op.w, op.alpha, op.delta, op.zp, bool(soft), op.wq, op.wd)
            if op.tc_phase is not None:
                op.expand_phase()

    def _probe_scales(self):
        """fp16 planes need a per-tensor power-of-two scale (include/rdo_ptq_hip.h, "H2 tensors"), fixed before the plan is recorded.
        TWO probe iterations bracket what the run will see: one at the initial soft weights (= the full-precision weights: the
        reconstruction error is whatever the quantised input alone causes -- nothing at all for the first unit of a model) and one at
        the HARD-rounded weights (every weight up to delta / 2 off: the error level the soft weights move towards while b decays --
        the practical upper end of the run's gradients).  The scale puts the larger of the two magnitudes at 2^7: x512 before fp16
        overflows, and fp32-chain accuracy down to 2^-2, i.e. for tensors up to 512 x smaller than probed (tools/f16_probe.hip).
        A tensor that is exactly zero in both probes gives no scale: the unit then runs on fp32 activations.  Should a run outgrow
        its scales all the same, `_recover` restarts the unit (scales from the recorded magnitudes, then fp32 activations): loud, never a lost run."""
        plan = self._plan_h2()
        if plan is None:
            return
        names = self.H2_PROBED[plan]
        amax = self._probe_amax(names)
        self._set_weights(soft=False)
        try:
            amax = torch.maximum(amax, self._probe_amax(names))
        finally:
            self._set_weights(soft=True)
        self._clear_probe()
        if self.world > 1:
            # every rank probes its own shard: agree on the magnitudes (MAX) so that all ranks record the same scales, take the same
            # "all-zero tensor" decision below and -- after an overflow, whose words are MAX-reduced too -- re-derive the same new scales
            a = amax.to(self.dev)
            a = torch.where(torch.isfinite(a), a, torch.full_like(a, float("inf")))
            torch.distributed.all_reduce(a, op=torch.distributed.ReduceOp.MAX, group=self.group)
            amax = a.cpu()
        if not torch.isfinite(amax).all():
            raise RuntimeError("calibration engine: non-finite activations in the probe iteration")
        if bool((amax <= 0).any()):
            import logging
            logging.getLogger("rdo_ptq.engine").warning(
                "unit with an all-zero tensor in both probe iterations (%s): no fp16 plane scale can be derived, running it on fp32 activations",
                [n for n, a in zip(names, amax) if float(a) <= 0])
            self.use_h2 = False
            return
        self._amax = {n: float(a) for n, a in zip(names, amax)}
        self.scales = {n: ops.pow2_scale(a) for n, a in self._amax.items()}

    def _clear_probe(self):
        """a probe leaves no trace: logs, counters, gradient slabs (the recorded kernels may want another split count)"""
        self.loss_log.zero_(); self.task_log.zero_(); self.round_log.zero_(); self._it2.zero_()
        for op in self.ops.values():
            op.slabs = None

    # ---- fp16 range: poll, restart ------------------------------------------------------------------------------------------------
    H2_POLL = int(os.environ.get("RDO_H2_POLL", 512))        # iterations between two reads of the overflow word inside one run() call

    H2_RESTARTS = 2                                           # restarts on re-scaled planes before the unit goes to fp32 activations

    def _overflowed(self):
        """Has an H2 producer of this unit met a value outside fp16's range?  (A 64-byte read: synchronises.)  Data parallel: MAX over
        the ranks -- every rank restarts, with the same new scales, or none does."""
        if not self.P:
            return False
        v = self._ovf.clone()
        if self.world > 1:
            torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX, group=self.group)
        self._ovf_seen = v.cpu().tolist()
        return any(self._ovf_seen)

    def _recover(self):
        """A value left the fp16 range of its planes: the iterations since the unit's start are invalid.  Restart the unit from its
        initial state (alpha from the weights, zero Adam moments, iteration 0: the index table and the counter-RNG masks make the
        re-run the same run).  The overflow words say what happened: a tensor's word 0 is the largest finite |x s| that did not fit
        -> its magnitude is now known exactly and its scale is re-derived from it with x8 head-room; a tensor that only met
        inf / NaN sits downstream of such an overflow and is re-scaled by the largest growth seen in the unit.  After H2_RESTARTS such
        restarts, or when nothing finite was recorded (the data themselves are not finite), the unit re-runs on fp32 activations
        (`use_h2 = False`: the split-bf16 / fp32-MFMA kernels cannot overflow).  The reference's arithmetic is plain fp32
        (quant_layer.py:123)."""
        import logging
        lg = logging.getLogger("rdo_ptq.engine")
        seen = self._ovf_seen
        self.h2_restarts += 1
        grown = {}                                            # tensor -> observed magnitude / the magnitude its scale was made for
        for name, k in self._ovf_slot.items():
            if seen[2 * k]:
                grown[name] = ops.overflow_magnitude(seen[2 * k]) / self.scales[name] / self._amax[name]
        if self.h2_restarts <= self.H2_RESTARTS and grown and not (seen[0] or seen[1]):
            r = max(grown.values())
            for name, k in self._ovf_slot.items():
                if name in grown or seen[2 * k + 1]:
                    self._amax[name] *= 8.0 * grown.get(name, r)
            self.scales = {n: ops.pow2_scale(a) for n, a in self._amax.items()}
            lg.warning("calibration unit outgrew the fp16 range of its H2 planes within %d iterations (%s): restarting it with scales %s",
                       self._done, {n: f"x{g:.3g}" for n, g in grown.items()}, self.scales)
        else:
            self.use_h2 = False
            lg.warning("calibration unit left the fp16 range of its H2 planes%s: restarting it on fp32 activations",
                       " again" if self.h2_restarts > 1 else " with non-finite values")
        for op in self.ops.values():
            ops.adaround_init_alpha(op.desc, op.w, op.delta, op.alpha)
            op.m.zero_(); op.v.zero_()
            op.wq_planes = op.wd_planes = op.wp_h2 = op.wp_planes = op.lin_fwd = op.lin_bwd = None
        self._set_weights(soft=True)
        self.P, self._ovf_slot = {}, {}
        self._clear_probe()
        self._ovf.zero_()
        self._done = 0
        self._ovf_checked = -1
        self._dp_graph = self._rd_graph = None
        self._record()
        for op in self.ops.values():
            op.refresh_planes()
        self._prime()

    def _check_overflow(self):
        """Where results leave the engine: an overflow not yet seen by the polls of `run` is handled here.  COLLECTIVE under data
        parallelism (the MAX all-reduce of `_overflowed`), so it must only be reached at rank-invariant points: it is run once per
        position of the unit (`sync_overflow`, called by recon.py behind `run()` on every rank, and by `finish`) and the verdict is
        cached -- `logs` / `logs_terms` called afterwards, on any subset of the ranks, issue no overflow collective.
        Restart ladder (`_recover`): H2_RESTARTS restarts on re-derived scales, then one on fp32 activations (`use_h2 = False`, no
        planes left, `_overflowed` is False by construction).  An overflow word raised with no plane tensor alive would be a library
        fault."""
        self.dp_drain()
        if self._ovf_checked == self._done:
            return
        while self._overflowed():
            if not self.use_h2:
                raise RuntimeError("calibration engine: an H2 overflow word is raised on the fp32-activation path (library fault)")
            target = self._done
            with ops.h2_flag(self._ovf[0:2]):
                self._recover()
            self.run(target)
            torch.cuda.synchronize()
        self._ovf_checked = self._done

    def sync_overflow(self):
        """The rank-synchronised fp16-range verdict for the iterations run so far (see `_check_overflow`): call it on EVERY rank at the
        same point of the schedule before any rank-dependent use of `logs*()`."""
        self._check_overflow()

    def _forward_backward(self):
        o, t, x = self.ops, self.t, self.x_in
        self.h2_plan = None if self._probing else self._plan_h2()
        if self.h2_plan == "rb":
            return self._fb_rb_h2()
        if self.h2_plan in ("rbws", "rbu"):
            return self._fb_gdn_block_h2()
        if self.h2_plan not in ("tconv", "layer"):            # (those plans gather straight into planes)
            self._gather(x)
        if self.kind == "layer" and o["layer"].is_gdn:
            # a GDN / IGDN that is its own unit (sequential Minnen2018-style coders): only gamma is trained, no dx needed
            op = o["layer"]
            if self.fused:
                self._gdn_pool(op, x, t["norm"])                                                 # norm pool only
                self._tail_gdn(x, t["norm"], None, op.inverse, t["dy"], t["t"])
            else:
                self._conv(op, x, t["y"], epilogue=L.EPI_IGDN if op.inverse else L.EPI_GDN, aux=x, pre=t["norm"], square=True)
                self._loss(t["y"], t["dy"])
                ops.gdn_bwd_t(t["dy"], x, t["norm"], op.inverse, t["t"])
            self._wgrad(op, x, t["t"], square=True)
        elif self.kind == "layer":
            op = o["layer"]
            if op.tconv is not None and op.tc_phase is None:
                s_, q_, Hu, Wu = self.tc_geom
                x = ops.zero_insert(x, s_, q_, q_, Hu, Wu, out=t["xu"])
            epi = op.qm.fused_epilogue() if self.include_act else None
            if epi is None and self.include_act and type(op.qm.activation_function).__name__ != "StraightThrough":
                raise NotImplementedError("calibration engine: only LeakyReLU(0.01) or ReLU may be fused into a layer unit")
            act = {None: ops.ACT_NONE, L.EPI_LRELU: ops.ACT_LRELU, L.EPI_RELU: ops.ACT_RELU}[epi]
            if self.h2_plan == "layer":                      # conv, tail and weight gradient on H2 tensors (x and dL/dpre exist as planes only)
                xs = tuple(x.shape)
                xp, dpp = self._h2("x", x), self._h2("dpre", t["dpre"])
                self._gather(None, xp)
                self._conv_h2(op, xp, xs, out=t["y"])                                               # pre-activation
                self._task_is_rec = True
                ops.loss_act_bwd(t["y"], None, self.co, self.idx, self.it, 2.0, act, self.loss_log, dpre_planes=dpp)
                self._wgrad_h2(op, xp, xs, dpp)
            elif self.h2_plan == "tconv":                    # ... with the phase conv and its weight gradient on H2 tensors
                ph, xs = op.tc_phase, tuple(x.shape)
                xp, dypp = self._h2("x", x), self._h2("dpre", t["dyp"])
                if op.wp_h2 is None:
                    op.wp_h2 = ops.H2(torch.empty((2,) + tuple(op.wp4), device=self.dev, dtype=torch.int16), op.wscale)
                self._gather(None, xp)
                ops.conv2d_fwd_h2(xp, xs, op.wp4, op.wp_h2, op.bias_p, 1, ph.pad, out=t["yp"])
                self._shuffle(t["yp"], 2, t["y"])
                self._tail_act(t["y"], None, act, t["dpre"])
                ops.pixel_unshuffle2(t["dpre"], out_planes=dypp)
                if op.slabs is None:
                    ns = ops.wgrad_nsplit(xs, op.wp4, 1, ph.pad)
                    op.slabs_p = self._buf(ns, *op.wp4)
                    op.slabs = self._buf(ns, *op.w4)
                ops.conv2d_wgrad_h2(xp, xs, dypp, op.wp4, 1, ph.pad, slabs=op.slabs_p)
                ops.tconv_fold(op.slabs_p, ph, op.rows, op.w4[3], out=op.slabs)
            elif op.tc_phase is not None:                    # transposed conv without zero insertion
                if self.fused:
                    self._tconv_forward(op, x, t["y"])                                             # pre-activation
                    self._tail_act(t["y"], None, act, t["dpre"])
                    self._tconv_wgrad(op, x, t["dpre"])
                else:
                    self._tconv_forward(op, x, t["y"], epilogue=L.EPI_NONE if epi is None else epi)
                    self._loss(t["y"], t["dy"])
                    g = t["dy"]
                    if epi is not None:
                        (ops.lrelu_bwd if epi == L.EPI_LRELU else ops.relu_bwd)(t["dy"], t["y"], t["dpre"])
                        g = t["dpre"]
                    self._tconv_wgrad(op, x, g)
            elif self._unit1x1_ok(op, x):                    # a 1 x 1 conv: forward, tail and weight-gradient slabs in one launch
                if op.slabs is None:
                    op.slabs = self._buf(ops.unit1x1_nslab(x.numel() // x.shape[-1], op.w4[0]), *op.w4)
                self._task_is_rec = True
                ops.unit1x1(x, op.wq4(), op.bias, self.co, self.idx, self.it, 2.0, act, self.loss_log, op.slabs)
            elif self.fused:
                self._conv_tail(op, x, t["y"], None, act, t["dpre"])                              # t["y"]: pre-activation, if it is stored
                self._wgrad(op, x, t["dpre"])
            elif epi is not None:
                self._conv(op, x, t["y"], epilogue=epi)
                self._loss(t["y"], t["dy"])
                (ops.lrelu_bwd if epi == L.EPI_LRELU else ops.relu_bwd)(t["dy"], t["y"], t["dpre"])
                self._wgrad(op, x, t["dpre"])
            else:
                self._conv(op, x, t["y"])
                self._loss(t["y"], t["dy"])
                self._wgrad(op, x, t["dy"])
        elif self.kind == "rb":
            c1, c2 = o["conv1"], o["conv2"]
            self._conv(c1, x, t["h1"], epilogue=L.EPI_LRELU)
            res = x
            if "skip" in o:
                self._conv(o["skip"], x, t["sk"])
                res = t["sk"]
            if self.fused:
                self._conv_tail(c2, t["h1"], t["pre2"], res, ops.ACT_LRELU, t["dpre2"], gout=t["dout"] if "skip" in o else None)
            else:
                self._conv(c2, t["h1"], t["out"], epilogue=L.EPI_LRELU, residual=res, pre=t["pre2"])
                self._loss(t["out"], t["dout"])
                ops.lrelu_bwd(t["dout"], t["pre2"], t["dpre2"])
            if "skip" in o:
                self._wgrad(o["skip"], x, t["dout"])
            self._wgrad(c2, t["h1"], t["dpre2"])
            self._dgrad(c2, t["dpre2"], t["dh1"], epilogue=L.EPI_LRELU_BWD, aux=t["h1"])
            self._split_point()
            self._wgrad(c1, x, t["dh1"])
        elif self.kind == "rbws":
            c1, c2, g = o["conv1"], o["conv2"], o["gdn"]
            self._conv(c1, x, t["h1"], epilogue=L.EPI_LRELU)
            self._conv(c2, t["h1"], t["c2"])
            res = x
            if "skip" in o:
                self._conv(o["skip"], x, t["sk"])
                res = t["sk"]
            if self.fused:
                self._gdn_pool(g, t["c2"], t["norm"])
                self._tail_gdn(t["c2"], t["norm"], res, False, t["dout"], t["t"])
            else:
                self._conv(g, t["c2"], t["out"], epilogue=L.EPI_GDN, aux=t["c2"], residual=res, pre=t["norm"], square=True)
                self._loss(t["out"], t["dout"])
            if "skip" in o:
                self._wgrad(o["skip"], x, t["dout"])
            self._gdn_backward(g, t["dout"], t["c2"], t["norm"], t["t"], t["acc"], t["dc2"], inverse=False)
            self._wgrad(c2, t["h1"], t["dc2"])
            self._dgrad(c2, t["dc2"], t["dh1"], epilogue=L.EPI_LRELU_BWD, aux=t["h1"])
            self._split_point()
            self._wgrad(c1, x, t["dh1"])
        elif self.kind == "rbu":
            sp, cv, g, up = o["subpel_conv"], o["conv"], o["igdn"], o["upsample"]
            r = self.r
            self._conv(sp, x, t["sp"], epilogue=L.EPI_LRELU)       # LeakyReLU commutes with the pixel shuffle
            self._shuffle(t["sp"], r, t["h1"])
            self._conv(cv, t["h1"], t["c"])
            self._conv(up, x, t["up"])
            self._shuffle(t["up"], r, t["ups"])
            if self.fused:
                self._gdn_pool(g, t["c"], t["norm"])
                self._tail_gdn(t["c"], t["norm"], t["ups"], True, t["dout"], t["t"])
            else:
                self._conv(g, t["c"], t["out"], epilogue=L.EPI_IGDN, aux=t["c"], residual=t["ups"], pre=t["norm"], square=True)
                self._loss(t["out"], t["dout"])
            self._unshuffle(t["dout"], r, t["dup"])
            self._wgrad(up, x, t["dup"])
            self._gdn_backward(g, t["dout"], t["c"], t["norm"], t["t"], t["acc"], t["dc"], inverse=True)
            self._wgrad(cv, t["h1"], t["dc"])
            self._dgrad(cv, t["dc"], t["dh1"], epilogue=L.EPI_LRELU_BWD, aux=t["h1"])
            self._unshuffle(t["dh1"], r, t["dsp"])
            self._split_point()
            self._wgrad(sp, x, t["dsp"])

    def _gdn_backward(self, g, dout, xin, norm, tbuf, acc, dx, inverse):
        if not self.fused:                                           # the fused tail has already written t
            ops.gdn_bwd_t(dout, xin, norm, inverse, tbuf)
        self._gdn_acc(g, tbuf, acc)                                   # t . gamma'
        if xin.numel() % 4 == 0:
            ops.gdn_bwd_dx_h2(dout, xin, norm, acc, inverse, dx=dx)   # 16-byte accesses
        else:
            ops.gdn_bwd_dx(dout, xin, norm, acc, inverse, dx)
        self._wgrad(g, xin, tbuf, square=True)                        # dgamma'[k][i] = sum_m t_k x_i^2

    def _items(self, opl):
        # (lin_fwd / lin_bwd: the fragment-ordered planes of a Linear / GDN gamma on rdo_linear_h2 -- written by the step's own launch)
        return [dict(d=op.desc, w=op.w, delta=op.delta, zp=op.zp, slabs=op.slabs, alpha=op.alpha, m=op.m, v=op.v, wq=op.wq, wd=op.wd,
                     wq_planes=op.wq_planes, wd_planes=op.wd_planes, dalpha=getattr(op, "dalpha", None),
                     lin_fwd=op.lin_fwd, lin_bwd=op.lin_bwd) for op in opl]

    STEP_BATCH = 8          # weight tensors per rdo_adaround_step_batch launch (kMaxBatch of adaround.hip)

    def _batchable(self, opl):
        """The tensors can go through rdo_adaround_step_batch: in ONE launch up to STEP_BATCH of them, else in ceil(n / STEP_BATCH) launches
        (an RSTB with six blocks has 37 trainable tensors: 37 single-tensor steps, 24 transposes and 48 plane splits per iteration before)."""
        return self.batch_step and len(opl) >= 1 and all(op.numel() % 4 == 0 for op in opl)

    def _step_batches(self, opl, scale, mode):
        """The batched step (mode 0) / update (mode 2) of `opl`.  The iteration counter moves by hand-over (the FIRST launch leaves
        it + 1 in the shadow word -- every launch of the step only reads the counter --, the next gather publishes it) or, with the
        hand-over switched off, with the last launch."""
        for i in range(0, len(opl), self.STEP_BATCH):
            last = i + self.STEP_BATCH >= len(opl)
            if self._folded:
                # every launch of the step reads the published word; the LAST one carries the next mini-batch and moves the real counter
                ops.adaround_step_batch(self._items(opl[i:i + self.STEP_BATCH]), scale, self.weight, self.sched, self.it_shadow, self.round_log, mode=mode,
                                        iter_shadow=self.it if last else None, gather=self._next_gather if last else None)
            elif self._handover:
                ops.adaround_step_batch(self._items(opl[i:i + self.STEP_BATCH]), scale, self.weight, self.sched, self.it, self.round_log, mode=mode,
                                        iter_shadow=self.it_shadow if i == 0 else None)
            else:
                ops.adaround_step_batch(self._items(opl[i:i + self.STEP_BATCH]), scale, self.weight, self.sched, self.it, self.round_log, mode=mode,
                                        advance_iter=self.it if last else None)

    @property
    def _handover(self):
        """The counter hand-over needs the unit's step as batched launches (it is the first of them that fills the shadow)."""
        return self.fold_iter and self._batchable(list(self.ops.values()))

    @property
    def _folded(self):
        """The step launch of iteration i assembles the mini-batch of iteration i + 1.  Counter words then: the loss / tail launch reads
        the real counter `it` and publishes it into `it_shadow` (ops.iter_bind_publish); the step reads `it_shadow` and its first thread
        stores it + 1 into `it` -- nobody reads a word that another thread of the same launch writes (include/rdo_ptq_hip.h)."""
        return self.fold_gather and self._handover and self.rd is None

    def _it_src(self):
        return self.it_shadow if (self._handover and not self._folded) else self.it

    def _it_pub(self):
        return self.it if (self._handover and not self._folded) else None

    def _gather(self, x, xp=None):
        """The unit's mini-batch for the current iteration: x_q / x_fp rows mixed by the QDrop mask into `x` (fp32, may be None with
        planes) and / or the planes `xp`.  Recorded as the iteration's first kernel -- or, folded, handed to the step launch of the
        PREVIOUS iteration (`_step_batches`) with `_prime` covering iteration 0."""
        if self._folded and not self._probing:
            self._next_gather = dict(cache_q=self.cq, cache_fp=self.cf, idx_table=self.idx, B=self.B, batch_offset=self.batch_offset,
                                     prob=self.input_prob, seed=self.seed, out=x, out_planes=xp)
            return
        if xp is not None:
            ops.gather_qdrop_h2(self.cq, self.cf, self.idx, self._it_src(), self.B, self.input_prob, self.seed, x, xp, self.batch_offset,
                                iter_publish=self._it_pub())
        else:
            ops.gather_qdrop(self.cq, self.cf, self.idx, self._it_src(), self.B, self.input_prob, self.seed, x, self.batch_offset,
                             iter_publish=self._it_pub())

    def _prime(self):
        """Folded gather: the mini-batch of the iteration the counter stands at, assembled eagerly (before the first iteration of a
        run and after a restart; every later one comes from the previous iteration's step launch)."""
        g = self._next_gather
        if g is None:
            return
        self.it_shadow.copy_(self.it)
        if g["out_planes"] is not None:
            ops.gather_qdrop_h2(self.cq, self.cf, self.idx, self.it, self.B, self.input_prob, self.seed, g["out"], g["out_planes"], self.batch_offset)
        else:
            ops.gather_qdrop(self.cq, self.cf, self.idx, self.it, self.B, self.input_prob, self.seed, g["out"], self.batch_offset)

    def _grad_ops(self, names):
        opl = [self.ops[n] for n in names]
        if self._batchable(opl):
            for i in range(0, len(opl), self.STEP_BATCH):
                ops.adaround_step_batch(self._items(opl[i:i + self.STEP_BATCH]), 1.0, self.weight, self.sched, self.it, self.round_log, mode=1)
            return
        for op in opl:
            ops.adaround_grad(op.desc, op.w, op.alpha, op.delta, op.zp, op.slabs, op.dalpha)

    def _step_ops(self):
        """AdaRound step of every op of the unit + iteration counter: one batched launch (+ one for the dgrad layouts) when the
        tensors allow it, else one launch per op.  The bf16 planes of the new weights are written by the same launches."""
        opl = list(self.ops.values())
        if self._batchable(opl):
            self._step_batches(opl, 1.0, 0)
            self._after_step(lin_done=True)
            return
        for op in opl:
            ops.adaround_step(op.desc, op.w, op.delta, op.zp, op.slabs, 1.0, self.weight, self.sched, self.it,
                              op.alpha, op.m, op.v, op.wq, op.wd, self.round_log, op.wq_planes, op.wd_planes)
        ops.iter_advance(self.it)
        self._after_step()

    def _split_point(self):
        """Called by the backward pass right before its last weight-gradient kernel.  Data-parallel recording only: the gradients
        of every other op are final here -> chain them into the front of the bucket, close plan A and continue in plan A2, so
        that `run` can start the all-reduce of the front while the last wgrad computes."""
        if self._probing or self._late is None or self.plan_a2 is not None:
            return
        self._grad_ops([n for n in self.ops if n != self._late])
        self._rec_ctx.__exit__(None, None, None)
        self.plan_a2 = Plan()
        self._rec_ctx = self.plan_a2.record()
        self._rec_ctx.__enter__()

    def _record(self):
        self.plan_a = Plan()
        self.plan_a2 = self.plan_b = self.plan_rd = None
        self._rec_ctx = self.plan_a.record()
        self._rec_ctx.__enter__()
        self._next_gather = None
        try:
            if self._folded:
                ops.iter_bind_publish(self.it_shadow)          # consumed by the iteration's first loss / tail launch
            self._forward_backward()
            if self._folded and (ops.iter_bind_publish(None) or self._next_gather is None):
                raise RuntimeError("calibration engine: the folded gather needs exactly one loss / tail launch and one gather per iteration")
            if not self.split:
                self._step_ops()
            elif self.plan_a2 is not None:
                self._grad_ops([self._late])
            else:
                self._grad_ops(list(self.ops))
        finally:
            ops.iter_bind_publish(None)        # (a recording that raised must not leave its word bound for the next engine's loss launch)
            self._rec_ctx.__exit__(None, None, None)
            self._rec_ctx = None
        if self.split:
            self.plan_b = Plan()
            with self.plan_b.record():
                opl = list(self.ops.values())
                if self._batchable(opl):
                    self._step_batches(opl, self.bucket.scale, 2)
                    self._after_step(lin_done=True)
                else:
                    for op in opl:
                        ops.adaround_apply(op.desc, op.w, op.delta, op.zp, op.dalpha, self.bucket.scale, self.weight, self.sched,
                                           self.it, op.alpha, op.m, op.v, op.wq, op.wd, self.round_log, op.wq_planes, op.wd_planes)
                    ops.iter_advance(self.it)
                    self._after_step()

    # ------------------------------------------------------------------------------------------------------------------
    def run(self, n_iters=None, idle=None):
        """Enqueue `n_iters` (default: all remaining) calibration iterations on the current stream; `idle()` (optional) is called after
        every enqueued piece, in front of the next poll -- host work that overlaps the GPU's (recon.py draws the next unit's mini-batch
        indices there).  Units on H2 planes read their
        overflow word every `H2_POLL` iterations of a long call (a 4-byte read; calls of at most H2_POLL iterations stay fully
        asynchronous, the word is then read by `logs` / `finish`) and restart themselves when it is set (`_recover`)."""
        done = self._done
        n = self.iters - done if n_iters is None else int(n_iters)
        if n < 0 or done + n > self.iters:
            raise ValueError(f"cannot run {n} iterations: {done} of {self.iters} already done")
        target = done + n
        with ops.h2_flag(self._ovf[0:2]):
            while self._done < target:
                k = target - self._done
                if self.P and self.H2_POLL > 0:
                    k = min(k, self.H2_POLL)
                if not (self.P and self.H2_POLL > 0) and idle is not None:
                    k = min(k, max(64, (target - done) // 16))      # no poll points: still enqueue in pieces so that `idle` gets its turns
                if self.plan_rd is not None:
                    self._run_rd(k)
                elif not self.split:
                    self.plan_a.run(k, graph=self.use_graph)
                else:
                    self._run_dp(k)
                self._done += k
                if idle is not None:
                    idle()                                      # host work while the GPU has the enqueued iterations to do
                if self._done < target and self.P and self.H2_POLL > 0:
                    self.dp_drain()
                if self._done < target and self._overflowed():
                    self._recover()                  # back to iteration 0 with new scales / on fp32 activations
        return n

    def prepare(self, n_iters=None):
        """Capture the graphs a `run(n_iters)` (default: the rest of the unit's iterations) will replay -- set-up work like the recording;
        `run` does it lazily otherwise (inside its first call).  Single-GPU plans only: the data-parallel and `rd` loops capture in
        their own first call (they need an eager iteration in front)."""
        if self.use_graph and not self.split and self.plan_rd is None:
            n = self.iters - self._done if n_iters is None else int(n_iters)
            k = min(n, self.H2_POLL) if (self.P and self.H2_POLL > 0) else n
            self.plan_a.prepare(k)
            if n % k:
                self.plan_a.prepare(n % k)

    def _dp_iteration(self, graph, first=True, last=True):
        """One data-parallel iteration on the current stream: plan A -> all-reduce of the front of the bucket (asynchronous, on the
        process group's stream) overlapped with plan A2 = the last weight gradient -> all-reduce of the rest -> plan B.
        Inside a host-driven run of several iterations plan B of iteration i and plan A of iteration i + 1 go out as ONE graph launch
        (`first=False`: plan A was enqueued with the previous iteration's plan B; `last=False`: enqueue the next plan A with this
        plan B): two host enqueues per iteration and collective instead of three -- the small units are host-bound in this loop."""
        if first:
            self.plan_a.run(1, graph=graph)
        self.bucket.reduce(between=None if self.plan_a2 is None else (lambda: self.plan_a2.run(1, graph=graph)))
        if last:
            self.plan_b.run(1, graph=graph)
        else:
            self.plan_b.run_then(self.plan_a, graph=graph)

    # backends whose collectives can be captured into a graph (torch's "nccl" = RCCL on ROCm); a test adds "gloo" to exercise the agreement
    DP_GRAPH_BACKENDS = ("nccl",)
    DP_HB_BATCH = 32                                                   # captured replays between two looks at the heartbeat
    DP_STALL_S = float(os.environ.get("RDO_DP_STALL_S", 120))          # no heartbeat for this long = the captured loop hangs

    def _dp_capture(self):
        """One data-parallel iteration -- the recorded kernels of the plans AND the collectives -- as one graph, closed by the heartbeat:
        a device counter incremented and copied to pinned host memory by the graph's last two nodes.  torch's NCCL watchdog does not see
        collectives replayed from a graph; the heartbeat is what tells the host that replays still complete (`_hb_wait`)."""
        if self._hb_dev is None:
            self._hb_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self._hb_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._dp_iteration(False)
            self._hb_dev.add_(1)
            self._hb_host.copy_(self._hb_dev, non_blocking=True)
        return g

    def _hb_wait(self, target):
        """Block until `target` captured replays have completed (heartbeat in pinned memory: no device synchronisation, which would hang
        with the loop).  No progress for DP_STALL_S seconds -> `DpStallError`: the process cannot recover a hung collective; the
        application decides (bench.py exits with status 86 and its launcher / supervisor starts FRESH children on the host loop)."""
        import time
        if self._hb_host is None or target <= 0:
            return
        seen, t_last = int(self._hb_host[0]), time.monotonic()
        while seen < target:
            time.sleep(0.0005)
            now = int(self._hb_host[0])
            if now != seen:
                seen, t_last = now, time.monotonic()
            elif time.monotonic() - t_last > self.DP_STALL_S:
                raise DpStallError(f"rank {self._rank()}: the captured data-parallel loop made no progress for {self.DP_STALL_S:.0f} s "
                                   f"({seen} of {self._hb_issued} replays completed) -- a collective inside the graph hangs")

    def dp_drain(self):
        """Wait (heartbeat, with the stall deadline) for every captured replay issued so far: call it before anything that synchronises
        the device behind a data-parallel run -- a plain synchronisation behind a hung graph would never return."""
        if self._dp_graph is not None:
            self._hb_wait(self._hb_issued)

    def _run_dp(self, n):
        """n data-parallel iterations.  With the RCCL backend (round 5: the DEFAULT; `RDO_DP_GRAPH=0` keeps the host loop) the whole
        iteration -- the recorded kernels of the three plans AND the collectives -- is captured once per unit into ONE graph
        (torch.cuda.graph) and replayed with no host work in between: the host-driven loop (plan A -> all-reduce -> plan B, each plan
        a graph replay, the collective enqueued by torch.distributed in between) costs two to three enqueues per unit-iteration, which
        the sixteen <= 50-us units cannot hide (+13 % on one rank against +5 % captured, `extra.dp_overhead_one_rank` of the bench
        line re-measures both on every run).  The ranks AGREE on the outcome of the capture (MIN all-reduce of an "ok" flag) before
        anyone replays: a rank never replays a graph with collectives while another runs the host loop; a unit whose capture fails
        on any rank runs the host loop on every rank (`dp_fallbacks` counts them).  The eager iteration in front of the capture is
        OUTSIDE the try: it is an ordinary iteration on every rank whatever the capture does afterwards, so all ranks run the same
        number of iterations (an exception in it is a real failure and propagates).  Replays are watched through a heartbeat
        (`_dp_capture`, `_hb_wait`).  Other backends (gloo in the CPU / one-GPU tests): host loop.  `self.dp_path` says which loop
        ran ("graph" / "host"); every rank logs it, the bench line carries it per unit."""
        import logging
        dist = torch.distributed
        comm = self.world > 1 or (dist.is_available() and dist.is_initialized())
        want = (self.use_graph and os.environ.get("RDO_DP_GRAPH", "1") == "1" and comm
                and dist.get_backend(self.group) in self.DP_GRAPH_BACKENDS)
        if want and self._dp_graph is None and not self._dp_graph_failed and n > 1:
            self._dp_iteration(False)                               # one eager iteration: warms RCCL and every lazy initialisation
            n -= 1
            ok, g = 1, None
            try:
                g = self._dp_capture()
            except Exception as e:      # pragma: no cover - depends on the RCCL / driver stack
                logging.warning("rank %s: data-parallel iteration could not be captured into a graph (%s)", self._rank(), e)
                ok = 0
                torch.cuda.synchronize()
            if comm:
                flag = torch.tensor([ok], device=self.dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                ok = int(flag.item())
            if ok:
                self._dp_graph, self._hb_issued = g, 0
                self._hb_dev.zero_()
                self._hb_host.zero_()
            else:
                self._dp_graph_failed = True
                self.dp_fallbacks += 1
                del g
        path = "graph" if self._dp_graph is not None else "host"
        if self.dp_path != path:
            self.dp_path = path
            logging.getLogger("rdo_ptq.dp").info("rank %s: data-parallel loop = %s (world %d)", self._rank(), path, self.world)
        if self._dp_graph is not None:
            for _ in range(n):
                self._dp_graph.replay()
                self._hb_issued += 1
                if self._hb_issued % self.DP_HB_BATCH == 0:         # at most two batches queued: the GPU stays fed, a hang is seen
                    self._hb_wait(self._hb_issued - self.DP_HB_BATCH)
            return
        merge = self.use_graph and os.environ.get("RDO_DP_MERGE", "1") != "0"
        for i in range(n):
            self._dp_iteration(self.use_graph, first=(i == 0 or not merge), last=(i == n - 1 or not merge))

    def _rank(self):
        dist = torch.distributed
        return dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0

    def _rd_iteration(self, graph):
        """One iteration of the R + lambda*D mode on the current stream: plan A (gather, unit forward, rec_loss and its gradient) ->
        the task term (the whole model behind the unit, `_rd_tail`) -> plan RD (+ g_task, backward, AdaRound step -- or, data
        parallel, the gradient bucket) [-> all-reduce -> plan B]."""
        self.plan_a.run(1, graph=graph)
        self._rd_tail()
        self.plan_rd.run(1, graph=graph)
        if self.split:
            self.bucket.reduce()
            self.plan_b.run(1, graph=graph)

    def _run_rd(self, n):
        """n iterations of the R + lambda*D mode.  Single process with graphs on: the WHOLE iteration -- the recorded kernels of both
        plans and the model tail with its autograd backward (every op of which is a device-side kernel: the mini-batch rows are
        selected with the device iteration counter) -- is captured once into one graph (torch.cuda.graph) and replayed, no host work
        per iteration.  `RDO_RD_GRAPH=0`, a refused capture, or the data-parallel split (a collective per iteration): host-driven."""
        import logging
        want = self.use_graph and not self.split and os.environ.get("RDO_RD_GRAPH", "1") != "0"
        if want and self._rd_graph is None and not self._rd_graph_failed and n > 2:
            self._rd_iteration(False)                 # eager: every lazy initialisation (scratch buffers, kernel attributes, autograd)
            n -= 1
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._rd_iteration(False)
                self._rd_graph = g
            except Exception as e:      # pragma: no cover - depends on the driver stack
                logging.warning("R + lambda*D iteration could not be captured into a graph (%s): host-driven loop", e)
                self._rd_graph_failed = True
                torch.cuda.synchronize()
        self.rd_path = "graph" if self._rd_graph is not None else "host"
        if self._rd_graph is not None:
            for _ in range(n):
                self._rd_graph.replay()
            return
        for _ in range(n):
            self._rd_iteration(self.use_graph)

    # ---- R + lambda*D: what does not depend on this unit is computed once per calibration image -----------------------------------
    RD_CACHE_LIMIT = 8 << 30            # bytes of cached module outputs per unit (a coder's latent for 256 images is 50-100 MB)

    @staticmethod
    def _tree(fn, t):
        if torch.is_tensor(t):
            return fn(t)
        if isinstance(t, dict):
            return {k: UnitEngine._tree(fn, v) for k, v in t.items()}
        if isinstance(t, (tuple, list)):
            return type(t)(UnitEngine._tree(fn, v) for v in t)
        return t

    def _rd_prepare(self):
        """Modules of the wrapped model whose INPUTS do not depend on this unit's output produce the same output for an image in every
        iteration (weights behind and before the unit are fixed for the unit's run): found by running the model twice on one
        mini-batch with two different random tensors in place of the unit's output, on torch's tape: a candidate is independent when
        its inputs are bit-equal in both runs AND none of them carries a grad_fn back to the substituted tensor (values alone are not
        enough: two different hyper-latents can round to the same z^; the tape alone is not either: a detached activation
        quantiser cuts it).  Independent modules are evaluated ONCE per calibration image (in mini-batches of B, the shape the iterations run at) and looked up by row afterwards.
        Candidates: the children of the model (the coders and entropy models) that do not contain the unit.  Inside the coder that
        does, the modules in front of the unit in an nn.Sequential only feed the unit, whose forward is replaced: they are skipped.
        A candidate holding a live dynamic activation quantiser (statistics over the mini-batch) is left alone.  `RDO_RD_CACHE=0`
        turns all of this off."""
        rd = self.rd
        rd["memo"], rd["skip"] = {}, []
        if os.environ.get("RDO_RD_CACHE", "1") == "0":
            return
        from .quant_block import BaseQuantBlock
        from .quant_layer import QuantModule
        model, unit, cali = rd["model"], rd["unit"], rd["cali"]
        root = model.model if isinstance(getattr(model, "model", None), torch.nn.Module) else model
        holds = lambda m: any(c is unit for c in m.modules())
        cands = []
        for _, child in root.named_children():
            if not holds(child):
                cands.append(child)
                continue
            box = child
            while box is not unit and isinstance(box, torch.nn.Sequential):       # descend to the unit: skip what only feeds it
                kids = list(box.children())
                k = next(i for i, c in enumerate(kids) if holds(c))
                rd["skip"] += kids[:k]
                box = kids[k]
        live_aq = lambda m: any(isinstance(c, (QuantModule, BaseQuantBlock)) and c.use_act_quant and c.trained
                                and not getattr(c, "disable_act_quant", False) for c in m.modules())
        cands = [c for c in cands if not live_aq(c)]
        if not cands:
            return
        was_training = model.training
        model.eval()
        B = self.B
        shape = tuple(self._rd_pred.permute(0, 3, 1, 2).shape)
        seen = [{}, {}]
        try:
            tainted = set()

            def note(m, a, k, run):
                flag = [False]
                self._tree(lambda t: flag.__setitem__(0, flag[0] or t.requires_grad), (a, k))
                if flag[0] or id(m) in seen[run]:
                    # depends on the unit -- or is called more than once per model forward (a shared / looped module): its outputs
                    # could not be told apart by image row, so it is never memoised
                    tainted.add(id(m))
                seen[run][id(m)] = self._tree(lambda t: t.detach().clone(), (a, k))
            for run in range(2):
                fake = torch.randn(shape, device=self.dev, generator=torch.Generator(device=self.dev).manual_seed(17 + run)).requires_grad_(True)
                hooks = [c.register_forward_pre_hook(lambda m, a, k, run=run: note(m, a, k, run), with_kwargs=True) for c in cands]
                unit.forward = lambda *a, fake=fake, **k: fake
                for s_ in rd["skip"]:
                    s_.forward = lambda x, *a, **k: x
                try:
                    with torch.enable_grad():
                        model(cali[:B])
                finally:
                    for h in hooks:
                        h.remove()
                    del unit.forward
                    for s_ in rd["skip"]:
                        del s_.forward
            with torch.no_grad():

                def same(a, b):
                    if torch.is_tensor(a):
                        return torch.is_tensor(b) and a.shape == b.shape and bool(torch.equal(a, b))
                    if isinstance(a, dict):
                        return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
                    if isinstance(a, (tuple, list)):
                        return isinstance(b, (tuple, list)) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
                    return a == b
                free = [c for c in cands if id(c) in seen[0] and id(c) in seen[1] and id(c) not in tainted and same(seen[0][id(c)], seen[1][id(c)])]
                seen = None
                if not free:
                    return
                # one pass over the calibration images: outputs of the unit-independent modules, image-major
                parts = {id(c): [] for c in free}
                hooks = [c.register_forward_hook(lambda m, a, o: parts[id(m)].append(self._tree(lambda t: t.detach().clone(), o))) for c in free]
                unit.forward = lambda *a, **k: torch.zeros(shape, device=self.dev)
                for s_ in rd["skip"]:
                    s_.forward = lambda x, *a, **k: x
                try:
                    n = cali.shape[0]
                    n_fwd = 0
                    for i in range(0, n, B):
                        xb = cali[i:i + B]
                        if xb.shape[0] < B:                                   # ragged end: pad the mini-batch with the first images
                            xb = torch.cat([xb, cali[:B - xb.shape[0]]])
                        model(xb)
                        n_fwd += 1
                finally:
                    for h in hooks:
                        h.remove()
                    del unit.forward
                    for s_ in rd["skip"]:
                        del s_.forward

                def cat(chunks):
                    first = chunks[0]
                    if torch.is_tensor(first):
                        return torch.cat(chunks)[:n] if first.dim() > 0 and first.shape[0] == B else None
                    if isinstance(first, dict):
                        out = {k: cat([c[k] for c in chunks]) for k in first}
                        return None if any(v is None for v in out.values()) else out
                    if isinstance(first, (tuple, list)):
                        out = [cat([c[j] for c in chunks]) for j in range(len(first))]
                        return None if any(v is None for v in out) else type(first)(out)
                    return None
                total = 0
                for c in free:
                    chunks = parts.pop(id(c))
                    if len(chunks) != n_fwd:                                  # exactly one call per model forward, or the rows of the
                        continue                                              # concatenation would not be the calibration images
                    memo = cat(chunks)
                    if memo is None:
                        continue                                              # an output that is not per-image: leave the module alone
                    nbytes = [0]
                    self._tree(lambda v: nbytes.__setitem__(0, nbytes[0] + v.numel() * v.element_size()), memo)
                    if total + nbytes[0] > self.RD_CACHE_LIMIT:
                        continue
                    total += nbytes[0]
                    rd["memo"][c] = memo
        finally:
            model.train(was_training)

    def _rd_tail(self):
        """The current iteration's task loss: the images of the mini-batch through the wrapped model with this unit's output replaced
        by the engine's soft-quantised output, `RateDistortionLoss` (lambda * 255^2 * MSE + bpp) on the result, gradient back to that
        output (HIP kernels under torch's tape: hipops.autograd; the factorised prior and the Gaussian conditional included).  The
        modules behind the unit are whatever the calibration flow left them: trained ones hard-quantised, the others full precision
        (layer_opt.py:15-43).  Device-side throughout (capturable): the mini-batch rows come from the index table through the device
        iteration counter."""
        from losses.losses import RateDistortionLoss
        rd = self.rd
        it = self.it.long()                                                   # [1], published by this iteration's gather
        rows = self.idx.index_select(0, it).view(-1).long()
        x = rd["cali"].index_select(0, rows)
        leaf = self._rd_pred.permute(0, 3, 1, 2).detach().requires_grad_(True)
        if "memo" not in rd:
            self._rd_prepare()
        was_training = rd["model"].training
        rd["model"].eval()
        patched = [rd["unit"]]
        rd["unit"].forward = lambda *a, **k: leaf      # the unit itself is not run: its output is the engine's
        for m_ in rd["skip"]:                          # ... nor what only feeds it
            m_.forward = lambda x_, *a, **k: x_
            patched.append(m_)
        for m_, memo in rd["memo"].items():            # ... and what does not depend on it is looked up per image
            m_.forward = lambda *a, memo=memo, **k: self._tree(lambda t: t.index_select(0, rows), memo)
            patched.append(m_)
        try:
            with torch.enable_grad():
                out = rd["model"](x)
                loss = RateDistortionLoss(lmbda=rd["lmbda"], metric="mse")(out, x)["loss"]
                (g,) = torch.autograd.grad(loss, [leaf], allow_unused=True)
        finally:
            for m_ in patched:
                del m_.forward
            rd["model"].train(was_training)
        if g is None:
            raise RuntimeError("loss_mode='rd': no gradient reached the unit's output -- a module behind it detaches the tape (dynamic "
                               "activation quantisers of trained modules do); run the RD task loss with act_quant=False")
        self.g_task.copy_(g.permute(0, 2, 3, 1))
        self.task_log.view(-1).index_add_(0, it * L.LOG_SLOTS, loss.detach().reshape(1))

    def _data_terms(self):
        """(rec, task) per iteration on the device, averaged over the data-parallel ranks."""
        rec, task = self.loss_log.sum(1), self.task_log.sum(1)
        if self._task_is_rec:
            rec = rec * 0.5
            task = rec.clone()
        if self.world > 1:
            both = torch.stack([rec, task])
            torch.distributed.all_reduce(both, group=self.group)
            rec, task = both[0] / self.world, both[1] / self.world
        return rec, task

    def logs(self):
        """(total, rec+task, round) per iteration as CPU tensors (synchronises)."""
        self._check_overflow()
        rec, task = self._data_terms()
        rt = (rec + task).cpu()
        rd = self.round_log.sum(1).cpu()
        return rt + rd, rt, rd

    def logs_terms(self):
        """(rec, task, round, b) per iteration as CPU tensors: the four numbers of the reference's periodic log line
        (layer_opt.py:168-170); b is the temperature of the schedule table (0 while the rounding loss is off)."""
        self._check_overflow()
        rec, task = self._data_terms()
        return rec.cpu(), task.cpu(), self.round_log.sum(1).cpu(), self.sched[:, 0].cpu()

    def alpha_of(self, name):
        """Trained alpha in the logical weight shape (OIHW view of the OHWI storage; [Cin,Cout,KH,KW] for transposed convs)."""
        op = self.ops[name]
        if op.tconv is not None:
            return op.alpha.flip(1, 2).permute(3, 0, 1, 2)
        if op.qm.kind in ("linear", "layernorm"):
            return op.alpha.reshape(op.qm.org_weight.shape)
        return op.alpha.permute(0, 3, 1, 2) if op.alpha.dim() == 4 else op.alpha

    def finish(self):
        """Hand the trained rounding back to the modules: AdaRoundQuantizer with hard targets, `trained` flags
        (layer_opt.py:313-316 / block_opt.py:316-321)."""
        self._check_overflow()
        for name, op in self.ops.items():
            qm = op.qm
            rows = op.alpha.flip(1, 2).contiguous() if op.tconv is not None else op.alpha
            if qm.kind == "linear":
                rows = op.alpha.reshape(op.alpha.shape[0], -1)
            if not op.channel_wise and not op.is_ln:
                # the per-tensor quantiser keeps alpha in the logical (OIHW) element order, flattened to one row
                rows = from_rows(rows, qm.org_weight.data, op.tconv is not None).contiguous().reshape(1, -1)
            ada = AdaRoundQuantizer(uaq=qm.weight_quantizer, round_mode="learned_hard_sigmoid",
                                    weight_tensor=qm.org_weight.data, alpha_rows=rows)
            ada.soft_targets = False
            qm.weight_quantizer = ada
            qm.act_quantizer.is_training = False
            qm.drop_weight_pack()              # packs of the pre-calibration weights (nearest rounding) are stale from here on
        if self.rd is not None:
            self.rd.pop("memo", None)          # per-image outputs of the unit-independent modules (up to RD_CACHE_LIMIT bytes)
