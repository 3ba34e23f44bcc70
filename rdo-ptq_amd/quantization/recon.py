"""Shared body of `layer_reconstruction` / `block_reconstruction` (reference: layer_opt.py:175-320, block_opt.py:176-324).

The choreography around the hot loop is the reference's; the loop itself runs on `engine.UnitEngine`."""
import logging
import time
import zlib

import torch

from . import dp
from hipops import ops
from .engine import UnitEngine
from .quant_block import BaseQuantBlock, QuantRSTB
from .swin_engine import TapeEngine
from .quant_layer import QuantModule, _nhwc
from .utils import LinearTempDecay, save_inp_oup_data, set_mode

_COOPS = ("g_a", "h_a", "h_s", "g_s")


def find_unquantized_module(model, _name_="g_a", module_list=None, name_list=None):
    """Switch every untrained unit to full precision and collect those of the current sub-coder (layer_opt.py:15-43).
    For Sequential-indexed models the names never contain g_a/h_a/h_s/g_s, so the lists stay empty."""
    module_list = [] if module_list is None else module_list
    name_list = [] if name_list is None else name_list
    for name, module in model.named_children():
        if isinstance(module, (QuantModule, BaseQuantBlock)):
            if not module.trained:
                module.set_quant_state(False, False)
                for tag in _COOPS:
                    if tag in _name_ and tag in name:
                        name_list.append(name)
                        module_list.append(module)
        else:
            find_unquantized_module(module, _name_, module_list, name_list)
    return module_list[1:], name_list[1:]


def fp_out(module_list, x, round_after, batch=8):
    """`fp_out` of the reference (layer_opt.py:45-75) over a whole cache: the remaining stages of the sub-coder in full
    precision (their quant state was switched off by find_unquantized_module), then round for analysis-transform units."""
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            h = x[i:i + batch]
            for m in module_list:
                h = m(h) if isinstance(m, QuantModule) else m(h, (h.shape[2], h.shape[3]))
            if round_after:
                h = ops.round_(_nhwc(h)).permute(0, 3, 1, 2)
            outs.append(h)
    return torch.cat(outs)


class LossFunction:
    """The objective of one reconstruction unit with the reference's call signature (layer_opt.py:87-173; the block variant,
    block_opt.py:87-173, sums the rounding term over the block's QuantModules).  The calibration engine evaluates the same three
    terms on the device inside the recorded iteration; this class is the inspection / logging form of it on torch tensors
    (used by the parity tests to evaluate the loss of a calibrated unit), not part of the hot loop."""

    def __init__(self, unit, round_loss="relaxation", weight=1., rec_loss="mse", max_count=2000, b_range=(10, 2),
                 decay_start=0.0, warmup=0.0, p=2., lmbda=None, metric=None):
        self.unit, self.round_loss, self.weight, self.rec_loss = unit, round_loss, weight, rec_loss
        self.layer = self.block = unit                 # attribute names of the two reference classes
        self.loss_start = max_count * warmup
        self.p, self.lmbda, self.metric = p, lmbda, metric
        self.temp_decay = LinearTempDecay(max_count, rel_start_decay=warmup + (1 - warmup) * decay_start,
                                          start_b=b_range[0], end_b=b_range[1])
        self.count = 0

    def _round_modules(self):
        if isinstance(self.unit, QuantModule):
            return [self.unit]
        return [m for m in self.unit.modules() if isinstance(m, QuantModule) and m.org_weight is not None]

    def __call__(self, pred, tgt, quant_net_out=None, cali_data=None, grad=None):
        from .quantizer import lp_loss
        self.count += 1
        if self.rec_loss != "mse":
            raise NotImplementedError("only rec_loss='mse' (the mode main2.py uses) is built")
        rec_loss = lp_loss(pred, tgt, p=self.p)
        task_loss = 0.
        if quant_net_out is not None:
            task_loss = lp_loss(quant_net_out, cali_data, p=self.metric)
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round_loss == "none":
            b = round_loss = 0
        elif self.round_loss == "relaxation":
            round_loss = 0
            for m in self._round_modules():
                round_vals = m.weight_quantizer.get_soft_targets()
                round_loss += self.weight * (1 - ((round_vals - .5).abs() * 2).pow(b)).sum()
        else:
            raise NotImplementedError
        total_loss = round_loss + rec_loss + task_loss
        if self.count % 500 == 0:
            logging.info("Total loss:\t{:.3f} ( task:{:.3f}, rec:{:.3f}, round:{:.3f})\tb={:.2f}\tcount={}".format(
                float(total_loss), float(task_loss), float(rec_loss), float(round_loss), b, self.count))
        return total_loss


def unit_seed(unit_name: str) -> int:
    """Seed of the QDrop stream of one unit: the process seed (main2.py seed_all -> torch.manual_seed) mixed with a CRC of the
    unit's name -- reproducible from run to run (Python's str hash is salted per process) and distinct per unit."""
    return (torch.initial_seed() ^ zlib.crc32(unit_name.encode())) & 0xFFFFFFFF


def _unit_modules(unit):
    """kind + the engine's module dict for a QuantModule or a Cheng2020 block."""
    if isinstance(unit, QuantModule):
        return "layer", {"layer": unit}
    kind = getattr(unit, "unit_kind", None)
    if kind == "rb":
        return kind, {"conv1": unit.conv1, "conv2": unit.conv2, "skip": unit.skip}
    if kind == "rbws":
        return kind, {"conv1": unit.conv1, "conv2": unit.conv2, "gdn": unit.gdn, "skip": unit.skip}
    if kind == "rbu":
        return kind, {"subpel_conv": unit.subpel_conv[0], "conv": unit.conv, "igdn": unit.igdn,
                      "upsample": unit.upsample[0], "upscale": unit.subpel_conv[1].upscale_factor}
    if kind == "rstb":
        return kind, {"rstb": unit}
    raise NotImplementedError(f"reconstruction of {type(unit).__name__} is not built yet")


def reconstruct(model, unit, unit_name, cali_data, *a, **kw):
    """`_reconstruct` with the one piece of cross-unit state tidied up on failure: the full-precision cache memo of the schedule
    (quantization/utils.py::_FpMemo) is dropped when a unit raises, so a schedule that dies half-way pins no device memory."""
    try:
        return _reconstruct(model, unit, unit_name, cali_data, *a, **kw)
    except BaseException:
        from .utils import _FpMemo
        _FpMemo.clear()
        raise


def _reconstruct(model, unit, unit_name, cali_data, batch_size=32, iters=20000, weight=0.01, opt_mode="mse", asym=False,
                 include_act_func=True, b_range=(20, 2), warmup=0.0, input_prob=1.0, act_quant=False, lr=4e-5, p=2.0,
                 config=None, args=None, is_block=False):
    if opt_mode != "mse":
        # The reference cannot run these modes on a compression model either: its LossFunction returns None for them whenever a coder
        # tail output is passed (layer_opt.py:146-151, always the case in its loops, so `err.backward()` fails), and GetLayerGrad feeds
        # the model's output DICT to log_softmax / kl_div (utils.py:316-321: BRECQ's classification code).
        raise NotImplementedError(f"opt_mode={opt_mode!r}: only 'mse' (the mode main2.py uses) is built -- the Fisher-weighted modes of the "
                                  "reference are classification left-overs that do not run on its compression models")
    task_p = getattr(args, "task_loss", 2.0) if args is not None else 2.0
    if float(p) != 2.0:
        raise NotImplementedError("rec_loss is built for p = 2 (the value main2.py passes); --task_loss may be any exponent >= 1")
    if float(task_p) < 1.0:
        raise ValueError("--task_loss < 1 has no finite gradient at zero error")
    rank, world_size = dp.world()
    if world_size > 1:                      # data parallel: this rank calibrates on its shard with its share of the batch
        if batch_size % world_size != 0:
            raise ValueError(f"data-parallel calibration: batch_size {batch_size} is not a multiple of the world size "
                             f"{world_size} (the global mini-batch is split evenly; pass a multiple, e.g. {world_size * max(1, batch_size // world_size)})")
        if cali_data.size(0) % world_size != 0:
            raise ValueError(f"data-parallel calibration: {cali_data.size(0)} calibration images do not split evenly over "
                             f"{world_size} ranks (uneven shards would weight samples unequally)")
        cali_data = dp.shard(cali_data)
        batch_size = batch_size // world_size
    # args.timing (optional list): the caller wants this unit's wall split -- cache building / plan recording / loop -- which costs
    # three device synchronisations (tools/full_schedule.py, bench.py's recon_model_wall_s)
    timing = getattr(args, "timing", None) if args is not None else None

    def _mark():
        if timing is not None:
            torch.cuda.synchronize()
        return time.time()
    t0 = _mark()
    # dynamic activation quantisation makes cached values depend on the caching batch: keep the reference's batch of 1 then
    cache_bs = 1 if act_quant else max(1, min(32, cali_data.size(0)))
    (inp_q, inp_fp), out_fp = save_inp_oup_data(model, unit, cali_data, asym, act_quant, batch_size=cache_bs, input_prob=True)
    t1 = _mark()
    logging.info("Cached init time: {}".format(t1 - t0))
    module_list, name_list = find_unquantized_module(model, unit_name, [], [])
    logging.info(name_list)
    # Lu2022 naming (g_a0 ... g_s7): the task term runs through the untrained rest of the sub-coder, and through round_ste for
    # analysis-transform units (layer_opt.py:45-75); its target is the same function of the cached FP outputs (:262-263)
    tail_round = "g_a" in unit_name
    task_cache = None
    model.set_quant_state(False, False)
    rd_mode = getattr(args, "loss_mode", "lp") == "rd"
    if (module_list or tail_round) and not rd_mode:          # (the R + lambda*D task term does not use the lp target)
        task_cache = _nhwc(fp_out(module_list, out_fp, tail_round))
    set_mode(model, act_quant)
    if not is_block and ("g_s7" in unit_name or "7" in unit_name):
        logging.info("=======last layer, close activation quantization=======")
        unit.set_quant_state(True, False)
    else:
        unit.set_quant_state(True, act_quant)
    if not is_block and unit.org_weight is None:
        return None                                   # PixelShuffle units carry nothing to train (layer_opt.py:245-246)
    kind, mods = _unit_modules(unit)
    # opt-in task loss: the R + lambda*D loss of the whole model that the reference sketches and comments out
    # (layer_opt.py:146-148).  args.loss_mode = 'rd' (default 'lp' = the reference's lp_loss pair)
    rd = None
    if getattr(args, "loss_mode", "lp") == "rd":
        rd = dict(model=model, unit=unit, cali=cali_data.to(next(model.parameters()).device), lmbda=float(getattr(args, "lmbda", 0.01)))
    elif getattr(args, "loss_mode", "lp") != "lp":
        raise ValueError(f"unknown loss_mode {args.loss_mode!r} ('lp' or 'rd')")
    # the CLI --lr is ignored by the reference (Adam default 1e-3, layer_opt.py:253-254); kept that way.
    common = dict(batch_size=batch_size, iters=iters, weight=weight, b_range=b_range, warmup=warmup, input_prob=input_prob,
                  lr=1e-3, seed=unit_seed(unit_name), include_act_func=include_act_func, batch_offset=rank * batch_size,
                  task_p=float(task_p))
    if rd is not None:
        # The rate-distortion task term replaces the lp term through the rest of the sub-coder: the whole wrapped model behind the unit
        # runs under torch's tape (hipops.autograd; Swin blocks included), so the Lu2022 units need no recorded FP tail here
        common["rd"] = rd
        if kind == "rstb":
            eng = TapeEngine(kind, mods, _nhwc(inp_q), _nhwc(inp_fp), _nhwc(out_fp), **common)
        else:
            eng = UnitEngine(kind, mods, _nhwc(inp_q), _nhwc(inp_fp), _nhwc(out_fp), **common)
    elif kind == "rstb" or task_cache is not None:
        eng = TapeEngine(kind, mods, _nhwc(inp_q), _nhwc(inp_fp), _nhwc(out_fp), tail=module_list, tail_round=tail_round,
                         task_cache=task_cache, **common)
    else:
        eng = UnitEngine(kind, mods, _nhwc(inp_q), _nhwc(inp_fp), _nhwc(out_fp), **common)
    eng.prepare()                # graph capture belongs to the set-up ("recording + graph capture" of the timing split), not to the loop
    t2 = _mark()
    # the next unit's index table is drawn while this unit's loop runs on the GPU (engine.IdxStream: adopted only if the next request
    # matches and nobody touched the CPU generator in between)
    from .engine import IdxStream
    IdxStream.begin(eng.cq.shape[0], eng.B, eng.iters)
    per_turn = max(256, eng.iters // 12)
    eng.run(idle=lambda: IdxStream.step(per_turn))
    t3 = _mark()
    if timing is not None:
        timing.append(dict(unit=unit_name, kind=kind, cache_s=t1 - t0, record_s=t2 - t1, loop_s=t3 - t2))
    # rank-invariant from here on: the overflow verdict of the unit (a MAX all-reduce under data parallelism) is formed ONCE, here, on
    # every rank; `logs_terms` is a collective as well when world > 1 (mean of the ranks' data terms), so with several ranks every rank
    # computes the log rows whether or not its own logger prints them -- the log level may differ from rank to rank, the sequence of
    # collectives must not
    eng.sync_overflow()
    want_log = logging.getLogger().isEnabledFor(logging.INFO)
    if iters >= 500 and (want_log or world_size > 1):
        rec, task, rd, b = eng.logs_terms()
        for c in (range(500, iters + 1, 500) if want_log else ()):
            logging.info("Total loss:\t{:.3f} ( task:{:.3f}, rec:{:.3f}, round:{:.3f})\tb={:.2f}\tcount={}".format(
                float(rec[c - 1] + task[c - 1] + rd[c - 1]), float(task[c - 1]), float(rec[c - 1]), float(rd[c - 1]),
                float(b[c - 1]), c))
    eng.finish()
    for m in ([unit] if not is_block else unit.modules()):
        if isinstance(m, (QuantModule, BaseQuantBlock)):
            m.trained = True
    return eng
