"""Cache building and schedule helpers of the reconstruction loops (reference surface: quantization/utils.py).

`save_inp_oup_data` runs the (partially quantised) model up to a target unit with a forward hook that captures the unit's
input/output and aborts the pass (utils.py:92-139, 175-258).  Differences from the reference, none of which change values:
tensors never bounce through host memory, and the per-sample `print` is dropped."""
import time
from typing import Union

import torch

from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule
from .quant_model import QuantModel


class StopForwardException(Exception):
    """Raised by the capture hook to abandon the rest of the forward pass."""


def set_mode(model, act_quant):
    """Re-enable quantisation on every QuantModule that has already been trained (utils.py:28-35)."""
    for _, module in model.named_children():
        if isinstance(module, QuantModule):
            if module.trained:
                module.set_quant_state(True, act_quant)
        else:
            set_mode(module, act_quant)


class LinearTempDecay:
    """Temperature b of the rounding regulariser: start_b during warm-up, then linear to end_b (utils.py:37-54)."""

    def __init__(self, t_max: int, rel_start_decay: float = 0.2, start_b: int = 10, end_b: int = 2):
        self.t_max = t_max
        self.start_decay = rel_start_decay * t_max
        self.start_b, self.end_b = start_b, end_b

    def __call__(self, t):
        if t < self.start_decay:
            return self.start_b
        frac = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + (self.start_b - self.end_b) * max(0.0, 1 - frac)


class DataSaverHook:
    def __init__(self, store_input=False, store_output=False, stop_forward=False):
        self.store_input, self.store_output, self.stop_forward = store_input, store_output, stop_forward
        self.input_store = self.output_store = None

    def __call__(self, module, input_batch, output_batch):
        if self.store_input:
            self.input_store = input_batch
        if self.store_output:
            self.output_store = output_batch
        if self.stop_forward:
            raise StopForwardException


class GetLayerInpOut:
    """Two truncated passes per batch: full precision (FP input + FP target), then with the trained prefix quantised
    (the asymmetric input of AdaRound/QDrop)."""

    def __init__(self, model: QuantModel, layer: Union[QuantModule, BaseQuantBlock], device, asym: bool = False,
                 act_quant: bool = False, input_prob: bool = False):
        self.model, self.layer, self.device = model, layer, device
        self.asym, self.act_quant, self.input_prob = asym, act_quant, input_prob
        self.data_saver = DataSaverHook(store_input=True, store_output=True, stop_forward=True)

    def _truncated_forward(self, x):
        try:
            self.model(x)
        except StopForwardException:
            pass

    def __call__(self, model_input, fp_pass=True):
        """fp_pass=False (asym only): skip the full-precision pass -- the caller has its rows already (`_FpMemo`); returns (inp_q,)."""
        self.model.eval()
        self.model.set_quant_state(False, False)
        handle = self.layer.register_forward_hook(self.data_saver)
        x = model_input.to(self.device)
        with torch.no_grad():
            if not fp_pass:
                set_mode(self.model, self.act_quant)
                self._truncated_forward(x)
                handle.remove()
                self.model.set_quant_state(False, False)
                set_mode(self.model, self.act_quant)
                self.layer.set_quant_state(True, self.act_quant)
                self.model.train()
                return (self.data_saver.input_store[0].detach(),)
            self._truncated_forward(x)
            input_sym = self.data_saver.input_store[0].detach() if self.input_prob else None
            if self.asym:
                self.data_saver.store_output = False
                set_mode(self.model, self.act_quant)
                self._truncated_forward(x)
            self.data_saver.store_output = True
        handle.remove()
        self.model.set_quant_state(False, False)
        set_mode(self.model, self.act_quant)
        self.layer.set_quant_state(True, self.act_quant)
        self.model.train()
        inp, out = self.data_saver.input_store[0].detach(), self.data_saver.output_store.detach()
        return (inp, out, input_sym) if self.input_prob else (inp, out)


# ---- full-precision caches of ALL units from one pass ---------------------------------------------------------------------------------
# The full-precision half of a unit's caches (x_fp, target) does not depend on what has been calibrated so far: the pass runs with every
# quantiser off (utils.py:219-221 of the reference).  The reference -- and `GetLayerInpOut` above -- recompute it for every unit with a
# truncated forward: over a model that is the sum of all prefixes, about half of the cache-building time (2.5 of 5.1 s for 256 images of
# Cheng2020 N=192).  `_FpMemo` runs ONE full-precision forward per caching batch with a capture hook on every reconstruction unit and
# hands each unit its rows when its turn comes (bit-identical: the same modules on the same batches).  Bounded: only when the extrapolated
# size fits RDO_FP_MEMO_GIB (default 64) and half of the free device memory; a unit's rows are released when it takes them; any change of
# model, calibration tensor or batch size drops the memo.  RDO_FP_MEMO=0 turns it off.
class _FpMemo:
    """One memo at a time.  What makes a hand-out safe (each a way the rows of the one full forward could differ from the unit's own
    truncated pass, in which case the unit simply gets no rows and runs its own pass):
      * a unit called more or less than ONCE in some batch (a shared / looped module: its rows would interleave) is dropped;
      * a captured tensor that any LATER op of the full forward modified in place (`_version` moved: the truncated pass stops at the
        unit and never runs that op) is dropped;
      * units with `ignore_reconstruction` are never asked for (main2.py:231,241) and are not captured.
    Identity: the memo holds a weak reference to the model and a STRONG one to the calibration tensor it was built from -- the tensor's
    storage cannot be freed and re-issued under the same `data_ptr()` while the memo lives, and a dead model drops the memo.
    `clear()` releases it explicitly (recon.reconstruct calls it when a unit fails; a schedule that stops early may call it too)."""
    current = None
    refused = None          # key of the last memo that could not be built (too large / the model's forward failed): not retried per unit

    def __init__(self, model, cali_data, batch_size):
        import weakref
        self.model_ref = weakref.ref(model)
        self.cali = cali_data                  # strong: pins the storage behind data_ptr()
        self.sig = self._sig(cali_data, batch_size)
        self.rows = {}

    @staticmethod
    def _sig(cali_data, batch_size):
        return (cali_data.data_ptr(), cali_data._version, tuple(cali_data.shape), tuple(cali_data.stride()), int(batch_size))

    def matches(self, model, cali_data, batch_size):
        return self.model_ref() is model and self.sig == self._sig(cali_data, batch_size)

    @classmethod
    def clear(cls):
        cls.current = cls.refused = None

    @staticmethod
    def units_of(model):
        """the reconstruction units in the order main2.py's recon_model visits them (main2.py:227-253); units the schedule skips
        (`ignore_reconstruction`, main2.py:231,241) are left out"""
        out = []

        def walk(m):
            for _, c in m.named_children():
                if isinstance(c, (QuantModule, BaseQuantBlock)):
                    if not getattr(c, "ignore_reconstruction", False):
                        out.append(c)
                else:
                    walk(c)
        walk(model.model if isinstance(getattr(model, "model", None), torch.nn.Module) else model)
        return out

    @classmethod
    def get(cls, model, layer, cali_data, batch_size, device):
        import os
        import weakref
        if os.environ.get("RDO_FP_MEMO", "1") == "0":
            return None
        memo = cls.current
        if memo is not None and memo.model_ref() is None:
            memo = cls.current = None                            # the model it belonged to is gone
        if memo is None or not memo.matches(model, cali_data, batch_size):
            r = cls.refused
            if r is not None and r[0]() is model and r[1] == cls._sig(cali_data, batch_size):
                return None
            memo = cls.current = cls._build(model, cali_data, batch_size, device)
            if memo is None:
                cls.refused = (weakref.ref(model), cls._sig(cali_data, batch_size))
                return None
            cls.refused = None
        got = memo.rows.pop(id(layer), None)
        if not memo.rows:
            cls.current = None
        return got

    @classmethod
    def _build(cls, model, cali_data, batch_size, device):
        import os
        units = cls.units_of(model)
        if len(units) < 2:
            return None
        store = {id(u): ([], []) for u in units}
        calls = {id(u): 0 for u in units}
        bad = set()

        def hook(m, i, o):
            k = id(m)
            calls[k] += 1
            if not (torch.is_tensor(o) and len(i) > 0 and torch.is_tensor(i[0])):
                bad.add(k)
                return None
            a, b = i[0].detach(), o.detach()
            store[k][0].append((a, a._version))
            store[k][1].append((b, b._version))
            return None
        hooks = [u.register_forward_hook(hook) for u in units]
        was_training = model.training
        states = [(m, m.use_weight_quant, m.use_act_quant) for m in model.modules() if isinstance(m, (QuantModule, BaseQuantBlock))]
        budget = min(float(os.environ.get("RDO_FP_MEMO_GIB", "64")) * 2 ** 30, 0.5 * torch.cuda.mem_get_info(device)[0])
        ok = True
        try:
            model.eval()
            model.set_quant_state(False, False)
            with torch.no_grad():
                n = cali_data.size(0)
                for i in range(0, n, batch_size):
                    for k in calls:
                        calls[k] = 0
                    model(cali_data[i:i + batch_size].to(device))
                    for k, c in calls.items():
                        if c != 1:                               # shared / looped / unreached module: its rows are not one per image
                            bad.add(k)
                    for k, (a, b) in store.items():              # a later op of this forward wrote into a captured tensor in place
                        if k not in bad and any(t._version != v for t, v in (a[-1:] + b[-1:])):
                            bad.add(k)
                    for k in bad:
                        store[k][0].clear(); store[k][1].clear()
                    if i == 0:                                   # extrapolate from the first batch
                        per = sum(t.numel() * t.element_size() for a, b in store.values() for t, _ in a + b) / min(batch_size, n)
                        if per * n > budget:
                            ok = False
                            break
        except Exception:                                        # a model whose full forward cannot run here: per-unit passes
            ok = False
        finally:
            for h in hooks:
                h.remove()
            for m, w, a_ in states:                              # leave every quant state as it was found
                m.use_weight_quant, m.use_act_quant = w, a_
            model.train(was_training)
        if not ok:
            return None
        memo = cls(model, cali_data, batch_size)
        nb = (cali_data.size(0) + batch_size - 1) // batch_size
        for u in units:
            a, b = store.pop(id(u))
            if id(u) not in bad and len(a) == nb and len(b) == nb:
                memo.rows[id(u)] = (torch.cat([t for t, _ in a]), torch.cat([t for t, _ in b]))
            del a, b
        return memo if memo.rows else None


def save_inp_oup_data(model: QuantModel, layer: Union[QuantModule, BaseQuantBlock], cali_data: torch.Tensor,
                      asym: bool = False, act_quant: bool = False, batch_size: int = 32, keep_gpu: bool = True,
                      input_prob: bool = False):
    """-> ((inp_q, inp_fp), out_fp) if input_prob else ((inp,), out).  Everything stays on the model's device."""
    device = next(model.parameters()).device
    if asym and input_prob and keep_gpu and cali_data.is_cuda:
        fp = _FpMemo.get(model, layer, cali_data, batch_size, device)
        if fp is not None:
            # the quantised-prefix pass only (x_q): the full-precision rows come from the memo
            inp_fp, out_fp = fp
            grab = GetLayerInpOut(model, layer, device=device, asym=True, act_quant=act_quant, input_prob=True)
            inp_q = torch.cat([grab(cali_data[i:i + batch_size], fp_pass=False)[0] for i in range(0, cali_data.size(0), batch_size)])
            return (inp_q, inp_fp), out_fp
    grab = GetLayerInpOut(model, layer, device=device, asym=asym, act_quant=act_quant, input_prob=input_prob)
    # every sample is cached, the last (short) batch included: the reference caches with batch 1 and drops nothing
    # (utils.py:112-121 there iterate range(size / batch_size) with batch_size = 1, layer_opt.py:212)
    parts = [grab(cali_data[i:i + batch_size]) for i in range(0, cali_data.size(0), batch_size)]
    cols = [torch.cat([p[k] for p in parts]) for k in range(len(parts[0]))]
    if not keep_gpu:
        cols = [c.cpu() for c in cols]
    if input_prob:
        return (cols[0], cols[2]), cols[1]
    return (cols[0],), cols[1]
