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

    def __call__(self, model_input):
        self.model.eval()
        self.model.set_quant_state(False, False)
        handle = self.layer.register_forward_hook(self.data_saver)
        x = model_input.to(self.device)
        with torch.no_grad():
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


def save_inp_oup_data(model: QuantModel, layer: Union[QuantModule, BaseQuantBlock], cali_data: torch.Tensor,
                      asym: bool = False, act_quant: bool = False, batch_size: int = 32, keep_gpu: bool = True,
                      input_prob: bool = False):
    """-> ((inp_q, inp_fp), out_fp) if input_prob else ((inp,), out).  Everything stays on the model's device."""
    device = next(model.parameters()).device
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
