"""ORACLE (test infrastructure, not product code): the reference driver's calibration FLOW restated on the CPU.

`main2.py:214-282` of /root/reference/task-oriented-PTQ -- wrap the model, visit every reconstruction unit in `recon_model` order,
calibrate each one on caches produced by the ALREADY CALIBRATED prefix, then evaluate the W8 and the W8A8 model -- for the
Sequential-indexed CompressAI families (Cheng2020-anchor, Minnen2018 mean-scale).  It chains the pieces the other oracle
modules restate one by one:

  * unit list, fused activations, the PixelShuffle wrapper's LeakyReLU            quant_model.py:23-62, quant_layer.py:100,107-111
  * cache building: one full-precision pass (x_fp, target) and one pass with the
    trained prefix hard-quantised and everything else full precision (x_q)        quantization/utils.py:175-258, :28-35 (set_mode),
                                                                                  layer_opt.py:15-43 (find_unquantized_module)
  * the hot loop of one unit                                                      rdo_oracle.reconstruct_unit (layer_opt.py:236-315)
  * evaluation: pad, forward, crop, clamp, PSNR, bpp                              test_datasets.py:76-117
  * which modules get the dynamic activation quantiser in the W8A8 evaluation     quant_layer.py:126-133 (`use_act_quant and trained`,
    `disable_act_quant`), quant_model.py:66-70 (last QuantModule of the child order), main2.py:258-263 (last decoder layer)

Only tests may import this module.  Parity pins: the per-unit pieces are pinned by tests/golden (see rdo_oracle.py); the flow itself
has no golden of its own (a verbatim reference flow needs CompressAI, absent here) -- it is the composition of pinned pieces."""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lic_oracle as L
from . import rdo_oracle as O
from .cheng_units import unit_of
from .rdo_oracle import QOp


@dataclass
class FlowUnit:
    name: str                    # dotted path, e.g. "g_a.0", "g_s.7.0"
    local: str                   # the name recon_model hands to layer_/block_reconstruction: the child's own name (main2.py:232-247)
    kind: str                    # 'layer' | 'rb' | 'rbws' | 'rbu'
    ops: Dict[str, QOp]
    shuffle: int = 0             # upscale factor of the PixelShuffle wrapper that follows in the same Sequential (applies LeakyReLU)
    act_quant_ok: bool = True
    trained: bool = False


class _Tap(Exception):
    def __init__(self, inp, out):
        self.inp, self.out = inp, out


def _layer_op(m, act):
    if isinstance(m, nn.ConvTranspose2d):
        return QOp("tconv", m.weight.detach().clone(), None if m.bias is None else m.bias.detach().clone(), stride=m.stride[0],
                   padding=m.padding[0], output_padding=m.output_padding[0], act=act)
    if isinstance(m, L.GDN):
        return QOp("igdn" if m.inverse else "gdn", m.gamma.detach().clone(), m.beta.detach().clone())
    return QOp("conv", m.weight.detach().clone(), None if m.bias is None else m.bias.detach().clone(), stride=m.stride[0],
               padding=m.padding[0], act=act)


class FlowOracle:
    """The wrapped model as a list of FlowUnits per coder + the reference's flow over them."""

    CODERS = ("g_a", "g_s", "h_a", "h_s", "entropy_parameters")        # CompressAI child order (entropy models in between carry no units)

    def __init__(self, model: nn.Module, n_bits=8, channel_wise=True, scale_method="max"):
        self.model = model.eval()
        self.plan: Dict[str, List[FlowUnit]] = {}
        blocks = (L.ResidualBlockWithStride, L.ResidualBlockUpsample, L.ResidualBlock)
        for coder in self.CODERS:
            seq = getattr(model, coder, None)
            if seq is None:
                continue
            mods = list(seq.named_children())
            units = []
            for i, (n, m) in enumerate(mods):
                nxt = mods[i + 1][1] if i + 1 < len(mods) else None
                if isinstance(m, blocks):
                    units.append(FlowUnit(f"{coder}.{n}", n, *unit_of(m)))
                elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, L.GDN)):
                    act = "lrelu" if isinstance(nxt, nn.LeakyReLU) else ("relu" if isinstance(nxt, nn.ReLU) else None)   # quant_model.py:51-54
                    units.append(FlowUnit(f"{coder}.{n}", n, "layer", {"layer": _layer_op(m, act)}))
                elif isinstance(m, nn.Sequential):           # subpel_conv3x3 = Sequential(conv, PixelShuffle): recon_model recurses into it
                    conv, ps = m[0], m[1]
                    units.append(FlowUnit(f"{coder}.{n}.0", "0", "layer", {"layer": _layer_op(conv, None)}, shuffle=ps.upscale_factor))
                elif isinstance(m, (nn.LeakyReLU, nn.ReLU)):
                    continue                                 # fused into the module in front of it / a no-op behind a PixelShuffle wrapper
                else:
                    raise TypeError(f"flow oracle: {type(m).__name__} in {coder}")
            self.plan[coder] = units
        cp = getattr(model, "context_prediction", None)
        if cp is not None:
            # the wrapper takes the MaskedConv2d's weight as it is at wrap time and never re-applies the mask
            self.plan["context_prediction"] = [FlowUnit("context_prediction", "context_prediction", "layer", {"layer": _layer_op(cp, None)})]
        self.units: List[FlowUnit] = [u for c in self.plan.values() for u in c]
        for u in self.units:
            for op in u.ops.values():
                op.n_bits, op.channel_wise, op.scale_method = n_bits, channel_wise, scale_method
                op.init_scale()
        self.units[-1].act_quant_ok = False                  # disable_network_output_quantization: last QuantModule of the child order
        self.plan["g_s"][-1].act_quant_ok = False            # main2.py:258-263: g_s[-1][0] / g_s[-1] keeps activation quantisation off
        self.by_name = {u.name: u for u in self.units}

    # ------------------------------------------------------------------------------------------------------------ forward
    def _set_modes(self, state: str):
        """'fp': everything full precision; 'prefix': trained units hard-quantised, the rest full precision (set_mode after
        find_unquantized_module); 'quant': every weight quantised (trained: learned rounding, untrained: nearest)."""
        for u in self.units:
            for op in u.ops.values():
                if state == "fp" or (state == "prefix" and not u.trained):
                    op.mode = "fp"
                elif u.trained:
                    op.mode, op.soft = "ada", False
                else:
                    op.mode = "uaq"

    def _run(self, coder, h, act_quant, tap):
        for u in self.plan[coder]:
            x_in = h
            aq = bool(act_quant and u.trained and u.act_quant_ok)
            h = O.UNIT_FORWARD[u.kind](u.ops, h, aq=aq, inner_aq=aq)
            if tap == u.name:
                raise _Tap(x_in, h)
            if u.shuffle:
                h = F.leaky_relu(F.pixel_shuffle(h, u.shuffle), 0.01)
        return h

    def forward(self, x, act_quant=False, tap=None):
        m = self.model
        y = self._run("g_a", x, act_quant, tap)
        z = self._run("h_a", y, act_quant, tap)
        z_hat, z_lik = m.entropy_bottleneck(z)
        params = self._run("h_s", z_hat, act_quant, tap)
        if "context_prediction" in self.plan:
            y_hat = m.gaussian_conditional.quantize(y, "dequantize")
            ctx = self._run("context_prediction", y_hat, act_quant, tap)
            gp = self._run("entropy_parameters", torch.cat((params, ctx), dim=1), act_quant, tap)
            scales_hat, means_hat = gp.chunk(2, 1)
            _, y_lik = m.gaussian_conditional(y, scales_hat, means=means_hat)
        else:
            scales_hat, means_hat = params.chunk(2, 1)
            y_hat, y_lik = m.gaussian_conditional(y, scales_hat, means=means_hat)
        x_hat = self._run("g_s", y_hat, act_quant, tap)
        return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}

    # ------------------------------------------------------------------------------------------------------------ caches + loop
    def caches(self, name, cali, batch=8):
        """-> (x_q, x_fp, target) of unit `name` for all calibration images (utils.py:175-258 with asym=True, act_quant=False)."""
        xq, xf, tg = [], [], []
        with torch.no_grad():
            for i in range(0, cali.shape[0], batch):
                x = cali[i:i + batch]
                for state, keep in (("fp", (xf, tg)), ("prefix", (xq, None))):
                    self._set_modes(state)
                    try:
                        self.forward(x, act_quant=False, tap=name)
                        raise RuntimeError(f"unit {name} was not reached")
                    except _Tap as t:
                        keep[0].append(t.inp.clone())
                        if keep[1] is not None:
                            keep[1].append(t.out.clone())
        return torch.cat(xq), torch.cat(xf), torch.cat(tg)

    @staticmethod
    def unit_seed(process_seed: int, local_name: str) -> int:
        """The build's QDrop stream key of a unit: process seed ^ CRC32 of the name handed to layer_/block_reconstruction."""
        return (process_seed ^ zlib.crc32(local_name.encode())) & 0xFFFFFFFF

    def recon_model(self, cali, idx_streams, process_seed, *, iters, batch_size, weight=0.01, input_prob=0.5, b_range=(20, 2),
                    warmup=0.2, on_unit=None):
        """main2.py:227-253 + layer_opt.py / block_opt.py around the loop: units in order, each calibrated on caches of the
        calibrated prefix.  `idx_streams[name]` = the [iters, B] mini-batch index table of that unit (layer_opt.py:289).
        `iters`: one count for all units (main2.py --iters_w) or a callable unit name -> count (tests with per-unit horizons)."""
        logs = {}
        for u in self.units:
            xq, xf, tg = self.caches(u.name, cali)
            self._set_modes("prefix")
            seed = self.unit_seed(process_seed, u.local)
            logs[u.name] = O.reconstruct_unit(u.kind, u.ops, xq, xf, tg, iters=iters(u.name) if callable(iters) else iters,
                                              batch_size=batch_size,
                                              idx_stream=idx_streams[u.name],
                                              mask_fn=lambda i, shape, seed=seed: O.qdrop_keep_mask_nhwc(seed, i, shape, input_prob),
                                              input_prob=input_prob, weight=weight, b_range=b_range, warmup=warmup)
            u.trained = True
            if on_unit is not None:
                on_unit(u)
        return logs

    # ------------------------------------------------------------------------------------------------------------ evaluation
    def evaluate(self, images, p=64, act_quant=False):
        """test_datasets.py:76-117: mean PSNR (dB) and bpp over `images` ([1,3,h,w] in [0,1]) of the fully weight-quantised model."""
        self._set_modes("quant")
        psnr = bpp = 0.0
        with torch.no_grad():
            for x in images:
                h, w = x.shape[2], x.shape[3]
                H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
                left, top = (W - w) // 2, (H - h) // 2
                xp = F.pad(x, (left, W - w - left, top, H - h - top))
                out = self.forward(xp, act_quant=act_quant)
                rec = out["x_hat"][:, :, top:top + h, left:left + w].clamp(0, 1)
                psnr += 10 * math.log10(1.0 / float(((x - rec) ** 2).mean()))
                bpp += sum(float((-torch.log2(v)).sum()) for v in out["likelihoods"].values()) / (xp.shape[0] * H * W)
        return psnr / len(images), bpp / len(images)
