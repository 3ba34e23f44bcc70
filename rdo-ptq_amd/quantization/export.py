"""True-integer export of a calibrated QuantModel (in the spirit of light-uniform-PTQ/quant_int/quant_layer.py:116-122,
which overwrites weights with uint8 levels): for every trained QuantModule the unsigned integer levels, the per-channel
scale `delta` and the zero point, such that  w_q = (levels - zero_point) * delta  is exactly the hard-rounded weight the
module uses at inference (`AdaRoundQuantizer.forward` with soft_targets=False, quantizer.py:441-449)."""
from collections import OrderedDict

import torch

from .quant_layer import QuantModule
from .quantizer import AdaRoundQuantizer


def integer_state(qnn) -> "OrderedDict[str, dict]":
    """name -> {levels (uint8, or int32 above 8 bits; logical weight shape), delta, zero_point, n_bits, bias, kind}."""
    out = OrderedDict()
    for name, m in qnn.named_modules():
        if not isinstance(m, QuantModule) or m.org_weight is None:
            continue
        q = m.weight_quantizer
        if not q.inited if hasattr(q, "inited") else False:
            continue
        w = m.org_weight
        d, z = q.delta.to(w.device), q.zero_point.to(w.device)
        if isinstance(q, AdaRoundQuantizer):
            up = (q.alpha.detach() >= 0).to(w.dtype)
            x_int = torch.floor(w / d) + up
        else:
            x_int = torch.round(w / d)
        levels = torch.clamp(x_int + z, 0, q.n_levels - 1)
        out[name] = {"levels": levels.to(torch.uint8 if q.n_bits <= 8 else torch.int32).cpu(), "delta": d.detach().cpu(), "zero_point": z.detach().cpu(),
                     "n_bits": q.n_bits, "bias": None if m.org_bias is None else m.org_bias.detach().cpu(), "kind": m.kind}
    return out


def dequantize(entry) -> torch.Tensor:
    return (entry["levels"].to(torch.float32) - entry["zero_point"]) * entry["delta"]
