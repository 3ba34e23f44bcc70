"""ORACLE helper (test infrastructure): the reconstruction-unit schedule of a Cheng2020-anchor model, expressed with the
oracle's QOps.  Unit order and fused activations follow what the reference's QuantModel surgery + recon_model produce
(quant_model.py:23-62, main2.py:227-253; pinned by tests/golden/surgery.npz)."""
import torch
import torch.nn as nn

from . import lic_oracle as L
from .rdo_oracle import QOp


def _conv_op(c: nn.Conv2d, act=None):
    return QOp("conv", c.weight.detach().clone(), None if c.bias is None else c.bias.detach().clone(),
               stride=c.stride[0], padding=c.padding[0], act=act)


def _gdn_op(g: L.GDN):
    return QOp("igdn" if g.inverse else "gdn", g.gamma.detach().clone(), g.beta.detach().clone())


def unit_of(module):
    """(kind, ops) for a Cheng2020 block or a bare conv."""
    if isinstance(module, L.ResidualBlockWithStride):
        ops = {"conv1": _conv_op(module.conv1), "conv2": _conv_op(module.conv2), "gdn": _gdn_op(module.gdn)}
        if module.skip is not None:
            ops["skip"] = _conv_op(module.skip)
        return "rbws", ops
    if isinstance(module, L.ResidualBlockUpsample):
        return "rbu", {"subpel_conv": _conv_op(module.subpel_conv[0]), "conv": _conv_op(module.conv),
                       "igdn": _gdn_op(module.igdn), "upsample": _conv_op(module.upsample[0])}
    if isinstance(module, L.ResidualBlock):
        ops = {"conv1": _conv_op(module.conv1), "conv2": _conv_op(module.conv2)}
        if module.skip is not None:
            ops["skip"] = _conv_op(module.skip)
        return "rb", ops
    raise TypeError(type(module))


def schedule(model: L.Cheng2020Anchor):
    """[(name, kind, ops, module)] in recon_model order; PixelShuffle pseudo-units are omitted (layer_opt.py:245-246)."""
    out = []

    def seq(prefix, s):
        mods = list(s.named_children())
        for i, (n, m) in enumerate(mods):
            full = f"{prefix}.{n}"
            if isinstance(m, (L.ResidualBlockWithStride, L.ResidualBlockUpsample, L.ResidualBlock)):
                out.append((full,) + unit_of(m) + (m,))
            elif isinstance(m, nn.Conv2d):
                fused = i + 1 < len(mods) and isinstance(mods[i + 1][1], nn.LeakyReLU)
                out.append((full, "layer", {"layer": _conv_op(m, "lrelu" if fused else None)}, m))
            elif isinstance(m, nn.Sequential):          # subpel_conv3x3 = Sequential(conv, PixelShuffle)
                out.append((f"{full}.0", "layer", {"layer": _conv_op(m[0])}, m[0]))
    for name in ("g_a", "g_s", "h_a", "h_s", "entropy_parameters"):
        seq(name, getattr(model, name))
    out.append(("context_prediction", "layer", {"layer": _conv_op(model.context_prediction)}, model.context_prediction))
    return out


def capture_io(model, sched, x):
    """One FP forward of the whole model capturing each unit's (input, output) -- synthetic caches for timing runs."""
    io, hooks = {}, []
    for name, kind, ops, mod in sched:
        def hook(m, inp, out, name=name, kind=kind, ops=ops):
            y = out
            if kind == "layer" and ops["layer"].act == "lrelu":
                y = torch.nn.functional.leaky_relu(out, 0.01)
            io[name] = (inp[0].detach().clone(), y.detach().clone())
        hooks.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        model.eval()
        model(x)
    for h in hooks:
        h.remove()
    return io
