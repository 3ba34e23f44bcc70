"""BatchNorm folding (reference: quantization/fold_bn.py:68-80).  LIC models carry no BatchNorm, so for them this is a
walk that changes nothing; the fold is still implemented because QuantModel always calls it (quant_model.py:15-16)."""
import torch
import torch.nn as nn

from .quantizer import StraightThrough

_BN = (nn.BatchNorm1d, nn.BatchNorm2d)
_ABSORBING = (nn.Conv2d, nn.Linear)


def _folded(conv, bn):
    std = torch.sqrt(bn.running_var + bn.eps)
    scale = (bn.weight / std) if bn.affine else (1.0 / std)
    shift = (bn.bias - bn.weight * bn.running_mean / std) if bn.affine else (-bn.running_mean / std)
    w = conv.weight.data * scale.view(-1, *([1] * (conv.weight.dim() - 1)))
    b = shift if conv.bias is None else conv.bias.data * scale + shift
    return w, b


def fold_bn_into_conv(conv, bn):
    w, b = _folded(conv, bn)
    conv.weight.data = w
    if conv.bias is None:
        conv.bias = nn.Parameter(b)
    else:
        conv.bias.data = b


def search_fold_and_remove_bn(model):
    model.eval()
    prev = None
    for name, child in model.named_children():
        if isinstance(child, _BN) and isinstance(prev, _ABSORBING):
            fold_bn_into_conv(prev, child)
            setattr(model, name, StraightThrough())
        elif isinstance(child, _ABSORBING):
            prev = child
        else:
            prev = search_fold_and_remove_bn(child)
    return prev
