"""`QuantModule`: one fake-quantised layer (reference surface: quantization/quant_layer.py:11-138), forward on the
HIP kernels: Conv2d and ConvTranspose2d (dilation 1, groups 1; a transposed conv whose output is stride x its input runs as a
stride-1 conv with the phase weight + pixel shuffle, `hipops.ops.TconvPhase`; other geometries as zero insertion + the forward conv
kernel), Linear (as a 1x1 conv), LayerNorm over the last dimension, GDN/IGDN (`f_gdn`, :142-154) and PixelShuffle."""
from typing import Union

import torch
import torch.nn as nn

from hipops import _lib as L
from hipops import ops
from lic import GDN as _LicGDN

from .quantizer import StraightThrough, UniformAffineQuantizer, to_rows

try:  # recognise real CompressAI modules when that package is installed
    from compressai.layers.gdn import GDN as _CaiGDN
    GDN_TYPES = (_LicGDN, _CaiGDN)
except Exception:  # pragma: no cover - compressai is absent in the build image
    GDN_TYPES = (_LicGDN,)
GDN = _LicGDN


def _nhwc(x):
    """NCHW tensor (any memory format) -> contiguous NHWC storage; free when x is channels_last."""
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw_view(y):
    return y.permute(0, 3, 1, 2)


def _sq(v):
    a, b = (v, v) if isinstance(v, int) else tuple(v)
    if a != b:
        raise ValueError(f"only square stride/padding is supported, got {v}")
    return int(a)


class QuantModule(nn.Module):
    def __init__(self, org_module: Union[nn.Conv2d, nn.ConvTranspose2d, nn.LayerNorm, nn.Linear, GDN, nn.PixelShuffle],
                 weight_quant_params: dict = {}, act_quant_params: dict = {}, disable_act_quant: bool = False, se_module=None):
        super().__init__()
        self.if_layer_norm = self.if_tconv = self.is_ps = False
        if isinstance(org_module, nn.ConvTranspose2d):
            self.kind, self.if_tconv = "tconv", True
            self.fwd_kwargs = dict(stride=org_module.stride, padding=org_module.padding,
                                   output_padding=org_module.output_padding, dilation=org_module.dilation,
                                   groups=org_module.groups)
        elif isinstance(org_module, nn.Conv2d):
            self.kind = "conv"
            self.fwd_kwargs = dict(stride=org_module.stride, padding=org_module.padding, dilation=org_module.dilation,
                                   groups=org_module.groups)
        elif isinstance(org_module, nn.Linear):
            self.kind, self.fwd_kwargs = "linear", dict()
        elif isinstance(org_module, nn.LayerNorm):
            self.kind, self.if_layer_norm = "layernorm", True
            self.fwd_kwargs = dict(normalized_shape=org_module.normalized_shape)
        elif isinstance(org_module, GDN_TYPES):
            self.kind = "gdn"
            self.fwd_kwargs = dict(inverse=org_module.inverse, gamma_reparam=org_module.gamma_reparam,
                                   beta_reparam=org_module.beta_reparam)
            # (bound, pedestal) of both parametrisers as Python floats, read ONCE: the buffers may live on the GPU, and a
            # float() of a device tensor inside forward is a host sync (not permitted while a stream is capturing)
            self._reparam_consts = {n: (float(p.lower_bound.bound), float(p.pedestal))
                                    for n, p in (("gamma_reparam", org_module.gamma_reparam), ("beta_reparam", org_module.beta_reparam))}
        elif isinstance(org_module, nn.PixelShuffle):
            self.kind, self.is_ps = "ps", True
            self.fwd_kwargs = org_module.upscale_factor
        else:
            raise ValueError("Not supported modules: {}".format(org_module))

        if self.kind == "gdn":
            self.weight, self.bias = org_module.gamma, org_module.beta
        elif self.is_ps:
            self.weight = self.bias = None
        else:
            self.weight, self.bias = org_module.weight, org_module.bias
        self.org_weight = None if self.weight is None else self.weight.data.clone()
        self.org_bias = None if self.bias is None else self.bias.data.clone()

        self.use_weight_quant = False
        self.use_act_quant = False
        self.disable_act_quant = disable_act_quant
        self.weight_quantizer = UniformAffineQuantizer(tconv=self.if_tconv, **weight_quant_params)
        self.act_quantizer = UniformAffineQuantizer(tconv=self.if_tconv, **act_quant_params)
        self.activation_function = nn.LeakyReLU(inplace=True) if self.is_ps else StraightThrough()
        self.ignore_reconstruction = False
        self.se_module = se_module
        self.extra_repr = org_module.extra_repr
        self.trained = False

    def _apply(self, fn, *args, **kwargs):
        """`.to()/.cuda()` also moves the detached FP copies (plain attributes in the reference, quant_layer.py:82-89)."""
        super()._apply(fn, *args, **kwargs)
        self._pack_memo = {}
        for name in ("org_weight", "org_bias"):
            t = getattr(self, name, None)
            if t is not None:
                setattr(self, name, fn(t))
        return self

    # geometry helpers used by the calibration engine ----------------------------------------------------------------
    def conv_geometry(self):
        if self.kind != "conv":
            raise ValueError(f"{self.kind} has no conv geometry")
        kw = self.fwd_kwargs
        if _sq(kw["dilation"]) != 1 or kw["groups"] != 1:
            raise NotImplementedError("dilated / grouped convolutions are not on the supported path")
        return _sq(kw["stride"]), _sq(kw["padding"])

    def fused_lrelu(self):
        return isinstance(self.activation_function, nn.LeakyReLU) and abs(self.activation_function.negative_slope - 0.01) < 1e-12

    def fused_epilogue(self):
        """Conv-kernel epilogue of the activation quant_model.py:51-54 fused into this module (None: not fusable)."""
        if self.fused_lrelu():
            return L.EPI_LRELU
        if type(self.activation_function) is nn.ReLU:
            return L.EPI_RELU
        return None

    def _reparam(self, which, t):
        """NonNegativeParametrizer forward of `fwd_kwargs[which]`, max(t, bound)^2 - pedestal, with the constants taken as Python
        floats: the reparam modules live in `fwd_kwargs` (as in the reference), so `.to(device)` on the wrapper does not move them."""
        bound, pedestal = self._reparam_consts[which]
        return torch.clamp(t, min=bound) ** 2 - pedestal

    def gdn_constants(self):
        """beta' (re-parametrised, fp32 tensor) and the gamma (bound, pedestal) pair."""
        beta = self._reparam("beta_reparam", self.bias.detach() if self.use_weight_quant else self.org_bias)
        return beta.contiguous(), self._reparam_consts["gamma_reparam"]

    # forward --------------------------------------------------------------------------------------------------------
    def _weights(self):
        if self.use_weight_quant:
            return self.weight_quantizer(self.weight), self.bias
        return self.org_weight, self.org_bias

    @staticmethod
    def _tkey(t):
        return (t.data_ptr(), t._version) if torch.is_tensor(t) else None

    def _weight_state(self):
        """Everything the effective (weight, bias) of the next forward depends on, as a comparable key; None while the weight
        quantiser has not seen its weight yet (its first call initialises the scales)."""
        if not self.use_weight_quant:
            return ("fp", self._tkey(self.org_weight), self._tkey(self.org_bias))
        q = self.weight_quantizer
        if not getattr(q, "inited", True):
            return None
        return ("q", id(q), q.n_levels, getattr(q, "soft_targets", None), self._tkey(self.weight), self._tkey(self.bias),
                self._tkey(getattr(q, "alpha", None)), self._tkey(q.delta), self._tkey(q.zero_point))

    def _weight_refs(self):
        """the tensor OBJECTS `_weight_state()` keys on: held by the memo, so that none of them can be freed and its address reused by
        another tensor with the same version count while the memo lives"""
        if not self.use_weight_quant:
            return (self.org_weight, self.org_bias)
        q = self.weight_quantizer
        return (self.weight, self.bias, getattr(q, "alpha", None), q.delta, q.zero_point)

    def weight_pack(self):
        """`ops.WeightPack` of the effective weight in kernel layout (conv: OHWI rows; transposed conv: to_rows(., tconv=True) with
        the bias; GDN: re-parametrised gamma as a 1x1 weight with beta' as its bias), memoised per quant state (one entry for the
        full-precision weights, one for the quantised ones: cache building toggles between the two for every batch) while
        `_weight_state()` is unchanged and the tensors it was made from are the SAME objects: a model evaluated or differentiated many
        times with fixed weights (cache building, evaluation, the R + lambda*D tail of the calibration loop) quantises, re-lays-out and
        splits each weight once.  A weight changed behind torch's version counters (`.data` writes, raw-pointer kernel writes) needs
        `drop_weight_pack()`: the calibration engine calls it when it hands its rounding back (`finish`), `bitwidth_refactor` and a
        new quantiser object change the key by themselves."""
        key = self._weight_state()
        memos = self.__dict__.get("_pack_memo")
        if not isinstance(memos, dict):            # (a pickled model of an earlier version carries None here)
            memos = self.__dict__["_pack_memo"] = {}
        slot = "q" if self.use_weight_quant else "fp"
        memo = memos.get(slot)
        if key is not None and memo is not None and memo[0] == key and all(a is b for a, b in zip(memo[2], self._weight_refs())):
            return memo[1]
        weight, bias = self._weights()
        weight = weight.detach()
        bias = None if bias is None else bias.detach()
        if self.kind == "gdn":
            c = weight.shape[0]
            pack = ops.WeightPack(self._reparam("gamma_reparam", weight).reshape(c, 1, 1, c),
                                  self._reparam("beta_reparam", bias))
        elif self.kind == "linear":
            pack = ops.WeightPack(weight.reshape(weight.shape[0], 1, 1, weight.shape[1]), bias)
        else:
            pack = ops.WeightPack(to_rows(weight, tconv=self.kind == "tconv"), bias)
        if key is None:
            key = self._weight_state()             # the quantiser initialised its scales inside _weights()
        memos[slot] = (key, pack, self._weight_refs())
        return pack

    def drop_weight_pack(self):
        self.__dict__["_pack_memo"] = {}

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_pack_memo"] = {}                       # derived data: not part of the saved artefact (main2.py:285-290)
        return d

    def _forward_autograd(self, input):
        """Forward whose output carries a grad_fn with respect to the INPUT (hipops.autograd; weights are constants): what the
        opt-in R + lambda*D task loss differentiates through the modules behind a unit.  The reference's dynamic activation
        quantiser works on a detached clone (quantizer.py:99-100), so an activation-quantised output has no gradient there either."""
        from hipops import autograd as A
        if self.is_ps:
            y = torch.nn.functional.pixel_shuffle(input, int(self.fwd_kwargs))
            return torch.nn.functional.leaky_relu(y, 0.01) if isinstance(self.activation_function, nn.LeakyReLU) else y
        epi = self.fused_epilogue() if self.se_module is None else None
        fuse = epi is not None
        epi = L.EPI_NONE if epi is None else epi
        if self.kind == "layernorm":
            weight, bias = self._weights()
            ns = tuple(self.fwd_kwargs["normalized_shape"])
            if len(ns) != 1 or ns[0] != input.shape[-1]:
                raise NotImplementedError("LayerNorm over more than the last dimension is not on the supported path")
            out = A.LayerNormFn.apply(input, None if weight is None else weight.detach().contiguous(),
                                      None if bias is None else bias.detach().contiguous(), float(self.fwd_kwargs.get("eps", 1e-5)))
            out = self.activation_function(out)
            if not self.disable_act_quant and self.use_act_quant and self.trained:
                out = self.act_quantizer(out, True)
            return out
        if self.kind not in ("conv", "tconv", "gdn", "linear"):
            raise NotImplementedError(f"QuantModule({self.kind}): no autograd forward")
        pack = self.weight_pack()
        if self.kind == "linear":
            out = A.LinearFn.apply(input, pack)
        elif self.kind == "conv":
            stride, pad = self.conv_geometry()
            out = A.Conv2dFn.apply(input, pack, pack.bias, stride, pad, epi)
        elif self.kind == "tconv":
            kw = self.fwd_kwargs
            out = A.ConvTranspose2dFn.apply(input, pack, pack.bias, _sq(kw["stride"]), _sq(kw["padding"]),
                                            _sq(kw["output_padding"]), epi)
        elif self.kind == "gdn":
            out = A.GDNFn.apply(input, pack, pack.bias, bool(self.fwd_kwargs["inverse"]))
        if self.se_module is not None:
            out = self.se_module(out)
        if not fuse:
            out = self.activation_function(out)
        if not self.disable_act_quant and self.use_act_quant and self.trained:
            out = self.act_quantizer(out, True)
        return out

    def forward(self, input: torch.Tensor):
        if not input.is_cuda:
            raise RuntimeError("QuantModule.forward runs on librdoptq_hip only: move the model and data to the GPU")
        if torch.is_grad_enabled() and input.requires_grad:
            return self._forward_autograd(input)
        if self.is_ps:
            y = ops.pixel_shuffle(_nhwc(input), int(self.fwd_kwargs))
            return _nchw_view(ops.lrelu(y) if isinstance(self.activation_function, nn.LeakyReLU) else y)
        epi = self.fused_epilogue() if self.se_module is None else None
        fuse = epi is not None
        epi = L.EPI_NONE if epi is None else epi
        if self.kind == "conv":
            stride, pad = self.conv_geometry()
            y = ops.conv2d_fwd_pack(_nhwc(input), self.weight_pack(), stride, pad, epilogue=epi)
            out = _nchw_view(y)
        elif self.kind == "gdn":
            x = _nhwc(input)
            y = ops.conv2d_fwd_pack(x, self.weight_pack(), 1, 0, epilogue=L.EPI_IGDN if self.fwd_kwargs["inverse"] else L.EPI_GDN,
                                    aux=x, square_input=True)
            out = _nchw_view(y)
        elif self.kind == "tconv":
            kw = self.fwd_kwargs
            if _sq(kw["dilation"]) != 1 or kw["groups"] != 1:
                raise NotImplementedError("dilated / grouped transposed convolutions are not on the supported path")
            y = ops.conv_transpose2d(_nhwc(input), None, None, _sq(kw["stride"]), _sq(kw["padding"]), _sq(kw["output_padding"]),
                                     epilogue=epi, pack=self.weight_pack())
            out = _nchw_view(y)
        elif self.kind == "layernorm":
            weight, bias = self._weights()
            bias = None if bias is None else bias.detach().contiguous()
            ns = tuple(self.fwd_kwargs["normalized_shape"])
            if len(ns) != 1 or ns[0] != input.shape[-1]:
                raise NotImplementedError("LayerNorm over more than the last dimension is not on the supported path")
            out = ops.layer_norm(input.contiguous(), None if weight is None else weight.detach().contiguous(), bias)
            fuse = False
        elif self.kind == "linear":
            pack = self.weight_pack()
            x = input.reshape(1, 1, -1, input.shape[-1]).contiguous()
            y = ops.conv2d_fwd_pack(x, pack, 1, 0, epilogue=epi)
            out = y.reshape(*input.shape[:-1], pack.w.shape[0])
        else:
            raise NotImplementedError(f"QuantModule({self.kind}) forward is not built yet (SURVEY 8f rows 3: Lu2022 / Minnen2018)")
        if self.se_module is not None:
            out = self.se_module(out)
        if not fuse:
            out = self.activation_function(out)
        if self.disable_act_quant:
            return out
        if self.use_act_quant and self.trained:
            out = self.act_quantizer(out, True)
        return out

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant


def f_gdn(x, gamma, beta, inverse, gamma_reparam, beta_reparam):
    """Functional GDN on the HIP conv kernel (reference: quant_layer.py:142-154)."""
    c = x.shape[1]
    xr = _nhwc(x)
    y = ops.conv2d_fwd(xr, gamma_reparam(gamma).reshape(c, 1, 1, c).contiguous(), beta_reparam(beta).contiguous(), 1, 0,
                       epilogue=L.EPI_IGDN if inverse else L.EPI_GDN, aux=xr, square_input=True)
    return _nchw_view(y)
