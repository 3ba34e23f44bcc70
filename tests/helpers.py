"""Shared builders for the parity tests: turn golden-fixture tensors into oracle QOps and into product modules."""
import numpy as np
import torch
import torch.nn as nn

from oracle import rdo_oracle as O

T = torch.from_numpy

UNIT_OPS = {"rbws": ["conv1", "conv2", "gdn", "skip"], "rb": ["conv1", "conv2", "skip"],
            "rbu": ["subpel_conv", "conv", "igdn", "upsample"], "layer": ["layer"]}
LAYER_GEOM = {"g_a.6": (2, 1), "g_s.7.0": (1, 1), "h_s.2.0": (1, 1), "entropy_parameters.0": (1, 0),
              "context_prediction": (1, 2)}
RECON_UNITS = [("g_a.0", "rbws"), ("g_a.1", "rb"), ("g_a.6", "layer"), ("g_s.1", "rbu"), ("g_s.7.0", "layer"),
               ("h_s.2.0", "layer"), ("entropy_parameters.0", "layer"), ("context_prediction", "layer")]
WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
AQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}


def oracle_ops(fx, tag, kind):
    ops = {}
    for n in UNIT_OPS[kind]:
        if f"{tag}/{n}.weight" not in fx:
            continue
        w = T(fx[f"{tag}/{n}.weight"])
        b = T(fx[f"{tag}/{n}.bias"]) if f"{tag}/{n}.bias" in fx else None
        if n in ("gdn", "igdn"):
            op = O.QOp(n, w, b)
        elif kind == "layer":
            s, p = LAYER_GEOM[tag]
            op = O.QOp("conv", w, b, stride=s, padding=p, act="lrelu" if int(fx[f"{tag}/{n}.act"]) else None)
        else:
            stride = 2 if (kind == "rbws" and n in ("conv1", "skip")) else 1
            op = O.QOp("conv", w, b, stride=stride, padding=w.shape[-1] // 2)
        op.delta, op.zp = T(fx[f"{tag}/{n}.delta"]), T(fx[f"{tag}/{n}.zp"])
        ops[n] = op
    return ops


def _conv(w, b, stride, pad, dev):
    co, ci, k, _ = w.shape
    c = nn.Conv2d(ci, co, k, stride=stride, padding=pad, bias=b is not None)
    with torch.no_grad():
        c.weight.copy_(w)
        if b is not None:
            c.bias.copy_(b)
    return c.to(dev)


def product_unit(fx, tag, kind, dev="cuda", wq=None):
    """-> (unit module of the product `quantization` package, engine kind, engine module dict) from fixture tensors."""
    import lic
    from quantization.quant_block import QuantRB, QuantRBU, QuantRBWS
    from quantization.quant_layer import QuantModule
    from quantization.recon import _unit_modules
    g = lambda n: T(fx[f"{tag}/{n}"]) if f"{tag}/{n}" in fx else None
    WQ = wq or globals()["WQ"]
    if kind == "layer":
        s, p = LAYER_GEOM[tag]
        qm = QuantModule(_conv(g("layer.weight"), g("layer.bias"), s, p, dev), WQ, AQ)
        if int(fx[f"{tag}/layer.act"]):
            qm.activation_function = nn.LeakyReLU(inplace=True)
        unit = qm
    else:
        w1 = g("conv1.weight") if kind != "rbu" else g("conv.weight")
        C = w1.shape[0]
        if kind == "rb":
            blk = lic.ResidualBlock(w1.shape[1], C)
            names = {"conv1": blk.conv1, "conv2": blk.conv2}
        elif kind == "rbws":
            blk = lic.ResidualBlockWithStride(w1.shape[1], C, stride=2)
            names = {"conv1": blk.conv1, "conv2": blk.conv2, "skip": blk.skip, "gdn": blk.gdn}
        else:
            cin = g("subpel_conv.weight").shape[1]
            blk = lic.ResidualBlockUpsample(cin, C, 2)
            names = {"subpel_conv": blk.subpel_conv[0], "conv": blk.conv, "upsample": blk.upsample[0], "igdn": blk.igdn}
        with torch.no_grad():
            for n, mod in names.items():
                if n in ("gdn", "igdn"):
                    mod.gamma.copy_(g(f"{n}.weight"))
                    mod.beta.copy_(g(f"{n}.bias"))
                else:
                    mod.weight.copy_(g(f"{n}.weight"))
                    mod.bias.copy_(g(f"{n}.bias"))
        blk = blk.to(dev)
        unit = {"rb": QuantRB, "rbws": QuantRBWS, "rbu": QuantRBU}[kind](blk, WQ, AQ)
    unit = unit.to(dev)
    k, mods = _unit_modules(unit)
    assert k == kind
    return unit, kind, mods


def nhwc(a, dev="cuda"):
    return T(np.ascontiguousarray(a.transpose(0, 2, 3, 1))).to(dev)


ATTN_UNITS = ["g_a.3.conv_a.0.conv.0", "g_a.3.conv_a.0.conv.2", "g_a.3.conv_a.0.conv.4", "g_a.3.conv_b.3"]
MINNEN_UNITS = ["g_a.0", "g_a.1", "g_s.0", "g_s.1", "h_s.0"]


def minnen_oracle_op(fx, tag):
    """QOp of one layer unit of the toy Minnen2018 golden (conv / tconv / gdn / igdn)."""
    kind = str(fx[f"{tag}/kind"])
    w, b = T(fx[f"{tag}/weight"]), T(fx[f"{tag}/bias"])
    act = {0: None, 1: "lrelu", 2: "relu"}[int(fx[f"{tag}/act"])]
    if kind == "gdn":
        op = O.QOp("igdn" if int(fx[f"{tag}/inverse"]) else "gdn", w, b)
    else:
        s, p, opad = (int(v) for v in fx[f"{tag}/geom"])
        op = O.QOp(kind, w, b, stride=s, padding=p, output_padding=opad, act=act)
    op.delta, op.zp = T(fx[f"{tag}/delta"]), T(fx[f"{tag}/zp"])
    return op


def minnen_product_module(fx, tag, dev="cuda"):
    """The product QuantModule for the same unit."""
    import lic
    from quantization.quant_layer import QuantModule
    kind = str(fx[f"{tag}/kind"])
    w, b = T(fx[f"{tag}/weight"]), T(fx[f"{tag}/bias"])
    if kind == "gdn":
        m = lic.GDN(w.shape[0], inverse=bool(int(fx[f"{tag}/inverse"])))
        with torch.no_grad():
            m.gamma.copy_(w); m.beta.copy_(b)
    else:
        s, p, opad = (int(v) for v in fx[f"{tag}/geom"])
        if kind == "tconv":
            m = nn.ConvTranspose2d(w.shape[0], w.shape[1], w.shape[2], stride=s, padding=p, output_padding=opad)
        else:
            m = nn.Conv2d(w.shape[1], w.shape[0], w.shape[2], stride=s, padding=p)
        with torch.no_grad():
            m.weight.copy_(w); m.bias.copy_(b)
    qm = QuantModule(m.to(dev), WQ, AQ).to(dev)
    if int(fx[f"{tag}/act"]):
        qm.activation_function = nn.LeakyReLU(inplace=True) if int(fx[f"{tag}/act"]) == 1 else nn.ReLU(inplace=True)
    return qm


# ---- natural-image statistics (missing 3 / next 3) ---------------------------------------------------------------------
def kodak_crops(golden_dir, n=None):
    """[n, 3, 256, 256] floats in [0, 1]: the committed crops of the reference's Kodak images (tools/make_kodak_fixture.py), divided by 255
    as the reference's ToTensor does (datasets/dataset.py:8-12)."""
    import os
    crops = np.load(os.path.join(golden_dir, "kodak_crops.npz"))["crops"]
    x = torch.from_numpy(crops[: n or len(crops)].astype(np.float32) / 255.0)
    return x.permute(0, 3, 1, 2).contiguous()


def trained_like_(ref, g, decades=2.5, probe=None):
    """'Trained-like' parameters for an oracle model (oracle/lic_oracle.py; the product side copies the state): there are no checkpoints
    offline (ReadMe.md:56-57), so the properties of trained codecs that matter to the plane path are drawn instead --
      * conv weights with Laplace tails (a few weights per channel far outside the bulk: they set delta = range / 255 and push most
        weights into a handful of levels) and per-output-channel scales spread log-uniformly over `decades` decades (channel-wise
        ranges over orders of magnitude), normalised so that the layer as a whole preserves the activation variance (He);
      * biases of the order of the activations;
      * GDN: a NON-diagonal gamma (diagonal 0.05-0.4, off-diagonal mass of the same order with Laplace tails) and beta log-uniform in
        [0.1, 10] -- stored through the re-parametrisation sqrt(. + 2^-36) like a trained CompressAI state.  An IGDN MULTIPLIES by
        sqrt(beta + gamma . x^2): a decoder that does not explode keeps gamma . x^2 of the order of beta, so with `probe` (a batch of
        images) every IGDN's gamma is scaled by the power of two that puts its mean gamma . x^2 at ~1 for the activations the model
        itself produces on that batch (one forward pass, in execution order; a power of two so that the last bits of the probe
        activations on another CPU do not change the parameters).
    Heavy-tailed activations behind the GDNs and smooth image regions then do the rest.  Seeded: both sides draw the same model."""
    from oracle import lic_oracle as L
    ped = 2.0 ** -36

    def laplace(shape):
        u = torch.rand(shape, generator=g) - 0.5
        return -torch.sign(u) * torch.log1p(-2 * u.abs().clamp(max=0.4999999))

    with torch.no_grad():
        for name, p in ref.named_parameters():
            if "entropy_bottleneck" in name:
                continue
            if p.dim() == 4:
                co, fan = p.shape[0], p[0].numel()
                s = 10.0 ** ((torch.rand(co, generator=g) - 0.5) * decades)
                s = s / s.pow(2).mean().sqrt()
                gain = 0.5 if name.startswith("g_s") else 1.0       # (a synthesis transform contracts towards [0, 1] pixels)
                p.copy_(laplace(p.shape) / 2 ** 0.5 * gain * (2.0 / fan) ** 0.5 * s.view(-1, 1, 1, 1))
            elif p.dim() == 1 and name.endswith("bias"):
                p.copy_(0.1 * laplace(p.shape))
        for m in ref.modules():
            if isinstance(m, L.GDN):
                c = m.gamma.shape[0]
                diag = 0.05 * 8.0 ** torch.rand(c, generator=g)
                off = (0.3 / c) * laplace((c, c)).abs() * 10.0 ** ((torch.rand(c, 1, generator=g) - 0.5) * 1.5)
                m.gamma.copy_(torch.sqrt(torch.diag(diag) + off + ped))
                m.beta.copy_(torch.sqrt(10.0 ** (2.0 * torch.rand(c, generator=g) - 1.0) + ped))
        cp = getattr(ref, "context_prediction", None)
        if cp is not None:
            cp.weight.data *= cp.mask          # a trained checkpoint carries the masked weight; the wrapper never re-applies the mask
        if probe is not None:
            import math

            def tame(m, args):
                x = args[0]
                gam = m.gamma.pow(2) - ped
                pool = float((x.pow(2).mean(dim=(0, 2, 3)) * gam.mean(dim=0)).sum())     # mean over rows of gamma . E[x^2]
                m.gamma.copy_(torch.sqrt(gam * 2.0 ** -round(math.log2(max(pool, 1e-30))) + ped))
            hooks = [m.register_forward_pre_hook(tame) for m in ref.modules() if isinstance(m, L.GDN) and m.inverse]
            try:
                ref(probe)
            finally:
                for h in hooks:
                    h.remove()


def trained_like_nic_(model, g, decades=2.0):
    """The same idea for the Lu2022 `NIC` (lic/nic.py; Swin blocks): Laplace-tailed conv / transposed-conv / Linear weights with
    per-output-channel scales over `decades` decades (variance-preserving per layer), LayerNorm gains log-uniform in [0.3, 3] with small
    offsets, relative-position tables and biases with mass.  The residual + LayerNorm structure keeps the activations bounded."""
    import torch.nn as nn

    def laplace(shape):
        u = torch.rand(shape, generator=g) - 0.5
        return -torch.sign(u) * torch.log1p(-2 * u.abs().clamp(max=0.4999999))

    def scales(n, dec):
        s = 10.0 ** ((torch.rand(n, generator=g) - 0.5) * dec)
        return s / s.pow(2).mean().sqrt()

    with torch.no_grad():
        for name, m in model.named_modules():
            if "entropy_bottleneck" in name:
                continue
            if isinstance(m, nn.ConvTranspose2d):
                cin, cout, k, _ = m.weight.shape
                fan = cin * k * k / (m.stride[0] * m.stride[1])
                m.weight.copy_(laplace(m.weight.shape) / 2 ** 0.5 * (1.0 / fan) ** 0.5 * scales(cout, decades).view(1, -1, 1, 1))
            elif isinstance(m, nn.Conv2d):
                fan = m.weight[0].numel()
                m.weight.copy_(laplace(m.weight.shape) / 2 ** 0.5 * (1.0 / fan) ** 0.5 * scales(m.weight.shape[0], decades).view(-1, 1, 1, 1))
                if hasattr(m, "mask"):
                    m.weight.mul_(m.mask)
            elif isinstance(m, nn.Linear):
                m.weight.copy_(laplace(m.weight.shape) / 2 ** 0.5 * (1.0 / m.weight.shape[1]) ** 0.5 * scales(m.weight.shape[0], 0.75 * decades).view(-1, 1))
            elif isinstance(m, nn.LayerNorm):
                m.weight.copy_(10.0 ** (torch.rand(m.weight.shape, generator=g) - 0.5))
                m.bias.copy_(0.1 * laplace(m.bias.shape))
                continue
            else:
                continue
            if getattr(m, "bias", None) is not None:
                m.bias.copy_(0.05 * laplace(m.bias.shape))
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
