#!/usr/bin/env python3
"""Generate golden vectors for the oracle by running the REFERENCE's own code (authoring container only).

    python tools/make_golden.py            # writes tests/golden/*.npz

The reference (`/root/reference/task-oriented-PTQ`) is imported, never copied.  Its absent third-party
imports are satisfied by stand-in modules registered in `sys.modules`:

  * `compressai.*`   -> the build's restatement in `oracle/lic_oracle.py` (parity unpinned, see there)
  * `timm.models.layers`, `pytorch_msssim` -> trivial stand-ins (not on the Cheng2020 path)

and the hard-coded `device='cuda'` of `layer_opt.py:211` / `block_opt.py:211` is neutralised by mapping
`.to('cuda')` to a no-op while the reference loop runs (SURVEY 8c recipe).  The `idx` stream
(`torch.randperm`, layer_opt.py:289) and the QDrop uniform stream (`torch.rand_like`, :292) are recorded so
that any other implementation can replay the identical run.

Only data (inputs + expected outputs) is written to `tests/golden/`; no reference source or bytecode.
"""
import os
import sys
import types
import argparse
import contextlib

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/task-oriented-PTQ"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import lic_oracle as L  # noqa: E402


# ----------------------------------------------------------------------------- shims
def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    layers_layers = mod("compressai.layers.layers", ResidualBlockWithStride=L.ResidualBlockWithStride,
                        ResidualBlockUpsample=L.ResidualBlockUpsample, ResidualBlock=L.ResidualBlock,
                        subpel_conv3x3=L.subpel_conv3x3, conv3x3=L.conv3x3, conv1x1=L.conv1x1)
    gdn = mod("compressai.layers.gdn", GDN=L.GDN)
    layers = mod("compressai.layers", MaskedConv2d=L.MaskedConv2d, GDN=L.GDN, layers=layers_layers, gdn=gdn)
    em = mod("compressai.entropy_models", EntropyBottleneck=L.EntropyBottleneck,
             GaussianConditional=L.GaussianConditional)
    ans = mod("compressai.ans", BufferedRansEncoder=object, RansDecoder=object)
    mod("compressai", layers=layers, entropy_models=em, ans=ans)

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    tl = mod("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
             trunc_normal_=lambda t, std=0.02, **k: nn.init.trunc_normal_(t, std=std))
    tm = mod("timm.models", layers=tl)
    mod("timm", models=tm)
    mod("pytorch_msssim", ms_ssim=lambda *a, **k: torch.tensor(0.0))


@contextlib.contextmanager
def _cuda_is_cpu():
    """Map Tensor.to('cuda') -> identity and silence torch.cuda.empty_cache while the reference loop runs."""
    orig_to, orig_empty = torch.Tensor.to, torch.cuda.empty_cache

    def to(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return self
        return orig_to(self, *a, **k)

    torch.Tensor.to = to
    torch.cuda.empty_cache = lambda: None
    try:
        yield
    finally:
        torch.Tensor.to = orig_to
        torch.cuda.empty_cache = orig_empty


@contextlib.contextmanager
def _record_rng(idx_log, rand_log):
    orig_perm, orig_rand = torch.randperm, torch.rand_like

    def randperm(n, *a, **k):
        r = orig_perm(n, *a, **k)
        idx_log.append(r.clone())
        return r

    def rand_like(x, *a, **k):
        r = orig_rand(x, *a, **k)
        rand_log.append(r.clone())
        return r

    torch.randperm, torch.rand_like = randperm, rand_like
    try:
        yield
    finally:
        torch.randperm, torch.rand_like = orig_perm, orig_rand


def _np(t):
    return t.detach().cpu().numpy().copy()


def _randomise(model, gen):
    """Seeded non-degenerate weights (default CompressAI init leaves gamma = 0.1*I, biases ~0)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("gamma"):
                C = p.shape[0]
                g = 0.1 * torch.eye(C) + 0.02 * torch.rand(C, C, generator=gen)
                p.copy_(torch.sqrt(g + 2 ** -36))
            elif name.endswith("beta"):
                p.copy_(torch.sqrt(0.5 + torch.rand(p.shape, generator=gen) + 2 ** -36))
            elif "quantiles" in name or "_matrix" in name or "_factor" in name:
                continue
            elif p.dim() >= 2:
                fan = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=gen) - 0.5) * 2 * (3.0 / fan) ** 0.5)
            else:
                p.copy_((torch.rand(p.shape, generator=gen) - 0.5) * 0.2)


# ----------------------------------------------------------------------------- fixture groups
def golden_quantizers(out_dir):
    from quantization.quantizer import (UniformAffineQuantizer, AdaRoundQuantizer, ActQuantizer, lp_loss,
                                        round_ste)
    g = torch.Generator().manual_seed(1005)
    fx = {}
    w_conv = torch.randn(8, 4, 3, 3, generator=g) * 0.2
    w_tconv = torch.randn(4, 6, 5, 5, generator=g) * 0.1
    w_gdn = (0.1 * torch.eye(8) + 0.02 * torch.rand(8, 8, generator=g)).sqrt()
    w_vec = torch.randn(16, generator=g)
    fx["w_conv"], fx["w_tconv"], fx["w_gdn"], fx["w_vec"] = map(_np, (w_conv, w_tconv, w_gdn, w_vec))
    for tag, w, tconv in (("conv", w_conv, False), ("tconv", w_tconv, True), ("gdn", w_gdn, False),
                          ("vec", w_vec, False)):
        for method in ("max", "mse", "l1", "l2"):
            for cw in (True, False):
                for bits in (8, 4):
                    q = UniformAffineQuantizer(n_bits=bits, channel_wise=cw, scale_method=method, tconv=tconv)
                    y = q(w)
                    key = f"uaq_{tag}_{method}_{'cw' if cw else 'lw'}_{bits}"
                    fx[key + "_delta"], fx[key + "_zp"], fx[key + "_out"] = _np(torch.as_tensor(q.delta)), _np(
                        torch.as_tensor(q.zero_point)), _np(y)
    # gaussian init (layer-wise only is meaningful)
    q = UniformAffineQuantizer(n_bits=8, channel_wise=False, scale_method="gaussian")
    y = q(w_conv)
    fx["uaq_conv_gaussian_lw_8_delta"], fx["uaq_conv_gaussian_lw_8_zp"], fx["uaq_conv_gaussian_lw_8_out"] = \
        _np(torch.as_tensor(q.delta)), _np(torch.as_tensor(q.zero_point)), _np(y)

    # AdaRound
    for tag, w, tconv in (("conv", w_conv, False), ("tconv", w_tconv, True), ("gdn", w_gdn, False)):
        uaq = UniformAffineQuantizer(n_bits=8, channel_wise=True, scale_method="max", tconv=tconv)
        uaq(w)
        ada = AdaRoundQuantizer(uaq=uaq, round_mode="learned_hard_sigmoid", weight_tensor=w)
        fx[f"ada_{tag}_alpha0"] = _np(ada.alpha)
        with torch.no_grad():
            ada.alpha.add_(torch.randn(ada.alpha.shape, generator=g) * 2.0)
        fx[f"ada_{tag}_alpha"] = _np(ada.alpha)
        ada.soft_targets = True
        ys = ada(w)
        gy = torch.randn(ys.shape, generator=g)
        (ys * gy).sum().backward()
        fx[f"ada_{tag}_soft"], fx[f"ada_{tag}_gy"], fx[f"ada_{tag}_galpha"] = _np(ys), _np(gy), _np(ada.alpha.grad)
        ada.soft_targets = False
        fx[f"ada_{tag}_hard"] = _np(ada(w))
        for b in (20, 11.5, 2.0):
            rv = ada.get_soft_targets()
            fx[f"ada_{tag}_roundloss_b{b}"] = _np(0.01 * (1 - ((rv - .5).abs() * 2).pow(b)).sum())

    # ActQuant
    a4 = torch.randn(2, 5, 6, 7, generator=g) * 3
    a4[:, 2] = 0.25                      # constant channel -> range clamp 1e-6
    a3 = torch.randn(2, 9, 6, generator=g)
    a2 = torch.randn(7, 4, generator=g)
    for tag, a in (("a4", a4), ("a3", a3), ("a2", a2)):
        fx[f"act_{tag}_in"], fx[f"act_{tag}_out"] = _np(a), _np(ActQuantizer(a))

    # lp_loss / round_ste
    p1, p2 = torch.randn(3, 4, 5, 5, generator=g), torch.randn(3, 4, 5, 5, generator=g)
    fx["lp_pred"], fx["lp_tgt"] = _np(p1), _np(p2)
    for p in (2.0, 1.0, 3.5):
        fx[f"lp_none_{p}"] = _np(lp_loss(p1, p2, p=p))
        fx[f"lp_all_{p}"] = _np(lp_loss(p1, p2, p=p, reduction="all"))
    fx["round_ste"] = _np(round_ste(p1 * 3))
    np.savez_compressed(os.path.join(out_dir, "quantizers.npz"), **fx)
    print("quantizers.npz", len(fx), "arrays")


def golden_quantizer_ties(out_dir):
    """Scale inits on weights whose per-channel ranges are exactly symmetric (min = -max, as kaiming-uniform initialised layers
    nearly are): -min/delta sits on x.5 and the zero point is decided by how the reference's expression is evaluated
    (quantizer.py:296: a Python float divided by a tensor = reciprocal-multiply in torch)."""
    from quantization.quantizer import UniformAffineQuantizer
    g = torch.Generator().manual_seed(77)
    fx = {}
    w_conv = (torch.rand(24, 16, 3, 3, generator=g) * 2 - 1) * torch.rand(24, 1, 1, 1, generator=g)
    w_tconv = (torch.rand(12, 20, 5, 5, generator=g) * 2 - 1) * 0.05
    w_lin = (torch.rand(40, 32, generator=g) * 2 - 1) * torch.rand(40, 1, generator=g)
    # force min = -max per output channel
    for w, dim in ((w_conv, 0), (w_tconv, 1), (w_lin, 0)):
        wt = w.transpose(0, dim) if dim else w
        flat = wt.reshape(wt.shape[0], -1)
        mx = flat.abs().amax(1)
        idx = flat.abs().argmax(1)
        for c in range(flat.shape[0]):
            flat[c, idx[c]] = mx[c]
            flat[c, (idx[c] + 1) % flat.shape[1]] = -mx[c]
    fx["w_conv"], fx["w_tconv"], fx["w_lin"] = map(_np, (w_conv, w_tconv, w_lin))
    for tag, w, tconv in (("conv", w_conv, False), ("tconv", w_tconv, True), ("lin", w_lin, False)):
        for cw in (True, False):
            for bits in (8, 6, 4):
                q = UniformAffineQuantizer(n_bits=bits, channel_wise=cw, scale_method="max", tconv=tconv)
                y = q(w)
                key = f"uaq_{tag}_{'cw' if cw else 'lw'}_{bits}"
                fx[key + "_delta"], fx[key + "_zp"], fx[key + "_out"] = _np(torch.as_tensor(q.delta)), _np(
                    torch.as_tensor(q.zero_point)), _np(y)
    # 'max_scale' (range * (n_bits+2)/8, evaluated in Python double) and symmetric grids on the same tie-prone weights
    for tag, w, tconv in (("conv", w_conv, False), ("lin", w_lin, False)):
        for method, sym in (("max_scale", False), ("max", True), ("max_scale", True)):
            for cw in (True, False):
                for bits in (8, 6, 4):
                    q = UniformAffineQuantizer(n_bits=bits, symmetric=sym, channel_wise=cw, scale_method=method, tconv=tconv)
                    y = q(w)
                    key = f"uaq_{tag}_{method}{'_sym' if sym else ''}_{'cw' if cw else 'lw'}_{bits}"
                    fx[key + "_delta"], fx[key + "_zp"], fx[key + "_out"] = _np(torch.as_tensor(q.delta)), _np(
                        torch.as_tensor(q.zero_point)), _np(y)
    np.savez_compressed(os.path.join(out_dir, "quantizer_ties.npz"), **fx)
    print("quantizer_ties.npz", len(fx), "arrays")


def golden_temp_decay(out_dir):
    from quantization.utils import LinearTempDecay
    fx = {}
    for t_max, warm in ((50, 0.2), (20000, 0.2), (10, 0.0)):
        d = LinearTempDecay(t_max, rel_start_decay=warm, start_b=20, end_b=2)
        fx[f"b_{t_max}_{warm}"] = np.array([float(d(t)) for t in range(1, t_max + 1)], dtype=np.float64)
    np.savez_compressed(os.path.join(out_dir, "temp_decay.npz"), **fx)
    print("temp_decay.npz")


def _toy_qnn(N, seed, wq=None, aq=None):
    from quantization import QuantModel
    torch.manual_seed(seed)
    model = L.Cheng2020Anchor(N=N)
    _randomise(model, torch.Generator().manual_seed(seed))
    model.eval()
    wq = wq or {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    aq = aq or {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq, is_cheng=True)
    qnn.eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    return model, qnn


def golden_model_surgery(out_dir):
    """Structure of QuantModel(Cheng2020Anchor) as the reference builds it: module order, types, fused act fns."""
    from quantization import QuantModule, BaseQuantBlock
    _, qnn = _toy_qnn(8, 1005)
    rows = []
    for name, m in qnn.model.named_modules():
        if isinstance(m, QuantModule):
            kind = "ps" if m.is_ps else ("gdn" if m.fwd_func.__name__ == "f_gdn" else m.fwd_func.__name__)
            rows.append(f"{name}|QuantModule|{kind}|{type(m.activation_function).__name__}|{int(m.disable_act_quant)}")
        elif isinstance(m, BaseQuantBlock):
            rows.append(f"{name}|{type(m).__name__}|||")
    units = []

    def walk(mod, prefix=""):
        for n, c in mod.named_children():
            if isinstance(c, (QuantModule, BaseQuantBlock)):
                units.append(prefix + n + "|" + type(c).__name__)
            else:
                walk(c, prefix + n + ".")
    walk(qnn)
    np.savez_compressed(os.path.join(out_dir, "surgery.npz"), modules=np.array(rows), units=np.array(units))
    print("surgery.npz", len(rows), "modules", len(units), "units")


def _export_unit(unit, kind):
    """Tensors of one reference unit (QuantModule or Cheng block) -> flat dict."""
    from quantization import QuantModule
    d = {}

    def one(tag, qm):
        d[f"{tag}.weight"] = _np(qm.org_weight)
        d[f"{tag}.act"] = np.array(int(isinstance(qm.activation_function, nn.LeakyReLU)))
        if qm.org_bias is not None:
            d[f"{tag}.bias"] = _np(qm.org_bias)
        if qm.weight_quantizer.delta is not None:
            d[f"{tag}.delta"] = _np(qm.weight_quantizer.delta)
            d[f"{tag}.zp"] = _np(qm.weight_quantizer.zero_point)
        if hasattr(qm.weight_quantizer, "alpha") and qm.weight_quantizer.alpha is not None:
            d[f"{tag}.alpha"] = _np(qm.weight_quantizer.alpha)

    if kind == "layer":
        one("layer", unit)
    else:
        for n, m in unit.named_modules():
            if isinstance(m, QuantModule) and not m.is_ps:
                one(n.replace(".0", ""), m)
    return d


def golden_recon(out_dir, iters=12):
    """Run the reference's layer_reconstruction / block_reconstruction verbatim on a toy Cheng2020 (N=8,
    64x64 crops, 6 calibration images, batch 2) for a representative set of units, in the order main2.py
    would visit them, recording caches, RNG streams and the trained alphas."""
    import logging
    from quantization import QuantModule, BaseQuantBlock, layer_reconstruction, block_reconstruction
    import quantization.layer_opt as lo
    import quantization.block_opt as bo
    import quantization.utils as qu

    N, n_img, B = 8, 6, 2
    model, qnn = _toy_qnn(N, 1005)
    g = torch.Generator().manual_seed(77)
    cali = torch.rand(n_img, 3, 64, 64, generator=g)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])                                   # scale init, main2.py:194-198
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Cheng2020")
    kwargs = dict(cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
                  b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)

    # capture the caches the reference builds (save_inp_oup_data is looked up in the opt modules' namespaces)
    captured = {}
    orig_save = qu.save_inp_oup_data

    def save_spy(*a, **k):
        r = orig_save(*a, **k)
        captured["inp_q"], captured["inp_fp"], captured["out"] = r[0][0].clone(), r[0][1].clone(), r[1].clone()
        return r
    lo.save_inp_oup_data = save_spy
    bo.save_inp_oup_data = save_spy

    # capture per-iteration losses from the reference LossFunction objects
    losses = []

    def wrap_loss(cls):
        orig_call = cls.__call__

        def call(self, pred, tgt, quant_net_out=None, cali_data=None, grad=None):
            r = orig_call(self, pred, tgt, quant_net_out, cali_data, grad)
            losses.append(float(r))
            return r
        cls.__call__ = call
    wrap_loss(lo.LossFunction)
    wrap_loss(bo.LossFunction)

    wanted = {"g_a.0": "rbws", "g_a.1": "rb", "g_a.6": "layer", "g_s.1": "rbu", "g_s.7.0": "layer",
              "h_s.2.0": "layer", "entropy_parameters.0": "layer", "context_prediction": "layer"}
    fx = {"cali": _np(cali), "meta": np.array([N, n_img, B, iters])}
    order = []
    # full toy-model state (original FP parameters and buffers) so that another implementation can rebuild the very same
    # QuantModel and replay the cache-building passes
    for k, v in model.state_dict().items():
        key = k.replace(".org_module", "")
        fx["state/" + key] = _np(v)
    for n_, m_ in qnn.model.named_modules():
        if isinstance(m_, QuantModule) and m_.org_weight is not None:
            fx["org/" + n_ + ".weight"] = _np(m_.org_weight)
            if m_.org_bias is not None:
                fx["org/" + n_ + ".bias"] = _np(m_.org_bias)
    full_order = []

    qnn.set_quant_state(True, False)
    qnn.model.g_s[-1][0].set_quant_state(True, False)

    def recon(mod, prefix=""):
        for name, m in mod.named_children():
            full = prefix + name
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                full_order.append(full)
                if full not in wanted:
                    # the reference would train it; for the fixture mark it trained with nearest rounding
                    # so later units see a quantised prefix (same effect on set_mode()).
                    if isinstance(m, QuantModule):
                        m.trained = True
                    else:
                        for mm in m.modules():
                            if isinstance(mm, (QuantModule, BaseQuantBlock)):
                                mm.trained = True
                    continue
                kind = wanted[full]
                idx_log, rand_log = [], []
                del losses[:]
                pre = _export_unit(m, kind)
                with _cuda_is_cpu(), _record_rng(idx_log, rand_log):
                    if isinstance(m, QuantModule):
                        layer_reconstruction(qnn, m, name, **kwargs)
                    else:
                        block_reconstruction(qnn, m, name, **kwargs)
                post = _export_unit(m, kind)
                tag = full
                order.append(f"{tag}|{kind}")
                for k, v in pre.items():
                    if not k.endswith(".alpha"):
                        fx[f"{tag}/{k}"] = v
                for k, v in post.items():
                    if k.endswith(".alpha"):
                        fx[f"{tag}/{k}_final"] = v
                fx[f"{tag}/inp_q"], fx[f"{tag}/inp_fp"], fx[f"{tag}/out"] = (_np(captured[k]) for k in
                                                                              ("inp_q", "inp_fp", "out"))
                fx[f"{tag}/idx"] = np.stack([_np(t[:B]) for t in idx_log]).astype(np.int64)
                fx[f"{tag}/rand"] = np.stack([_np(t) for t in rand_log]).astype(np.float32)
                fx[f"{tag}/loss"] = np.array(losses, dtype=np.float64)
                # hard-rounded output of the trained unit on the first two cached inputs
                with torch.no_grad():
                    m.set_quant_state(True, False)
                    fx[f"{tag}/hard_out"] = _np(m(captured["inp_q"][:2]))
                print(f"  {tag:24s} {kind:6s} loss[0]={losses[0]:.6f} loss[-1]={losses[-1]:.6f}")
            else:
                recon(m, full + ".")

    logging.disable(logging.CRITICAL)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        pass
    import io
    buf = io.StringIO()
    _stdout = sys.stdout
    recon_out = []

    class _Filter:
        def write(self, s):
            if "Forward init time" not in s and s.strip():
                _stdout.write(s if s.endswith("\n") else s + "\n")

        def flush(self):
            _stdout.flush()
    sys.stdout = _Filter()
    try:
        recon(qnn.model)
    finally:
        sys.stdout = _stdout
    fx["order"] = np.array(order)
    fx["full_order"] = np.array(full_order)
    np.savez_compressed(os.path.join(out_dir, "recon_toy.npz"), **fx)
    print("recon_toy.npz", len(fx), "arrays")


def golden_recon_minnen(out_dir, iters=10):
    """Verbatim layer_reconstruction runs of the reference on a toy Minnen2018 mean-scale model (N=8, M=12, 64x64 crops):
    5x5 stride-2 conv, GDN as its own unit, transposed conv, IGDN, transposed conv with a fused LeakyReLU."""
    import logging
    from quantization import QuantModel, QuantModule, layer_reconstruction
    import quantization.layer_opt as lo
    import quantization.utils as qu
    torch.manual_seed(2018)
    N, M, n_img, B = 8, 12, 6, 2
    model = L.MeanScaleHyperprior(N=N, M=M)
    _randomise(model, torch.Generator().manual_seed(2018))
    model.eval()
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    aq = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq)
    qnn.eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    cali = torch.rand(n_img, 3, 64, 64, generator=torch.Generator().manual_seed(78))
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Minnen2018")
    kwargs = dict(cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
                  b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    captured = {}
    orig_save = qu.save_inp_oup_data

    def save_spy(*a, **k):
        r = orig_save(*a, **k)
        captured["inp_q"], captured["inp_fp"], captured["out"] = r[0][0].clone(), r[0][1].clone(), r[1].clone()
        return r
    lo.save_inp_oup_data = save_spy
    losses = []
    orig_call = lo.LossFunction.__call__

    def call(self, pred, tgt, quant_net_out=None, cali_data=None, grad=None):
        r = orig_call(self, pred, tgt, quant_net_out, cali_data, grad)
        losses.append(float(r))
        return r
    lo.LossFunction.__call__ = call
    wanted = ["g_a.0", "g_a.1", "g_s.0", "g_s.1", "h_s.0"]
    fx = {"cali": _np(cali), "meta": np.array([N, M, n_img, B, iters])}
    for n_, m_ in qnn.model.named_modules():
        if isinstance(m_, QuantModule) and m_.org_weight is not None:
            fx["org/" + n_ + ".weight"] = _np(m_.org_weight)
            if m_.org_bias is not None:
                fx["org/" + n_ + ".bias"] = _np(m_.org_bias)
    for k, v in model.entropy_bottleneck.state_dict().items():
        fx["state/entropy_bottleneck." + k] = _np(v)
    full_order = []
    qnn.set_quant_state(True, False)
    qnn.model.g_s[-1].set_quant_state(True, False)
    logging.disable(logging.CRITICAL)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        for coder in ("g_a", "g_s", "h_a", "h_s"):
            for name, m in getattr(qnn.model, coder).named_children():
                if not isinstance(m, QuantModule):
                    continue
                full = f"{coder}.{name}"
                full_order.append(full)
                if full not in wanted:
                    m.trained = True
                    continue
                idx_log, rand_log = [], []
                del losses[:]
                with _cuda_is_cpu(), _record_rng(idx_log, rand_log):
                    layer_reconstruction(qnn, m, name, **kwargs)
                fx[f"{full}/kind"] = np.array("tconv" if m.if_tconv else ("gdn" if m.fwd_func.__name__ == "f_gdn" else "conv"))
                fx[f"{full}/act"] = np.array(int(isinstance(m.activation_function, nn.LeakyReLU)))
                if m.fwd_func.__name__ == "f_gdn":
                    fx[f"{full}/inverse"] = np.array(int(m.fwd_kwargs["inverse"]))
                else:
                    fx[f"{full}/geom"] = np.array([m.fwd_kwargs["stride"][0], m.fwd_kwargs["padding"][0],
                                                   m.fwd_kwargs.get("output_padding", (0, 0))[0]])
                fx[f"{full}/weight"] = _np(m.org_weight)
                fx[f"{full}/bias"] = _np(m.org_bias)
                fx[f"{full}/delta"] = _np(m.weight_quantizer.delta)
                fx[f"{full}/zp"] = _np(m.weight_quantizer.zero_point)
                fx[f"{full}/alpha_final"] = _np(m.weight_quantizer.alpha)
                for k in ("inp_q", "inp_fp", "out"):
                    fx[f"{full}/{k}"] = _np(captured[k])
                fx[f"{full}/idx"] = np.stack([_np(t[:B]) for t in idx_log]).astype(np.int64)
                fx[f"{full}/rand"] = np.stack([_np(t) for t in rand_log]).astype(np.float32)
                fx[f"{full}/loss"] = np.array(losses, dtype=np.float64)
                with torch.no_grad():
                    m.set_quant_state(True, False)
                    fx[f"{full}/hard_out"] = _np(m(captured["inp_q"][:2]))
    finally:
        sys.stdout = _stdout
        logging.disable(logging.NOTSET)
        lo.LossFunction.__call__ = orig_call
        lo.save_inp_oup_data = orig_save
    fx["full_order"] = np.array(full_order)
    fx["order"] = np.array(wanted)
    np.savez_compressed(os.path.join(out_dir, "recon_minnen.npz"), **fx)
    print("recon_minnen.npz", len(fx), "arrays;", " ".join(f"{w}:{float(fx[w + '/loss'][0]):.4g}->{float(fx[w + '/loss'][-1]):.4g}" for w in wanted))


def golden_recon_attn(out_dir, iters=10):
    """Verbatim layer_reconstruction runs of the reference inside an attention block of a toy Cheng2020-attn (N=8, 64x64
    crops): 1x1 conv + ReLU, 3x3 conv + ReLU, 1x1 conv without activation, and the mask branch's closing 1x1 conv.
    recon_model (main2.py:227-253) reaches these by recursing through AttentionBlock -> Sequential -> ResidualUnit ->
    Sequential and passes the LOCAL child name ('0', '2', '4', '3') as layer_name."""
    import logging
    from quantization import QuantModel, QuantModule, BaseQuantBlock, layer_reconstruction
    import quantization.layer_opt as lo
    import quantization.utils as qu
    torch.manual_seed(2020)
    N, n_img, B = 8, 6, 2
    model = L.Cheng2020Attention(N=N)
    _randomise(model, torch.Generator().manual_seed(2020))
    model.eval()
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    aq = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq, is_cheng=True)
    qnn.eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    cali = torch.rand(n_img, 3, 64, 64, generator=torch.Generator().manual_seed(79))
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Cheng2020")
    kwargs = dict(cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
                  b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    captured = {}
    orig_save = qu.save_inp_oup_data

    def save_spy(*a, **k):
        r = orig_save(*a, **k)
        captured["inp_q"], captured["inp_fp"], captured["out"] = r[0][0].clone(), r[0][1].clone(), r[1].clone()
        return r
    lo.save_inp_oup_data = save_spy
    losses = []
    orig_call = lo.LossFunction.__call__

    def call(self, pred, tgt, quant_net_out=None, cali_data=None, grad=None):
        r = orig_call(self, pred, tgt, quant_net_out, cali_data, grad)
        losses.append(float(r))
        return r
    lo.LossFunction.__call__ = call
    wanted = ["g_a.3.conv_a.0.conv.0", "g_a.3.conv_a.0.conv.2", "g_a.3.conv_a.0.conv.4", "g_a.3.conv_b.3"]
    fx = {"cali": _np(cali), "meta": np.array([N, n_img, B, iters])}
    for k, v in model.state_dict().items():
        fx["state/" + k.replace(".org_module", "")] = _np(v)
    for n_, m_ in qnn.model.named_modules():
        if isinstance(m_, QuantModule) and m_.org_weight is not None:
            fx["org/" + n_ + ".weight"] = _np(m_.org_weight)
            if m_.org_bias is not None:
                fx["org/" + n_ + ".bias"] = _np(m_.org_bias)
    full_order = []
    qnn.set_quant_state(True, False)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    logging.disable(logging.CRITICAL)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")

    def recon(mod, prefix=""):
        for name, m in mod.named_children():
            full = prefix + name
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                full_order.append(full)
                if full not in wanted:
                    for mm in m.modules():
                        if isinstance(mm, (QuantModule, BaseQuantBlock)):
                            mm.trained = True
                    continue
                idx_log, rand_log = [], []
                del losses[:]
                with _cuda_is_cpu(), _record_rng(idx_log, rand_log):
                    layer_reconstruction(qnn, m, name, **kwargs)
                act = m.activation_function
                fx[f"{full}/kind"] = np.array("conv")
                fx[f"{full}/act"] = np.array(1 if isinstance(act, nn.LeakyReLU) else (2 if isinstance(act, nn.ReLU) else 0))
                fx[f"{full}/geom"] = np.array([m.fwd_kwargs["stride"][0], m.fwd_kwargs["padding"][0], 0])
                fx[f"{full}/weight"] = _np(m.org_weight)
                fx[f"{full}/bias"] = _np(m.org_bias)
                fx[f"{full}/delta"] = _np(m.weight_quantizer.delta)
                fx[f"{full}/zp"] = _np(m.weight_quantizer.zero_point)
                fx[f"{full}/alpha_final"] = _np(m.weight_quantizer.alpha)
                for k in ("inp_q", "inp_fp", "out"):
                    fx[f"{full}/{k}"] = _np(captured[k])
                fx[f"{full}/idx"] = np.stack([_np(t[:B]) for t in idx_log]).astype(np.int64)
                fx[f"{full}/rand"] = np.stack([_np(t) for t in rand_log]).astype(np.float32)
                fx[f"{full}/loss"] = np.array(losses, dtype=np.float64)
                with torch.no_grad():
                    m.set_quant_state(True, False)
                    fx[f"{full}/hard_out"] = _np(m(captured["inp_q"][:2]))
            else:
                recon(m, full + ".")
    try:
        recon(qnn.model)
        # W8 forward of the whole model after these units are trained (the rest nearest-rounded): x_hat and likelihoods
        qnn.set_quant_state(True, False)
        qnn.eval()           # layer_reconstruction leaves the model in train mode; main2.py:269 evaluates `qnn.eval()`
        with torch.no_grad():
            out = qnn(cali[:2])
            y = qnn.model.g_a(cali[:2])
        fx["w8/x_hat"] = _np(out["x_hat"])
        fx["w8/y"] = _np(y)                         # analysis latents (continuous: comparable at fp32 tolerance)
        fx["w8/y_hat"] = _np(qnn.model.gaussian_conditional.quantize(y, "dequantize"))
        fx["w8/lik_y"] = _np(out["likelihoods"]["y"])
        fx["w8/lik_z"] = _np(out["likelihoods"]["z"])
        with torch.no_grad():                        # synthesis transform stage by stage, from the rounded latents
            h = torch.from_numpy(fx["w8/y_hat"])
            for k, stage in enumerate(qnn.model.g_s):
                h = stage(h)
                fx[f"w8/g_s.{k}"] = _np(h)
    finally:
        sys.stdout = _stdout
        logging.disable(logging.NOTSET)
        lo.LossFunction.__call__ = orig_call
        lo.save_inp_oup_data = orig_save
    fx["full_order"] = np.array(full_order)
    fx["order"] = np.array(wanted)
    np.savez_compressed(os.path.join(out_dir, "recon_attn.npz"), **fx)
    print("recon_attn.npz", len(fx), "arrays;", " ".join(f"{w.split('.', 2)[2]}:{float(fx[w + '/loss'][0]):.4g}->{float(fx[w + '/loss'][-1]):.4g}" for w in wanted))


def golden_recon_nic(out_dir, iters=6):
    """Lu2022 path on the reference's OWN model code (models/nic_cvt.py: NIC, models/layers.py: RSTB...) at toy width
    (embed 16, latent 32, 64x64 crops): verbatim layer_reconstruction / block_reconstruction runs for a 5x5 stride-2 conv
    with a long FP tail, a shifted-window RSTB with a tail, an RSTB at window == resolution, a 1x1-token RSTB, a transposed
    conv, and the closing 5x5 transposed conv; then stage-wise W8 and W8A8 forwards of the calibrated model."""
    import logging
    from models.nic_cvt import NIC
    from quantization import QuantModel, QuantModule, BaseQuantBlock, layer_reconstruction, block_reconstruction
    import quantization.layer_opt as lo
    import quantization.block_opt as bo
    import quantization.utils as qu
    cfg = dict(height=64, width=64, in_chans=3, embed_dim=16, latent_dim=32, window_size=8, mlp_ratio=2.0, qkv_bias=True,
               qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False)
    torch.manual_seed(2022)
    model = NIC(cfg)
    gen = torch.Generator().manual_seed(2022)
    _randomise(model, gen)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".norm" in name and name.endswith("weight"):
                p.copy_(1.0 + (torch.rand(p.shape, generator=gen) - 0.5) * 0.6)
    model.eval()
    n_img, B = 6, 2
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    aq = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    fx = {"meta": np.array([cfg["embed_dim"], cfg["latent_dim"], cfg["window_size"], n_img, B, iters])}
    for k, v in model.state_dict().items():
        fx["state/" + k] = _np(v)
    cali = torch.rand(n_img, 3, 64, 64, generator=torch.Generator().manual_seed(80))
    fx["cali"] = _np(cali)
    with torch.no_grad():
        out_fp = model(cali[:2])
        fx["fp/y"] = _np(model.g_a(cali[:2]))
        fx["fp/x_hat"] = _np(out_fp["x_hat"])
        fx["fp/lik_y"] = _np(out_fp["likelihoods"]["y"])
        fx["fp/lik_z"] = _np(out_fp["likelihoods"]["z"])
    qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq)
    qnn.eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Lu2022")
    kwargs = dict(cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
                  b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    captured = {}
    orig_save = qu.save_inp_oup_data

    def save_spy(*a, **k):
        r = orig_save(*a, **k)
        captured["inp_q"], captured["inp_fp"], captured["out"] = r[0][0].clone(), r[0][1].clone(), r[1].clone()
        return r
    lo.save_inp_oup_data = save_spy
    bo.save_inp_oup_data = save_spy
    losses = []
    originals = {}
    for mod_ in (lo, bo):
        cls = mod_.LossFunction
        originals[mod_] = cls.__call__

        def make(orig_call):
            def call(self, pred, tgt, quant_net_out=None, cali_data=None, grad=None):
                r = orig_call(self, pred, tgt, quant_net_out, cali_data, grad)
                losses.append(float(r))
                return r
            return call
        cls.__call__ = make(cls.__call__)
    wanted = ["g_a0", "g_a1", "g_a7", "h_a3", "h_s1", "g_s7"]
    full_order = []
    qnn.set_quant_state(True, False)
    qnn.model.g_s7.set_quant_state(True, False)
    logging.disable(logging.CRITICAL)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        for name, m in qnn.model.named_children():
            if not isinstance(m, (QuantModule, BaseQuantBlock)):
                continue
            full_order.append(name)
            if name not in wanted:
                for mm in m.modules():
                    if isinstance(mm, (QuantModule, BaseQuantBlock)):
                        mm.trained = True
                continue
            idx_log, rand_log = [], []
            del losses[:]
            with _cuda_is_cpu(), _record_rng(idx_log, rand_log):
                (layer_reconstruction if isinstance(m, QuantModule) else block_reconstruction)(qnn, m, name, **kwargs)
            for k in ("inp_q", "inp_fp", "out"):
                fx[f"{name}/{k}"] = _np(captured[k])
            fx[f"{name}/idx"] = np.stack([_np(t[:B]) for t in idx_log]).astype(np.int64)
            fx[f"{name}/rand"] = np.stack([_np(t) for t in rand_log]).astype(np.float32)
            fx[f"{name}/loss"] = np.array(losses, dtype=np.float64)
            inner = [("", m)] if isinstance(m, QuantModule) else \
                [(n_ + ".", mm) for n_, mm in m.named_modules() if isinstance(mm, QuantModule)]
            for n_, mm in inner:
                fx[f"{name}/{n_}delta"] = _np(mm.weight_quantizer.delta)
                fx[f"{name}/{n_}zp"] = _np(mm.weight_quantizer.zero_point)
                fx[f"{name}/{n_}alpha_final"] = _np(mm.weight_quantizer.alpha)
            with torch.no_grad():
                m.set_quant_state(True, False)
                xin = captured["inp_q"][:2]
                fx[f"{name}/hard_out"] = _np(m(xin) if isinstance(m, QuantModule) else m(xin, tuple(xin.shape[2:4])))
        qnn.eval()

        def stagewise(tag):
            with torch.no_grad():
                h = cali[:2]
                for nm in [n for n in full_order if n.startswith("g_a")]:
                    mod = getattr(qnn.model, nm)
                    h = mod(h) if isinstance(mod, QuantModule) else mod(h, tuple(h.shape[2:4]))
                    fx[f"{tag}/{nm}"] = _np(h)
                h = torch.round(h)
                fx[f"{tag}/y_hat"] = _np(h)
                for nm in [n for n in full_order if n.startswith("g_s")]:
                    mod = getattr(qnn.model, nm)
                    h = mod(h) if isinstance(mod, QuantModule) else mod(h, tuple(h.shape[2:4]))
                    fx[f"{tag}/{nm}"] = _np(h)
        qnn.set_quant_state(True, False)
        stagewise("w8")
        qnn.set_quant_state(True, True)
        qnn.model.g_s7.set_quant_state(True, False)
        stagewise("w8a8")
    finally:
        sys.stdout = _stdout
        logging.disable(logging.NOTSET)
        for mod_, c in originals.items():
            mod_.LossFunction.__call__ = c
        lo.save_inp_oup_data = orig_save
        bo.save_inp_oup_data = orig_save
    fx["full_order"] = np.array(full_order)
    fx["order"] = np.array(wanted)
    np.savez_compressed(os.path.join(out_dir, "recon_nic.npz"), **fx)
    print("recon_nic.npz", len(fx), "arrays;", " ".join(f"{w}:{float(fx[w + '/loss'][0]):.4g}->{float(fx[w + '/loss'][-1]):.4g}" for w in wanted))


def golden_recon_toy_aq(out_dir, iters=6):
    """`main2.py --act_quant` on the toy Cheng2020 (N=8): the reference calibrates g_a.0 and g_a.1 with act_quant=True (their
    caches are built from the W8A8 prefix, batch 1), then builds the caches of g_a.2.  Records the trained alphas of the two
    blocks and the three cache tensors of g_a.2 -- the activation-quantised cache-building pass (utils.py:195-258, set_mode :28-35)."""
    import logging
    from quantization import QuantModule, BaseQuantBlock, block_reconstruction
    import quantization.block_opt as bo
    import quantization.utils as qu
    N, n_img, B = 8, 4, 2
    model, qnn = _toy_qnn(N, 1005)
    cali = torch.rand(n_img, 3, 64, 64, generator=torch.Generator().manual_seed(81))
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Cheng2020")
    kwargs = dict(cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
                  b_range=(20, 2), warmup=0.2, act_quant=True, opt_mode="mse", config=None, args=args)
    fx = {"cali": _np(cali), "meta": np.array([N, n_img, B, iters])}
    for n_, m_ in qnn.model.named_modules():
        if isinstance(m_, QuantModule) and m_.org_weight is not None:
            fx["org/" + n_ + ".weight"] = _np(m_.org_weight)
            if m_.org_bias is not None:
                fx["org/" + n_ + ".bias"] = _np(m_.org_bias)
    for k, v in model.entropy_bottleneck.state_dict().items():
        fx["state/entropy_bottleneck." + k] = _np(v)
    qnn.set_quant_state(True, True)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    logging.disable(logging.CRITICAL)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        units = list(qnn.model.g_a.named_children())
        for name, u in units[:2]:
            with _cuda_is_cpu():
                block_reconstruction(qnn, u, name, **kwargs)
            for n_, mm in u.named_modules():
                if isinstance(mm, QuantModule) and mm.org_weight is not None:
                    fx[f"g_a.{name}/{n_}.alpha_final"] = _np(mm.weight_quantizer.alpha)
        with _cuda_is_cpu():
            (inp_q, inp_fp), out = qu.save_inp_oup_data(qnn, units[2][1], cali, True, True, batch_size=1, input_prob=True)
        fx["g_a.2/inp_q"], fx["g_a.2/inp_fp"], fx["g_a.2/out"] = _np(inp_q), _np(inp_fp), _np(out)
    finally:
        sys.stdout = _stdout
        logging.disable(logging.NOTSET)
    np.savez_compressed(os.path.join(out_dir, "recon_toy_aq.npz"), **fx)
    d = float(np.abs(fx["g_a.2/inp_q"] - fx["g_a.2/inp_fp"]).max())
    print("recon_toy_aq.npz", len(fx), "arrays; max |inp_q - inp_fp| =", d)


def golden_blocks(out_dir):
    """Forward (and input/weight gradients) of the reference Cheng2020 quant blocks with nearest-rounded weights."""
    from quantization.quant_block import QuantRBWS, QuantRBU, QuantRB
    from quantization.quant_layer import QuantModule
    g = torch.Generator().manual_seed(4242)
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    aq = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
    fx = {}
    specs = (("rbws", L.ResidualBlockWithStride(8, 8, stride=2), QuantRBWS), ("rbws3", L.ResidualBlockWithStride(3, 8, stride=2), QuantRBWS),
             ("rbu", L.ResidualBlockUpsample(8, 8, 2), QuantRBU), ("rb", L.ResidualBlock(8, 8), QuantRB))
    for tag, blk, Q in specs:
        _randomise(blk, g)
        qb = Q(blk, wq, aq)
        cin = 3 if tag == "rbws3" else 8
        x = torch.randn(2, cin, 12, 12, generator=g)
        fx[f"{tag}/x"] = _np(x)
        for state, (w, a, trained) in {"fp": (False, False, False), "w8": (True, False, False),
                                       "w8a8": (True, True, True)}.items():
            qb.set_quant_state(w, a)
            qb.trained = trained
            for m in qb.modules():
                if isinstance(m, QuantModule):
                    m.trained = trained
            with torch.no_grad():
                fx[f"{tag}/y_{state}"] = _np(qb(x.clone()))
        for k, v in _export_unit(qb, "block").items():
            fx[f"{tag}/{k}"] = v
    np.savez_compressed(os.path.join(out_dir, "blocks.npz"), **fx)
    print("blocks.npz", len(fx), "arrays")


def golden_bd(out_dir):
    """BD_RATE / BD_PSNR of the reference's BD-rate.py (imported by file path) on synthetic RD curves."""
    import importlib.util
    import warnings
    spec = importlib.util.spec_from_file_location("ref_bd_rate", os.path.join(REF, "BD-rate.py"))
    bd = importlib.util.module_from_spec(spec)
    if not hasattr(np, "trapz"):
        np.trapz = np.trapezoid
    spec.loader.exec_module(bd)
    rng = np.random.default_rng(5)
    fx = {}
    for case in range(4):
        r1 = np.sort(rng.uniform(0.1, 1.2, 6))
        p1 = 28 + 6 * np.log(1 + 4 * r1) + rng.normal(0, 0.02, 6)
        r2 = r1 * rng.uniform(1.0, 1.08)
        p2 = p1 - rng.uniform(0.0, 0.4)
        fx[f"c{case}_R1"], fx[f"c{case}_P1"], fx[f"c{case}_R2"], fx[f"c{case}_P2"] = r1, p1, r2, p2
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for pw in (0, 1):
                fx[f"c{case}_bdrate_{pw}"] = np.float64(bd.BD_RATE(r1, p1, r2, p2, piecewise=pw))
                fx[f"c{case}_bdpsnr_{pw}"] = np.float64(bd.BD_PSNR(r1, p1, r2, p2, piecewise=pw))
    np.savez_compressed(os.path.join(out_dir, "bd_rate.npz"), **fx)
    print("bd_rate.npz", len(fx), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None, help="run a single generator, e.g. recon_attn")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    _install_shims()
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    if a.only:
        globals()["golden_" + a.only](a.out)
        return
    golden_bd(a.out)
    golden_quantizers(a.out)
    golden_quantizer_ties(a.out)
    golden_temp_decay(a.out)
    golden_model_surgery(a.out)
    golden_blocks(a.out)
    golden_recon(a.out)
    golden_recon_minnen(a.out)
    golden_recon_attn(a.out)
    golden_recon_nic(a.out)
    golden_recon_toy_aq(a.out)


if __name__ == "__main__":
    main()
