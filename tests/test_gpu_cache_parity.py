"""End-to-end parity of the cache-building path: product `QuantModel` + `save_inp_oup_data` on the GPU
against the caches the REFERENCE built for the same toy Cheng2020 (tests/golden/recon_toy.npz: full model state, calibration
images, and for each visited unit the reference's (inp_q, inp_fp, out) plus its trained alphas).

Units are visited in the reference's recon_model order; after each one the reference's trained rounding (alpha_final) is
installed so that the quantised prefix seen by later units is the same as in the reference run."""
import os

import numpy as np
import pytest
import torch

from helpers import AQ, WQ, T

pytestmark = pytest.mark.gpu


def _get(root, dotted):
    m = root
    for p in dotted.split("."):
        m = m[int(p)] if p.isdigit() else getattr(m, p)
    return m


def test_save_inp_oup_data_matches_reference_caches(golden_dir):
    import lic
    from quantization import BaseQuantBlock, QuantModel, QuantModule
    from quantization.quantizer import AdaRoundQuantizer, to_rows
    from quantization.utils import save_inp_oup_data
    fx = np.load(os.path.join(golden_dir, "recon_toy.npz"))
    N, n_img, B, iters = (int(v) for v in fx["meta"])
    torch.manual_seed(0)
    qnn = QuantModel(lic.Cheng2020Anchor(N=N), WQ, AQ, is_cheng=True).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    with torch.no_grad():
        for name, m in qnn.model.named_modules():
            if isinstance(m, QuantModule) and m.org_weight is not None:
                w = T(fx[f"org/{name}.weight"]).cuda()
                m.weight.data.copy_(w); m.org_weight.copy_(w)
                if m.org_bias is not None:
                    b = T(fx[f"org/{name}.bias"]).cuda()
                    m.bias.data.copy_(b); m.org_bias.copy_(b)
        eb = qnn.model.entropy_bottleneck
        for k, v in eb.state_dict().items():
            v.copy_(T(fx[f"state/entropy_bottleneck.{k}"]).cuda())
    cali = T(fx["cali"]).cuda()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])                                               # scale init (main2.py:194-198)
    qnn.set_quant_state(True, False)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    wanted = {str(o).split("|")[0]: str(o).split("|")[1] for o in fx["order"]}
    checked = 0
    for full in (str(s) for s in fx["full_order"]):
        unit = _get(qnn.model, full)
        inner = [unit] if isinstance(unit, QuantModule) else [m for m in unit.modules() if isinstance(m, QuantModule)]
        if full in wanted:
            (inp_q, inp_fp), out = save_inp_oup_data(qnn, unit, cali, asym=True, act_quant=False, batch_size=2, input_prob=True)
            for got, key in ((inp_q, "inp_q"), (inp_fp, "inp_fp"), (out, "out")):
                ref = T(fx[f"{full}/{key}"])
                assert tuple(got.shape) == tuple(ref.shape), (full, key)
                err = float((got.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
                assert err < 3e-5, (full, key, err)
            checked += 1
            # install the reference's trained rounding for this unit
            for n, m in (("layer", unit),) if isinstance(unit, QuantModule) else \
                    [(nm.replace(".0", ""), mm) for nm, mm in unit.named_modules() if isinstance(mm, QuantModule) and not mm.is_ps]:
                alpha = T(fx[f"{full}/{n}.alpha_final"]).cuda()
                ada = AdaRoundQuantizer(uaq=m.weight_quantizer, round_mode="learned_hard_sigmoid",
                                        weight_tensor=m.org_weight.data, alpha_rows=to_rows(alpha))
                ada.soft_targets = False
                m.weight_quantizer = ada
        for m in ([unit] if isinstance(unit, QuantModule) else unit.modules()):
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = True
    assert checked == len(wanted) == 8


def test_attention_model_caches_and_w8_forward_match_reference(golden_dir):
    """Toy Cheng2020-attn (tests/golden/recon_attn.npz): the product QuantModel built from the reference's state reproduces
    the reference's caches for four layer units inside the first attention block (ReLU-fused 1x1 / 3x3 convs, bare 1x1 convs,
    sigmoid-gated residual around them), and -- with the reference's trained roundings installed -- its W8 forward
    (x_hat, both likelihood tensors)."""
    import lic
    from quantization import BaseQuantBlock, QuantModel, QuantModule
    from quantization.quantizer import AdaRoundQuantizer, to_rows
    from quantization.utils import save_inp_oup_data
    fx = np.load(os.path.join(golden_dir, "recon_attn.npz"))
    N, n_img, B, iters = (int(v) for v in fx["meta"])
    torch.manual_seed(0)
    qnn = QuantModel(lic.Cheng2020Attention(N=N), WQ, AQ, is_cheng=True).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    with torch.no_grad():
        for name, m in qnn.model.named_modules():
            if isinstance(m, QuantModule) and m.org_weight is not None:
                w = T(fx[f"org/{name}.weight"]).cuda()
                m.weight.data.copy_(w); m.org_weight.copy_(w)
                if m.org_bias is not None:
                    b = T(fx[f"org/{name}.bias"]).cuda()
                    m.bias.data.copy_(b); m.org_bias.copy_(b)
        for k, v in qnn.model.entropy_bottleneck.state_dict().items():
            v.copy_(T(fx[f"state/entropy_bottleneck.{k}"]).cuda())
    cali = T(fx["cali"]).cuda()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    qnn.set_quant_state(True, False)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    wanted = [str(o) for o in fx["order"]]
    # the unit schedule of the product surgery equals the reference's
    order = []

    def walk(mod, prefix=""):
        for name, m in mod.named_children():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                order.append(prefix + name)
            else:
                walk(m, prefix + name + ".")
    walk(qnn.model)
    assert order == [str(s) for s in fx["full_order"]]
    for full in order:
        unit = _get(qnn.model, full)
        if full in wanted:
            assert type(unit.activation_function).__name__ == {0: "StraightThrough", 2: "ReLU"}[int(fx[f"{full}/act"])]
            (inp_q, inp_fp), out = save_inp_oup_data(qnn, unit, cali, asym=True, act_quant=False, batch_size=2, input_prob=True)
            for got, key in ((inp_q, "inp_q"), (inp_fp, "inp_fp"), (out, "out")):
                ref = T(fx[f"{full}/{key}"])
                assert tuple(got.shape) == tuple(ref.shape), (full, key)
                err = float((got.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
                assert err < 3e-5, (full, key, err)
            alpha = T(fx[f"{full}/alpha_final"]).cuda()
            ada = AdaRoundQuantizer(uaq=unit.weight_quantizer, round_mode="learned_hard_sigmoid",
                                    weight_tensor=unit.org_weight.data, alpha_rows=to_rows(alpha))
            ada.soft_targets = False
            unit.weight_quantizer = ada
        for m in ([unit] if isinstance(unit, QuantModule) else unit.modules()):
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = True
    # W8 forward, split at the rounding of the latents (one flipped round() would change x_hat everywhere): the analysis
    # transform must reproduce the reference's continuous y, and the synthesis transform its x_hat from the reference's y_hat
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        y = qnn.model.g_a(cali[:2])
        stages, h = [], T(fx["w8/y_hat"]).cuda()
        for stage in qnn.model.g_s:
            h = stage(h)
            stages.append(h)
    for got, key in [(y, "w8/y")] + [(h, f"w8/g_s.{k}") for k, h in enumerate(stages)] + [(stages[-1], "w8/x_hat")]:
        ref = T(fx[key])
        err = float((got.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 1e-4, (key, err)


def test_activation_quantised_cache_building_matches_reference(golden_dir):
    """`--act_quant` cache pass (tests/golden/recon_toy_aq.npz): with the reference's trained roundings of g_a.0 / g_a.1
    installed, the product's save_inp_oup_data(act_quant=True, batch 1) reproduces the reference's quantised input, FP input and
    FP target of g_a.2.  The W8A8 prefix goes through five dynamic 8-bit activation quantisers: a value on a rounding boundary
    may land one level off on the GPU, so the quantised input is compared on its mean deviation."""
    import lic
    from quantization import BaseQuantBlock, QuantModel, QuantModule
    from quantization.quantizer import AdaRoundQuantizer, to_rows
    from quantization.utils import save_inp_oup_data
    fx = np.load(os.path.join(golden_dir, "recon_toy_aq.npz"))
    N, n_img, B, iters = (int(v) for v in fx["meta"])
    torch.manual_seed(0)
    qnn = QuantModel(lic.Cheng2020Anchor(N=N), WQ, AQ, is_cheng=True).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    with torch.no_grad():
        for name, m in qnn.model.named_modules():
            if isinstance(m, QuantModule) and m.org_weight is not None:
                w = T(fx[f"org/{name}.weight"]).cuda()
                m.weight.data.copy_(w); m.org_weight.copy_(w)
                if m.org_bias is not None:
                    b = T(fx[f"org/{name}.bias"]).cuda()
                    m.bias.data.copy_(b); m.org_bias.copy_(b)
    cali = T(fx["cali"]).cuda()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    qnn.set_quant_state(True, True)
    qnn.model.g_s[-1][0].set_quant_state(True, False)
    units = list(qnn.model.g_a.named_children())
    for name, u in units[:2]:
        for n_, m in u.named_modules():
            if isinstance(m, QuantModule) and m.org_weight is not None:
                alpha = T(fx[f"g_a.{name}/{n_}.alpha_final"]).cuda()
                ada = AdaRoundQuantizer(uaq=m.weight_quantizer, round_mode="learned_hard_sigmoid",
                                        weight_tensor=m.org_weight.data, alpha_rows=to_rows(alpha))
                ada.soft_targets = False
                m.weight_quantizer = ada
        for m in u.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = True
    (inp_q, inp_fp), out = save_inp_oup_data(qnn, units[2][1], cali, asym=True, act_quant=True, batch_size=1, input_prob=True)
    for got, key, tol in ((inp_fp, "inp_fp", 3e-5), (out, "out", 3e-5)):
        ref = T(fx[f"g_a.2/{key}"])
        assert float((got.cpu() - ref).abs().max()) / float(ref.abs().max()) < tol, key
    ref = T(fx["g_a.2/inp_q"])
    dev = (inp_q.cpu() - ref).abs()
    assert float(dev.max()) / float(ref.abs().max()) < 2e-2 and float(dev.mean()) / float(ref.abs().max()) < 1e-3
    # and it is the quantised path that was compared: the quantised input is measurably away from the FP input
    assert float((ref - T(fx["g_a.2/inp_fp"])).abs().max()) > 5e-3


def test_fp_cache_memo_gives_the_rows_of_the_per_unit_passes(monkeypatch):
    """`quantization.utils._FpMemo` (one full-precision forward for the x_fp / target rows of ALL units) against the per-unit truncated
    passes it replaces (RDO_FP_MEMO=0): bit-identical caches for units across the model, also behind a calibrated prefix, a short last
    batch included; a unit asked for twice falls back to its own pass."""
    import lic
    from helpers import WQ, AQ
    from quantization import QuantModel
    from quantization import utils as U
    torch.manual_seed(4)
    model = lic.Cheng2020Anchor(N=8).cuda().eval()
    qnn = QuantModel(model=model, weight_quant_params=WQ, act_quant_params=AQ, is_cheng=True).cuda().eval()
    cali = torch.rand(10, 3, 64, 64, device="cuda")
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:4])
    units = U._FpMemo.units_of(qnn)
    assert len(units) >= 29
    for u in units[:5]:                              # a "calibrated" prefix: the quantised pass differs from the full-precision one
        for m in u.modules():
            if hasattr(m, "trained"):
                m.trained = True
    byname = {n: m for n, m in qnn.model.named_modules()}
    named = [byname[n] for n in ("h_a.2", "h_a.8", "h_s.0", "h_s.4", "entropy_parameters.0", "entropy_parameters.4", "context_prediction")]
    assert all(any(u is m for u in units) for m in named)        # hyper-analysis / hyper-synthesis / entropy-parameter units too
    picks = [units[0], units[3], units[7], units[12], units[-1]] + [m for m in named if all(m is not q for q in (units[0], units[3], units[7], units[12], units[-1]))]
    monkeypatch.setenv("RDO_FP_MEMO", "0")
    ref = [U.save_inp_oup_data(qnn, u, cali, True, False, batch_size=4, input_prob=True) for u in picks]
    monkeypatch.setenv("RDO_FP_MEMO", "1")
    U._FpMemo.current = U._FpMemo.refused = None
    got = [U.save_inp_oup_data(qnn, u, cali, True, False, batch_size=4, input_prob=True) for u in picks]
    assert U._FpMemo.current is not None and id(picks[0]) not in U._FpMemo.current.rows        # built once, rows handed out
    for ((q0, f0), o0), ((q1, f1), o1) in zip(ref, got):
        assert torch.equal(q0, q1) and torch.equal(f0, f1) and torch.equal(o0, o1)
    (q2, f2), o2 = U.save_inp_oup_data(qnn, picks[1], cali, True, False, batch_size=4, input_prob=True)     # second request: own pass
    assert torch.equal(f2, ref[1][0][1]) and torch.equal(o2, ref[1][1])
    U._FpMemo.current = None


def test_fp_cache_memo_refuses_rows_it_cannot_vouch_for(monkeypatch):
    """The three ways the rows of the one full forward could differ from a unit's own truncated pass: a unit that the
    model calls TWICE per forward (its rows would interleave), a unit whose captured output a LATER op of the full forward modifies in
    place (the truncated pass stops before that op), and a unit the schedule skips (`ignore_reconstruction`).  Each must get no rows from
    the memo -- its caches then come from its own pass and equal the RDO_FP_MEMO=0 ones bit for bit -- while the other units keep theirs."""
    import torch.nn as nn
    from helpers import WQ, AQ
    from quantization import QuantModel
    from quantization import utils as U

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Conv2d(3, 8, 3, padding=1)
            self.shared = nn.Conv2d(8, 8, 3, padding=1)
            self.b = nn.Conv2d(8, 8, 3, padding=1)
            self.c = nn.Conv2d(8, 8, 3, padding=1)
            self.d = nn.Conv2d(8, 4, 3, padding=1)

        def forward(self, x):
            h = self.a(x)
            h = self.shared(self.shared(h))          # called twice per forward
            h = self.b(h)
            h.mul_(0.5)                              # in place on b's output, AFTER b's hook has fired
            h = self.c(h)
            return self.d(h)
    torch.manual_seed(9)
    qnn = QuantModel(model=Net().cuda().eval(), weight_quant_params=WQ, act_quant_params=AQ).cuda().eval()
    cali = torch.rand(6, 3, 32, 32, device="cuda")
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:4])
    m = qnn.model
    m.c.ignore_reconstruction = True
    assert [u for u in U._FpMemo.units_of(qnn)] == [m.a, m.shared, m.b, m.d]
    m.a.trained = True
    monkeypatch.setenv("RDO_FP_MEMO", "0")
    ref = {n: U.save_inp_oup_data(qnn, getattr(m, n), cali, True, False, batch_size=4, input_prob=True) for n in ("a", "shared", "b", "d")}
    monkeypatch.setenv("RDO_FP_MEMO", "1")
    U._FpMemo.clear()
    (q, f), o = U.save_inp_oup_data(qnn, m.a, cali, True, False, batch_size=4, input_prob=True)
    memo = U._FpMemo.current
    assert memo is not None and set(memo.rows) == {id(m.d)}, "only `d` may keep rows: `shared` runs twice, `b`'s output is overwritten in place"
    assert torch.equal(f, ref["a"][0][1]) and torch.equal(o, ref["a"][1])
    for n in ("shared", "b", "d"):
        (q, f), o = U.save_inp_oup_data(qnn, getattr(m, n), cali, True, False, batch_size=4, input_prob=True)
        assert torch.equal(q, ref[n][0][0]) and torch.equal(f, ref[n][0][1]) and torch.equal(o, ref[n][1]), n
    assert U._FpMemo.current is None                 # every row handed out -> released
    # identity: a different model object, or a dead one, never reuses the memo; clear() drops it
    U.save_inp_oup_data(qnn, m.a, cali, True, False, batch_size=4, input_prob=True)
    assert U._FpMemo.current is not None and U._FpMemo.current.matches(qnn, cali, 4)
    assert not U._FpMemo.current.matches(qnn, cali.clone(), 4) and not U._FpMemo.current.matches(qnn, cali, 2)
    U._FpMemo.clear()
    assert U._FpMemo.current is None

