"""Lu2022 path end to end on the GPU against the reference's own model code (tests/golden/recon_nic.npz was produced by
models/nic_cvt.py:NIC + quantization/* of the reference): state-dict compatibility of lic.NIC, unit schedule of the surgery,
cache building for six units (convs, transposed convs, RSTBs incl. shifted windows and a single-token window), and stage-wise
W8 / W8A8 forwards of the calibrated model through the Quant* Swin wrappers (HIP attention / LayerNorm / GELU kernels)."""
import os

import numpy as np
import pytest
import torch

from helpers import AQ, WQ, T

pytestmark = pytest.mark.gpu
CFG = dict(height=64, width=64, in_chans=3, embed_dim=16, latent_dim=32, window_size=8, mlp_ratio=2.0, qkv_bias=True,
           qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False)


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def build(golden_dir):
    import lic
    from quantization import QuantModel
    fx = np.load(os.path.join(golden_dir, "recon_nic.npz"))
    model = lic.NIC(CFG)
    state = {k[len("state/"):]: T(fx[k]) for k in fx.files if k.startswith("state/")}
    model.load_state_dict(state, strict=True)                 # parameter / buffer names are the reference's
    model = model.cuda().eval()
    return fx, model


def install_trained(fx, name, unit):
    from quantization import QuantModule
    from quantization.quantizer import AdaRoundQuantizer, to_rows
    inner = [("", unit)] if isinstance(unit, QuantModule) else \
        [(n + ".", m) for n, m in unit.named_modules() if isinstance(m, QuantModule)]
    for n, m in inner:
        alpha = T(fx[f"{name}/{n}alpha_final"]).cuda()
        np.testing.assert_array_equal(m.weight_quantizer.delta.reshape(-1).cpu().numpy(), fx[f"{name}/{n}delta"].reshape(-1))
        ada = AdaRoundQuantizer(uaq=m.weight_quantizer, round_mode="learned_hard_sigmoid", weight_tensor=m.org_weight.data,
                                alpha_rows=to_rows(alpha, tconv=m.if_tconv))
        ada.soft_targets = False
        m.weight_quantizer = ada


def test_nic_fp_model_matches_reference(golden_dir):
    fx, model = build(golden_dir)
    x = T(fx["cali"])[:2].cuda()
    with torch.no_grad():
        y = model.g_a(x)
        out = model(x)
    assert _rel(y.cpu(), T(fx["fp/y"])) < 2e-5
    assert _rel(out["x_hat"].cpu(), T(fx["fp/x_hat"])) < 1e-4


def test_nic_caches_and_quantised_forwards_match_reference(golden_dir):
    from quantization import BaseQuantBlock, QuantModel, QuantModule
    from quantization.utils import save_inp_oup_data
    fx, model = build(golden_dir)
    B = int(fx["meta"][4])
    qnn = QuantModel(model=model, weight_quant_params=WQ, act_quant_params=AQ).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    cali = T(fx["cali"]).cuda()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B])
    order = [n for n, m in qnn.model.named_children() if isinstance(m, (QuantModule, BaseQuantBlock))]
    assert order == [str(s) for s in fx["full_order"]]
    wanted = [str(o) for o in fx["order"]]
    qnn.set_quant_state(True, False)
    qnn.model.g_s7.set_quant_state(True, False)
    for name in order:
        unit = getattr(qnn.model, name)
        if name in wanted:
            (inp_q, inp_fp), out = save_inp_oup_data(qnn, unit, cali, asym=True, act_quant=False, batch_size=2, input_prob=True)
            for got, key in ((inp_q, "inp_q"), (inp_fp, "inp_fp"), (out, "out")):
                ref = T(fx[f"{name}/{key}"])
                assert tuple(got.shape) == tuple(ref.shape), (name, key)
                assert _rel(got.cpu(), ref) < 5e-5, (name, key, _rel(got.cpu(), ref))
            install_trained(fx, name, unit)
            unit.set_quant_state(True, False)
            with torch.no_grad():
                xin = T(fx[f"{name}/inp_q"])[:2].cuda()
                y = unit(xin) if isinstance(unit, QuantModule) else unit(xin, tuple(xin.shape[2:4]))
            assert _rel(y.cpu(), T(fx[f"{name}/hard_out"])) < 5e-5, name
        for m in unit.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = True
    qnn.eval()
    for tag, aq in (("w8", False), ("w8a8", True)):
        qnn.set_quant_state(True, aq)
        qnn.model.g_s7.set_quant_state(True, False)
        h = cali[:2]
        with torch.no_grad():
            for coder in ("g_a", "g_s"):
                for name in [n for n in order if n.startswith(coder)]:
                    unit = getattr(qnn.model, name)
                    h = unit(h) if isinstance(unit, QuantModule) else unit(h, tuple(h.shape[2:4]))
                    ref = T(fx[f"{tag}/{name}"])
                    err = _rel(h.cpu(), ref)
                    if not aq:
                        assert err < 1e-4, (tag, name, err)
                    else:
                        # W8A8: each Swin block re-quantises its activations at seven points with dynamic 8-bit grids.  A value
                        # on a rounding boundary may land one level off (1/255 of the channel range); window attention then
                        # spreads that to the window's tokens and later quantisers flip more.  From identical inputs every
                        # single stage is exact or flips < 0.3 % of its elements (tools/dbg_nic_gpu.py); over a whole RSTB the
                        # deviation must stay well below one quantisation level on average
                        mean_dev = float((h.cpu() - ref).abs().mean() / ref.abs().max())
                        assert err < 3e-2 and mean_dev < 2e-3, (tag, name, err, mean_dev)
                    h = ref.cuda()
                if coder == "g_a":
                    h = T(fx[f"{tag}/y_hat"]).cuda()
