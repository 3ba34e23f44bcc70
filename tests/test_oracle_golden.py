"""Pin the oracle (oracle/rdo_oracle.py) to golden vectors produced by the reference's own code
(tools/make_golden.py).  CPU only.  Tolerances: bit-exact where the restatement runs the same torch ops,
1e-6 relative where the vectorised form reorders nothing but goes through a different kernel."""
import os

import numpy as np
import pytest
import torch

from oracle import rdo_oracle as O

T = torch.from_numpy


@pytest.fixture(scope="module")
def qz(golden_dir):
    return np.load(os.path.join(golden_dir, "quantizers.npz"))


@pytest.mark.parametrize("tag,tconv", [("conv", False), ("tconv", True), ("gdn", False), ("vec", False)])
@pytest.mark.parametrize("method", ["max", "mse", "l1", "l2"])
@pytest.mark.parametrize("cw", [True, False])
@pytest.mark.parametrize("bits", [8, 4])
def test_uaq_init_and_fakequant(qz, tag, tconv, method, cw, bits):
    w = T(qz[f"w_{tag}"])
    key = f"uaq_{tag}_{method}_{'cw' if cw else 'lw'}_{bits}"
    delta, zp = O.uaq_init(w, bits, cw, method, tconv=tconv)
    np.testing.assert_array_equal(delta.numpy().reshape(-1), qz[key + "_delta"].reshape(-1))
    np.testing.assert_array_equal(zp.numpy().reshape(-1), qz[key + "_zp"].reshape(-1))
    assert tuple(delta.shape) == tuple(qz[key + "_delta"].shape)
    out = O.uaq_fakequant(w, delta, zp, 2 ** bits)
    np.testing.assert_array_equal(out.numpy(), qz[key + "_out"])


@pytest.mark.parametrize("tag,tconv", [("conv", False), ("tconv", True), ("lin", False)])
@pytest.mark.parametrize("cw", [True, False])
@pytest.mark.parametrize("bits", [8, 6, 4])
def test_uaq_max_init_on_symmetric_ranges(golden_dir, tag, tconv, cw, bits):
    """Channels with min = -max put -min/delta on x.5: the zero point then depends on torch evaluating the reference's
    `-x_min / delta` (Python float over tensor, quantizer.py:296) as delta.reciprocal() * (-x_min).  Vectors from the reference."""
    fx = np.load(os.path.join(golden_dir, "quantizer_ties.npz"))
    w = T(fx[f"w_{tag}"])
    key = f"uaq_{tag}_{'cw' if cw else 'lw'}_{bits}"
    delta, zp = O.uaq_init(w, bits, cw, "max", tconv=tconv)
    np.testing.assert_array_equal(delta.numpy().reshape(-1), fx[key + "_delta"].reshape(-1))
    np.testing.assert_array_equal(zp.numpy().reshape(-1), fx[key + "_zp"].reshape(-1))
    np.testing.assert_array_equal(O.uaq_fakequant(w, delta, zp, 2 ** bits).numpy(), fx[key + "_out"])


@pytest.mark.parametrize("tag", ["conv", "lin"])
@pytest.mark.parametrize("method,sym", [("max_scale", False), ("max", True), ("max_scale", True)])
@pytest.mark.parametrize("cw", [True, False])
@pytest.mark.parametrize("bits", [8, 6, 4])
def test_uaq_scaled_and_symmetric_init_on_tie_prone_ranges(golden_dir, tag, method, sym, cw, bits):
    """'max_scale' scales the range by (n_bits+2)/8 in Python double before the single rounding to fp32 (quantizer.py:284-293);
    symmetric grids mirror the larger side.  Vectors from the reference on weights whose -min/delta sits on x.5."""
    fx = np.load(os.path.join(golden_dir, "quantizer_ties.npz"))
    w = T(fx[f"w_{tag}"])
    key = f"uaq_{tag}_{method}{'_sym' if sym else ''}_{'cw' if cw else 'lw'}_{bits}"
    delta, zp = O.uaq_init(w, bits, cw, method, sym=sym)
    np.testing.assert_array_equal(delta.numpy().reshape(-1), fx[key + "_delta"].reshape(-1))
    np.testing.assert_array_equal(zp.numpy().reshape(-1), fx[key + "_zp"].reshape(-1))
    np.testing.assert_array_equal(O.uaq_fakequant(w, delta, zp, 2 ** bits).numpy(), fx[key + "_out"])


def test_uaq_gaussian(qz):
    w = T(qz["w_conv"])
    delta, zp = O.uaq_init(w, 8, False, "gaussian")
    np.testing.assert_array_equal(delta.numpy(), qz["uaq_conv_gaussian_lw_8_delta"])
    np.testing.assert_array_equal(zp.numpy(), qz["uaq_conv_gaussian_lw_8_zp"])


@pytest.mark.parametrize("tag,tconv", [("conv", False), ("tconv", True), ("gdn", False)])
def test_adaround(qz, tag, tconv):
    w = T(qz[f"w_{tag}"])
    delta, zp = O.uaq_init(w, 8, True, "max", tconv=tconv)
    np.testing.assert_array_equal(O.adaround_init_alpha(w, delta).numpy(), qz[f"ada_{tag}_alpha0"])
    alpha = T(qz[f"ada_{tag}_alpha"]).clone().requires_grad_(True)
    soft = O.adaround_forward(w, alpha, delta, zp, 256, True)
    np.testing.assert_array_equal(soft.detach().numpy(), qz[f"ada_{tag}_soft"])
    (soft * T(qz[f"ada_{tag}_gy"])).sum().backward()
    np.testing.assert_array_equal(alpha.grad.numpy(), qz[f"ada_{tag}_galpha"])
    hard = O.adaround_forward(w, alpha.detach(), delta, zp, 256, False)
    np.testing.assert_array_equal(hard.numpy(), qz[f"ada_{tag}_hard"])
    for b in (20, 11.5, 2.0):
        np.testing.assert_array_equal(O.round_loss_term(alpha.detach(), b, 0.01).numpy(),
                                      qz[f"ada_{tag}_roundloss_b{b}"])


@pytest.mark.parametrize("tag", ["a4", "a3", "a2"])
def test_act_quant(qz, tag):
    out = O.act_quant(T(qz[f"act_{tag}_in"]))
    np.testing.assert_array_equal(out.numpy(), qz[f"act_{tag}_out"])


def test_lp_loss_round_ste(qz):
    p, t = T(qz["lp_pred"]), T(qz["lp_tgt"])
    for e in (2.0, 1.0, 3.5):
        np.testing.assert_array_equal(O.lp_loss(p, t, p=e).numpy(), qz[f"lp_none_{e}"])
        np.testing.assert_array_equal(O.lp_loss(p, t, p=e, reduction="all").numpy(), qz[f"lp_all_{e}"])
    np.testing.assert_array_equal(O.round_ste(p * 3).numpy(), qz["round_ste"])


def test_temp_decay(golden_dir):
    d = np.load(os.path.join(golden_dir, "temp_decay.npz"))
    for t_max, warm in ((50, 0.2), (20000, 0.2), (10, 0.0)):
        got = [float(O.linear_temp_decay(t, t_max, warm, 20, 2)) for t in range(1, t_max + 1)]
        np.testing.assert_array_equal(np.array(got), d[f"b_{t_max}_{warm}"])


# ----------------------------------------------------------------------------- blocks
def _ops_from(fx, tag, names, mode):
    ops = {}
    for n in names:
        if f"{tag}/{n}.weight" not in fx:
            continue
        w = T(fx[f"{tag}/{n}.weight"])
        b = T(fx[f"{tag}/{n}.bias"]) if f"{tag}/{n}.bias" in fx else None
        if n in ("gdn", "igdn"):
            op = O.QOp(n, w, b)
        else:
            k = w.shape[-1]
            stride = 2 if (n in ("conv1", "skip") and tag.startswith("rbws")) else 1
            op = O.QOp("conv", w, b, stride=stride, padding=k // 2)
        if f"{tag}/{n}.delta" in fx:
            op.delta, op.zp = T(fx[f"{tag}/{n}.delta"]), T(fx[f"{tag}/{n}.zp"])
        op.mode = mode
        ops[n] = op
    return ops


BLOCK_OPS = {"rbws": ["conv1", "conv2", "gdn", "skip"], "rbws3": ["conv1", "conv2", "gdn", "skip"],
             "rbu": ["subpel_conv", "conv", "igdn", "upsample"], "rb": ["conv1", "conv2", "skip"]}


@pytest.mark.parametrize("tag", ["rbws", "rbws3", "rbu", "rb"])
def test_block_forward(golden_dir, tag):
    fx = np.load(os.path.join(golden_dir, "blocks.npz"))
    kind = "rbws" if tag.startswith("rbws") else tag
    x = T(fx[f"{tag}/x"])
    for state, mode, aq in (("fp", "fp", False), ("w8", "uaq", False), ("w8a8", "uaq", True)):
        ops = _ops_from(fx, tag, BLOCK_OPS[tag], mode)
        if aq:
            # reference: inner QuantModules with disable_act_quant=False and trained also act-quantise
            # their own outputs (quant_layer.py:130-133): conv2+gdn / conv+igdn / skip / upsample[0].
            y = _block_forward_w8a8(kind, ops, x)
        else:
            with torch.no_grad():
                y = O.UNIT_FORWARD[kind](ops, x)
        np.testing.assert_allclose(y.numpy(), fx[f"{tag}/y_{state}"], rtol=0, atol=0)


def _block_forward_w8a8(kind, ops, x):
    with torch.no_grad():
        return O.UNIT_FORWARD[kind](ops, x, aq=True, inner_aq=True)


# ----------------------------------------------------------------------------- the reconstruction loop
UNIT_OPS = {"rbws": ["conv1", "conv2", "gdn", "skip"], "rb": ["conv1", "conv2", "skip"],
            "rbu": ["subpel_conv", "conv", "igdn", "upsample"], "layer": ["layer"]}
LAYER_GEOM = {"g_a.6": (2, 1), "g_s.7.0": (1, 1), "h_s.2.0": (1, 1), "entropy_parameters.0": (1, 0),
              "context_prediction": (1, 2)}


def _unit_ops(fx, tag, kind):
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


@pytest.fixture(scope="module")
def recon(golden_dir):
    return np.load(os.path.join(golden_dir, "recon_toy.npz"))


@pytest.mark.parametrize("tag,kind", [("g_a.0", "rbws"), ("g_a.1", "rb"), ("g_a.6", "layer"), ("g_s.1", "rbu"),
                                      ("g_s.7.0", "layer"), ("h_s.2.0", "layer"), ("entropy_parameters.0", "layer"),
                                      ("context_prediction", "layer")])
def test_reconstruction_loop_matches_reference(recon, tag, kind):
    """Replay the reference's layer_/block_reconstruction run (same caches, idx stream, QDrop uniforms)."""
    fx = recon
    _, _, B, iters = (int(v) for v in fx["meta"])
    ops = _unit_ops(fx, tag, kind)
    rand = T(fx[f"{tag}/rand"])
    log = O.reconstruct_unit(kind, ops, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]),
                             iters=iters, batch_size=B, idx_stream=fx[f"{tag}/idx"],
                             mask_fn=lambda i, shape: rand[i] < 0.5, input_prob=0.5, weight=0.01, b_range=(20, 2),
                             warmup=0.2, p=2.0, task_p=2.0)
    np.testing.assert_allclose(np.array(log.total), fx[f"{tag}/loss"], rtol=1e-6, atol=1e-9)
    for n, op in ops.items():
        np.testing.assert_allclose(op.alpha.numpy(), fx[f"{tag}/{n}.alpha_final"], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        y = O.UNIT_FORWARD[kind](ops, T(fx[f"{tag}/inp_q"])[:2])
    np.testing.assert_allclose(y.numpy(), fx[f"{tag}/hard_out"], rtol=1e-5, atol=1e-6)


def test_counter_rng_is_uniform_and_reproducible():
    m1 = O.qdrop_keep_mask_nhwc(1005, 3, (2, 8, 16, 16), 0.5)
    m2 = O.qdrop_keep_mask_nhwc(1005, 3, (2, 8, 16, 16), 0.5)
    m3 = O.qdrop_keep_mask_nhwc(1005, 4, (2, 8, 16, 16), 0.5)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    assert abs(float(m1.float().mean()) - 0.5) < 0.03
    assert bool(O.qdrop_keep_mask_nhwc(1, 0, (1, 4, 4, 4), 1.0).all())
    assert not bool(O.qdrop_keep_mask_nhwc(1, 0, (1, 4, 4, 4), 0.0).any())


@pytest.mark.parametrize("tag", ["g_a.0", "g_a.1", "g_s.0", "g_s.1", "h_s.0"])
def test_reconstruction_loop_minnen_units(golden_dir, tag):
    """5x5 stride-2 conv, GDN unit, transposed conv, IGDN unit, transposed conv + LeakyReLU: oracle replay of the reference's
    layer_reconstruction runs on the toy Minnen2018 mean-scale model."""
    from helpers import minnen_oracle_op
    fx = np.load(os.path.join(golden_dir, "recon_minnen.npz"))
    B, iters = int(fx["meta"][3]), int(fx["meta"][4])
    op = minnen_oracle_op(fx, tag)
    rand = T(fx[f"{tag}/rand"])
    log = O.reconstruct_unit("layer", {"layer": op}, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]),
                             iters=iters, batch_size=B, idx_stream=fx[f"{tag}/idx"], mask_fn=lambda i, shape: rand[i] < 0.5)
    np.testing.assert_allclose(np.array(log.total), fx[f"{tag}/loss"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(op.alpha.numpy(), fx[f"{tag}/alpha_final"], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        y = op(T(fx[f"{tag}/inp_q"])[:2])
    np.testing.assert_allclose(y.numpy(), fx[f"{tag}/hard_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["g_a.3.conv_a.0.conv.0", "g_a.3.conv_a.0.conv.2", "g_a.3.conv_a.0.conv.4", "g_a.3.conv_b.3"])
def test_reconstruction_loop_attention_units(golden_dir, tag):
    """Layer units inside a Cheng2020-attn attention block (1x1 + ReLU, 3x3 + ReLU, bare 1x1, the mask branch's last 1x1):
    oracle replay of the reference's layer_reconstruction runs (tests/golden/recon_attn.npz)."""
    from helpers import minnen_oracle_op
    fx = np.load(os.path.join(golden_dir, "recon_attn.npz"))
    B, iters = int(fx["meta"][2]), int(fx["meta"][3])
    op = minnen_oracle_op(fx, tag)
    rand = T(fx[f"{tag}/rand"])
    log = O.reconstruct_unit("layer", {"layer": op}, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]),
                             iters=iters, batch_size=B, idx_stream=fx[f"{tag}/idx"], mask_fn=lambda i, shape: rand[i] < 0.5)
    np.testing.assert_allclose(np.array(log.total), fx[f"{tag}/loss"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(op.alpha.numpy(), fx[f"{tag}/alpha_final"], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        y = op(T(fx[f"{tag}/inp_q"])[:2])
    np.testing.assert_allclose(y.numpy(), fx[f"{tag}/hard_out"], rtol=1e-5, atol=1e-6)


def test_attention_unit_schedule(golden_dir):
    """Unit order the reference's recon_model recursion (main2.py:227-253) produced on the toy Cheng2020-attn: residual blocks
    are block units, every conv inside an AttentionBlock is its own layer unit (7 convs per branch pair -> 19 per block)."""
    fx = np.load(os.path.join(golden_dir, "recon_attn.npz"))
    order = [str(s) for s in fx["full_order"]]
    assert order[:4] == ["g_a.0", "g_a.1", "g_a.2", "g_a.3.conv_a.0.conv.0"]
    for blk in ("g_a.3", "g_a.8", "g_s.0", "g_s.5"):
        inner = [o for o in order if o.startswith(blk + ".")]
        assert len(inner) == 19 and inner[-1] == blk + ".conv_b.3", (blk, inner)
    assert order.index("g_a.3.conv_b.3") < order.index("g_a.4")


# ----------------------------------------------------------------------------- Lu2022 (NIC / RSTB): reference's own model code
NIC_CFG = dict(height=64, width=64, in_chans=3, embed_dim=16, latent_dim=32, window_size=8)


def _nic(golden_dir):
    from oracle import swin_oracle as S
    fx = np.load(os.path.join(golden_dir, "recon_nic.npz"))
    state = {k[len("state/"):]: T(fx[k]) for k in fx.files if k.startswith("state/")}
    return fx, S.NicOracle(state, NIC_CFG)


def test_nic_fp_forward_matches_reference_model(golden_dir):
    """Analysis + synthesis transforms of the functional restatement against models/nic_cvt.py:NIC.forward (FP32)."""
    fx, nic = _nic(golden_dir)
    x = T(fx["cali"])[:2]
    with torch.no_grad():
        y = nic.run(nic.coder("g_a"), x)
        x_hat = nic.run(nic.coder("g_s"), torch.round(y))
    np.testing.assert_allclose(y.numpy(), fx["fp/y"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x_hat.numpy(), fx["fp/x_hat"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["g_a0", "g_a1", "g_a7", "h_a3", "h_s1", "g_s7"])
def test_reconstruction_loop_nic_units(golden_dir, name):
    """layer_/block_reconstruction of the reference on its toy NIC, replayed by the oracle: conv with a 7-stage FP tail,
    shifted-window RSTB with a tail, RSTB at window == resolution (round_ste-only tail), single-token RSTB, transposed conv
    with a tail, closing 5x5 transposed conv."""
    from oracle import swin_oracle as S
    fx, nic = _nic(golden_dir)
    B, iters = int(fx["meta"][4]), int(fx["meta"][5])
    unit = nic.stages[name]
    rand = T(fx[f"{name}/rand"])
    if isinstance(unit, S.RstbOracle):
        ops, fwd = unit.ops, (lambda ops_, x: unit(x))
    else:
        ops, fwd = {"layer": unit}, "layer"
    log = O.reconstruct_unit(fwd, ops, T(fx[f"{name}/inp_q"]), T(fx[f"{name}/inp_fp"]), T(fx[f"{name}/out"]), iters=iters,
                             batch_size=B, idx_stream=fx[f"{name}/idx"], mask_fn=lambda i, shape: rand[i] < 0.5,
                             tail=nic.tail_of(name))
    np.testing.assert_allclose(np.array(log.total), fx[f"{name}/loss"], rtol=2e-5, atol=1e-7)
    for k, op in ops.items():
        pre = "" if k == "layer" else k + "."
        np.testing.assert_allclose(op.delta.numpy().reshape(-1), fx[f"{name}/{pre}delta"].reshape(-1), rtol=0, atol=0)
        np.testing.assert_allclose(op.alpha.numpy(), fx[f"{name}/{pre}alpha_final"], rtol=1e-5, atol=2e-6)
    with torch.no_grad():
        y = unit(T(fx[f"{name}/inp_q"])[:2])
    np.testing.assert_allclose(y.numpy(), fx[f"{name}/hard_out"], rtol=1e-4, atol=1e-5)


def _install_calibrated_state(fx, nic):
    """Reference end state: the six calibrated units carry their trained hard rounding, every other unit nearest rounding."""
    from oracle import swin_oracle as S
    wanted = [str(o) for o in fx["order"]]
    for name, st in nic.stages.items():
        ops = st.ops if isinstance(st, S.RstbOracle) else {"layer": st}
        for k, op in ops.items():
            if name in wanted:
                pre = "" if k == "layer" else k + "."
                op.init_scale()
                op.alpha, op.mode, op.soft = T(fx[f"{name}/{pre}alpha_final"]), "ada", False
            else:
                op.mode = "uaq"


@pytest.mark.parametrize("tag", ["w8", "w8a8"])
def test_nic_quantised_forward_stagewise(golden_dir, tag):
    """Stage-by-stage W8 and W8A8 forwards (g_a, rounding, g_s) of the calibrated reference model.  W8A8 exercises every
    activation-quantisation point of the Swin wrappers (QuantModule outputs, attention probabilities, attn @ v, block and
    RSTB outputs; quant_block.py:343-346,410-416,545-546,635-636)."""
    from oracle import swin_oracle as S
    fx, nic = _nic(golden_dir)
    _install_calibrated_state(fx, nic)
    aq = tag == "w8a8"
    h = T(fx["cali"])[:2]
    with torch.no_grad():
        for coder in ("g_a", "g_s"):
            for name in nic.coder(coder):
                st = nic.stages[name]
                if isinstance(st, S.RstbOracle):
                    st.aq = aq
                    h = st(h)
                else:
                    h = st(h)
                    if aq and name != "g_s7":
                        h = O.act_quant(h)
                ref = fx[f"{tag}/{name}"]
                err = float(np.abs(h.numpy() - ref).max() / (np.abs(ref).max() + 1e-12))
                # an activation landing on a rounding boundary may flip one 8-bit level: (1/255 of a channel range)
                assert err < (1e-4 if not aq else 6e-3), (name, err)
                h = T(ref)                                   # continue from the reference's tensor: no error build-up
            if coder == "g_a":
                h = T(fx[f"{tag}/y_hat"])


def test_nic_whole_model_oracle_matches_reference_forward(golden_dir):
    """`swin_oracle.NicModelOracle` (the whole NIC forward with the entropy models: the checker of the R + lambda*D task loss on the Lu2022
    coders) against models/nic_cvt.py:NIC.forward of the reference on the golden weights: x_hat and both likelihood tensors."""
    from oracle import swin_oracle as S
    fx = np.load(os.path.join(golden_dir, "recon_nic.npz"))
    state = {k[len("state/"):]: T(fx[k]) for k in fx.files if k.startswith("state/")}
    mo = S.NicModelOracle(state, NIC_CFG, masked_context=True)           # the full-precision model applies the context mask
    x = T(fx["cali"])[:fx["fp/x_hat"].shape[0]]
    with torch.no_grad():
        o = mo.forward(x)
    np.testing.assert_allclose(o["x_hat"].numpy(), fx["fp/x_hat"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o["likelihoods"]["y"].numpy(), fx["fp/lik_y"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o["likelihoods"]["z"].numpy(), fx["fp/lik_z"], rtol=1e-4, atol=1e-6)
    # substitution leaves everything in front of the stage alone and replaces its output
    with torch.no_grad():
        y1 = mo.nic.run(["g_a0", "g_a1"], x)
        o2 = mo.forward(x, substitute=("g_a1", y1))
    np.testing.assert_allclose(o2["x_hat"].numpy(), o["x_hat"].numpy(), rtol=0, atol=0)


def test_committed_oracle_trajectories_cover_the_test_tables(golden_dir):
    """tests/golden/long_horizon.npz and flow_n192.npz (tools/make_long_horizon_golden.py, tools/make_flow_golden.py) hold an entry for every
    run the GPU tests look up, with the iteration counts the tests use -- a stale fixture fails here, on the CPU, not on the GPU box."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import flow_common as F
    import long_horizon_common as C
    lh = np.load(os.path.join(golden_dir, "long_horizon.npz"))
    for (stats, name), (iters, every) in C.RUNS.items():
        key = f"{stats}/{name}"
        assert tuple(lh[f"{key}/iters"]) == (iters, every), key
        n = len(C.picks(iters, every))
        assert lh[f"{key}/total"].shape == (n,) and lh[f"{key}/rt"].shape == (n,) and lh[f"{key}/round"].shape == (n,), key
        assert lh[f"{key}/cache_sig"].shape == (6,) and np.isfinite(lh[f"{key}/total"]).all(), key
        assert any(k.startswith(f"{key}/bits/") for k in lh.files), key
    for stats in ("uniform", "kodak"):
        assert lh[f"{stats}/y_hat/fp"].shape == (C.N_IMG, 192, 16, 16) and lh[f"{stats}/y_hat/prefix"].dtype == np.int16
    fl = np.load(os.path.join(golden_dir, "flow_n192.npz"))
    for stats in ("uniform", "kodak"):
        units = list(fl[f"{stats}/units"])
        assert len(units) == 29 and units[0] == "g_a.0"
        for u in units:
            assert fl[f"{stats}/{u}/idx"].shape == (F.iters_of(u), F.B), (stats, u)
            assert fl[f"{stats}/{u}/total_first_last"].shape == (2,)
        assert fl[f"{stats}/w8"].shape == (2,) and fl[f"{stats}/w8a8"].shape == (2,)
    crops = np.load(os.path.join(golden_dir, "kodak_crops.npz"))["crops"]
    assert crops.shape == (16, 256, 256, 3) and crops.dtype == np.uint8
