"""Hot-loop parity: the HIP calibration engine against the oracle's reconstruct_unit (which is itself pinned to the
reference's layer_/block_reconstruction by tests/test_oracle_golden.py), on the toy-Cheng2020 golden caches with the same
mini-batch index stream and the same counter-RNG QDrop masks.

Tolerances (fp32, different reduction orders): per-iteration loss 2e-4 relative; alpha 2e-3 absolute after 12 Adam
steps of 1e-3 (Adam normalises the gradient, so a noisy tiny gradient may move one element by up to lr per step);
hard rounding decisions (alpha >= 0) must agree on >= 99.5 % of the weights."""
import os

import numpy as np
import pytest
import torch

from helpers import RECON_UNITS, nhwc, oracle_ops, product_unit, T

pytestmark = pytest.mark.gpu
SEED = 1005


@pytest.fixture(scope="module")
def recon(golden_dir):
    return np.load(os.path.join(golden_dir, "recon_toy.npz"))


@pytest.mark.parametrize("tag,kind", RECON_UNITS)
@pytest.mark.parametrize("graph", [True, False])
def test_engine_matches_oracle(recon, tag, kind, graph):
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    fx = recon
    _, n_img, B, iters = (int(v) for v in fx["meta"])
    idx = fx[f"{tag}/idx"]
    # --- oracle
    ops_o = oracle_ops(fx, tag, kind)
    log = O.reconstruct_unit(kind, ops_o, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]), iters=iters,
                             batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5), input_prob=0.5,
                             weight=0.01, b_range=(20, 2), warmup=0.2)
    # --- product
    unit, k, mods = product_unit(fx, tag, kind)
    eng = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]), batch_size=B,
                     iters=iters, weight=0.01, b_range=(20, 2), warmup=0.2, input_prob=0.5, seed=SEED,
                     idx_table=torch.from_numpy(idx), use_graph=graph)
    # scales initialised by the HIP min/max kernel must equal the reference's
    for n, op in eng.ops.items():
        np.testing.assert_array_equal(op.delta.cpu().numpy(), fx[f"{tag}/{n}.delta"].reshape(-1))
        np.testing.assert_array_equal(op.zp.cpu().numpy(), fx[f"{tag}/{n}.zp"].reshape(-1))
    eng.run()
    torch.cuda.synchronize()
    total, rt, rd = eng.logs()
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(rd.numpy(), np.array(log.round), rtol=2e-4, atol=1e-7)
    flips = tot = 0
    for n, op in ops_o.items():
        a_gpu = eng.alpha_of(n).cpu()
        assert a_gpu.shape == op.alpha.shape
        np.testing.assert_allclose(a_gpu.numpy(), op.alpha.numpy(), rtol=0, atol=2e-3)
        flips += int(((a_gpu >= 0) != (op.alpha >= 0)).sum())
        tot += a_gpu.numel()
    assert flips <= 0.005 * tot, f"{flips}/{tot} rounding decisions differ"
    # hand-back: hard-rounded forward of the trained unit through the module surface
    eng.finish()
    for m in unit.modules():
        if hasattr(m, "trained"):
            m.trained = True
    unit.set_quant_state(True, False)
    with torch.no_grad():
        y = unit(T(fx[f"{tag}/inp_q"][:2]).cuda())
        y_ref = O.UNIT_FORWARD[kind](ops_o, T(fx[f"{tag}/inp_q"][:2]))
    torch.cuda.synchronize()
    err = (y.cpu() - y_ref).abs().max() / (y_ref.abs().max() + 1e-12)
    # a flipped rounding decision moves one weight by one step: allow the few flips counted above
    assert float(err) < (5e-3 if flips else 2e-5), float(err)


@pytest.mark.parametrize("tag,kind", [("g_a.0", "rbws"), ("g_s.1", "rbu"), ("g_a.6", "layer")])
def test_engine_layer_wise_scales(recon, tag, kind):
    """main2.py without --channel_wise (main2.py:42,175): one (delta, zero_point) per weight tensor.  Engine vs oracle with
    the oracle's own per-tensor scale init (pinned by the quantiser goldens), then the hand-back through the per-tensor
    AdaRoundQuantizer, whose alpha lives in the logical OIHW order."""
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    fx = recon
    _, _, B, iters = (int(v) for v in fx["meta"])
    idx = fx[f"{tag}/idx"]
    ops_o = oracle_ops(fx, tag, kind)
    for op in ops_o.values():
        op.delta = op.zp = None
        op.channel_wise = False
    log = O.reconstruct_unit(kind, ops_o, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]), iters=iters,
                             batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5), input_prob=0.5)
    unit, k, mods = product_unit(fx, tag, kind, wq={"n_bits": 8, "channel_wise": False, "scale_method": "max"})
    eng = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]), batch_size=B,
                     iters=iters, input_prob=0.5, seed=SEED, idx_table=torch.from_numpy(idx))
    for n, op in eng.ops.items():
        assert float(op.delta.min()) == float(op.delta.max()) == float(ops_o[n].delta)
        assert float(op.zp.min()) == float(op.zp.max()) == float(ops_o[n].zp)
    eng.run()
    torch.cuda.synchronize()
    total, _, rd = eng.logs()
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(rd.numpy(), np.array(log.round), rtol=2e-4, atol=1e-7)
    flips = tot = 0
    for n, op in ops_o.items():
        a_gpu = eng.alpha_of(n).cpu()
        np.testing.assert_allclose(a_gpu.numpy(), op.alpha.numpy(), rtol=0, atol=2e-3)
        flips += int(((a_gpu >= 0) != (op.alpha >= 0)).sum())
        tot += a_gpu.numel()
    assert flips <= 0.005 * tot
    eng.finish()
    for m in unit.modules():
        if hasattr(m, "trained"):
            m.trained = True
    unit.set_quant_state(True, False)
    with torch.no_grad():
        y = unit(T(fx[f"{tag}/inp_q"][:2]).cuda())
        y_ref = O.UNIT_FORWARD[kind](ops_o, T(fx[f"{tag}/inp_q"][:2]))
    err = (y.cpu() - y_ref).abs().max() / (y_ref.abs().max() + 1e-12)
    assert float(err) < (5e-3 if flips else 2e-5), float(err)


@pytest.mark.parametrize("tag,kind", [("g_a.0", "rbws"), ("g_a.6", "layer")])
@pytest.mark.parametrize("task_p", [1.0, 2.4])
def test_engine_task_loss_exponent(recon, tag, kind, task_p):
    """main2.py --task_loss is the exponent of the task term (layer_opt.py:150,274): rec_loss keeps p = 2."""
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    fx = recon
    _, _, B, iters = (int(v) for v in fx["meta"])
    idx = fx[f"{tag}/idx"]
    ops_o = oracle_ops(fx, tag, kind)
    log = O.reconstruct_unit(kind, ops_o, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]), iters=iters,
                             batch_size=B, idx_stream=idx, task_p=task_p,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5), input_prob=0.5)
    unit, k, mods = product_unit(fx, tag, kind)
    eng = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]), batch_size=B,
                     iters=iters, input_prob=0.5, seed=SEED, idx_table=torch.from_numpy(idx), task_p=task_p)
    eng.run()
    torch.cuda.synchronize()
    total = eng.logs()[0]
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=2e-4, atol=1e-7)
    # the four numbers of the reference's periodic log line (layer_opt.py:168-170), term by term
    rec, task, rd, b = eng.logs_terms()
    np.testing.assert_allclose(rec.numpy(), np.array(log.rec), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(task.numpy(), np.array(log.task), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(rd.numpy(), np.array(log.round), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(b.numpy(), np.array(log.b), rtol=1e-6)
    flips = tot = 0
    for n, op in ops_o.items():
        a_gpu = eng.alpha_of(n).cpu()
        # p = 1: the gradient is sign(d), so an error that is ~0 in one implementation may flip sign in the other
        np.testing.assert_allclose(a_gpu.numpy(), op.alpha.numpy(), rtol=0, atol=2e-3 if task_p != 1.0 else 2.5e-2)
        flips += int(((a_gpu >= 0) != (op.alpha >= 0)).sum())
        tot += a_gpu.numel()
    assert flips <= 0.005 * tot


def test_lp_loss_kernel_matches_autograd():
    from hipops import ops
    g = torch.Generator().manual_seed(3)
    B, H, W, C, n = 3, 4, 8, 12, 5
    tgt = torch.randn(n, H, W, C, generator=g)
    pred = torch.randn(B, H, W, C, generator=g)
    idx = torch.tensor([[4, 0, 2]], dtype=torch.int32)
    for c2, cp, p in ((1.0, 1.0, 1.5), (0.0, 1.0, 3.0), (1.0, 1.0, 1.0)):
        pr = pred.clone().requires_grad_(True)
        d = pr - tgt[idx[0].long()]
        loss = c2 * d.pow(2).sum(-1).mean() + cp * d.abs().pow(p).sum(-1).mean()
        loss.backward()
        grad = torch.empty_like(pred).cuda()
        log = torch.zeros(1, 32, device="cuda")
        ops.lp_loss_grad(pred.cuda(), tgt.cuda(), idx.cuda(), torch.zeros(1, dtype=torch.int32, device="cuda"), c2, cp, p, grad, log)
        torch.testing.assert_close(grad.cpu(), pr.grad, rtol=2e-5, atol=1e-7)
        torch.testing.assert_close(log.sum().cpu(), loss.detach(), rtol=1e-5, atol=0)
        # with a second log the |d|^p term goes there and only the p = 2 term stays in the first
        log2, logp = torch.zeros(1, 32, device="cuda"), torch.zeros(1, 32, device="cuda")
        ops.lp_loss_grad(pred.cuda(), tgt.cuda(), idx.cuda(), torch.zeros(1, dtype=torch.int32, device="cuda"), c2, cp, p, grad, log2, logp)
        dd = pred - tgt[idx[0].long()]
        torch.testing.assert_close(log2.sum().cpu(), c2 * dd.pow(2).sum(-1).mean(), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(logp.sum().cpu(), cp * dd.abs().pow(p).sum(-1).mean(), rtol=1e-5, atol=1e-7)


def test_engine_rejects_cpu_tensors(recon):
    from quantization.engine import UnitEngine
    fx = recon
    with pytest.raises(RuntimeError):
        UnitEngine("layer", {}, torch.zeros(2, 4, 4, 4), torch.zeros(2, 4, 4, 4), torch.zeros(2, 4, 4, 4), batch_size=1, iters=1)


@pytest.mark.parametrize("tag,kind", [("g_a.0", "rbws"), ("g_s.1", "rbu"), ("g_a.6", "layer")])
def test_dp_split_sequence_equals_fused_step(recon, tag, kind):
    """The data-parallel op sequence (rdo_adaround_grad -> bucket -> rdo_adaround_apply, engine plan A / plan B) run on
    one rank must reproduce the fused rdo_adaround_step path bit for bit (same kernels' arithmetic, different packaging)."""
    from quantization.engine import UnitEngine
    fx = recon
    _, _, B, iters = (int(v) for v in fx["meta"])
    res = []
    for split in (False, True):
        unit, k, mods = product_unit(fx, tag, kind)
        eng = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]), batch_size=B,
                         iters=iters, seed=SEED, idx_table=torch.from_numpy(fx[f"{tag}/idx"]), force_dp_split=split)
        eng.run()
        torch.cuda.synchronize()
        res.append(({n: eng.alpha_of(n).clone() for n in eng.ops}, eng.logs()[0]))
    for n in res[0][0]:
        torch.testing.assert_close(res[0][0][n], res[1][0][n], rtol=0, atol=0)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-6, atol=0)


@pytest.mark.parametrize("tag", ["g_a.0", "g_a.1", "g_s.0", "g_s.1", "h_s.0"])
def test_engine_minnen_layer_units_match_oracle(golden_dir, tag):
    """Minnen2018-style sequential coders: 5x5 stride-2 conv, GDN / IGDN as their own units, transposed convs (one with a
    fused LeakyReLU) -- HIP engine vs oracle on the reference's toy-model caches (tests/golden/recon_minnen.npz)."""
    from helpers import minnen_oracle_op, minnen_product_module
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    fx = np.load(os.path.join(golden_dir, "recon_minnen.npz"))
    B, iters = int(fx["meta"][3]), int(fx["meta"][4])
    idx = fx[f"{tag}/idx"]
    op_o = minnen_oracle_op(fx, tag)
    log = O.reconstruct_unit("layer", {"layer": op_o}, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]),
                             iters=iters, batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5))
    qm = minnen_product_module(fx, tag)
    eng = UnitEngine("layer", {"layer": qm}, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]),
                     batch_size=B, iters=iters, seed=SEED, idx_table=torch.from_numpy(idx))
    np.testing.assert_array_equal(eng.ops["layer"].delta.cpu().numpy(), fx[f"{tag}/delta"].reshape(-1))
    eng.run()
    torch.cuda.synchronize()
    total, _, _ = eng.logs()
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=2e-4, atol=1e-7)
    a_gpu = eng.alpha_of("layer").cpu()
    assert a_gpu.shape == op_o.alpha.shape
    np.testing.assert_allclose(a_gpu.numpy(), op_o.alpha.numpy(), rtol=0, atol=2e-3)
    flips = int(((a_gpu >= 0) != (op_o.alpha >= 0)).sum())
    assert flips <= 0.005 * a_gpu.numel()
    eng.finish()
    qm.trained = True
    qm.set_quant_state(True, False)
    with torch.no_grad():
        y = qm(T(fx[f"{tag}/inp_q"][:2]).cuda())
        y_ref = op_o(T(fx[f"{tag}/inp_q"][:2]))
    err = float((y.cpu() - y_ref).abs().max() / (y_ref.abs().max() + 1e-12))
    assert err < (5e-3 if flips else 2e-5), err


@pytest.mark.parametrize("tag", ["g_a.3.conv_a.0.conv.0", "g_a.3.conv_a.0.conv.2", "g_a.3.conv_a.0.conv.4", "g_a.3.conv_b.3"])
def test_engine_attention_layer_units_match_oracle(golden_dir, tag):
    """Layer units inside a Cheng2020-attn attention block (ReLU fused into 1x1 / 3x3 convs, bare 1x1 convs): HIP engine vs
    oracle on the reference's toy-model caches (tests/golden/recon_attn.npz)."""
    from helpers import minnen_oracle_op, minnen_product_module
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    fx = np.load(os.path.join(golden_dir, "recon_attn.npz"))
    B, iters = int(fx["meta"][2]), int(fx["meta"][3])
    idx = fx[f"{tag}/idx"]
    op_o = minnen_oracle_op(fx, tag)
    log = O.reconstruct_unit("layer", {"layer": op_o}, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]),
                             iters=iters, batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5))
    qm = minnen_product_module(fx, tag)
    eng = UnitEngine("layer", {"layer": qm}, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]),
                     batch_size=B, iters=iters, seed=SEED, idx_table=torch.from_numpy(idx))
    np.testing.assert_array_equal(eng.ops["layer"].delta.cpu().numpy(), fx[f"{tag}/delta"].reshape(-1))
    eng.run()
    torch.cuda.synchronize()
    total, _, _ = eng.logs()
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=2e-4, atol=1e-7)
    a_gpu = eng.alpha_of("layer").cpu()
    np.testing.assert_allclose(a_gpu.numpy(), op_o.alpha.numpy(), rtol=0, atol=2e-3)
    flips = int(((a_gpu >= 0) != (op_o.alpha >= 0)).sum())
    assert flips <= 0.005 * a_gpu.numel()
    eng.finish()
    qm.trained = True
    qm.set_quant_state(True, False)
    with torch.no_grad():
        y = qm(T(fx[f"{tag}/inp_q"][:2]).cuda())
        y_ref = op_o(T(fx[f"{tag}/inp_q"][:2]))
    err = float((y.cpu() - y_ref).abs().max() / (y_ref.abs().max() + 1e-12))
    assert err < (5e-3 if flips else 2e-5), err


def test_engine_ten_bit_weights_match_oracle(golden_dir):
    """W10 (BASELINE config 3; an extension beyond the reference's 8-bit assert): 10-bit channel-wise grid, scale init,
    AdaRound trajectory and hard-rounded forward of the HIP engine against the oracle at n_bits = 10."""
    import torch.nn as nn
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    from quantization.quant_layer import QuantModule
    from quantization.export import integer_state, dequantize
    tag = "g_a.3.conv_a.0.conv.2"
    fx = np.load(os.path.join(golden_dir, "recon_attn.npz"))
    B, iters = int(fx["meta"][2]), int(fx["meta"][3])
    idx = fx[f"{tag}/idx"]
    w, b = T(fx[f"{tag}/weight"]), T(fx[f"{tag}/bias"])
    op_o = O.QOp("conv", w, b, stride=1, padding=1, act="relu", n_bits=10)
    log = O.reconstruct_unit("layer", {"layer": op_o}, T(fx[f"{tag}/inp_q"]), T(fx[f"{tag}/inp_fp"]), T(fx[f"{tag}/out"]),
                             iters=iters, batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5))
    conv = nn.Conv2d(w.shape[1], w.shape[0], 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)
    wq10 = {"n_bits": 10, "channel_wise": True, "scale_method": "max"}
    qm = QuantModule(conv.cuda(), wq10, {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}).cuda()
    qm.activation_function = nn.ReLU(inplace=True)
    eng = UnitEngine("layer", {"layer": qm}, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]),
                     batch_size=B, iters=iters, seed=SEED, idx_table=torch.from_numpy(idx))
    np.testing.assert_array_equal(eng.ops["layer"].delta.cpu().numpy(), op_o.delta.reshape(-1).numpy())
    assert float(op_o.zp.max()) > 255            # the grid really is wider than 8 bits
    eng.run()
    torch.cuda.synchronize()
    total, _, _ = eng.logs()
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=2e-4, atol=1e-7)
    a_gpu = eng.alpha_of("layer").cpu()
    np.testing.assert_allclose(a_gpu.numpy(), op_o.alpha.numpy(), rtol=0, atol=2e-3)
    flips = int(((a_gpu >= 0) != (op_o.alpha >= 0)).sum())
    assert flips <= 0.005 * a_gpu.numel()
    eng.finish()
    qm.trained = True
    qm.set_quant_state(True, False)
    ent = integer_state(qm)[""]
    assert ent["levels"].dtype == torch.int32 and int(ent["levels"].max()) > 255 and int(ent["levels"].max()) <= 1023
    torch.testing.assert_close(dequantize(ent).cuda(), qm.weight_quantizer(qm.weight).detach(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag,kind", RECON_UNITS)
def test_fused_tails_and_batched_step_change_nothing(recon, tag, kind, monkeypatch):
    """The fused unit tails (csrc/fused_tail.hip), the one-launch AdaRound step (tile form: dgrad layout in the same launch) and the
    next iteration's gather riding in that launch (round 6: rdo_adaround_step_batch_gather, counter published by the tail) execute
    the same fp32 operations as the chains of separate kernels they replace: trained alphas are bit-identical, the logged loss
    differs only in summation order."""
    from quantization.engine import UnitEngine
    fx = recon
    _, _, B, iters = (int(v) for v in fx["meta"])
    idx = torch.from_numpy(fx[f"{tag}/idx"])
    res = []
    for fuse, batch, fold in ((True, True, True), (False, False, True), (True, False, True), (True, True, False)):
        monkeypatch.setenv("RDO_GATHER_IN_STEP", "1" if fold else "0")
        unit, k, mods = product_unit(fx, tag, kind)
        eng = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]), batch_size=B,
                         iters=iters, input_prob=0.5, seed=SEED, idx_table=idx, fuse_tail=fuse, batch_step=batch)
        assert eng.fused == fuse and eng._folded == (fold and batch)
        assert any(t == "ada_step_gather" for t, _, _ in eng.plan_a.op_info()) == eng._folded
        assert any(t.startswith("gather_qdrop") for t, _, _ in eng.plan_a.op_info()) != eng._folded
        eng.run()
        torch.cuda.synchronize()
        res.append(({n: eng.alpha_of(n).clone() for n in eng.ops}, eng.logs()[0]))
    for other in res[1:]:
        for n in res[0][0]:
            assert torch.equal(res[0][0][n], other[0][n]), n
        torch.testing.assert_close(res[0][1], other[1], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("where", ["g_a.1", "g_s.2", "h_a.0"])
def test_rd_task_loss_mode_matches_oracle(where):
    """loss_mode='rd' (opt-in): task term = lambda * 255^2 * MSE(x_hat, x) + bpp of the WHOLE model with the unit's soft-quantised
    output substituted -- the criterion the reference sketches and comments out (layer_opt.py:146-148, losses/losses.py:8-35).
    Engine (HIP kernels under torch's tape for the modules behind the unit) against the oracle loop running the oracle model on the
    CPU, for an analysis block, a synthesis block and a hyper-analysis layer of a toy Cheng2020; entropy models in evaluation mode
    with straight-through rounding on both sides."""
    import lic
    from helpers import AQ, WQ
    from oracle import lic_oracle as LO
    from oracle import rdo_oracle as O
    from oracle.cheng_units import schedule
    from quantization import QuantModel
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    torch.manual_seed(41)
    N, n_img, B, iters, lmbda = 8, 6, 2, 6, 0.0483
    ref = LO.Cheng2020Anchor(N=N).eval()
    g = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("gamma"):
                c = p.shape[0]
                p.copy_(torch.sqrt(0.1 * torch.eye(c) + 0.01 * torch.rand(c, c, generator=g) + 2.0 ** -36))
            elif p.dim() == 4 and "entropy_bottleneck" not in name:
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 2 * (3.0 / p[0].numel()) ** 0.5)
    ref.context_prediction.mask.fill_(1.0)                     # the wrapper bypasses the mask
    cali = torch.rand(n_img, 3, 64, 64, generator=g)
    prod = lic.Cheng2020Anchor(N=N).eval()
    sd = ref.state_dict()
    with torch.no_grad():
        for k, v in prod.state_dict().items():
            v.copy_(sd[k])
    qnn = QuantModel(prod.cuda(), WQ, AQ, is_cheng=True).cuda().eval()
    qnn.set_quant_state(False, False)
    # the unit, on both sides
    sched = {n: (k, o, m) for n, k, o, m in schedule(ref)}
    kind, ops_o, ref_mod = sched[where]
    seq, pos = where.split(".")
    unit = getattr(qnn.model, seq)[int(pos)]
    # caches: FP input / output of the unit for every calibration image (the quantised-prefix input = FP + small noise)
    store = {}
    h = ref_mod.register_forward_hook(lambda m, i, o: store.update(inp=i[0].detach().clone(), out=o.detach().clone()))
    with torch.no_grad():
        ref(cali)
    h.remove()
    inp, out = store["inp"], store["out"]
    if kind == "layer" and ops_o["layer"].act == "lrelu":
        out = torch.nn.functional.leaky_relu(out, 0.01)
    inp_q = inp + 1e-3 * torch.randn(inp.shape, generator=g)
    idx = np.stack([np.random.RandomState(i).permutation(n_img)[:B] for i in range(iters)])

    # ---- oracle: the loop of reconstruct_unit with the RD task term through the oracle model
    LO.STE_ROUND = True
    try:
        def task_fn(out_quant, ix):
            x = cali[ix]
            y = out_quant
            # a layer unit carries its fused LeakyReLU (quant_model.py:51-54): substitute behind the activation module
            fused_act = kind == "layer" and ops_o["layer"].act == "lrelu"
            hook_mod = getattr(ref, seq)[int(pos) + 1] if fused_act else ref_mod
            hk = hook_mod.register_forward_hook(lambda m, i, o: y)
            try:
                o = ref(x)
            finally:
                hk.remove()
            n_pix = x.shape[0] * x.shape[2] * x.shape[3]
            bpp = sum((-torch.log2(v)).sum() for v in o["likelihoods"].values()) / n_pix
            # reference quirk: the wrapped PixelShuffle carries a LeakyReLU, so the wrapped model's output is
            # leaky_relu(x_hat); the product reproduces it, the oracle model mirrors it here
            xh = torch.nn.functional.leaky_relu(o["x_hat"], 0.01)
            return lmbda * 255 ** 2 * ((xh - x) ** 2).mean() + bpp
        log = O.reconstruct_unit(kind, ops_o, inp_q, inp, out, iters=iters, batch_size=B, idx_stream=idx,
                                 mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5), input_prob=0.5, weight=0.01,
                                 b_range=(20, 2), warmup=0.2, task_fn=task_fn)
    finally:
        LO.STE_ROUND = False

    # ---- product engine in rd mode
    k, mods = _unit_modules(unit)
    assert k == kind
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    eng = UnitEngine(k, mods, nh(inp_q), nh(inp), nh(out), batch_size=B, iters=iters, weight=0.01, b_range=(20, 2), warmup=0.2,
                     input_prob=0.5, seed=SEED, idx_table=torch.from_numpy(idx),
                     rd=dict(model=qnn, unit=unit, cali=cali.cuda(), lmbda=lmbda))
    for n_, op in eng.ops.items():
        np.testing.assert_array_equal(op.delta.cpu().numpy(), ops_o[n_].init_scale().delta.reshape(-1).numpy())
    eng.run()
    torch.cuda.synchronize()
    assert eng.rd_path == "graph"          # the iteration (both plans + the model tail with its backward) replays from ONE captured graph
    rec, task, rd_, _ = eng.logs_terms()
    np.testing.assert_allclose(rec.numpy(), np.array(log.rec), rtol=3e-4, atol=1e-7)
    # the rate term counts rounded latents: one latent within float noise of .5 moves the loss by a few bits of several thousand
    np.testing.assert_allclose(task.numpy(), np.array(log.task), rtol=2e-3)
    np.testing.assert_allclose(rd_.numpy(), np.array(log.round), rtol=2e-4, atol=1e-7)
    flips = tot = 0
    for n_, op in ops_o.items():
        a_gpu = eng.alpha_of(n_).cpu()
        far = ((a_gpu - op.alpha).abs() > 2e-3).float().mean()
        assert float(far) < 2e-2, (n_, float(far))
        flips += int(((a_gpu >= 0) != (op.alpha >= 0)).sum())
        tot += a_gpu.numel()
    assert flips <= 0.01 * tot


@pytest.mark.parametrize("path,n_img", [("g_s.2", 6), ("h_a.2", 6), ("g_a.3", 6), ("entropy_parameters.2", 6), ("h_s.2.0", 5)])
def test_rd_mode_graph_host_and_data_parallel_sequences_agree(monkeypatch, path, n_img):
    """loss_mode='rd' three ways on the same unit (g_s.2 of a toy Cheng2020): the captured-graph iteration, the host-driven iteration
    (RDO_RD_GRAPH=0) and the data-parallel op sequence on one rank (gradient bucket -> apply): the same alphas bit for bit, the
    same losses."""
    import lic
    from helpers import AQ, WQ
    from quantization import QuantModel
    from quantization.engine import UnitEngine
    from quantization.recon import _unit_modules
    torch.manual_seed(43)
    N, B, iters, lmbda = 8, 2, 7, 0.0483          # n_img = 5: the cache pass pads its last mini-batch
    g = torch.Generator().manual_seed(44)
    cali = torch.rand(n_img, 3, 64, 64, generator=g).cuda()
    idx = torch.from_numpy(np.stack([np.random.RandomState(i).permutation(n_img)[:B] for i in range(iters)]))
    res = {}
    for mode in ("graph", "host", "dp", "nocache"):
        torch.manual_seed(43)
        prod = lic.Cheng2020Anchor(N=N).eval()
        qnn = QuantModel(prod.cuda(), WQ, AQ, is_cheng=True).cuda().eval()
        qnn.set_quant_state(False, False)
        unit = qnn.model
        for part in path.split("."):
            unit = unit[int(part)] if part.isdigit() else getattr(unit, part)
        store = {}
        h = unit.register_forward_hook(lambda m, i, o: store.update(inp=i[0].detach().clone(), out=o.detach().clone()))
        with torch.no_grad():
            qnn(cali)
        h.remove()
        nh = lambda t: t.permute(0, 2, 3, 1).contiguous()
        inp, out = nh(store["inp"]), nh(store["out"])
        inp_q = inp + 1e-3 * torch.randn(inp.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        monkeypatch.setenv("RDO_RD_GRAPH", "0" if mode == "host" else "1")
        monkeypatch.setenv("RDO_RD_CACHE", "0" if mode == "nocache" else "1")      # per-image cache of the unit-independent modules
        k, mods = _unit_modules(unit)
        eng = UnitEngine(k, mods, inp_q, inp, out, batch_size=B, iters=iters, weight=0.01, b_range=(20, 2), warmup=0.2, input_prob=0.5, seed=SEED,
                         idx_table=idx, force_dp_split=(mode == "dp"), rd=dict(model=qnn, unit=unit, cali=cali, lmbda=lmbda))
        eng.run()
        torch.cuda.synchronize()
        assert eng.rd_path == ("graph" if mode in ("graph", "nocache") else "host")
        # g_s / hyper-path units leave whole coders unit-independent; a g_a unit only has the modules in front of it to skip
        assert bool(eng.rd["memo"]) == (mode != "nocache" and not path.startswith("g_a")), (path, mode, len(eng.rd["memo"]))
        assert bool(eng.rd["skip"]) == (mode != "nocache")
        res[mode] = ({n: eng.alpha_of(n).clone() for n in eng.ops}, [t.clone() for t in eng.logs_terms()[:3]])
    for mode in ("host", "dp", "nocache"):
        for n in res["graph"][0]:
            assert torch.equal(res[mode][0][n], res["graph"][0][n]), (mode, n)
        for a, b in zip(res[mode][1], res["graph"][1]):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
