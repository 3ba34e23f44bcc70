"""BASELINE config 4 at FULL size (missing 1 / weak 1): Lu2022 `NIC` with embed 192 / latent 320 on 256 x 256 crops,
batch 4 -- the size bench.py times -- through the PUBLIC `layer_reconstruction` / `block_reconstruction` on the tape engine, against
`oracle/swin_oracle.py` + `oracle.rdo_oracle.reconstruct_unit(kind=callable, tail=...)` (quant_block.py:350-553, models/layers.py:260-300,
layer_opt.py:45-75 restated on the CPU) on the SAME caches, mini-batch index stream and counter-RNG QDrop masks:

    g_a0   5x5-s2 stem, FP tail = the whole rest of g_a (14 Swin blocks, 3 strided convs) + round_ste
    g_a1   RSTB at 128^2 (2 blocks, 4 heads, shifted windows) + tail + round_ste
    g_s4   RSTB at 64^2 (4 blocks) with the tail  tconv 3x3 -> RSTB 128^2 -> tconv 5x5  (transposed-conv TAIL stages: the class the
           uninitialised phase-weight planes of round 4 lived in -- no parity test could see that fault then)
    g_s7   the closing 5x5 transposed conv (phase form on H2 / bf16x6 planes), no tail
    h_a1   RSTB at 8^2, window 4, with a strided-conv + single-window RSTB tail
    h_s1   transposed conv with an RSTB + transposed-conv tail

What is compared: per-iteration total loss (3e-4 relative; analysis-transform units end in round_ste, where a latent within fp32 noise
of x.5 may round the other way on the GPU -- a few such events are allowed for, as in test_gpu_nic.py), the trained alphas (fraction
further than 2e-3 from the oracle's < 2e-3, hard rounding decisions that differ < 5e-3), delta / zero point bit for bit, and the
FIRST-iteration gradient d(rec + task)/d alpha of every weight tensor of the unit (data-parallel op sequence of the engine, plan A)
against the oracle's autograd: 2e-5 of the tensor's largest entry (measured: <= 1e-5) (5 to 14 Swin blocks of bf16x6 / fp32-MFMA GEMMs, softmax and
LayerNorm backward lie between the loss and the weights).  The units in front of the tested one count as calibrated (nearest
rounding), so x_q carries real quantisation noise of the prefix.  The checker is the oracle only."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(height=256, width=256, in_chans=3, embed_dim=192, latent_dim=320, window_size=8, mlp_ratio=2.0, qkv_bias=True,
           qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False)
N_IMG, B = 4, 4
ITERS = {"g_a0": 6, "g_a1": 6, "g_s4": 8, "g_s7": 8, "h_a1": 8, "h_s1": 8}
WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
AQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}


@pytest.fixture(scope="module")
def nic_full():
    import lic
    torch.manual_seed(1005)
    model = lic.NIC(CFG).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        # default initialisation leaves the relative-position tables and every bias near zero: give them mass so that the bias /
        # shifted-window-mask paths of the attention kernels and the bias gradients carry signal
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias") and p.dim() == 1 and "norm" not in n:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cali = torch.rand(N_IMG, 3, 256, 256, generator=g)
    return model, state, cali


@pytest.fixture(scope="module")
def nic_full_kodak(golden_dir):
    """The same model class with natural-image statistics (missing 3 / next 3): crops of the reference's Kodak images
    (tests/golden/kodak_crops.npz) and 'trained-like' parameters (helpers.trained_like_nic_: Laplace-tailed weights, per-channel scales
    over two decades, LayerNorm gains in [0.3, 3]) -- activations with kurtosis 8-20 and magnitudes growing to 10^2-10^3 through the
    residual stacks, where the per-token scales of rdo_linear_h2 and the probed plane scales of the conv units are exercised."""
    import lic
    from helpers import kodak_crops, trained_like_nic_
    torch.manual_seed(1005)
    model = lic.NIC(CFG).eval()
    g = torch.Generator().manual_seed(7)
    trained_like_nic_(model, g)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model, state, kodak_crops(golden_dir, N_IMG)


@pytest.mark.parametrize("name", ["g_a1", "g_s4"])
def test_lu2022_full_size_units_natural_statistics(nic_full_kodak, name):
    """g_a1 (RSTB at 128^2 + the rest of g_a + round_ste) and g_s4 (RSTB at 64^2 + transposed-conv / RSTB tail) on Kodak crops with
    trained-like parameters: losses, trained alphas, scales as on uniform noise; the first-iteration gradients against the FLOAT64
    oracle with the fp32 oracle's own error as the yardstick (see there)."""
    test_lu2022_full_size_units_match_oracle(nic_full_kodak, name, natural=True)


def _logical(op, g):
    """engine gradient buffer (kernel layout) -> the oracle's logical weight shape"""
    g = g.reshape(op.alpha.shape)
    if op.qm.kind in ("linear", "layernorm"):
        return g.reshape(op.qm.org_weight.shape)
    if op.tconv is not None:
        return g.flip(1, 2).permute(3, 0, 1, 2)
    return g.permute(0, 3, 1, 2)


LONG = {"h_a1": 300, "h_s1": 300, "g_a7": 120}


@pytest.mark.parametrize("name", list(LONG))
def test_lu2022_full_size_long_horizon(nic_full, name):
    """The same comparison over a realistic stretch of the schedule for three full-width units whose oracle iteration is cheap (an RSTB
    of h_a with a tail, a transposed conv of h_s with a tail, the 320-channel RSTB of g_a with its round_ste-only tail): 300 / 120
    iterations, warm-up boundary inside, b decaying, Adam's moments carried -- losses at every 25th iteration and the final hard
    rounding decisions (>= 99 % per tensor equal to the oracle's; the run itself moves several per cent against nearest rounding)."""
    test_lu2022_full_size_units_match_oracle(nic_full, name, iters=LONG[name], long=True)


def _oracle_first_gradient(state, dtype, name, inp_q, inp_fp, out_fp, idx, seed, zr_fixed):
    """d(rec + task)/d alpha of every weight tensor of unit `name` in the FIRST iteration from the oracle evaluated in `dtype` (float32:
    the reference's arithmetic; float64: the value both fp32 implementations approximate) on the given caches.  zr_fixed (units in g_a):
    the latents the ENGINE rounded to -- an input of the comparison, see the caller -- and the count of latents this evaluation would
    have rounded differently."""
    from oracle import rdo_oracle as O, swin_oracle as S
    B_ = idx.shape[1]
    nic2 = S.NicOracle({k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}, CFG)
    unit2 = nic2.stages[name]
    ops2, fwd2 = (unit2.ops, (lambda ops_, x: unit2(x))) if isinstance(unit2, S.RstbOracle) else ({"layer": unit2}, "layer")
    seen = {"flips": 0}
    if zr_fixed is not None:
        rest = nic2.coder(name[:3])[nic2.coder(name[:3]).index(name) + 1:]

        def tail(t):
            z = nic2.run(rest, t)
            seen["flips"] = int((torch.round(z.detach()) != zr_fixed.to(dtype)).sum())
            return z + (zr_fixed.to(dtype) - z).detach()
    else:
        tail = nic2.tail_of(name)
    grads = []
    cast = lambda t: t.to(dtype)
    O.reconstruct_unit(fwd2, ops2, cast(inp_q), cast(inp_fp), cast(out_fp), iters=1, batch_size=B_, idx_stream=idx[:1],
                       mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(seed, i, shape, 0.5), tail=tail,
                       fp_net_out=nic2.tail_of(name)(cast(out_fp)), input_prob=0.5, weight=0.0, b_range=(20, 2), warmup=0.2,
                       grad_hook=lambda gs: grads.extend(g.clone() for g in gs))          # (weight 0: no rounding term in a 1-iteration run)
    return list(ops2), grads, seen["flips"]


@pytest.mark.parametrize("name", list(ITERS))
def test_lu2022_full_size_units_match_oracle(nic_full, name, iters=None, long=False, natural=False):
    import copy
    from oracle import rdo_oracle as O, swin_oracle as S
    from quantization import BaseQuantBlock, QuantModel, QuantModule, block_reconstruction, layer_reconstruction
    from quantization.recon import unit_seed
    from quantization.swin_engine import TapeEngine
    model, state, cali = nic_full
    iters = ITERS[name] if iters is None else iters
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
    torch.manual_seed(1005)                                   # recon.unit_seed mixes the process seed into the QDrop stream
    qnn = QuantModel(model=copy.deepcopy(model).cuda().eval(), weight_quant_params=WQ, act_quant_params=AQ).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    xc = cali.cuda()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(xc[:B])                                           # scale init
    order = [n for n, m in qnn.model.named_children() if isinstance(m, (QuantModule, BaseQuantBlock))]
    for n in order[:order.index(name)]:                       # the prefix counts as calibrated: nearest-rounded W8 weights in front of the unit
        for m in getattr(qnn.model, n).modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = True
    unit = getattr(qnn.model, name)
    fn = layer_reconstruction if isinstance(unit, QuantModule) else block_reconstruction
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Lu2022")
    eng = fn(qnn, unit, name, cali_data=xc, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
             b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    assert isinstance(eng, TapeEngine) == (name != "g_s7")    # the closing layer has no tail: plain unit engine, transposed conv in phase form
    torch.cuda.synchronize()
    total, _, _ = eng.logs()
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
    inp_q, inp_fp, out_fp = nchw(eng.cq), nchw(eng.cf), nchw(eng.co)
    assert float((inp_q - inp_fp).abs().max()) > 0 or name in ("g_a0",)       # a quantised prefix is in front of every unit but the first
    idx = eng.idx.cpu().numpy()
    seed = unit_seed(name)
    # --- oracle on the same caches
    nic = S.NicOracle({k: v.clone() for k, v in state.items()}, CFG)
    unit_o = nic.stages[name]
    if isinstance(unit_o, S.RstbOracle):
        ops_o, fwd = unit_o.ops, (lambda ops_, x: unit_o(x))
    else:
        ops_o, fwd = {"layer": unit_o}, "layer"
    grads = []
    log = O.reconstruct_unit(fwd, ops_o, inp_q, inp_fp, out_fp, iters=iters, batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(seed, i, shape, 0.5), tail=nic.tail_of(name),
                             input_prob=0.5, weight=0.01, b_range=(20, 2), warmup=0.2,
                             grad_hook=lambda gs: (grads.extend(g.clone() for g in gs) if not grads else None))
    assert log.round[0] == 0.0 and log.round[-1] > 0.0        # the warm-up boundary lies inside the run
    for k, op in ops_o.items():
        e = eng.ops[k]
        if e.channel_wise and not e.is_ln:
            np.testing.assert_array_equal(e.delta.cpu().numpy(), op.delta.reshape(-1).numpy(), err_msg=k)
            np.testing.assert_array_equal(e.zp.cpu().numpy(), op.zp.reshape(-1).numpy(), err_msg=k)
    tail_round = name.startswith("g_a")
    # a round_ste flip moves the task term by (2|d| + 1) / (B H' W'): allow three of them
    tc = getattr(eng, "task_cache", None)
    atol = 3.0 * 3.0 / (B * tc.shape[1] * tc.shape[2]) if (tail_round and tc is not None) else 1e-7
    if long:
        pick = list(range(0, iters, 25)) + [iters - 1]
        # (round_ste flips accumulate over a long run: each moves the task term by ~(2|d| + 1) / n -- allow a handful)
        np.testing.assert_allclose(total.numpy()[pick], np.array(log.total)[pick], rtol=1e-3, atol=4 * atol)
        moved = same = tot = 0
        for k, op in ops_o.items():
            a_gpu, a_ref = eng.alpha_of(k).cpu(), op.alpha
            a0 = O.adaround_init_alpha(op.weight.clone(), op.delta)
            flips = float(((a_gpu >= 0) != (a_ref >= 0)).float().mean())
            assert flips < 1e-2, (k, flips)
            moved += int(((a_ref >= 0) != (a0 >= 0)).sum()); same += int(((a_gpu >= 0) == (a_ref >= 0)).sum()); tot += a_ref.numel()
        print(f"{name}: {iters} iterations, decisions moved against nearest rounding {moved / tot:.4f}, product == oracle {same / tot:.6f}")
        return
    np.testing.assert_allclose(total.numpy(), np.array(log.total), rtol=3e-4, atol=atol)
    for k, op in ops_o.items():
        a_gpu, a_ref = eng.alpha_of(k).cpu(), op.alpha
        assert a_gpu.shape == a_ref.shape, k
        far = ((a_gpu - a_ref).abs() > 2e-3).float().mean()
        flips = ((a_gpu >= 0) != (a_ref >= 0)).float().mean()
        assert float(far) < 2e-3 and float(flips) < 5e-3, (k, float(far), float(flips))
    # --- first-iteration gradient: the engine's data-parallel op sequence leaves d(rec + task)/d alpha in the bucket (plan A)
    from quantization.quantizer import UniformAffineQuantizer
    for m in unit.modules():                                  # back to the pre-calibration quantisers (same delta / zero point)
        if isinstance(m, QuantModule) and hasattr(m.weight_quantizer, "alpha"):
            u = UniformAffineQuantizer(**WQ, tconv=m.if_tconv)
            u.delta, u.zero_point, u.inited = m.weight_quantizer.delta, m.weight_quantizer.zero_point, True
            m.weight_quantizer = u
    tape = dict(tail=eng.tail, tail_round=eng.tail_round, task_cache=eng.task_cache) if isinstance(eng, TapeEngine) else {}
    eng2 = type(eng)(eng.kind, eng.mods, eng.cq, eng.cf, eng.co, batch_size=B, iters=iters, seed=seed, idx_table=torch.from_numpy(idx),
                     force_dp_split=True, weight=0.01, b_range=(20, 2), warmup=0.2, input_prob=0.5, **tape)
    eng2.plan_a.run(1, graph=False)
    torch.cuda.synchronize()
    # round_ste makes the task term's gradient 2 (round(z) - target) / n DISCONTINUOUS in z: one latent within fp32 noise of x.5 that
    # rounds the other way on the GPU moves every gradient of the unit by ~1 % (the target is the rounded FP latent, so only the few
    # elements where the quantised prefix changed a rounding contribute at all).  The gradient comparison therefore hands the
    # oracle the latents the ENGINE rounded to in this iteration (an input: the backward pass is what is compared) and bounds the
    # number of such events separately.
    zr_gpu = eng2.z_rounded.permute(0, 3, 1, 2).contiguous().cpu() if tail_round else None
    if tail_round or natural:
        _, grads, flips = _oracle_first_gradient(state, torch.float32, name, inp_q, inp_fp, out_fp, idx, seed, zr_gpu)
        if tail_round:
            print(f"{name}: latents rounded differently by the two sides in iteration 0: {flips} of {zr_gpu.numel()}")
            assert flips <= (8 if natural else 3)      # (latents of magnitude 10^2: an fp32 ulp is 10^2 x nearer to the next x.5)
    got = [_logical(op, op.dalpha).cpu() for op in eng2.ops.values()]
    rel = lambda a_, b_: float((a_.double() - b_.double()).abs().max() / (b_.double().abs().max() + 1e-300))
    if not natural:
        worst, bad = 0.0, []
        for (k, op), g, g_o in zip(eng2.ops.items(), got, grads):
            assert g.shape == g_o.shape, k
            r = rel(g, g_o)
            worst = max(worst, r)
            if not r < 2e-5:
                bad.append((k, r, float(g_o.abs().max())))
        assert not bad, bad
        print(f"{name}: loss rel {float(np.max(np.abs(total.numpy() - np.array(log.total)) / np.abs(np.array(log.total)))):.2e}, "
              f"worst first-iteration gradient rel {worst:.2e}")
        return
    # Natural statistics: heavy-tailed activations (kurtosis 10-20), LayerNorm gains up to 3, magnitudes of 10^2 through 5-14 Swin blocks
    # -- the gradients are ill-conditioned enough that fp32 ARITHMETIC ITSELF is only good to a few 1e-4 of a tensor's largest entry
    # (measured in the authoring container: the fp32 oracle against the fp64 oracle 2-3.5e-4 on g_a1, against 5e-6 on uniform noise).
    # A bar against the fp32 oracle alone would measure that noise twice.  So the float64 oracle is the reference here, and the fp32
    # oracle's own error against it the yardstick, tensor by tensor, for the Linear / conv weights.  FINDING (round 6): on
    # the g_a1 statistics (errors of 2-7e-4 on BOTH sides: the unit is ill-conditioned) the product's gradients are 1.0-3.8 x as far
    # from float64 as the fp32 oracle's (g_s4, errors of 2-5e-6: 1.0 x; uniform noise: within 1 x) -- the Swin path's GEMMs run on two fp16 planes per operand, 22 significant bits against fp32's 24, and heavy-tailed
    # sums are dominated by a few large terms whose operand rounding no longer averages out.  That is the format's bound (4 x the
    # operand rounding of fp32), not a fault to tune away: the bar is 5 x the fp32 oracle's error + 2e-5, the ratios are printed.  LayerNorm gains are one quantisation row per tensor: floor(w / delta) of a gain that sits on a grid point differs
    # between an fp32 and an fp64 evaluation (a different soft weight, not an arithmetic error), so they are compared with the fp32
    # oracle under the worst Linear-tensor bar.
    keys, g64, _ = _oracle_first_gradient(state, torch.float64, name, inp_q, inp_fp, out_fp, idx, seed, zr_gpu)
    rows, noise = [], 0.0
    for k, g, g32, g_64 in zip(keys, got, grads, g64):
        if not eng2.ops[k].is_ln:
            noise = max(noise, rel(g32, g_64))
    bad = []
    for k, g, g32, g_64 in zip(keys, got, grads, g64):
        assert g.shape == g32.shape, k
        if eng2.ops[k].is_ln:
            e_prod, e_ref, bar = rel(g, g32), None, 5.0 * noise + 2e-5
        else:
            e_prod, e_ref = rel(g, g_64), rel(g32, g_64)
            bar = 5.0 * e_ref + 2e-5
        rows.append((k, e_prod, e_ref))
        if not e_prod <= bar:
            bad.append((k, e_prod, e_ref, bar))
    print(f"{name} (natural statistics): first-iteration gradient error against the fp64 oracle, product / fp32 oracle: "
          + ", ".join(f"{k.split('.')[-1] if '.' in k else k} {e:.1e}/{('%.1e' % r) if r is not None else 'ln'}" for k, e, r in rows))
    assert not bad, bad
