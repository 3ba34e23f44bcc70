"""Chained end-to-end parity: the WHOLE calibration flow of main2.py:214-282 -- every unit
calibrated on caches produced by the already calibrated prefix, then the W8 and the W8A8 evaluation -- on the product (HIP, public
`layer_reconstruction` / `block_reconstruction` API) against `oracle.flow_oracle.FlowOracle` running the same flow on the CPU from
the same weights, calibration images, mini-batch index streams (recorded from the product's engines: inputs, not expected values)
and counter-RNG QDrop masks.

Compared: the hard rounding decision of EVERY weight of every layer, W8 and W8A8 bpp / PSNR on held-out images.  Tolerances (fp32
summation order only; a decision flips where an alpha ends within Adam noise of zero; in W8A8 a value within float noise of a boundary
of one of the cascaded dynamic 8-bit grids moves one level): per layer at least 99.5 % identical decisions (layers of fewer than 400
weights: at most 2 differing), over the model >= 99.8 %; W8 bpp 1e-3 relative, PSNR 0.02 dB; W8A8 bpp 2e-3 relative, PSNR 0.05 dB."""
import math
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
SEED = 1005


def _seed_model(ref, g):
    from oracle import lic_oracle as L
    with torch.no_grad():
        # variance-preserving conv weights: with torch's default init the activations shrink layer by layer and the latents of a
        # random model collapse to zero (bpp and x_hat would not respond to the weights at all)
        for name, p in ref.named_parameters():
            if p.dim() == 4 and "entropy_bottleneck" not in name:
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 2 * (3.0 / p[0].numel()) ** 0.5)
        for m in ref.modules():
            if isinstance(m, L.GDN):
                c = m.gamma.shape[0]
                m.gamma.copy_(torch.sqrt(0.1 * torch.eye(c) + 0.01 * torch.rand(c, c, generator=g) + 2.0 ** -36))
                m.beta.copy_(torch.sqrt(0.5 + torch.rand(c, generator=g) + 2.0 ** -36))
        cp = getattr(ref, "context_prediction", None)
        if cp is not None:
            cp.weight.data *= cp.mask          # a trained checkpoint carries the masked weight; the wrapper never re-applies the mask


def _sync_state(dst, src):
    sd = src.state_dict()
    with torch.no_grad():
        for k, v in dst.state_dict().items():
            v.copy_(sd[k])


def _product_flow(qnn, cali, B, iters, last_layer):
    """main2.py:214-263 on the drop-in package; -> {dotted unit name: engine} in recon_model order."""
    from quantization import BaseQuantBlock, QuantModule, block_reconstruction, layer_reconstruction
    args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Cheng2020")
    base = dict(cali_data=cali, batch_size=B, weight=0.01, input_prob=0.5, lr=4e-5, asym=True, b_range=(20, 2),
                warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
    it_of = iters if callable(iters) else (lambda full_name: iters)       # iterations per unit (main2.py passes one --iters_w for all)
    engines = {}

    def recon_model(m: nn.Module, prefix):
        for name, module in m.named_children():
            kwargs = dict(base, iters=it_of(prefix + name))
            if isinstance(module, QuantModule):
                eng = layer_reconstruction(qnn, module, name, **kwargs)
                if eng is not None:
                    engines[prefix + name] = eng
            elif isinstance(module, BaseQuantBlock):
                engines[prefix + name] = block_reconstruction(qnn, module, name, **kwargs)
            else:
                recon_model(module, prefix + name + ".")
    qnn.set_quant_state(weight_quant=True, act_quant=False)
    last_layer(qnn).set_quant_state(True, False)
    recon_model(qnn.model, "")
    return engines


def _compare(engines, flow, qnn, last_layer, test_imgs):
    from test_datasets import evaluate_images
    assert list(engines) == [u.name for u in flow.units]
    from oracle import rdo_oracle as O
    same = total = moved = 0
    for u in flow.units:
        eng = engines[u.name]
        for n, op in u.ops.items():
            a_gpu, a_ref = eng.alpha_of(n).cpu(), op.alpha
            moved += int(((a_ref >= 0) != (O.adaround_init_alpha(op.weight.clone(), op.delta) >= 0)).sum())   # vs nearest rounding (diagnostic)
            assert a_gpu.shape == a_ref.shape, (u.name, n)
            diff = int(((a_gpu >= 0) != (a_ref >= 0)).sum())
            numel = a_ref.numel()
            assert diff <= max(2, 0.005 * numel), (u.name, n, diff, numel)
            same += numel - diff
            total += numel
    assert same >= 0.998 * total, (same, total)
    print(f"decisions the calibration changed against nearest rounding: {moved} of {total} ({100.0 * moved / total:.3f} %); product != oracle: {total - same}")
    res = {}
    for act, (tol_bpp, tol_psnr) in ((False, (1e-3, 0.02)), (True, (2e-3, 0.05))):
        qnn.set_quant_state(weight_quant=True, act_quant=act)
        last_layer(qnn).set_quant_state(True, False)
        psnr, bpp = evaluate_images(qnn.eval(), test_imgs, p=64)
        psnr_o, bpp_o = flow.evaluate(test_imgs, p=64, act_quant=act)
        assert math.isfinite(psnr) and bpp > 0
        assert abs(bpp - bpp_o) <= tol_bpp * bpp_o, (act, bpp, bpp_o)
        assert abs(psnr - psnr_o) <= tol_psnr, (act, psnr, psnr_o)
        res[act] = (psnr, bpp, psnr_o, bpp_o)
    return same / total, res


def _run(arch, N, iters, n_img=8, crop=64, test_hw=((96, 80), (96, 80)), loss_rtol=5e-3):
    import lic
    from oracle import lic_oracle as L
    from oracle.flow_oracle import FlowOracle
    from quantization import QuantModel
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED)
    if arch == "cheng":
        ref, prod = L.Cheng2020Anchor(N=N).eval(), lic.Cheng2020Anchor(N=N).eval()
        last_layer = lambda q: q.model.g_s[-1][0]
    else:
        ref, prod = L.MeanScaleHyperprior(N=N, M=N * 3 // 2).eval(), lic.MeanScaleHyperprior(N=N, M=N * 3 // 2).eval()
        last_layer = lambda q: q.model.g_s[-1]
    _seed_model(ref, g)
    _sync_state(prod, ref)
    B = 4
    cali = torch.rand(n_img, 3, crop, crop, generator=g)
    test_imgs = [torch.rand(1, 3, h, w, generator=g) for h, w in test_hw]
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(model=prod.cuda(), weight_quant_params=wq, act_quant_params=dict(wq, leaf_param=False), is_cheng=arch == "cheng").cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:B].cuda())
    torch.manual_seed(SEED)            # main2.py seed_all: unit seeds and the randperm stream start here
    engines = _product_flow(qnn, cali.cuda(), B, iters, last_layer)
    flow = FlowOracle(ref)
    idx = {name: e.idx.cpu().numpy() for name, e in engines.items()}
    for u in flow.units:               # the two sides derive the same QDrop key for every unit
        assert engines[u.name].seed == FlowOracle.unit_seed(SEED, u.local), u.name
    logs = flow.recon_model(cali, idx, SEED, iters=(lambda n: engines[n].iters) if callable(iters) else iters, batch_size=B)
    # first and last iteration's loss of every unit, product vs oracle: the chain has not drifted apart
    for u in flow.units:
        tot = engines[u.name].logs()[0].numpy()
        np.testing.assert_allclose(tot[[0, -1]], np.array(logs[u.name].total)[[0, -1]], rtol=loss_rtol, atol=1e-6, err_msg=u.name)
    return _compare(engines, flow, qnn, last_layer, test_imgs)


def test_chained_flow_toy_cheng2020_matches_oracle_flow():
    agree, res = _run("cheng", 16, 120)
    print("cheng2020 toy: identical rounding decisions", agree, "W8", res[False], "W8A8", res[True])


def test_chained_flow_toy_minnen2018_matches_oracle_flow():
    agree, res = _run("minnen", 16, 120)
    print("minnen2018 toy: identical rounding decisions", agree, "W8", res[False], "W8A8", res[True])


def _compare_golden(engines, gold, key, qnn, last_layer, test_imgs):
    """`_compare` against the committed oracle flow (tests/golden/flow_n192.npz, tools/make_flow_golden.py): the same bars -- per weight
    tensor at most max(2, 0.5 %) differing decisions, over the model >= 99.8 % identical -- on the fixture's fixed 1-in-8 sample of
    every tensor; W8 / W8A8 bpp and PSNR against the stored oracle values."""
    import flow_common as F
    from test_datasets import evaluate_images
    assert list(engines) == list(gold[f"{key}/units"])
    same = total = moved = moved_of = 0
    for name, eng in engines.items():
        np.testing.assert_array_equal(eng.idx.cpu().numpy(), gold[f"{key}/{name}/idx"], err_msg=f"{name}: the engines drew other index "
                                      "tables than the fixture's oracle flow (tests/golden/flow_n192.npz is stale)")
        tot = eng.logs()[0].numpy()
        # Units BEHIND a rounded latent (z_hat, y_hat): their first loss is the effect of the few latents whose rounding the quantised
        # prefix changed -- 0.1 against last losses of 10^3-10^4 -- and ONE latent within fp32 noise of x.5 that rounds the other way
        # on the GPU than on the CPU the fixture was made on moves it by 10 % and more (observed: h_s.2.0 0.120 against 0.105).  With
        # a live oracle that was a ~5 % event per run; against a constant fixture it is either always or never there, so those units'
        # FIRST loss is only checked for its order of magnitude; their last loss, every unit's rounding decisions and the evaluation
        # metrics below carry the parity statement.
        behind = name.startswith(("h_s", "g_s", "entropy_parameters", "context_prediction"))
        want = gold[f"{key}/{name}/total_first_last"]
        np.testing.assert_allclose(tot[-1], want[1], rtol=5e-3, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(tot[0], want[0], rtol=0.5 if behind else 5e-3, atol=1e-6, err_msg=name)
        if behind and abs(tot[0] - want[0]) > 5e-3 * abs(want[0]):
            print(f"{key}/{name}: first loss {tot[0]:.6g} against the fixture's {want[0]:.6g} (a latent rounded the other way)")
        for n in eng.ops:
            a = (eng.alpha_of(n).cpu() >= 0).numpy().reshape(-1)
            assert a.size == int(gold[f"{key}/{name}/numel/{n}"]), (name, n)
            ref = np.unpackbits(gold[f"{key}/{name}/bits/{n}"])[:len(a[::F.SAMPLE])].astype(bool)
            diff, numel = int((a[::F.SAMPLE] != ref).sum()), ref.size
            assert diff <= max(2, 0.005 * numel), (name, n, diff, numel)
            same += numel - diff
            total += numel
            moved += int(gold[f"{key}/{name}/moved/{n}"]); moved_of += a.size
    assert same >= 0.998 * total, (same, total)
    print(f"{key}: decisions the calibration changed against nearest rounding: {moved} of {moved_of} ({100.0 * moved / moved_of:.3f} %); "
          f"product != oracle on the 1-in-{F.SAMPLE} sample: {total - same} of {total}")
    res = {}
    for act, gk, (tol_bpp, tol_psnr) in ((False, "w8", (1e-3, 0.02)), (True, "w8a8", (2e-3, 0.05))):
        qnn.set_quant_state(weight_quant=True, act_quant=act)
        last_layer(qnn).set_quant_state(True, False)
        psnr, bpp = evaluate_images(qnn.eval(), test_imgs, p=64)
        psnr_o, bpp_o = (float(v) for v in gold[f"{key}/{gk}"])
        assert math.isfinite(psnr) and bpp > 0
        assert abs(bpp - bpp_o) <= tol_bpp * bpp_o, (act, bpp, bpp_o)
        assert abs(psnr - psnr_o) <= tol_psnr, (act, psnr, psnr_o)
        res[act] = (psnr, bpp, psnr_o, bpp_o)
    return same / total, res


@pytest.mark.parametrize("stats", ["uniform", "kodak"])
def test_chained_flow_full_size_cheng2020_n192_matches_oracle_flow(stats, golden_dir):
    """BASELINE config 2 at FULL width: Cheng2020-anchor N=192 on 256 x 256 crops, all 29 units through the
    public layer_/block_reconstruction API -- the H2 / halo / row / split-K / fused-tail kernels at the sizes the bench runs them --,
    every unit on caches of the product's own calibrated prefix, against the oracle flow doing the same on the CPU; then W8 and W8A8
    bpp / PSNR on held-out images.  Same bars as the toy-width flows.  `kodak` (missing 3): crops of the reference's
    Kodak images and trained-like parameters (tests/flow_common.py) -- the chain of 29 units, each calibrated behind the quantisation
    noise of the calibrated prefix, on heavy-tailed activations.

    Iterations per unit (round 5): 12 on the 128^2 / 64^2 units (their long horizons are
    tests/test_gpu_long_horizon.py), 80 on everything from 32^2 down -- g_a.4-6, the hyper path, g_s.0-2, the entropy-parameter and
    context layers: the rounding loss is on for 64 of them and alphas near zero change sign, so "identical decisions" is a statement
    about the trained rounding, not about two implementations of nearest rounding (the print shows the share that moved).

    The oracle flow is a constant of the seeds and comes from tests/golden/flow_n192.npz (round 6; RDO_LIVE_ORACLE_FLOW=1 recomputes
    it here, ~80 s on the GPU box's host cores, and cross-checks the fixture)."""
    import os
    import lic
    import flow_common as F
    from quantization import QuantModel
    ref, cali, test_imgs = F.build(stats)
    prod = lic.Cheng2020Anchor(N=192).eval()
    _sync_state(prod, ref)
    last_layer = lambda q: q.model.g_s[-1][0]
    wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
    qnn = QuantModel(model=prod.cuda(), weight_quant_params=wq, act_quant_params=dict(wq, leaf_param=False), is_cheng=True).cuda().eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(cali[:F.B].cuda())
    torch.manual_seed(SEED)            # main2.py seed_all: unit seeds and the randperm stream start here
    engines = _product_flow(qnn, cali.cuda(), F.B, F.iters_of, last_layer)
    restarts = {n: e.h2_restarts for n, e in engines.items() if e.h2_restarts}
    print(f"{stats}: plane plans {sorted(n for n, e in engines.items() if getattr(e, 'h2_plan', None))}; restarts {restarts or 'none'}")
    gold = np.load(os.path.join(golden_dir, "flow_n192.npz"))
    if os.environ.get("RDO_LIVE_ORACLE_FLOW") == "1":
        from oracle.flow_oracle import FlowOracle
        idx = {name: e.idx.cpu().numpy() for name, e in engines.items()}
        flow, logs, evals, _ = F.oracle_flow(ref, cali, test_imgs, idx=idx)
        for u in flow.units:
            assert engines[u.name].seed == FlowOracle.unit_seed(SEED, u.local), u.name
            np.testing.assert_allclose(np.array(logs[u.name].total)[[0, -1]], gold[f"{stats}/{u.name}/total_first_last"], rtol=1e-4)
        agree, res = _compare(engines, flow, qnn, last_layer, test_imgs)
    else:
        agree, res = _compare_golden(engines, gold, stats, qnn, last_layer, test_imgs)
    print(f"cheng2020 N=192 ({stats}): identical rounding decisions", agree, "W8", res[False], "W8A8", res[True])
