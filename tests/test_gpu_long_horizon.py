"""Realistic-horizon trajectories of the block units of Cheng2020-anchor N=192 against the ORACLE (VERDICT round 3, weak 1c /
next 1b): g_a.4 (ResidualBlockWithStride -> 32^2), g_a.5, g_s.2 (ResidualBlocks at 32^2) run on H2 tensors -- fp16 planes whose
power-of-two scales are fixed by ONE probe iteration before the plan is recorded and must hold while the temperature b decays from
20 to 2, the rounding loss switches on and Adam's moments build up --, g_s.0 (ResidualBlock at 16^2) on the fp32-MFMA kernels.

1000 iterations per unit for the <= 32^2 units (warm-up boundary at 200) -- round 5 adds the 64^2 and 128^2 units, see UNITS below --, QDrop on, on the product `UnitEngine` with its DEFAULT switches and on
`oracle.reconstruct_unit` (block_opt.py:287-311 restated, torch CPU fp32) from the same caches, mini-batch index stream and counter-RNG
masks.  The quantised-prefix input x_q is what the reference would cache for the unit when every unit in front of it is
hard-quantised (`FlowOracle.caches` with the prefix at its initial -- nearest -- rounding): real quantisation noise, not a perturbed copy.

Checked: rec + task, round and total loss at every 50th iteration (and the last) to 1e-3 relative; final hard rounding decisions
>= 99 % identical per weight tensor; no H2 overflow (`logs()` raises on the sticky flag).  The oracle costs 0.02-0.05 s per iteration
of these units on the GPU box's 16 host cores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 1005
N_IMG, B = 8, 4
# unit -> (iterations, loss check every ...).  Round 5 (missing 3 / weak 2) adds the ResidualBlockUpsample class (IGDN +
# sub-pixel convs: g_s.1 over 1000 iterations), the four 64^2 units over 300 iterations (warm-up boundary at 60) and one 128^2 unit of
# each kind that carries the step's time -- g_a.1 (ResidualBlock) and g_s.5 (ResidualBlockUpsample) -- over 100 iterations: the units
# where the fp16 range of the H2 planes matters most (largest reductions, largest gradients), with the probe-time scales held.
UNITS = {"g_a.4": (1000, 50), "g_a.5": (1000, 50), "g_s.0": (1000, 50), "g_s.2": (1000, 50), "g_s.1": (1000, 25),
         "g_a.2": (300, 25), "g_a.3": (300, 25), "g_s.3": (300, 25), "g_s.4": (300, 25), "g_a.1": (100, 25), "g_s.5": (100, 25)}
FP32_UNITS = {"g_s.0"}                                     # 16^2: below the plane path's size threshold


@pytest.fixture(scope="module")
def long_caches():
    from oracle import lic_oracle as L
    from oracle.cheng_units import schedule
    from oracle.flow_oracle import FlowOracle
    from test_gpu_chained_flow import _seed_model
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED)
    model = L.Cheng2020Anchor(N=192).eval()
    _seed_model(model, g)
    cali = torch.rand(N_IMG, 3, 256, 256, generator=g)
    flow = FlowOracle(model)
    for u in flow.units:                      # the whole prefix "calibrated": AdaRound at its initial logits = nearest rounding, hard
        for op in u.ops.values():
            op.to_adaround()
        u.trained = True
    mods = {n: m for n, _, _, m in schedule(model)}
    return flow, cali, mods


@pytest.mark.parametrize("name", list(UNITS))
def test_long_horizon_unit_matches_oracle(long_caches, name):
    from oracle import rdo_oracle as O
    from quantization.engine import UnitEngine
    from test_gpu_fullsize_units import _product_unit
    import os
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
    flow, cali, mods = long_caches
    ITERS, every = UNITS[name]
    u = flow.by_name[name]
    xq, xf, tg = flow.caches(name, cali)
    assert float((xq - xf).abs().max()) > 0.0            # the prefix really is quantised
    idx = np.stack([np.random.RandomState(500 + i).permutation(N_IMG)[:B] for i in range(ITERS)])
    for op in u.ops.values():
        op.init_scale()
    log = O.reconstruct_unit(u.kind, u.ops, xq, xf, tg, iters=ITERS, batch_size=B, idx_stream=idx,
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5), input_prob=0.5,
                             weight=0.01, b_range=(20, 2), warmup=0.2)
    pm = _product_unit(u.kind, mods[name])
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    eng = UnitEngine(u.kind, pm, nh(xq), nh(xf), nh(tg), batch_size=B, iters=ITERS, weight=0.01, b_range=(20, 2), warmup=0.2,
                     input_prob=0.5, seed=SEED, idx_table=torch.from_numpy(idx))
    assert (eng.h2_plan == u.kind) == (name not in FP32_UNITS), (name, eng.h2_plan)
    scales0 = dict(eng.scales)
    eng.run()
    torch.cuda.synchronize()
    total, rt, rd = eng.logs()                            # raises if a value left the fp16 range of its planes
    assert eng.scales == scales0 or getattr(eng, "h2_restarts", 0) > 0
    pick = list(range(0, ITERS, every)) + [ITERS - 1]
    np.testing.assert_allclose(rt.numpy()[pick], (np.array(log.rec) + np.array(log.task))[pick], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(rd.numpy()[pick], np.array(log.round)[pick], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(total.numpy()[pick], np.array(log.total)[pick], rtol=1e-3, atol=1e-7)
    assert log.round[0] == 0.0 and log.round[-1] > 0.0
    for n, op in u.ops.items():
        a_gpu, a_ref = eng.alpha_of(n).cpu(), op.alpha
        assert a_gpu.shape == a_ref.shape, n
        a0 = O.adaround_init_alpha(op.weight.clone(), op.delta)
        moved = float(((a_ref >= 0) != (a0 >= 0)).float().mean())      # decisions the run changed against nearest rounding (diagnostic)
        flips = float(((a_gpu >= 0) != (a_ref >= 0)).float().mean())
        assert flips < 1e-2, (name, n, flips, moved)
        print(f"{name}.{n}: decisions changed vs nearest {moved:.4f}, product != oracle {flips:.5f}")
