"""Realistic-horizon trajectories of the block units of Cheng2020-anchor N=192 against the ORACLE (VERDICT round 3, weak 1c /
next 1b; round 5: missing 3, weak 2): the units run on H2 tensors -- fp16 planes whose power-of-two scales are fixed by the probe
iterations before the plan is recorded and must hold while the temperature b decays from 20 to 2, the rounding loss switches on and
Adam's moments build up -- and g_s.0 (ResidualBlock at 16^2) on the fp32-MFMA kernels.

Two input sets (tests/long_horizon_common.py): `uniform` (torch.rand images, variance-preserving weights: rounds 3-5) and `kodak` --
crops of the reference's Kodak images through a model with trained-like parameters (Laplace-tailed weights, per-channel scales over
2.5 decades, non-diagonal GDN gamma, beta in [0.1, 10]): heavy-tailed activations (kurtosis 10-70 against 1.8 for uniform noise),
per-channel ranges over 2-3 orders of magnitude, smooth regions -- where static plane scales, the second plane's small values and the
restart ladder are exercised.  1000 iterations for the <= 32^2 units (warm-up boundary at 200), 300 at 64^2, 100 at 128^2, QDrop on,
product `UnitEngine` with its DEFAULT switches against `oracle.reconstruct_unit` (block_opt.py:287-311 restated, torch CPU fp32) from
the same caches, mini-batch index stream and counter-RNG masks.  The quantised-prefix input x_q is what the reference would cache for
the unit when every unit in front of it is hard-quantised (`FlowOracle.caches`): real quantisation noise, not a perturbed copy.

The oracle side is a constant of the seeds: for the units in `LIVE` (one per unit class + one natural-statistics unit) it is computed
in the test, for the others it comes from tests/golden/long_horizon.npz (tools/make_long_horizon_golden.py; the caches are rebuilt
here and must carry the fixture's signature).  Checked: rec + task, round and total loss at every pick point to 1e-3 relative; final
hard rounding decisions >= 99 % identical per weight tensor; the plane path taken and what it did (scales held, or a restart that still
matches: reported, never tuned away); no unreported H2 overflow (`logs()` handles the sticky flag)."""
import os

import numpy as np
import pytest
import torch

import long_horizon_common as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    cache = {}

    def get(stats):
        if stats not in cache:
            cache[stats] = C.build(stats)
        return cache[stats]
    return get


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(C.GOLDEN, "long_horizon.npz"))


@pytest.mark.parametrize("stats,name", list(C.RUNS), ids=[f"{s}-{n}" for s, n in C.RUNS])
def test_long_horizon_unit_matches_oracle(built, golden, stats, name):
    from quantization.engine import UnitEngine
    from test_gpu_fullsize_units import _product_unit
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
    flow, cali, mods = built(stats)
    iters, every = C.RUNS[(stats, name)]
    key = f"{stats}/{name}"
    u = flow.by_name[name]
    # synthesis units: caches from the committed rounded latents (a latent within fp32 noise of x.5 rounds differently on another CPU)
    lat = {k: golden[f"{stats}/y_hat/{k}"] for k in ("fp", "prefix")}
    if (stats, name) in C.LIVE:
        log, u, (xq, xf, tg) = C.oracle_run(flow, cali, name, iters, lat=lat)
        want = C.summary(log, u, iters, every, (xq, xf, tg))
        if f"{key}/total" in golden:       # the committed trajectory of a live unit is the same constant
            np.testing.assert_allclose(want["total"], golden[f"{key}/total"], rtol=1e-3, atol=1e-7)
    else:
        assert tuple(golden[f"{key}/iters"]) == (iters, every), "tests/golden/long_horizon.npz is stale: run tools/make_long_horizon_golden.py"
        want = {k[len(key) + 1:]: golden[k] for k in golden.files if k.startswith(key + "/")}
        xq, xf, tg = C.caches(flow, cali, name, lat)
        np.testing.assert_allclose(C.cache_signature(xq, xf, tg), want["cache_sig"], rtol=1e-5, err_msg="caches differ from the fixture's")
    assert name == "g_a.0" or float((xq - xf).abs().max()) > 0.0      # the prefix really is quantised (the first unit has none)
    pm = _product_unit(u.kind, mods[name])
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    eng = UnitEngine(u.kind, pm, nh(xq), nh(xf), nh(tg), batch_size=C.B, iters=iters, weight=0.01, b_range=(20, 2), warmup=0.2,
                     input_prob=0.5, seed=C.SEED, idx_table=torch.from_numpy(C.idx_stream(iters)))
    assert (eng.h2_plan == u.kind) == (name not in C.FP32_UNITS), (name, eng.h2_plan)
    scales0 = dict(eng.scales)
    eng.run()
    torch.cuda.synchronize()
    total, rt, rd = eng.logs()                            # an overflow of the planes' fp16 range restarts the unit in here
    # what the plane path did: the probe-time scales held for the whole run, or the restart ladder fired -- a finding to report
    # (DESIGN 4), and the result must match the oracle either way
    held = eng.scales == scales0 and eng.h2_restarts == 0
    print(f"{key}: plan {eng.h2_plan}, scales {'held' if held else 'RE-DERIVED'}, restarts {eng.h2_restarts}, use_h2 {eng.use_h2}, "
          f"probe magnitudes { {k: f'{v:.3g}' for k, v in eng._amax.items()} }")
    assert held or eng.h2_restarts > 0
    if stats == "uniform":
        assert held, "uniform-noise inputs have never outgrown their probed scales"
    np.testing.assert_allclose(rt.numpy()[C.picks(iters, every)], want["rt"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(rd.numpy()[C.picks(iters, every)], want["round"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(total.numpy()[C.picks(iters, every)], want["total"], rtol=1e-3, atol=1e-7)
    assert want["round_first_last"][0] == 0.0 and want["round_first_last"][1] > 0.0
    for n in u.ops:
        a_gpu = eng.alpha_of(n).cpu()
        assert tuple(a_gpu.shape) == tuple(want[f"shape/{n}"]), n
        ref_bits = np.unpackbits(want[f"bits/{n}"])[:a_gpu.numel()].astype(bool)
        flips = float(((a_gpu >= 0).numpy().reshape(-1) != ref_bits).mean())
        assert flips < 1e-2, (key, n, flips, float(want[f"moved/{n}"]))
        print(f"{key}.{n}: decisions changed vs nearest {float(want[f'moved/{n}']):.4f}, product != oracle {flips:.5f}")
