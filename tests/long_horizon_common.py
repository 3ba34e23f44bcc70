"""Shared by tests/test_gpu_long_horizon.py and tools/make_long_horizon_golden.py: the seeded inputs of the long-horizon unit runs and
the oracle side of them.  Test infrastructure (imports oracle/): never imported by the product.

Two input sets ("stats"):
  uniform  the round-3..5 set: torch.rand calibration images, variance-preserving uniform weights (test_gpu_chained_flow._seed_model);
  kodak    natural-image statistics (missing 3): crops of the reference's Kodak images (tests/golden/kodak_crops.npz)
           through a model with 'trained-like' parameters (helpers.trained_like_: Laplace-tailed weights, per-channel scales over 2.5
           decades, non-diagonal GDN gamma, beta in [0.1, 10]).
The oracle trajectory of a (stats, unit) pair is a constant of the seeds: tools/make_long_horizon_golden.py stores what the checks need
(losses at the pick points, final hard decisions as bits, a signature of the caches) in tests/golden/long_horizon.npz so that the GPU
suite does not spend its time limit on CPU oracle loops; one unit per class stays on the LIVE oracle (`LIVE`)."""
import os

import numpy as np
import torch

SEED = 1005
N_IMG, B = 8, 4
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# (stats, unit) -> (iterations, loss check every ...)
RUNS = {
    ("uniform", "g_a.4"): (1000, 50), ("uniform", "g_a.5"): (1000, 50), ("uniform", "g_s.0"): (1000, 50), ("uniform", "g_s.2"): (1000, 50),
    ("uniform", "g_s.1"): (1000, 25), ("uniform", "g_a.2"): (300, 25), ("uniform", "g_a.3"): (300, 25), ("uniform", "g_s.3"): (300, 25),
    ("uniform", "g_s.4"): (300, 25), ("uniform", "g_a.1"): (100, 25), ("uniform", "g_s.5"): (100, 25),
    # the four 128^2 units over 100 iterations, one 64^2 unit of each GDN kind over 300
    ("kodak", "g_a.0"): (100, 25), ("kodak", "g_a.1"): (100, 25), ("kodak", "g_s.5"): (100, 25), ("kodak", "g_s.6"): (100, 25),
    ("kodak", "g_a.2"): (300, 25), ("kodak", "g_s.3"): (300, 25),
    ("kodak", "g_a.5"): (300, 25),         # a cheap natural-statistics unit that stays on the live oracle in every GPU run
}
FP32_UNITS = {"g_s.0"}                                     # 16^2: below the plane path's size threshold
# one unit per class on the live oracle in every GPU run (ResidualBlock on fp32 MFMA, ResidualBlock on planes, ResidualBlockWithStride,
# ResidualBlockUpsample, and one natural-statistics unit); the others against the committed trajectories
LIVE = {("uniform", "g_s.0"), ("uniform", "g_a.5"), ("uniform", "g_a.4"), ("uniform", "g_s.1"), ("kodak", "g_a.5")}


def build(stats):
    """-> (FlowOracle with the whole prefix 'calibrated' at nearest rounding, calibration images, {unit name: oracle module})"""
    from oracle import lic_oracle as L
    from oracle.cheng_units import schedule
    from oracle.flow_oracle import FlowOracle
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED)
    model = L.Cheng2020Anchor(N=192).eval()
    if stats == "uniform":
        from test_gpu_chained_flow import _seed_model
        _seed_model(model, g)
        cali = torch.rand(N_IMG, 3, 256, 256, generator=g)
    elif stats == "kodak":
        from helpers import kodak_crops, trained_like_
        crops = kodak_crops(GOLDEN)
        with torch.no_grad():
            trained_like_(model, g, probe=crops[N_IMG:N_IMG + 4])
        cali = crops[:N_IMG]
    else:
        raise KeyError(stats)
    flow = FlowOracle(model)
    for u in flow.units:                      # the whole prefix "calibrated": AdaRound at its initial logits = nearest rounding, hard
        for op in u.ops.values():
            op.to_adaround()
        u.trained = True
    mods = {n: m for n, _, _, m in schedule(model)}
    return flow, cali, mods


def latents(flow, cali):
    """The rounded latents y_hat = round(g_a(x)) of the calibration images in the two states a synthesis unit's caches are built from
    (full precision / hard-quantised prefix), as int16.  Rounding is a discontinuity: a latent within fp32 noise of x.5 rounds the
    other way on another CPU and moves every cache behind it by a whole quantisation step (observed between the authoring container
    and the GPU box's host: 1.7e-3 of g_s.0's first loss).  The committed trajectories of the g_s units therefore come WITH the
    latents they were computed from, and the test builds the caches of those units from the stored latents (`caches`)."""
    out = {}
    with torch.no_grad():
        for state in ("fp", "prefix"):
            flow._set_modes(state)
            y = flow._run("g_a", cali, False, None)
            yh = flow.model.gaussian_conditional.quantize(y, "dequantize")
            assert float(yh.abs().max()) < 32000 and bool((yh == yh.round()).all())
            out[state] = yh.to(torch.int16).numpy()
    return out


def caches(flow, cali, name, lat=None):
    """(x_q, x_fp, target) of unit `name`: FlowOracle.caches, or -- for a synthesis unit when the latents are given -- the synthesis
    prefix alone run from them (same modules, same batch: bit-identical to FlowOracle.caches on the machine the latents come from)."""
    if lat is None or not name.startswith("g_s"):
        return flow.caches(name, cali)
    from oracle.flow_oracle import _Tap
    got = {}
    with torch.no_grad():
        for state in ("fp", "prefix"):
            flow._set_modes(state)
            try:
                flow._run("g_s", torch.from_numpy(lat[state]).float(), False, name)
                raise RuntimeError(f"unit {name} was not reached")
            except _Tap as t:
                got[state] = (t.inp.clone(), t.out.clone())
    return got["prefix"][0], got["fp"][0], got["fp"][1]


def idx_stream(iters):
    return np.stack([np.random.RandomState(500 + i).permutation(N_IMG)[:B] for i in range(iters)])


def picks(iters, every):
    return list(range(0, iters, every)) + [iters - 1]


def cache_signature(xq, xf, tg):
    """six numbers that identify the caches a trajectory belongs to (double sums: insensitive to the last bits a different CPU gives)"""
    return np.array([float(t.double().sum()) for t in (xq, xf, tg)] + [float(t.double().abs().sum()) for t in (xq, xf, tg)])


def oracle_run(flow, cali, name, iters, lat=None):
    """the oracle's trajectory of unit `name` (block_opt.py:287-311 restated, torch CPU fp32) -> (log, unit, caches)"""
    import copy
    from oracle import rdo_oracle as O
    xq, xf, tg = caches(flow, cali, name, lat)
    # a COPY of the unit is trained: the flow stays what `build` made it (every unit at nearest rounding), so the caches of a unit do
    # not depend on which other units were run before it -- in the generator or in the test, in any order or selection
    u = copy.deepcopy(flow.by_name[name])
    for op in u.ops.values():
        op.init_scale()
    log = O.reconstruct_unit(u.kind, u.ops, xq, xf, tg, iters=iters, batch_size=B, idx_stream=idx_stream(iters),
                             mask_fn=lambda i, shape: O.qdrop_keep_mask_nhwc(SEED, i, shape, 0.5), input_prob=0.5,
                             weight=0.01, b_range=(20, 2), warmup=0.2)
    return log, u, (xq, xf, tg)


def summary(log, u, iters, every, caches):
    """what the test compares, as arrays: losses at the pick points, the final hard decisions (alpha >= 0) of every weight tensor as
    packed bits, the fraction of decisions the run moved against nearest rounding, the cache signature"""
    from oracle import rdo_oracle as O
    pk = picks(iters, every)
    out = {"rt": (np.array(log.rec) + np.array(log.task))[pk], "round": np.array(log.round)[pk], "total": np.array(log.total)[pk],
           "round_first_last": np.array([log.round[0], log.round[-1]]), "cache_sig": cache_signature(*caches)}
    for n, op in u.ops.items():
        a0 = O.adaround_init_alpha(op.weight.clone(), op.delta)
        out[f"bits/{n}"] = np.packbits((op.alpha >= 0).numpy().reshape(-1))
        out[f"shape/{n}"] = np.array(op.alpha.shape)
        out[f"moved/{n}"] = np.array(float(((op.alpha >= 0) != (a0 >= 0)).float().mean()))
    return out
