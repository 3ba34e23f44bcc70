"""Shared by tests/test_gpu_chained_flow.py and tools/make_flow_golden.py: the seeded inputs of the N=192 chained calibration flows and
the ORACLE side of them (test infrastructure: imports oracle/).

The oracle flow does not depend on the product: weights, calibration images, the mini-batch index stream (the draws main2.py's seed_all
fixes: one torch.randperm(n) per iteration, unit after unit -- `idx_tables` replays them; the test checks that the product's engines
drew the same tables) and the QDrop keys are all functions of the seeds.  tools/make_flow_golden.py stores what the comparison needs in
tests/golden/flow_n192.npz -- first / last loss of every unit, the final hard decisions of every weight tensor on a fixed 1-in-8
sample, the count of decisions moved, W8 / W8A8 bpp and PSNR -- so that the GPU suite does not spend 80 s per flow on the CPU oracle
(VERDICT round 5, weak 2 / next 7).  Input sets as in tests/long_horizon_common.py: `uniform` and `kodak`."""
import os

import numpy as np
import torch

SEED = 1005
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_IMG, B = 8, 4
SAMPLE = 8                                                # decisions kept in the fixture: flat index % SAMPLE == 0
BIG = {"g_a.0", "g_a.1", "g_a.2", "g_a.3", "g_s.3", "g_s.4", "g_s.5", "g_s.6", "g_s.7.0"}


def iters_of(name):
    """12 iterations on the 128^2 / 64^2 units (their long horizons: tests/test_gpu_long_horizon.py), 80 from 32^2 down"""
    return 12 if name in BIG else 80


def build(stats):
    """-> (oracle Cheng2020-anchor N=192, calibration images [8,3,256,256], held-out evaluation images)"""
    from oracle import lic_oracle as L
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED)
    ref = L.Cheng2020Anchor(N=192).eval()
    if stats == "uniform":
        from test_gpu_chained_flow import _seed_model
        _seed_model(ref, g)
        cali = torch.rand(N_IMG, 3, 256, 256, generator=g)
        test_imgs = [torch.rand(1, 3, 512, 768, generator=g)]
    elif stats == "kodak":
        from helpers import kodak_crops, trained_like_
        crops = kodak_crops(GOLDEN)
        with torch.no_grad():
            trained_like_(ref, g, probe=crops[N_IMG:N_IMG + 4])
        cali = crops[:N_IMG]
        test_imgs = [crops[12:13].clone(), crops[13:14].clone()]      # held out: neither calibration nor probe images
    else:
        raise KeyError(stats)
    return ref, cali, test_imgs


def idx_tables(unit_names):
    """the index tables the product's engines draw: torch.manual_seed(SEED), then per unit `iters` x torch.randperm(n)[:B]
    (layer_opt.py:289 through engine.IdxStream)"""
    torch.manual_seed(SEED)
    return {n: torch.stack([torch.randperm(N_IMG)[:B] for _ in range(iters_of(n))]).numpy() for n in unit_names}


def oracle_flow(ref, cali, test_imgs, idx=None):
    from oracle.flow_oracle import FlowOracle
    flow = FlowOracle(ref)
    if idx is None:
        idx = idx_tables([u.name for u in flow.units])
    logs = flow.recon_model(cali, idx, SEED, iters=iters_of, batch_size=B)
    evals = {act: flow.evaluate(test_imgs, p=64, act_quant=act) for act in (False, True)}
    return flow, logs, evals, idx


def summary(flow, logs, evals, idx):
    from oracle import rdo_oracle as O
    out = {"units": np.array([u.name for u in flow.units]), "w8": np.array(evals[False]), "w8a8": np.array(evals[True])}
    for u in flow.units:
        out[f"{u.name}/total_first_last"] = np.array(logs[u.name].total)[[0, -1]]
        out[f"{u.name}/idx"] = idx[u.name].astype(np.int16)
        for n, op in u.ops.items():
            a0 = O.adaround_init_alpha(op.weight.clone(), op.delta)
            out[f"{u.name}/moved/{n}"] = np.array(int(((op.alpha >= 0) != (a0 >= 0)).sum()))
            out[f"{u.name}/numel/{n}"] = np.array(op.alpha.numel())
            out[f"{u.name}/bits/{n}"] = np.packbits((op.alpha >= 0).numpy().reshape(-1)[::SAMPLE])
    return out
