#!/usr/bin/env python3
"""The whole calibration schedule on natural-image statistics, product only (no oracle in the loop: the per-unit parity on these statistics is
tests/test_gpu_long_horizon.py / test_gpu_chained_flow.py): Cheng2020-anchor N=192 with trained-like parameters (helpers.trained_like_),
calibration images = the 16 committed Kodak crops and their three flips (64 images), all 29 units through layer_/block_reconstruction
(--arch attn: Cheng2020-attn, W10A10, 105 units -- BASELINE config 3, the attention blocks' 1 x 1 units on the one-launch kernel).
Reports per unit the plane plan, whether a restart fired, the share of soft targets that ended in {0, 1}, and the W8 / W8A8 fidelity.
Not a pytest file (a measurement run; lives under tests/ because the parameters come from the oracle-side helper).

    python tests/run_kodak_schedule.py [--iters 2000]"""
import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "rdo-ptq_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--arch", default="anchor", choices=["anchor", "attn"], help="attn: Cheng2020-attn with W10A10 (BASELINE config 3)")
    a = ap.parse_args()
    import flow_common as F
    import lic
    from full_schedule import run_schedule
    from helpers import kodak_crops
    from test_gpu_chained_flow import _sync_state
    c = kodak_crops(F.GOLDEN)
    if a.arch == "attn":
        from helpers import trained_like_
        from oracle import lic_oracle as L
        torch.manual_seed(F.SEED)
        g = torch.Generator().manual_seed(F.SEED)
        ref = L.Cheng2020Attention(N=192).eval()
        with torch.no_grad():
            trained_like_(ref, g, probe=c[F.N_IMG:F.N_IMG + 4])
        prod = lic.Cheng2020Attention(N=192).eval()
    else:
        ref, _, _ = F.build("kodak")
        prod = lic.Cheng2020Anchor(N=192).eval()
    _sync_state(prod, ref)
    cali = torch.cat([c, c.flip(-1), c.flip(-2), c.flip(-1, -2)])
    bits = 10 if a.arch == "attn" else 8
    r = run_schedule(iters=a.iters, batch=4, quality=True, model=prod, cali=cali, eval_hw=(256, 256), n_eval=2, arch=a.arch, w_bits=bits, a_bits=bits)
    fired = {u["unit"]: u["h2_restarts"] for u in r["units"] if u["h2_restarts"]}
    print(f"kodak schedule: {r['n_units']} units x {a.iters} iterations on {cali.shape[0]} images: {r['recon_model_wall_s']:.1f} s wall, "
          f"{r['loop_ms_per_step']:.3f} ms/step; plane plans on {sum(1 for u in r['units'] if u['h2_plan'])} units, restarts: {fired or 'none'}, "
          f"fp32 fall-backs: {[u['unit'] for u in r['units'] if u['h2_plan'] and not u['use_h2']] or 'none'}")


if __name__ == "__main__":
    main()
