#!/usr/bin/env python3
"""tests/golden/flow_n192.npz: the ORACLE's chained calibration flow of Cheng2020-anchor N=192 (main2.py:214-282 restated:
oracle/flow_oracle.py) for the seeded input sets of tests/flow_common.py.  A constant of the seeds; computing it in the GPU suite cost
~80 s per flow of the driver's time limit.  Runs anywhere the repository does.

    python tools/make_flow_golden.py [--only kodak]        (about 5-10 minutes per flow on 8 cores)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import flow_common as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(F.GOLDEN, "flow_n192.npz"))
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    out = dict(np.load(a.out)) if os.path.exists(a.out) else {}
    for stats in ("uniform", "kodak"):
        if a.only and a.only != stats:
            continue
        t0 = time.time()
        ref, cali, test_imgs = F.build(stats)
        flow, logs, evals, idx = F.oracle_flow(ref, cali, test_imgs)
        out = {k: v for k, v in out.items() if not k.startswith(stats + "/")}
        for k, v in F.summary(flow, logs, evals, idx).items():
            out[f"{stats}/{k}"] = v
        moved = sum(int(out[k]) for k in out if k.startswith(stats + "/") and "/moved/" in k)
        total = sum(int(out[k]) for k in out if k.startswith(stats + "/") and "/numel/" in k)
        print(f"{stats}: {len(flow.units)} units in {time.time() - t0:.0f} s; decisions moved against nearest rounding {moved} of {total}; "
              f"W8 (PSNR, bpp) {evals[False]}, W8A8 {evals[True]}", flush=True)
        np.savez_compressed(a.out, **out)
    print(f"{a.out}: {os.path.getsize(a.out) / 1e6:.2f} MB")


if __name__ == "__main__":
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    main()
