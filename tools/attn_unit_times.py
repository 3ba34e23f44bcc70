#!/usr/bin/env python3
"""Per unit-iteration wall (us) of the attention-block layer units of BASELINE config 3 inside a short schedule (Cheng2020-attn N=192 W10A10,
16 images, 400 iterations per unit): the first residual unit and the closing 1x1 conv of each of the four attention blocks."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(R, "rdo-ptq_amd"), R, os.path.join(R, "tools")):
    sys.path.insert(0, p)
import full_schedule as FS  # noqa: E402

r = FS.run_schedule(images=16, iters=400, batch=4, quality=False, arch="attn", w_bits=10, a_bits=10, per_unit_log=False, roofline=False, log=lambda *a: None)
print("RDO_UNIT1X1", os.environ.get("RDO_UNIT1X1", "(default)"), "ms/step", round(r["loop_ms_per_step"], 3))
sel = [u for u in r["units"] if ("conv_a.0" in u["unit"] or "conv_b.3" in u["unit"])]
print(" ".join(f"{u['unit'].replace('conv_', '')}={u['loop_ms_per_iter'] * 1e3:.0f}" for u in sel))
att = [u for u in r["units"] if "conv_a" in u["unit"] or "conv_b" in u["unit"]]
print(f"{len(att)} attention-block units: {sum(u['loop_ms_per_iter'] for u in att):.3f} ms per step")
