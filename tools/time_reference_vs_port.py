#!/usr/bin/env python3
"""AUTHORING CONTAINER ONLY (imports /root/reference): the cross-timing BASELINE.md section 2 asks for -- iterations per second of the
VERBATIM reference loop (`layer_reconstruction` / `block_reconstruction` of /root/reference/task-oriented-PTQ run on the CPU with the
shims of tools/make_golden.py) against the build's CPU port (`oracle.reconstruct_unit`, the `cpu_baseline` of bench.py) on the same
units, caches and hyper-parameters.  Times only the loop (first to last loss evaluation), not the cache building.

    python tools/time_reference_vs_port.py [--N 48] [--crop 128] [--iters 24]"""
import argparse
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=48)
ap.add_argument("--crop", type=int, default=128)
ap.add_argument("--iters", type=int, default=24)
a = ap.parse_args()

MG._install_shims()
sys.path.insert(0, MG.REF)
from quantization import BaseQuantBlock, QuantModule, block_reconstruction, layer_reconstruction  # noqa: E402  (the REFERENCE's package)
import quantization.block_opt as bo  # noqa: E402
import quantization.layer_opt as lo  # noqa: E402
import quantization.utils as qu  # noqa: E402
from oracle import rdo_oracle as O  # noqa: E402

torch.set_num_threads(os.cpu_count() or 1)
n_img, B = 8, 4
model, qnn = MG._toy_qnn(a.N, 1005)
cali = torch.rand(n_img, 3, a.crop, a.crop, generator=torch.Generator().manual_seed(77))
qnn.set_quant_state(True, False)
with torch.no_grad():
    qnn(cali[:B])
args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Cheng2020")
kwargs = dict(cali_data=cali, batch_size=B, iters=a.iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True, b_range=(20, 2), warmup=0.2,
              act_quant=False, opt_mode="mse", config=None, args=args)
captured, stamps = {}, []
orig_save = qu.save_inp_oup_data


def save_spy(*x, **k):
    r = orig_save(*x, **k)
    captured.update(inp_q=r[0][0].clone(), inp_fp=r[0][1].clone(), out=r[1].clone())
    return r


lo.save_inp_oup_data = bo.save_inp_oup_data = save_spy
for cls in (lo.LossFunction, bo.LossFunction):
    orig = cls.__call__

    def call(self, *x, _o=orig, **k):
        stamps.append(time.perf_counter())
        return _o(self, *x, **k)
    cls.__call__ = call


def oracle_ops(unit, kind):
    def conv(m):
        return O.QOp("conv", m.org_weight.clone(), None if m.org_bias is None else m.org_bias.clone(), stride=m.fwd_kwargs["stride"][0],
                     padding=m.fwd_kwargs["padding"][0], act="lrelu" if type(m.activation_function).__name__ == "LeakyReLU" else None)

    def gdn(m, inv):
        return O.QOp("igdn" if inv else "gdn", m.org_weight.clone(), m.org_bias.clone())
    if kind == "layer":
        return {"layer": conv(unit)}
    if kind == "rb":
        ops = {"conv1": conv(unit.conv1), "conv2": conv(unit.conv2)}
    elif kind == "rbws":
        ops = {"conv1": conv(unit.conv1), "conv2": conv(unit.conv2), "gdn": gdn(unit.gdn, False)}
    else:
        return {"subpel_conv": conv(unit.subpel_conv[0]), "conv": conv(unit.conv), "igdn": gdn(unit.igdn, True), "upsample": conv(unit.upsample[0])}
    if getattr(unit, "skip", None) is not None:
        ops["skip"] = conv(unit.skip)
    for o in ops.values():
        o.act = None                 # inside blocks the activation is the block's own
    return ops


wanted = {"g_a.0": "rbws", "g_a.1": "rb", "g_a.2": "rbws", "g_a.6": "layer", "g_s.1": "rbu", "h_a.0": "layer"}
qnn.set_quant_state(True, False)
qnn.model.g_s[-1][0].set_quant_state(True, False)
rows = []


def recon(mod, prefix=""):
    for name, m in mod.named_children():
        full = prefix + name
        if isinstance(m, (QuantModule, BaseQuantBlock)):
            if full not in wanted:
                for mm in m.modules():
                    if isinstance(mm, (QuantModule, BaseQuantBlock)):
                        mm.trained = True
                continue
            kind = wanted[full]
            ops = oracle_ops(m, kind)                       # before the reference replaces the quantisers
            del stamps[:]
            with MG._cuda_is_cpu():
                (layer_reconstruction if isinstance(m, QuantModule) else block_reconstruction)(qnn, m, name, **kwargs)
            t_ref = (stamps[-1] - stamps[0]) / (len(stamps) - 1)
            ts = []
            O.reconstruct_unit(kind, ops, captured["inp_q"], captured["inp_fp"], captured["out"], iters=a.iters, batch_size=B, input_prob=0.5,
                               weight=0.01, b_range=(20, 2), warmup=0.2, grad_hook=lambda g: ts.append(time.perf_counter()))
            t_port = (ts[-1] - ts[0]) / (len(ts) - 1)
            rows.append((full, kind, t_ref, t_port))
            print(f"{full:8s} {kind:5s} reference {t_ref * 1e3:8.2f} ms/iteration   port {t_port * 1e3:8.2f} ms/iteration   ratio port/ref {t_port / t_ref:5.2f}", flush=True)
        else:
            recon(m, full + ".")


recon(qnn.model)
tr, tp = sum(r[2] for r in rows), sum(r[3] for r in rows)
print(f"sum over {len(rows)} units: reference {tr * 1e3:.1f} ms, port {tp * 1e3:.1f} ms per iteration of each => the port runs at {tr / tp:.2f} x the "
      f"reference loop's speed (Cheng2020-anchor N={a.N}, {a.crop}x{a.crop}, batch {B}, {torch.get_num_threads()} threads)")
