#!/usr/bin/env python3
"""Full-size Lu2022 (embed 192, latent 320, 256x256 crops, batch 4) calibration iterations on the tape engine: per-unit ms per
iteration (difference of two runs with different iteration counts, so cache building and plan recording cancel)."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rdo-ptq_amd"))
import lic  # noqa: E402
from quantization import BaseQuantBlock, QuantModel, QuantModule, block_reconstruction, layer_reconstruction  # noqa: E402

cfg = dict(height=256, width=256, in_chans=3, embed_dim=192, latent_dim=320, window_size=8, mlp_ratio=2.0, qkv_bias=True,
           qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_checkpoint=False)
units = sys.argv[1:] or ["g_a0", "g_a1", "g_a7", "g_s0", "g_s6"]
ITERS = tuple(int(v) for v in os.environ.get("LU_ITERS", "4,24").split(","))
torch.manual_seed(0)
model = lic.NIC(cfg).cuda().eval()
wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
aq = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=aq).cuda().eval()
qnn.set_first_last_layer_to_8bit()
qnn.disable_network_output_quantization()
B, n_img = 4, 8
cali = torch.rand(n_img, 3, 256, 256, device="cuda")
qnn.set_quant_state(True, False)
with torch.no_grad():
    qnn(cali[:B])
args = types.SimpleNamespace(lmbda=0.0483, task_loss=2.0, arch="Lu2022", loss_mode=os.environ.get("LU_LOSS_MODE", "lp"))   # LU_LOSS_MODE=rd: R + lambda*D task term
for name in units:
    unit = getattr(qnn.model, name)
    fn = layer_reconstruction if isinstance(unit, QuantModule) else block_reconstruction
    ts = []
    order = [n for n, m in qnn.model.named_children() if isinstance(m, (QuantModule, BaseQuantBlock))]
    for n in order:                                   # units are calibrated in order: everything before `name` counts as trained
        for m in getattr(qnn.model, n).modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = order.index(n) < order.index(name)
    for iters in (ITERS[0],) + ITERS:                 # (a throw-away call first: one-time costs must not sit in the first timed call only)
        for m in unit.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.trained = False
        for m in unit.modules():
            if isinstance(m, QuantModule) and hasattr(m.weight_quantizer, "alpha"):
                from quantization.quantizer import UniformAffineQuantizer
                u = UniformAffineQuantizer(**wq, tconv=m.if_tconv)
                u.delta, u.zero_point, u.inited = m.weight_quantizer.delta, m.weight_quantizer.zero_point, True
                m.weight_quantizer = u
        torch.cuda.synchronize()
        t0 = time.time()
        fn(qnn, unit, name, cali_data=cali, batch_size=B, iters=iters, weight=0.01, input_prob=0.5, lr=4e-5, asym=True,
           b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode="mse", config=None, args=args)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    print(f"{name}: {(ts[2] - ts[1]) / (ITERS[1] - ITERS[0]) * 1e3:8.2f} ms/iteration   (setup+{ITERS[0]} iters {ts[1]:.2f} s)   peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
