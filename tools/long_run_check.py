"""One layer unit (3x3 conv 192 -> 192 at 16x16 + LeakyReLU) for the reference's full 20 000 iterations in one graph-replay call:
timing per iteration, loss trajectory sanity and convergence of the soft rounding targets to {0, 1}."""
import sys, os, time, torch, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R + '/rdo-ptq_amd')
import torch.nn as nn
from quantization.quant_layer import QuantModule
from quantization.engine import UnitEngine
WQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
AQ = {"n_bits": 8, "channel_wise": True, "scale_method": "max", "leaf_param": False}
torch.manual_seed(0)
conv = nn.Conv2d(192, 192, 3, padding=1).cuda()
qm = QuantModule(conv, WQ, AQ).cuda()
qm.activation_function = nn.LeakyReLU(inplace=True)
n, B, iters = 64, 4, 20000
cq = torch.randn(n, 16, 16, 192, device="cuda"); cf = cq + 0.01 * torch.randn_like(cq)
with torch.no_grad():
    co = qm(cf.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
eng = UnitEngine("layer", {"layer": qm}, cq, cf, co, batch_size=B, iters=iters, seed=1)
torch.cuda.synchronize(); t0 = time.time()
eng.run()
torch.cuda.synchronize(); dt = time.time() - t0
total, rt, rd = eng.logs()
print(f"{iters} iterations in {dt:.2f} s = {dt / iters * 1e6:.1f} us/iteration; loss first {float(total[0]):.4e} at-warmup-end {float(total[3999]):.4e} last {float(total[-1]):.4e}; round last {float(rd[-1]):.4e}")
a = eng.alpha_of("layer")
h = torch.clamp(torch.sigmoid(a) * 1.2 - 0.1, 0, 1)
print("soft targets converged to {0,1}:", float(((h < 1e-3) | (h > 1 - 1e-3)).float().mean()))
