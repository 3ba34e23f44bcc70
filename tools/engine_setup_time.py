import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/rdo-ptq_amd")
import bench
from quantization import QuantModel
from quantization.engine import UnitEngine
from quantization.recon import _unit_modules
dev = torch.device("cuda", 0)
model = bench.seeded_model(192, 1005, dev)
wq = {"n_bits": 8, "channel_wise": True, "scale_method": "max"}
qnn = QuantModel(model=model, weight_quant_params=wq, act_quant_params=dict(wq, leaf_param=False), is_cheng=True).to(dev).eval()
cali = torch.rand(8, 3, 256, 256).to(dev)
units = [(n, u) for n, u in bench.unit_list(qnn) if n in ("h_a.0", "g_a.1")]
caches = bench.build_caches(qnn, units, cali, bs=8)
for name, u in units:
    kind, mods = _unit_modules(u)
    cq, cf, co = caches[name]
    for trial in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        idx = torch.stack([torch.randperm(8)[:4] for _ in range(20000)])
        t1 = time.time()
        e = UnitEngine(kind, mods, cq, cf, co, batch_size=4, iters=20000, idx_table=idx, seed=1)
        torch.cuda.synchronize(); t2 = time.time()
        e.run(64); torch.cuda.synchronize(); t3 = time.time()
        e.run(64); torch.cuda.synchronize(); t4 = time.time()
        print(f"{name} trial {trial}: idx table {t1-t0:.3f} s | engine init (probe+record) {t2-t1:.3f} s | first run(64) incl. graph capture {t3-t2:.3f} s | second run(64) {t4-t3:.3f} s")
import cProfile, pstats
name, u = [x for x in units if x[0] == "h_a.0"][0]
kind, mods = _unit_modules(u)
cq, cf, co = caches[name]
idx = torch.stack([torch.randperm(8)[:4] for _ in range(20000)])
pr = cProfile.Profile(); pr.enable()
e = UnitEngine(kind, mods, cq, cf, co, batch_size=4, iters=20000, idx_table=idx, seed=1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
