import os, sys, torch
sys.path.insert(0, "/root/repo/rdo-ptq_amd")
from hipops import ops
def timeit(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B,H,Cin,Cout) in [(4,128,192,192),(4,64,192,192)]:
    torch.manual_seed(1)
    x = torch.randn(B,H,H,Cin,device="cuda"); w = torch.randn(Cout,3,3,Cin,device="cuda")/(Cin*9)**0.5
    wpl, xp = ops.split_h2_conv(w), ops.split_h2(x)
    opl = ops.h2_empty((B,H,H,Cout),"cuda",16.0)
    r = {1:[],5:[],6:[]}
    for _ in range(5):
        for v in (1,5,6):
            ops.set_tuning("h2_stagger", v)
            r[v].append(timeit(lambda: ops.conv2d_fwd_h2(xp, tuple(x.shape), tuple(w.shape), wpl, None, 1, 1, out_planes=opl)))
    ops.set_tuning("h2_stagger", 1)
    print(B,H,Cin,Cout, {v: round(sorted(l)[len(l)//2],1) for v,l in r.items()})
