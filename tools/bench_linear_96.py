import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "rdo-ptq_amd"))
from hipops import ops
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for rows, K, N in [(16384, 192, 96), (16384, 96, 192), (16384, 192, 192), (16384, 96, 96)]:
    x = torch.randn(rows, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5; b = torch.randn(N, device="cuda", generator=g)
    pl = ops.split_h2_linear(w); y = torch.empty(rows, N, device="cuda")
    t = timed(lambda: ops.linear_h2(x, pl, b, out=y))
    w4 = w.reshape(N, 1, 1, K).contiguous(); x4 = x.view(1, 1, rows, K); y4 = torch.empty(1, 1, rows, N, device="cuda")
    t2 = timed(lambda: ops.conv2d_fwd(x4, w4, b, 1, 0, out=y4))
    print(f"{rows} x {K} -> {N}: linear_h2 {t:6.1f} us   conv kernel {t2:6.1f} us")
