import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "rdo-ptq_amd"))
from hipops import ops, _lib as L
g = torch.Generator(device="cuda").manual_seed(0)
def check(name, fn, n=12):
    ref = fn().clone()
    bad = 0
    for _ in range(n):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
    print(f"{name:40s} {'bit-identical over %d runs' % n if bad == 0 else 'DIFFERS in %d runs' % bad}", flush=True)
    return bad
tot = 0
for rows, cin, cout in [(65536, 192, 576), (16384, 384, 192), (4096 + 32 * 3, 192, 192)]:
    x = torch.randn(1, 1, rows, cin, device="cuda", generator=g); dy = torch.randn(1, 1, rows, cout, device="cuda", generator=g) * torch.exp(3 * torch.randn(1, 1, rows, 1, device="cuda", generator=g))
    tot += check(f"linear_wgrad_h2 {rows}x{cin}x{cout}", lambda: ops.conv2d_wgrad(x, dy, (cout, 1, 1, cin), 1, 0))
for rows, K, N in [(65536, 192, 576), (65536 + 192, 192, 192), (32768, 192, 384), (16384, 576, 192)]:
    x = torch.randn(rows, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    pl = ops.split_h2_linear(w); b = torch.randn(N, device="cuda", generator=g)
    tot += check(f"linear_h2 {rows}x{K}->{N}", lambda: ops.linear_h2(x, pl, b))
for B, H, C, heads, shift in [(4, 128, 192, 4, 4), (4, 64, 192, 8, 0), (2, 32, 192, 8, 4), (4, 16, 192, 16, 4)]:
    d = ops.attn_desc(B, H, H, C, heads, 8, shift)
    qkv = torch.randn(B, H, H, 3 * C, device="cuda", generator=g); bias = torch.randn(heads, 64, 64, device="cuda", generator=g); dout = torch.randn(B, H, H, C, device="cuda", generator=g)
    tot += check(f"attention fwd {H}^2 x{heads}", lambda: ops.window_attention(d, qkv, bias))
    tot += check(f"attention bwd {H}^2 x{heads}", lambda: ops.window_attention_bwd(d, qkv, bias, dout))
print("TOTAL differing:", tot)
