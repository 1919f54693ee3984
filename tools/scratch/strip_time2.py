import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dinounet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(B, H, W, C1, C2, Cout, st, bias_on=True):
    Cin = C1 + C2
    per = B * H * W * (Cin + Cout) * 2
    ring = int(600e6 // per) + 1
    xs = [torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16) for _ in range(ring)]
    x2s = [torch.randn(B, H, W, C2, device=dev).to(torch.bfloat16) for _ in range(ring)] if C2 else None
    wp = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev) if bias_on else None
    for i in range(ring): ops.conv3x3_halo(xs[i], wp, bias, x2s[i] if C2 else None, want_stats=st)
    torch.cuda.synchronize()
    n = 6 * ring
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): ops.conv3x3_halo(xs[i % ring], wp, bias, x2s[i % ring] if C2 else None, want_stats=st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tf = 2.0 * B * H * W * Cin * Cout * 9 / us / 1e6
    print(f"{H}x{W} {C1}+{C2}->{Cout} stats {st} bias {bias_on}: {us:.1f} us  {per/us/1e3:.0f} GB/s  frac {per/us/1e3/8000:.3f}  {tf:.0f} TF/s (DBG={os.environ.get('DU_STRIP_DEBUG','0')})", flush=True)
for a in [(8, 256, 256, 64, 0, 64, True), (8, 256, 256, 64, 0, 64, False, False), (8, 512, 512, 32, 32, 32, True), (8, 512, 512, 32, 0, 64, False, False)]:
    timeit(*a)
