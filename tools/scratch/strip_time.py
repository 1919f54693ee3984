import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dinounet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for Cout in (32, 64):
    B, H, W = 8, 512, 512
    per = B * H * W * (32 + Cout) * 2
    ring = int(600e6 // per) + 1
    xs = [torch.randn(B, H, W, 32, device=dev).to(torch.bfloat16) for _ in range(ring)]
    wp = (torch.randn(Cout, 288, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    for st in (True,):
        for i in range(ring): ops.conv3x3_halo(xs[i], wp, bias, None, want_stats=st)
        torch.cuda.synchronize()
        n = 6 * ring
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): ops.conv3x3_halo(xs[i % ring], wp, bias, None, want_stats=st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"512x512 32->{Cout} stats {st}: {us:.1f} us  {per/us/1e3:.0f} GB/s  frac {per/us/1e3/8000:.3f}  (DU_STRIP_DEBUG={os.environ.get('DU_STRIP_DEBUG','0')})", flush=True)
