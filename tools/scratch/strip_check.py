import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dinounet_amd import ops, _lib
import ctypes as C
dev = torch.device("cuda:0")
torch.manual_seed(0)
def check(B, H, W, Cout, bias_on=True):
    x = torch.randn(B, H, W, 32, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 32, 3, 3, device=dev) * 0.1)
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    bias = torch.randn(Cout, device=dev) if bias_on else None
    r = ops.conv3x3_halo(x, wp, bias, None, want_stats=True)
    y, part = r
    yr = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, 1, 1).permute(0, 2, 3, 1)
    err = (y.float() - yr).abs().max().item() / yr.abs().max().item()
    sums = torch.empty(B, Cout, 2, device=dev)
    _lib.check(_lib.lib().du_strip_finalize(C.c_void_p(part.data_ptr()), C.c_void_p(sums.data_ptr()), B, part.shape[0] // B, Cout,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fin")
    yf = y.float()
    ref = torch.stack([yf.sum((1, 2)), (yf * yf).sum((1, 2))], -1)
    serr = ((sums - ref).abs().max() / ref.abs().max()).item()
    bad = (y.float() - yr).abs() > 0.05 * yr.abs().max()
    print(f"B{B} {H}x{W} 32->{Cout} bias {bias_on}: parts {part.shape[0]} max rel err {err:.3e} stats err {serr:.3e} bad {int(bad.sum())}", flush=True)
    if bad.any():
        idx = bad.nonzero()
        print("  first bad", idx[:8].tolist(), "rows", idx[:,1].unique()[:20].tolist(), "cols", idx[:,2].unique()[:40].tolist())
for args in [(1, 8, 128, 32), (2, 24, 128, 32), (1, 16, 256, 64), (2, 40, 384, 32, False), (8, 512, 512, 32), (8, 512, 512, 64, False),
             (8, 512, 512, 32, False), (8, 512, 512, 64), (8, 512, 512, 64, False), (8, 512, 512, 32), (3, 256, 256, 64), (8, 512, 512, 64, False)]:
    check(*args)
# timing (ring of buffers > infinity cache)
for Cout in (32, 64):
    B, H, W = 8, 512, 512
    per = B * H * W * (32 + Cout) * 2
    ring = int(600e6 // per) + 1
    xs = [torch.randn(B, H, W, 32, device=dev).to(torch.bfloat16) for _ in range(ring)]
    wp = (torch.randn(Cout, 288, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    for st in (True, False):
        for i in range(ring): ops.conv3x3_halo(xs[i], wp, bias, None, want_stats=st)
        torch.cuda.synchronize()
        n = 6 * ring
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): ops.conv3x3_halo(xs[i % ring], wp, bias, None, want_stats=st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"512x512 32->{Cout} stats {st}: {us:.1f} us  {per/us/1e3:.0f} GB/s  frac {per/us/1e3/8000:.3f}  (DU_CONV_STRIP={os.environ.get('DU_CONV_STRIP','1')})", flush=True)
