import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dinounet_amd import ops, _lib
import ctypes as C
dev = torch.device("cuda:0")
torch.manual_seed(0)
def check(B, H, W, C1, C2, Cout, bias_on=True, reps=1):
    Cin = C1 + C2
    for rep in range(reps):
        x = torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16)
        x2 = torch.randn(B, H, W, C2, device=dev).to(torch.bfloat16) if C2 else None
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1)
        wp = ops.pack_conv_weight(w, torch.bfloat16)
        bias = torch.randn(Cout, device=dev) if bias_on else None
        y, part = ops.conv3x3_halo(x, wp, bias, x2, want_stats=True)
        xin = torch.cat([x, x2], -1) if C2 else x
        yr = F.conv2d(xin.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, 1, 1).permute(0, 2, 3, 1)
        err = (y.float() - yr).abs().max().item() / yr.abs().max().item()
        sums = torch.empty(B, Cout, 2, device=dev)
        _lib.check(_lib.lib().du_strip_finalize(C.c_void_p(part.data_ptr()), C.c_void_p(sums.data_ptr()), B, part.shape[0] // B, Cout,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fin")
        yf = y.float()
        ref = torch.stack([yf.sum((1, 2)), (yf * yf).sum((1, 2))], -1)
        serr = ((sums - ref).abs().max() / ref.abs().max()).item()
        bad = ~((y.float() - yr).abs() <= 0.05 * yr.abs().max())
        print(f"B{B} {H}x{W} {C1}+{C2}->{Cout} bias {bias_on}: parts {part.shape[0]} max rel err {err:.3e} stats err {serr:.3e} bad {int(bad.sum())}", flush=True)
        if bad.any():
            idx = bad.nonzero()
            print("  first bad", idx[:4].tolist(), "rows", idx[:,1].unique()[:20].tolist(), "cols", idx[:,2].unique()[:40].tolist(), "ch", idx[:,3].unique()[:64].tolist())
for args in [(1, 8, 128, 32, 0, 64), (1, 16, 128, 64, 0, 32), (2, 24, 256, 32, 32, 32), (1, 16, 128, 64, 0, 64), (2, 40, 384, 64, 0, 64, False), (3, 64, 128, 32, 32, 64),
             (8, 512, 512, 32, 0, 64, True, 2), (8, 512, 512, 32, 32, 32, True, 2), (8, 256, 256, 64, 0, 64, True, 3), (8, 256, 256, 64, 0, 64, False, 2), (8, 512, 512, 64, 0, 32, False, 2)]:
    check(*args)
def timeit(B, H, W, C1, C2, Cout, st):
    Cin = C1 + C2
    per = B * H * W * (Cin + Cout) * 2
    ring = int(600e6 // per) + 1
    xs = [torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16) for _ in range(ring)]
    x2s = [torch.randn(B, H, W, C2, device=dev).to(torch.bfloat16) for _ in range(ring)] if C2 else None
    wp = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    for i in range(ring): ops.conv3x3_halo(xs[i], wp, bias, x2s[i] if C2 else None, want_stats=st)
    torch.cuda.synchronize()
    n = 6 * ring
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): ops.conv3x3_halo(xs[i % ring], wp, bias, x2s[i % ring] if C2 else None, want_stats=st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tf = 2.0 * B * H * W * Cin * Cout * 9 / us / 1e6
    print(f"{H}x{W} {C1}+{C2}->{Cout} stats {st}: {us:.1f} us  {per/us/1e3:.0f} GB/s  frac {per/us/1e3/8000:.3f}  {tf:.0f} TF/s (STRIP={os.environ.get('DU_CONV_STRIP','1')} DBG={os.environ.get('DU_STRIP_DEBUG','0')})", flush=True)
for a in [(8, 512, 512, 32, 0, 32, True), (8, 512, 512, 32, 0, 64, False), (8, 512, 512, 32, 32, 32, True), (8, 256, 256, 64, 0, 64, True), (8, 256, 256, 32, 0, 64, True), (8, 256, 256, 64, 0, 64, False)]:
    timeit(*a)
