import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
def timed(fn, reps=7):
    gr = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for _ in range(10): fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[len(ts) // 2]
g = torch.Generator(device="cpu").manual_seed(0)
names = {0: "base", 1: "-exp", 2: "-QK", 4: "-PV", 6: "-QK-PV", 193: "-exp-sum-cvt", 8: "-lds reads", 48: "-dma-barrier", 56: "-lds-dma-barrier", 199: "-all math", 255: "empty", 249: "MFMA only"}
B, H, N, Dh = 8, 16, 1024, 64
Npad = (N + 7) // 8 * 8
q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
for base, tag in ((1000, "64 q/wave, 2 waves/SIMD"), (2000, "32 q/wave, 4 waves/SIMD")):
    line = f"{tag}:"
    for bits, nm in names.items():
        L.du_set_option(8, (base + bits) if bits else (0 if base == 1000 else 2))
        t = timed(lambda: run(q, k, v, out, B, H, N, Npad, Dh))
        line += f"  {nm} {t:.1f}"
    L.du_set_option(8, 0)
    print(line, flush=True)
