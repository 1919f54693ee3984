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
variants = [int(x) for x in sys.argv[1].split(",")]
Dh = 64
for N in (1024, 2048, 512):
  for B, H in [(1, 8), (1, 16), (2, 16), (4, 16), (8, 16), (16, 16)]:
    Npad = (N + 7) // 8 * 8
    q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
    k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    line = f"N{N} B{B} H{H} ({B*H*((N+255)//256)} wgs of 256 q):"
    fl = 4.0 * B * H * N * N * Dh
    for var in variants:
        L.du_set_option(8, var)
        t = timed(lambda: run(q, k, v, out, B, H, N, Npad, Dh))
        line += f"  v{var} {t:.1f} us ({fl/t/1e6/2500*100:.1f} %)"
    L.du_set_option(8, 0)
    print(line, flush=True)
