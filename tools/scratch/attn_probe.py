import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
def timed(fn, reps=5):
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
names = ["dma issue", "K reads + QK", "exp + sums + cvt", "V reads + PV", "vmcnt wait", "barrier"]
for B, H, N, Dh in [(8, 16, 1024, 64), (4, 16, 1024, 64), (2, 16, 1024, 64), (1, 16, 1024, 64), (8, 16, 1029, 64)]:
    Npad = (N + 7) // 8 * 8
    q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
    k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    t = timed(lambda: run(q, k, v, out, B, H, N, Npad, Dh))
    L.du_set_option(4, 64)
    run(q, k, v, out, B, H, N, Npad, Dh); torch.cuda.synchronize()
    tp = timed(lambda: run(q, k, v, out, B, H, N, Npad, Dh))
    buf = (C.c_uint64 * 8)()
    L.du_debug_attn_probe(buf)
    L.du_set_option(4, 0)
    nt = (N + 63) // 64 - 1
    segs = [buf[i] / nt for i in range(6)]
    wgs = B * H * ((N + 255) // 256)
    print(f"B{B} H{H} N{N}: {t:.1f} us ({4.0*B*H*N*N*Dh/t/1e6:.0f} TF/s), {wgs} workgroups = {wgs/512:.2f} per slot; probed build {tp:.1f} us; cycles per step (wave 0 of one workgroup, {nt} steps): " +
          ", ".join(f"{n} {s:.0f}" for n, s in zip(names, segs)) + f"; total {sum(segs):.0f}")
