import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
g = torch.Generator(device="cpu").manual_seed(0)
cfgs = [("r3", 1, 0), ("w64 v0", 0, 0), ("w64 v7", 0, 7), ("w64 v2 (32q, 4 waves)", 0, 2), ("pipe v9", 0, 9)] + [(f"pipe {nm}", 0, 3000 + b) for b, nm in
        ((1, "-exp"), (64, "-sums"), (192, "-sums-cvt"), (193, "-exp-sums-cvt"), (8, "-lds reads"), (48, "-dma-barrier"), (56, "-lds-dma-barrier"), (57, "-exp-lds-dma-barrier"), (249, "MFMA only"))]
Dh = 64
for N, B, H in ((1024, 8, 16), (1024, 4, 16), (1029, 8, 16)):
    Npad = (N + 7) // 8 * 8
    q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
    k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    graphs = []
    for nm, impl, var in cfgs:
        L.du_set_option(6, impl); L.du_set_option(8, var)
        run(q, k, v, out, B, H, N, Npad, Dh)
        gr = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(20): run(q, k, v, out, B, H, N, Npad, Dh)
        graphs.append(gr)
    L.du_set_option(6, 0); L.du_set_option(8, 0)
    # warm the clocks, then interleave the variants
    t0 = time.time()
    while time.time() - t0 < 1.0:
        graphs[0].replay()
    torch.cuda.synchronize()
    res = [[] for _ in cfgs]
    for rnd in range(6):
        for j, gr in enumerate(graphs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[j].append(e0.elapsed_time(e1) / 40 * 1e3)
    print(f"N{N} B{B} H{H}: " + "  ".join(f"{nm} {sorted(r)[len(r)//2]:.1f} (min {min(r):.1f})" for (nm, _, _), r in zip(cfgs, res)), flush=True)
