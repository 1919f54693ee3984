import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
g = torch.Generator(device="cpu").manual_seed(0)
cfgs = [("r3", 1, 0), ("w64", 0, 0), ("32q tail-first", 0, 2), ("32q tail-split", 0, 3)]
for N, B, H, Dh in ((1029, 8, 16, 64), (1024, 8, 16, 64), (1029, 8, 12, 64), (1029, 16, 6, 64), (261, 2, 16, 64), (1029, 4, 16, 64), (1100, 8, 16, 64)):
    Npad = (N + 7) // 8 * 8
    q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
    k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    s = torch.einsum("bhqd,bhkd->bhqk", q[:, :, :N].float(), k[:, :, :N].float()) * math.log(2.0)
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v[:, :, :N].float()).permute(0, 2, 1, 3).reshape(B * N, H * Dh)
    del s
    graphs, errs = [], []
    for nm, impl, var in cfgs:
        L.du_set_option(6, impl); L.du_set_option(8, var)
        out.zero_(); run(q, k, v, out, B, H, N, Npad, Dh)
        errs.append(float((out.float() - ref).abs().max() / ref.abs().max()))
        gr = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(20): run(q, k, v, out, B, H, N, Npad, Dh)
        graphs.append(gr)
    L.du_set_option(6, 0); L.du_set_option(8, 0)
    t0 = time.time()
    while time.time() - t0 < 0.5: graphs[0].replay()
    torch.cuda.synchronize()
    res = [[] for _ in cfgs]
    for rnd in range(5):
        for j, gr in enumerate(graphs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[j].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * B * H * N * N * Dh
    print(f"N{N} B{B} H{H} Dh{Dh}: " + "  ".join(f"{nm} {sorted(r)[len(r)//2]:.1f} us ({fl/sorted(r)[len(r)//2]/1e6/2500*100:.1f} %, err {e:.0e})" for (nm, _, _), r, e in zip(cfgs, res, errs)), flush=True)
