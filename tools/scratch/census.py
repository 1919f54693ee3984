import ctypes as C, math, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
g = torch.Generator(device="cpu").manual_seed(0)
for B, H, N, Dh in [(8, 16, 1024, 64)]:
    Npad = (N + 7) // 8 * 8
    q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
    k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    L.du_set_option(4, 64)
    for _ in range(3):
        run(q, k, v, out, B, H, N, Npad, Dh)
    torch.cuda.synchronize()
    nwg = B * H * ((N + 255) // 256)
    buf = (C.c_uint64 * (8 * nwg))()
    L.du_debug_attn_census(buf, nwg)
    L.du_set_option(4, 0)
    rec = [tuple(buf[8 * i + j] for j in range(6)) for i in range(nwg)]
    byx = collections.defaultdict(list)
    for i, (lb, le, hw, xcc, te, tx) in enumerate(rec):
        byx[xcc & 0xf].append((te, lb, le, tx, i, (hw >> 8) & 0xf, (hw >> 13) & 7, hw & 0xf))
    for x in sorted(byx)[:2]:
        lst = byx[x]
        t0 = min(r[0] for r in lst)
        ent = sorted(r[0] - t0 for r in lst); lbs = sorted(r[1] - t0 for r in lst); les = sorted(r[2] - t0 for r in lst); txs = sorted(r[3] - t0 for r in lst)
        def q3(a): return f"min {a[0]} med {a[len(a)//2]} max {a[-1]}"
        print(f"XCC {x}: {len(lst)} workgroups; ticks from the first entry: entry {q3(ent)}; loop begin {q3(lbs)}; loop end {q3(les)}; exit {q3(txs)}")
        pro = sorted(r[1] - r[0] for r in lst); loop = sorted(r[2] - r[1] for r in lst); epi = sorted(r[3] - r[2] for r in lst)
        print(f"     per workgroup: prologue {q3(pro)}; loop {q3(loop)}; epilogue {q3(epi)}")
