import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
g = torch.Generator(device="cpu").manual_seed(0)
Dh = 64
B, H, N = 1, 8, 1029
Npad = (N + 7) // 8 * 8
q = (torch.randn(B, H, Npad, Dh, generator=g) * 0.3).to(dev, torch.bfloat16)
k = (torch.randn(B, H, Npad, Dh, generator=g) * 0.3).to(dev, torch.bfloat16)
v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
for h in range(H):
    for j, key in enumerate((100 + 37 * h, 700 + 11 * h)):
        qrow = 5 + 64 * h + 300 * j
        k[0, h, key] = (q[0, h, qrow].float() * (300.0 * (j + 1)) / (q[0, h, qrow].float().norm() ** 2)).to(torch.bfloat16)
out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
run(q, k, v, out, B, H, N, Npad, Dh)
s = torch.einsum("bhqd,bhkd->bhqk", q[:, :, :N].float(), k[:, :, :N].float())
o = out.float().reshape(N, H, Dh)
for h in range(H):
    bad = (~torch.isfinite(o[:, h])).any(dim=1).nonzero().flatten().tolist()
    first = s[0, h, :, 1024:N].max(dim=1).values
    k1, k2 = 100 + 37 * h, 700 + 11 * h
    print(f"head {h}: n bad {len(bad)}; first: {bad[:8]}")
    for r in bad[:6]:
        row = s[0, h, r]
        tmax = [float(row[t * 64:(t + 1) * 64].max() - first[r]) for t in range(16)]
        print(f"   row {r}: first-tile max {float(first[r]):.1f}  s[k1]-f {float(row[k1]-first[r]):.1f} s[k2]-f {float(row[k2]-first[r]):.1f}  per-tile max excess {[round(x) for x in tmax]}")
