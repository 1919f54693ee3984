import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
L.du_set_option(8, int(os.environ.get("ATT_VAR", "0")))
def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
Dh = 64; B, H, N = 1, 1, 1029
Npad = (N + 7) // 8 * 8
def case(name, spikes, scale=0.02):
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
    k = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    # orthogonal spike directions: query row r gets direction e_{r % 64}
    for (qrow, key, val) in spikes:
        q[0, 0, qrow] = 0; q[0, 0, qrow, qrow % 64] = 1.0
    for (qrow, key, val) in spikes:
        k[0, 0, key] = 0; k[0, 0, key, qrow % 64] = val
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    run(q, k, v, out, B, H, N, Npad, Dh)
    s = torch.einsum("bhqd,bhkd->bhqk", q[:, :, :N].float(), k[:, :, :N].float())
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s * math.log(2.0), -1), v[:, :, :N].float()).permute(0, 2, 1, 3).reshape(B * N, H * Dh)
    bad = (~torch.isfinite(out.float())).any(dim=1).nonzero().flatten().tolist()
    fin = torch.isfinite(out.float()).all(dim=1)
    err = float((out.float() - ref)[fin].abs().max() / ref.abs().max())
    print(f"{name}: bad rows {bad[:12]} n={len(bad)} err(finite) {err:.1e}")
case("one spike 300 tile1", [(5, 100, 300.0)])
case("two rows same block, tiles 1 and 10", [(5, 100, 300.0), (9, 700, 300.0)])
case("two rows other blocks (qb0/qb1)", [(5, 100, 300.0), (40, 700, 300.0)])
case("two rows other waves", [(5, 100, 300.0), (70, 700, 300.0)])
case("two rows same block same tile", [(5, 100, 300.0), (9, 101, 300.0)])
case("same row twice: 100 then 300", [(5, 100, 100.0)] )
case("spike 100 (finite p, over thresh)", [(5, 100, 100.0), (9, 700, 100.0)])
case("spike 140 x2", [(5, 100, 140.0), (9, 700, 140.0)])
case("spike 300 in tile 0", [(5, 10, 300.0), (9, 700, 300.0)])
case("half 1 key (key 104: r&4)", [(5, 104, 300.0), (9, 708, 300.0)])
def dump(name, spikes, scale=0.02):
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
    k = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
    v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
    for (qrow, key, val) in spikes:
        q[0, 0, qrow] = 0; q[0, 0, qrow, qrow % 64] = 1.0
    for (qrow, key, val) in spikes:
        k[0, 0, key] = 0; k[0, 0, key, qrow % 64] = val
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
    for th in (60, 20, -2000):
        L.du_set_option(7, th)
        run(q, k, v, out, B, H, N, Npad, Dh)
        r = spikes[0][0]
        print(name, "thresh", th, "row", r, out[r].float().tolist()[:16], "v[key]", v[0, 0, spikes[0][1]].float().tolist()[:8])
    L.du_set_option(7, 60)
dump("half1 spike", [(5, 100, 300.0)])
dump("half0 spike", [(5, 104, 300.0)])
