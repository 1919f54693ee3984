import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dinounet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, S, M, D, L, Lq, P = 8, 1024, 16, 32, 1, 5376, 4
value = torch.randn(N, S, M, D, device=dev).to(torch.bfloat16)
shapes = torch.tensor([[32, 32]], dtype=torch.int64, device=dev)
lsi = torch.zeros(1, dtype=torch.int64, device=dev)
# reference points on a grid + small learned offsets, like the adapter
ref = torch.rand(1, Lq, 1, 1, 1, 2, device=dev)
loc = (ref + 0.05 * torch.randn(N, Lq, M, L, P, 2, device=dev)).contiguous()
attn = torch.softmax(torch.randn(N, Lq, M, L * P, device=dev), -1).view(N, Lq, M, L, P).contiguous()
go = torch.randn(N, Lq, M * D, device=dev).to(torch.bfloat16)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tf = t(lambda: ops.msda_forward_raw(value, shapes, lsi, loc, attn))
tb = t(lambda: ops.msda_backward_raw(value, shapes, lsi, loc, attn, go, True))
print(f"msda fwd {tf:.1f} us, bwd (all) {tb:.1f} us   NO_PLANE={os.environ.get('DU_MSDA_NO_PLANE')}")
# correctness of plane vs q8 path
