import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
impl, var, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, H, Dh = 8, 16, 64
dev = torch.device("cuda", 0)
L = _lib.lib()
L.du_set_option(6, impl); L.du_set_option(8, var)
Npad = (N + 7) // 8 * 8
g = torch.Generator(device="cpu").manual_seed(0)
q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
for _ in range(4):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
torch.cuda.synchronize()
