#!/usr/bin/env python
"""One shape of du_attention_fwd, a few launches: the target of rocprofv3 --pmc passes (tools/pmc_summary.py reads the CSV)."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import _lib  # noqa: E402

B, H, N, Dh = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (8, 16, 1029, 64))]
dev = torch.device("cuda", 0)
L = _lib.lib()
Npad = (N + 127) // 128 * 128
g = torch.Generator(device="cpu").manual_seed(0)
q = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(dev, torch.bfloat16)
k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
for _ in range(6):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()),
                                  B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "du_attention_fwd")
torch.cuda.synchronize()
print("done")
