"""qkv product + RoPE/head split: persistent kernel with the rotation in its drain vs plain persistent kernel + du_qkv_rope_split (ViT-L shape)"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import ops, _lib
dev = torch.device("cuda", 0); bf = torch.bfloat16
B, H, hp, wp, D, Dh, prefix = 8, 16, 32, 32, 1024, 64, 5
N = prefix + hp * wp
g = torch.Generator().manual_seed(0)
h = torch.randn(B * N, D, generator=g).to(dev, bf); w = (torch.randn(3 * H * Dh, D, generator=g) * D ** -0.5).to(dev, bf)
bias = torch.randn(3 * H * Dh, generator=g).to(dev)
periods = 100.0 ** (2 * torch.arange(Dh // 4, dtype=torch.float32) / (Dh // 2))
ch = (torch.arange(0.5, hp) / hp * 2 - 1); cw = (torch.arange(0.5, wp) / wp * 2 - 1)
coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), -1).flatten(0, 1)
ang = (2 * math.pi * coords[:, :, None] / periods[None, None, :]).flatten(1, 2).tile(2)
sin, cos = torch.sin(ang).to(dev).contiguous(), torch.cos(ang).to(dev).contiguous()
ws = {}
def heads(): return ops.qkv_attention(h, w, bias, sin, cos, B, N, H, Dh, prefix, ws)
def drain():
    ops._QKV_ROPE_DRAIN = True
    try:
        return ops.qkv_attention(h, w, bias, sin, cos, B, N, H, Dh, prefix, ws, grid=(hp, wp))
    finally:
        ops._QKV_ROPE_DRAIN = False
def unfused(): return ops.attention(ops.mm(h, w, bias=bias), sin, cos, B, N, H, Dh, prefix, ws)
res = {}
for name, fn in (("heads+inplace", heads), ("rope in drain", drain), ("unfused", unfused)):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): fn()
    res[name] = gr
for _ in range(3):
    for name, gr in res.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        print(name, f"{e0.elapsed_time(e1) * 100:.1f} us per (qkv + rope + attention)")
