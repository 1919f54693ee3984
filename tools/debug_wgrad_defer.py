#!/usr/bin/env python
"""Debug aid: the same pinned dinounet train step with the weight-gradient queue on and off; lists every parameter whose gradients differ.
usage: python tools/debug_wgrad_defer.py [model] [size] [batch]"""
import os
import sys
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import ops  # noqa: E402
from dinounet_amd.network_architecture import DinoUNet  # noqa: E402
from dinounet_amd.plans import PLANS_2D  # noqa: E402
from dinounet_amd.training import dc_and_ce_loss  # noqa: E402
from dinounet_amd.dinov3.adapter import DropPath  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "dinounet_l"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.manual_seed(0)
net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name=model, precision="bf16").cuda().train()
for m in net.modules():
    if isinstance(m, DropPath):
        m.drop_prob = 0.0
net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, size, size, generator=g).cuda()
t = torch.randint(0, 2, (B, 1, size, size), generator=g).cuda()


def grads(enabled):
    ops.WGRAD.enabled = enabled
    net.zero_grad(set_to_none=True)
    q0, l0 = ops.WGRAD.queued, ops.WGRAD.launches
    dc_and_ce_loss(net(x), t).backward()
    torch.cuda.synchronize()
    print(f"queue {'on' if enabled else 'off'}: {ops.WGRAD.queued - q0} products queued, {ops.WGRAD.launches - l0} flushes")
    return {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}


g_off = grads(False)
g_on = grads(True)
g_on2 = grads(True)
gmax = max(float(v.norm()) for v in g_off.values())
bad = 0
for k in g_off:
    e = float((g_on[k] - g_off[k]).norm()) / max(float(g_off[k].norm()), 1e-3 * gmax)
    e2 = float((g_on2[k] - g_off[k]).norm()) / max(float(g_off[k].norm()), 1e-3 * gmax)
    if e > 1e-3 or e2 > 1e-3:
        bad += 1
        print(f"  MISMATCH {k:80s} rel {e:.3e} / {e2:.3e}  |off| {float(g_off[k].norm()):.3e} |on| {float(g_on[k].norm()):.3e} shape {tuple(g_off[k].shape)}")
print(f"{bad} of {len(g_off)} parameters differ")
