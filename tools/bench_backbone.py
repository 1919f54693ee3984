#!/usr/bin/env python
"""BASELINE.json configs[4]: frozen DINOv3 backbone forward only (default dinounet_7b, 1024x1024 patches, bf16), one replica per GPU.
Reports tokens/s, ms per forward and the achieved TF/s of the GEMM and attention kernels (HIP events around every launch).
usage: python tools/bench_backbone.py [--model dinounet_7b] [--size 1024] [--batch 2] [--steps 3]"""
import argparse
import json
import os
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE", "1")
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinounet_7b")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    from dinounet_amd import ops
    from dinounet_amd.dinov3 import build_backbone
    from dinounet_amd.network_architecture.dinounet import DINOv3_INTERACTION_INDEXES
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    with torch.device(dev):                      # initialise the 6.7 B parameters on the GPU, not on the host
        vit = build_backbone(a.model)
    vit = vit.eval()
    for blk in vit.blocks:                       # non-trivial LayerScale so the residual branches matter
        blk.ls1.gamma.data.fill_(0.1); blk.ls2.gamma.data.fill_(0.1)
    nparam = sum(p.numel() for p in vit.parameters())
    print(f"[backbone] {a.model}: {nparam / 1e9:.2f} B params built in {time.perf_counter() - t0:.1f} s", flush=True)
    x = torch.randn(a.batch, 3, a.size, a.size, device=dev)
    idx = DINOv3_INTERACTION_INDEXES[a.model]
    outs = vit.get_intermediate_layers(x, n=idx, return_class_token=True, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert all(torch.isfinite(o[0].float()).all() for o in outs)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs = vit.get_intermediate_layers(x, n=idx, return_class_token=True, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    ntok = a.batch * (1 + vit.n_storage_tokens + (a.size // 16) ** 2)
    ops.PROFILE = ops.KernelProfile()
    vit.get_intermediate_layers(x, n=idx, return_class_token=True, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for name, e0, e1, fl, nb in prof.rec:
        v = agg.setdefault(name, [0.0, 0, 0.0])
        v[0] += e0.elapsed_time(e1) * 1e-3; v[1] += 1; v[2] += fl
    out = {"workload": f"{a.model} frozen backbone forward, {a.size}x{a.size}, batch {a.batch}, bf16, random-init weights", "tokens_per_s": round(ntok / dt, 1),
           "ms_per_forward": round(dt * 1e3, 2), "tokens": ntok,
           "kernels": {k: {"launches": v[1], "ms": round(v[0] * 1e3, 2), "tflops": round(v[2] / v[0] / 1e12, 1)} for k, v in agg.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
