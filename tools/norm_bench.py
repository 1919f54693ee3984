#!/usr/bin/env python
"""Bandwidth of the InstanceNorm + LeakyReLU kernels (csrc/norm.hip) on the decoder's NHWC bf16 tensors of a dinounet_l 512^2 batch-8
step: statistics, normalise + activate, backward statistics, backward dx.  Algorithmic bytes = every tensor read / written once.
usage: python tools/norm_bench.py [rounds]      (DU_NORM_SLOTS / DU_STRIP_TARGET: launch-geometry tuning aids of norm.hip)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import ops  # noqa: E402
from dinounet_amd._lib import ACT_LEAKY  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def timeit(fn, rounds):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    return sorted(ts)[len(ts) // 2]


def main(rounds):
    print(f"# DU_NORM_SLOTS={os.environ.get('DU_NORM_SLOTS', '-')} DU_STRIP_TARGET={os.environ.get('DU_STRIP_TARGET', '-')}")
    print(f"{'tensor':>22} {'MB':>6} | {'stats us':>9} {'TB/s':>5} | {'fwd us':>8} {'TB/s':>5} | {'fwd+bwd us':>10} {'TB/s':>5}")
    for B, H, W, C in [(8, 512, 512, 32), (8, 256, 256, 64), (8, 128, 128, 128), (8, 64, 64, 256), (8, 32, 32, 256)]:
        x = torch.randn(B, H, W, C, device=dev).to(bf)
        w = torch.ones(C, device=dev, requires_grad=True)
        b = torch.zeros(C, device=dev, requires_grad=True)
        mb = x.numel() * 2 / 1e6
        t_stats = timeit(lambda: ops.chan_stats(x, B), rounds)
        t_fwd = timeit(lambda: ops.norm_act(x, w, b, "in", act=ACT_LEAKY), rounds)       # stats + finalize + apply
        go = torch.randn_like(x)
        xr = x.clone().requires_grad_(True)

        def fb():
            y = ops.norm_act(xr, w, b, "in", act=ACT_LEAKY)
            torch.autograd.grad(y, (xr, w, b), go)
        t_fb = timeit(fb, rounds)
        # bytes: stats 1 read; fwd = stats + read + write (3); bwd = stats(2 reads) + dx (2 reads + 1 write) = 5 -> fwd+bwd 8
        print(f"{str((B, H, W, C)):>22} {mb:6.1f} | {t_stats:9.1f} {mb / t_stats:5.2f} | {t_fwd:8.1f} {3 * mb / t_fwd:5.2f} | {t_fb:10.1f} {8 * mb / t_fb:5.2f}",
              flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
