#!/usr/bin/env python
"""Times the GPU-side 2D training augmentation chain (dinounet_amd.augment.GPUAugment2D) on a batch of the benchmark
shape already resident in HBM: draw() (host-side parameter draws) + apply() (du_aug_* kernels), against the train step it feeds.
usage: python tools/bench_augment.py [--batch 8] [--size 512] [--reps 20]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from dinounet_amd.augment import GPUAugment2D
    dev = torch.device("cuda", 0)
    aug = GPUAugment2D((a.size, a.size), seed=1)
    g = torch.Generator().manual_seed(0)
    data = torch.randn(a.batch, 3, a.size, a.size, generator=g).to(dev)
    seg = torch.randint(0, 2, (a.batch, 1, a.size, a.size), generator=g).to(dev).float()
    for _ in range(3):
        aug.apply(data, seg, aug.draw(a.batch, 3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        d = aug.draw(a.batch, 3)
    t_draw = (time.perf_counter() - t0) / a.reps
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = aug.apply(data, seg, aug.draw(a.batch, 3))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"metric": "GPU 2D training augmentation (spatial+mirror, noise, blur, brightness, contrast, low-res, gamma x2)",
                      "batch": a.batch, "size": a.size, "ms_per_batch": round(dt * 1e3, 3), "ms_host_draw": round(t_draw * 1e3, 3),
                      "slices_per_s": round(a.batch / dt, 1), "out_shapes": [list(t.shape) for t in out]}))


if __name__ == "__main__":
    main()
