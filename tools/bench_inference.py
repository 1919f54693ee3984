#!/usr/bin/env python
"""Times sliding-window inference (dinounet_amd.inference.predict_sliding_window_logits) of the benchmark
network on a synthetic volume: 512 x 512 windows at step 0.5 with Gaussian blending, windows batched through the eval-mode forward.
usage: python tools/bench_inference.py [--model dinounet_l] [--slices 4] [--size 1024] [--batch 12] [--reps 3]"""
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
    ap.add_argument("--model", default="dinounet_l")
    ap.add_argument("--slices", type=int, default=4)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture of the window forward")
    ap.add_argument("--mirror", action="store_true", help="test-time mirroring over both in-plane axes (4 forwards per window batch)")
    a = ap.parse_args()
    from dinounet_amd.plans import PLANS_2D
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd import inference as INF
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name=a.model, precision="bf16").to(dev).eval()
    data = torch.randn(3, a.slices, a.size, a.size, generator=torch.Generator().manual_seed(1)).to(dev)
    nwin = len(INF.sliding_window_origins((a.slices, a.size, a.size), (512, 512), 0.5))
    mirror = (0, 1) if a.mirror else None
    INF.predict_sliding_window_logits(net, data, (512, 512), 0.5, True, a.batch, graph=not a.eager, mirror_axes=mirror)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = INF.predict_sliding_window_logits(net, data, (512, 512), 0.5, True, a.batch, graph=not a.eager, mirror_axes=mirror)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"metric": "sliding-window inference, 512x512 windows, step 0.5, gaussian", "model": a.model, "volume": [a.slices, a.size, a.size],
                      "windows": nwin, "window_batch": a.batch, "s_per_volume": round(dt, 4), "windows_per_s": round(nwin / dt, 1),
                      "slices_per_s": round(a.slices / dt, 2), "logits_shape": list(out.shape), "dtype": "bf16", "hipgraph": not a.eager,
                      "mirror_axes": list(mirror) if mirror else None}))


if __name__ == "__main__":
    main()
