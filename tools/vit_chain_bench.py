#!/usr/bin/env python
"""Frozen ViT forward as 1 / 2 / 4 half-batch chains on side streams (dinov3/vision_transformer.py: begin_intermediate_layers):
hipGraph-captured forward of the backbone alone, variants replayed interleaved in one process.  Also: the DVFS sensitivity of the big-GEMM
figures to the operand fill (random normal / uniform / zeros), which decides what a "fraction of 2.5 PF" on random data can be.

usage: python tools/vit_chain_bench.py [--model dinounet_l] [--batch 8] [--rounds 7] [--fills]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE", "1")
import torch  # noqa: E402
from dinounet_amd import _lib, ops  # noqa: E402
from dinounet_amd.dinov3.vision_transformer import build_backbone  # noqa: E402


def chains(a):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    vit = build_backbone(a.model).to(dev).eval()
    x = torch.randn(a.batch, 3, a.size, a.size, device=dev)
    taps = {"dinounet_s": [2, 5, 8, 11], "dinounet_b": [2, 5, 8, 11], "dinounet_l": [4, 11, 17, 23], "dinounet_7b": [9, 19, 29, 39]}[a.model]
    graphs = {}
    for nc in a.chains:
        vit.chains = nc
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                vit.get_intermediate_layers(x, n=taps, dtype=torch.bfloat16)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = vit.get_intermediate_layers(x, n=taps, dtype=torch.bfloat16)
        graphs[nc] = (g, out)
    ref = [t[0].float().clone() for t in graphs[a.chains[0]][1]]
    graphs[a.chains[0]][0].replay()
    torch.cuda.synchronize()
    ref = [t[0].float().clone() for t in graphs[a.chains[0]][1]]
    for nc in a.chains[1:]:
        graphs[nc][0].replay()
        torch.cuda.synchronize()
        err = max(float((t[0].float() - r).abs().max() / r.abs().max()) for t, r in zip(graphs[nc][1], ref))
        print(f"# chains {nc} vs {a.chains[0]}: max rel diff of the tap outputs {err:.2e}")
    t0 = time.time()
    while time.time() - t0 < 0.5:
        graphs[a.chains[0]][0].replay()
    torch.cuda.synchronize()
    ts = {nc: [] for nc in a.chains}
    for _ in range(a.rounds):
        for nc in a.chains:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                graphs[nc][0].replay()
            e1.record()
            torch.cuda.synchronize()
            ts[nc].append(e0.elapsed_time(e1) / 3)
    base = None
    for nc in a.chains:
        v = sorted(ts[nc])
        med = v[len(v) // 2]
        base = base or med
        print(f"{a.model} B{a.batch} {a.size}^2 forward, {nc} chain(s): median {med:8.3f} ms  [{v[0]:.3f} .. {v[-1]:.3f}]   x{base / med:.3f} vs {a.chains[0]} chain(s)", flush=True)


def fills(a):
    dev = torch.device("cuda", 0)
    bf = torch.bfloat16
    for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 4096, 1024)]:
        res = {}
        ops_ = {}
        for fill in ("normal", "uniform", "zeros"):
            if fill == "normal":
                x, w = torch.randn(M, K, device=dev).to(bf), torch.randn(N, K, device=dev).to(bf)
            elif fill == "uniform":
                x, w = (torch.rand(M, K, device=dev) * 2 - 1).to(bf), (torch.rand(N, K, device=dev) * 2 - 1).to(bf)
            else:
                x, w = torch.zeros(M, K, device=dev, dtype=bf), torch.zeros(N, K, device=dev, dtype=bf)
            out = torch.empty(M, N, device=dev, dtype=bf)
            ops.mm(x, w, out=out)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    ops.mm(x, w, out=out)
            ops_[fill] = (g, x, w, out)
        t0 = time.time()
        while time.time() - t0 < 0.5:
            ops_["normal"][0].replay()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for fill, (g, *_r) in ops_.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(fill, []).append(e0.elapsed_time(e1) / 10 * 1e3)
        fl = 2.0 * M * N * K
        print(f"M{M} N{N} K{K} bf16 NT (auto tile): " + "  ".join(f"{f} {sorted(v)[len(v) // 2]:8.1f} us = {fl / sorted(v)[len(v) // 2] / 1e6:7.1f} TF/s" for f, v in res.items()),
              flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinounet_l")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--fills", action="store_true")
    a = ap.parse_args()
    chains(a)
    if a.fills:
        fills(a)
