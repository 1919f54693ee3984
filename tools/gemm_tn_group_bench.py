#!/usr/bin/env python
"""Grouped weight gradients (du_gemm_tn_group: ops.WgradQueue) against one launch per product, on the linear weight-gradient shapes of one
dinounet_l 512^2 training step (adapter: 6 extractors x {value_proj, sampling_offsets + attention_weights, output_proj, ffn.fc1}; ffn.fc2
carries DropPath's contraction scale and stays on the immediate path).  Both arms are captured into hipGraphs (zero fills included) and
replayed interleaved.
usage: python tools/gemm_tn_group_bench.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def main(rounds):
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).to(bf)
    # (rows, N of dy, K of x, with bias gradient) per extractor
    per_extractor = [(8192, 512, 1024, True), (43008, 192, 1024, True), (43008, 1024, 512, True), (43008, 256, 1024, True)]
    sets = {"1 extractor": 1, "3 extractors": 3, "6 extractors": 6}
    W = ops.WGRAD
    for name, reps in sets.items():
        jobs = [(rnd(r, n), rnd(r, k), cs) for _ in range(reps) for (r, n, k, cs) in per_extractor]
        flops = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _ in jobs)

        def run(group):
            if group:
                W._armed = True
                for a, b, cs in jobs:
                    ops.mm_wgrad(a, b, with_colsum=cs, defer=True)
                W._armed = False
                W.flush()
            else:
                for a, b, cs in jobs:
                    ops.mm_wgrad(a, b, with_colsum=cs)
        graphs = {}
        for tag, grp in (("per-product", False), ("grouped", True)):
            run(grp)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                run(grp)
            graphs[tag] = gr
        ts = {k: [] for k in graphs}
        for _ in range(rounds):
            for k, gr in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record()
                torch.cuda.synchronize()
                ts[k].append(e0.elapsed_time(e1) * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        print(f"{name:>14}: {len(jobs):3d} products {flops / 1e9:7.1f} GF | per-product {med['per-product']:8.1f} us ({flops / med['per-product'] / 1e6:6.1f} TF/s)"
              f" | grouped {med['grouped']:8.1f} us ({flops / med['grouped'] / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
