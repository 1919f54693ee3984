#!/usr/bin/env python
"""Weight-gradient products (contraction-major operands, split-K fp32 atomics) on the multi-phase kernel's TN form (csrc/gemm_p8.hip)
against the 128 x 128 register-staged kernel (csrc/gemm_bf16.hip): within-process interleaved timing on the shapes of one dinounet_l
512^2 training step (tools/step_detail.py), 10 launches per hipGraph replay.
usage: python tools/gemm_tn_bench.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16
L = _lib.lib()


def main(rounds):
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).to(bf)
    # (rows, N of dy, K of x, calls per step)
    shapes = [(43008, 1024, 512, 6), (43008, 1024, 256, 6), (43008, 256, 1024, 6), (43008, 192, 1024, 6), (8192, 512, 1024, 7),
              (131072, 512, 1024, 1), (131072, 512, 256, 1), (32768, 512, 1024, 1), (2048, 256, 256, 3), (8192, 128, 128, 2),
              (131072, 1024, 64, 1), (43008, 1024, 1024, 0), (43008, 4096, 1024, 0)]
    variants = [("old128", 0, 0), ("tn256", 2, 0), ("tn noepi", 2, 2)]
    print(f"{'rows x N x K':>24} {'calls':>5} " + " ".join(f"{n:>10}" for n, _, _ in variants) + "   (us median) | TF/s new | us/step saved")
    saved = 0.0
    for rows, N, K, calls in shapes:
        dy, x = rnd(rows, N), rnd(rows, K)
        graphs, ts = {}, {n: [] for n, _, _ in variants}
        for n, flag, dbg in variants:
            L.du_set_option(5, flag)
            L.du_set_option(3, dbg)
            ops.mm_wgrad(dy, x)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10):
                    ops.mm_wgrad(dy, x)
            graphs[n] = gr
        L.du_set_option(5, 1)
        L.du_set_option(3, 0)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for n, _, _ in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graphs[n].replay()
                e1.record()
                torch.cuda.synchronize()
                ts[n].append(e0.elapsed_time(e1) / 10 * 1e3)
        med = {n: sorted(v)[len(v) // 2] for n, v in ts.items()}
        # note: each captured call also zero-fills its result (ops.ZEROS / torch.zeros), identical for every variant
        saved += calls * (med["old128"] - med["tn256"])
        print(f"{rows:>8} x{N:>5} x{K:>5} {calls:>5} " + " ".join(f"{med[n]:10.1f}" for n, _, _ in variants)
              + f"   | {2.0 * rows * N * K / med['tn256'] / 1e6:7.1f} | {calls * (med['old128'] - med['tn256']):8.1f}", flush=True)
    print(f"sum over the step's calls: {saved:.0f} us/step")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
