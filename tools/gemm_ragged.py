#!/usr/bin/env python
"""How much does the ragged last row tile of the ViT products cost?  M = 8 * 1029 = 64 * 128 + 40 rows puts 65 x (N/128) tiles on
the 512 resident workgroup slots (2 per CU); this times the four ViT-L linear shapes at M = 8192 (exact rounds) and M = 8232.
usage: python tools/gemm_ragged.py [reps]      (DU_GEMM_NO_RAGGED_SPLIT=1: without the skinny-tail split of gemm_skinny.hip;
                                                DU_GLDS_VARIANT=3 forces the 3-workgroups-per-CU ring kernel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import ops  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).to(bf)
    print(f"# DU_GLDS_VARIANT={os.environ.get('DU_GLDS_VARIANT', '-')} DU_GEMM_NO_RAGGED_SPLIT={os.environ.get('DU_GEMM_NO_RAGGED_SPLIT', '-')}")
    print(f"{'shape':34s} {'M=8192 us':>10s} {'TF/s':>7s} {'M=8232 us':>10s} {'TF/s':>7s} {'penalty':>8s}")
    for N, K, od in [(3072, 1024, bf), (1024, 1024, torch.float32), (4096, 1024, bf), (1024, 4096, torch.float32)]:
        w = rnd(N, K)
        res = []
        for M in (8192, 8232):
            x = rnd(M, K)
            out = torch.empty((M, N), dtype=od, device=dev)
            t = timeit(lambda: ops.mm(x, w, out=out), reps)
            res.append((t, 2.0 * M * N * K / t * 1e-6))
        pen = res[1][0] / res[0][0] / (8232 / 8192) - 1
        print(f"N{N} K{K} {'f32' if od == torch.float32 else 'bf16'}out".ljust(34) +
              f" {res[0][0]:10.1f} {res[0][1]:7.1f} {res[1][0]:10.1f} {res[1][1]:7.1f} {100 * pen:7.1f}%")


if __name__ == "__main__":
    main()
