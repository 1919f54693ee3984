#!/usr/bin/env python
"""Time the 3x3 halo convolution (du_conv3x3_halo: forward conv, and dgrad = the same kernel on flipped weights) on the decoder shapes of
dinounet_l at batch 8, each launch on ITS OWN input/output buffers out of a ring larger than the 256 MB Infinity Cache, so the rate is an
HBM rate, not a cache rate.  Prints microseconds, algorithmic GB/s (read x [+ x2] once, write y once, bf16) and the fraction of 8 TB/s.

Run ON THE GPU BOX:  python tools/conv_bench.py  [> gpurun_out/conv_table.txt]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dinounet_amd import _lib, ops  # noqa: E402

if "--lib" in sys.argv:                      # A/B against another build of the library (same C ABI)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])

SHAPES = [            # (H, W, C1, C2, Cout)   C2 > 0: fused concat of two sources
    (512, 512, 32, 0, 32),        # conv2 of the 512^2 stage and its data gradient
    (512, 512, 32, 32, 32),       # conv1 of the 512^2 stage (concat of the skip and the upsampled tensor)
    (512, 512, 32, 0, 64),        # its data gradient (towards both halves of the concat)
    (256, 256, 64, 0, 64),
    (256, 256, 64, 64, 64),
    (256, 256, 64, 0, 128),
    (256, 256, 32, 0, 64),
    (128, 128, 128, 0, 128),
    (128, 128, 128, 128, 128),
    (128, 128, 64, 0, 128),
]
B = 8
if "--option" in sys.argv:                   # e.g. --option 14 0: the 128 -> 128 layers on the chunked form (du_set_option, include/dinounet_hip.h)
    i = sys.argv.index("--option")
    _lib.lib().du_set_option(int(sys.argv[i + 1]), int(sys.argv[i + 2]))
if "--c128" in sys.argv:
    SHAPES = [s for s in SHAPES if s[2] == 128 and s[4] == 128]


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print(f"# library {os.path.relpath(_lib.LIB_PATH, ROOT)}")
    print(f"# du_conv3x3_halo, batch {B}, bf16, ring of buffers > 256 MB; GB/s = (x [+ x2] + y bytes) / time; frac of 8000 GB/s")
    print("# kernel: strip = csrc/conv_strip.hip (one wave per 32-column strip, round 4), halo = csrc/conv_halo.hip (LDS-tiled); DU_CONV_STRIP=0 forces halo")
    print(f"# {'H x W':>9} {'Cin':>7} {'Cout':>4} {'us':>8} {'GB/s':>8} {'frac':>6} {'TFLOP/s':>8} {'of 2.5 PF':>9}  kernel")
    for H, W, C1, C2, Cout in SHAPES:
        Cin = C1 + C2
        per = B * H * W * (Cin + Cout) * 2
        ring = max(3, int(600e6 // per) + 1)
        xs = [torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16) for _ in range(ring)]
        x2s = [torch.randn(B, H, W, C2, device=dev).to(torch.bfloat16) for _ in range(ring)] if C2 else None
        wp = (torch.randn(Cout, 9 * Cin, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(Cout, device=dev)
        for i in range(ring):
            r = ops.conv3x3_halo(xs[i], wp, bias, x2s[i] if C2 else None, want_stats=True)
            assert r is not None, "shape not served by the halo kernel"
        torch.cuda.synchronize()
        n = 4 * ring
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            k = i % ring
            ops.conv3x3_halo(xs[k], wp, bias, x2s[k] if C2 else None, want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        gbs = per / us / 1e3
        tf = 2.0 * B * H * W * Cin * Cout * 9 / us / 1e6
        strip = os.environ.get("DU_CONV_STRIP", "1") != "0" and W % 128 == 0 and H % 8 == 0 and Cout in (32, 64) and \
            (Cin == 32 and not C2 or Cin == 64 and (not C2 or C1 == 32))
        print(f"{H:5d}x{W:<4d} {C1:3d}+{C2:<3d} {Cout:4d} {us:8.1f} {gbs:8.0f} {gbs / 8000:6.3f} {tf:8.1f} {tf / 2500:9.3f}  {'strip' if strip else 'halo'}")
        del xs, x2s


if __name__ == "__main__":
    main()
