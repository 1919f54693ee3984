#!/usr/bin/env python
"""GPU tuning aid (round 3 kernel): where does a K/V tile of du_attention_fwd spend its time?  (Any bit set in option 4 routes the launch
to the round-3 kernel `attn_fwd_kernel` and its ablation instantiation; the round-4 product kernel's ablations are compile-time variants
driven by tools/scratch/attn_abl.py / attn_abl2.py, summary in profiles/r04_attention_ablation_v1.txt.)  Times the kernel with pieces of the tile program switched
off through du_set_option(4, bits) (results are wrong then; timing only) at three occupancies: one workgroup per CU, the dinounet_l
step's grid, and a long sequence.
Column 1 is the product kernel (4 workgroups / CU); every other column is the ablation instantiation (3 workgroups / CU: its
branches cost registers), to be compared with 'abl base'.
bits: 1 no exp2, 2 no running-max bookkeeping, 4 no PV MFMAs, 8 no QK^T MFMAs, 16 no end-of-tile wait + barrier, 32 no DMA in the loop, 128 no s_setprio around the MFMA runs.
usage: python tools/attn_ablate.py"""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
VARIANTS = [(0, "product"), (256, "abl base"), (1, "-exp"), (2, "-max"), (3, "-exp-max"), (4, "-PV"), (8, "-QK"), (12, "-QK-PV"), (15, "-all math"), (16, "-barrier"),
            (48, "-barrier-dma"), (63, "empty loop"), (128, "-setprio")]


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"{'shape':>28} " + " ".join(f"{n:>12}" for _, n in VARIANTS) + "   (us, 10 launches per graph replay, median of 7)")
    for B, H, N, Dh, name in [(2, 16, 1029, 64, "1 wg/CU"), (4, 16, 1029, 64, "2.25 wg/CU"), (8, 16, 1029, 64, "dinounet_l step"), (2, 16, 4101, 64, "long N")]:
        Npad = (N + 127) // 128 * 128
        scale = Dh ** -0.5 * math.log2(math.e)
        q = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
        k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
        v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
        out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()),
                                                    C.c_void_p(out.data_ptr()), B, H, N, Npad, Dh, st()), "du_attention_fwd")
        row = []
        for bits, _ in VARIANTS:
            L.du_set_option(4, bits)
            run()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10):
                    run()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 100.0)
            row.append(sorted(ts)[3])
        # cycle probe: wave 0 of workgroup (1, 0), per-tile means of the segment cycle counts
        L.du_set_option(4, 64)
        run()
        torch.cuda.synchronize()
        host = (C.c_uint64 * 8)()
        _lib.check(L.du_debug_attn_probe(host), "du_debug_attn_probe")
        L.du_set_option(4, 0)
        nt = (N + 63) // 64
        segs = ["dma issue", "K reads + QK MFMA", "max / rescale", "exp2 + row sums", "cvt + V reads + PV MFMA", "wait + barrier"]
        print(f"{name + f' B{B} H{H} N{N}':>28} " + " ".join(f"{t:12.1f}" for t in row), flush=True)
        print(f"{'':>28}   probe, cycles per tile (one wave, {nt} tiles): " + ", ".join(f"{n} {host[j] / nt:.0f}" for j, n in enumerate(segs))
              + f"; total {sum(host[:6]) / nt:.0f}", flush=True)


if __name__ == "__main__":
    main()
