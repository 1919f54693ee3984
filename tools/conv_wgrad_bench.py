#!/usr/bin/env python
"""3 x 3 weight-gradient kernels of the decoder layers (csrc/conv_halo.hip): the round-5 kernel (conv3x3_wgrad_rows_kernel: all 9 taps per
wave, fragments read once per tile, two tiles in flight) against the round-3 one (du_set_option(13, 0)) on the shapes of a dinounet_l step
(batch 8): check against the fp32 weight gradient of the same bf16 operands (torch autograd), repeat-run determinism, interleaved timing
of hipGraph-captured launches, algorithmic GB/s = (Cin + Cout) * 2 bytes per pixel / time and TF/s.

usage: python tools/conv_wgrad_bench.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from dinounet_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16
L = _lib.lib()
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)

CASES = [  # B, H, W, C1, C2 (fused concat, 0 = none), Cout
    (8, 512, 512, 32, 0, 32), (8, 512, 512, 32, 32, 32), (8, 256, 256, 64, 0, 64), (8, 256, 256, 64, 64, 64),
    (2, 64, 48, 32, 0, 64), (1, 16, 32, 64, 0, 32), (3, 40, 80, 96, 0, 32), (2, 24, 16, 32, 32, 64), (8, 256, 256, 32, 0, 64),
    # round 6: 128 output channels (first decoder stage): "old" = not served (the step ran them on the grouped launch: 3 jobs, 400 us)
    (8, 128, 128, 128, 0, 128), (8, 128, 128, 128, 128, 128)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ok = True
    print(f"{'case':>30} {'old us':>9} {'new us':>9} {'speedup':>8} {'new GB/s':>9} {'of 8 TB/s':>9} {'TF/s':>7}   err_new  err_old  db_err  deterministic")
    for B, H, W, C1, C2, Co in CASES:
        Cin = C1 + C2
        x = rnd(B, H, W, C1).to(bf)
        x2 = rnd(B, H, W, C2).to(bf) if C2 else None
        dy = rnd(B, H, W, Co).to(bf)
        # fp32 reference: d/dw of sum(conv(x, w) * dy)
        xf = (torch.cat([x, x2], -1) if C2 else x).float().permute(0, 3, 1, 2)
        w = torch.zeros(Co, Cin, 3, 3, device=dev, requires_grad=True)
        (F.conv2d(xf, w, padding=1) * dy.float().permute(0, 3, 1, 2)).sum().backward()
        ref = w.grad.permute(0, 2, 3, 1).reshape(Co, 9 * Cin)          # (tap, ci) column order
        ref_db = dy.float().sum((0, 1, 2))
        scale = ref.abs().max().item()
        fn = lambda: ops.conv3x3_wgrad_halo(x, dy, x2, with_db=True)
        res = {}
        for tag, mode in (("old", 0), ("new", 2)):
            L.du_set_option(13, mode)
            out = fn()
            if out is None:
                res[tag] = None
                continue
            dw, db = out
            res[tag] = (dw.clone(), db.clone())
        if res["new"] is None:
            print(f"{f'{H}x{W} {C1}+{C2}->{Co} b{B}':>30}  not served")
            continue
        L.du_set_option(13, 2)
        same = all(torch.equal(fn()[0], res["new"][0]) for _ in range(6))
        e_new = (res["new"][0] - ref).abs().max().item() / scale
        e_old = (res["old"][0] - ref).abs().max().item() / scale if res["old"] is not None else float("nan")
        e_db = (res["new"][1] - ref_db).abs().max().item() / ref_db.abs().max().item()
        good = e_new < 2e-3 and same and e_db < 1e-3
        ok &= good
        graphs = {}
        for tag, mode in (("old", 0), ("new", 2)):
            if res[tag] is None:
                continue
            L.du_set_option(13, mode)
            fn(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(5):
                    fn()
            graphs[tag] = gr
        t0 = time.time()
        while time.time() - t0 < 0.2:
            graphs["new"].replay()
        torch.cuda.synchronize()
        ts = {k: [] for k in graphs}
        for _ in range(rounds):
            for tag in graphs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); graphs[tag].replay(); e1.record()
                torch.cuda.synchronize()
                ts[tag].append(e0.elapsed_time(e1) / 5 * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        old = med.get("old", float("nan"))
        gbs = B * H * W * (Cin + Co) * 2 / med["new"] / 1e3
        tf = 2.0 * B * H * W * Cin * Co * 9 / med["new"] / 1e6
        print(f"{f'{H}x{W} {C1}+{C2}->{Co} b{B}':>30} {old:9.1f} {med['new']:9.1f} {old / med['new']:8.2f} {gbs:9.0f} {gbs / 8000:9.3f} {tf:7.0f}   "
              f"{e_new:.2e} {e_old:.2e} {e_db:.2e}  {same}{'' if good else '  <-- FAIL'}   (launch + finalize)", flush=True)
    L.du_set_option(13, 1)
    print("CHECK", "PASSED" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
