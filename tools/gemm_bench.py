#!/usr/bin/env python
"""Times du_gemm on the shapes that dominate the dinounet_l train step (taken from tools/debug_step.py --shapes).
Run twice to compare engines: DU_GEMM_GENERIC=1 forces the generic kernel.  usage: python tools/gemm_bench.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinounet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rows = []
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).to(bf)
    # linear forward (NT)
    for M, N, K, od in [(8232, 4096, 1024, bf), (8232, 1024, 4096, torch.float32), (8232, 3072, 1024, bf), (8232, 1024, 1024, torch.float32),
                        (43008, 1024, 256, bf), (43008, 1024, 512, bf), (43008, 192, 1024, torch.float32), (43008, 256, 1024, bf),
                        (131072, 1024, 64, bf), (131072, 512, 1024, bf), (524288, 128, 64, bf), (2097152, 8, 32, torch.float32)]:
        x, w = rnd(M, K), rnd(N, K)
        out = torch.empty((M, N), dtype=od, device=dev)
        t = timeit(lambda: ops.mm(x, w, out=out), reps)
        rows.append((f"linear      M{M} N{N} K{K} {'f32' if od == torch.float32 else 'bf16'}out", t, 2.0 * M * N * K, (M * K + N * K) * 2 + M * N * out.element_size()))
    # dgrad (A row, B col)
    for M, N, K in [(43008, 1024, 512), (43008, 256, 1024), (43008, 192, 1024), (43008, 1024, 256), (131072, 512, 1024)]:
        dy, w = rnd(M, N), rnd(N, K)
        t = timeit(lambda: ops.mm_dgrad(dy, w), reps)
        rows.append((f"lin_dgrad   M{M} K(out){K} N(contr){N}", t, 2.0 * M * N * K, (M * N + N * K + M * K) * 2))
    # wgrad (A col, B col)
    for Mr, N, K in [(43008, 1024, 512), (43008, 1024, 256), (43008, 256, 1024), (43008, 192, 1024), (131072, 512, 1024), (8192, 512, 1024), (131072, 32, 256)]:
        dy, x = rnd(Mr, N), rnd(Mr, K)
        t = timeit(lambda: ops.mm_wgrad(dy, x), reps)
        rows.append((f"lin_wgrad   rows{Mr} N{N} K{K}", t, 2.0 * Mr * N * K, (Mr * N + Mr * K) * 2 + N * K * 4))
    # conv 3x3 fwd / dgrad / wgrad, convT
    for B, H, Cin, Cout in [(8, 512, 64, 32), (8, 512, 32, 32), (8, 256, 128, 64), (8, 256, 64, 64), (8, 128, 256, 128), (8, 256, 8, 64)]:
        x = rnd(B, H, H, Cin)
        w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev)
        wp = ops.pack_conv_weight(w, bf)
        wd = ops.pack_conv_weight_dgrad(w, bf)
        dy = rnd(B, H, H, Cout)
        fl = 2.0 * B * H * H * Cin * Cout * 9
        by = B * H * H * (Cin + Cout) * 2
        rows.append((f"conv3x3 fwd   {H}^2 {Cin}->{Cout}", timeit(lambda: ops.conv_fwd(x, wp, None, 3, 3, 1, 1), reps), fl, by))
        rows.append((f"conv3x3 dgrad {H}^2 {Cin}->{Cout} (implicit GEMM)", timeit(lambda: ops.conv_dgrad(dy, wd, 3, 3, 1, 1, H, H), reps), fl, by))
        wf = ops.pack_conv_weight_dgrad_flipped(w, bf)
        if ops.conv3x3_halo(dy, wf, None) is not None:
            rows.append((f"conv3x3 dgrad {H}^2 {Cin}->{Cout} (halo)", timeit(lambda: ops.conv3x3_halo(dy, wf, None), reps), fl, by))
        if ops.conv3x3_halo(x, wp, None, None, True) is not None:
            rows.append((f"conv3x3 fwd+stats {H}^2 {Cin}->{Cout} (halo)", timeit(lambda: ops.conv3x3_halo(x, wp, None, None, True), reps), fl, by))
        rows.append((f"conv3x3 wgrad {H}^2 {Cin}->{Cout}", timeit(lambda: ops.conv_wgrad(x, dy, 3, 3, 1, 1), reps), fl, by))
    for B, H, Cin, Cout in [(8, 64, 1024, 1024), (8, 256, 64, 32), (8, 256, 32, 32)]:
        x = rnd(B, H, H, Cin).requires_grad_(True)
        w = torch.randn(Cin, Cout, 2, 2, generator=g).to(dev).requires_grad_(True)
        bias = torch.zeros(Cout, device=dev)
        fl = 2.0 * B * H * H * Cin * Cout * 4
        by = B * H * H * (Cin + 4 * Cout) * 2
        rows.append((f"convT2x2 fwd  {H}^2 {Cin}->{Cout}", timeit(lambda: ops.conv_transpose2x2(x, w, bias), reps), fl, by))
        y = ops.conv_transpose2x2(x, w, bias)
        go = rnd(*y.shape)
        rows.append((f"convT2x2 bwd  {H}^2 {Cin}->{Cout} (dx+dw+db)", timeit(lambda: torch.autograd.grad(y, (x, w), go, retain_graph=True), max(reps // 4, 3)), 2 * fl, 2 * by))
    print(f"{'us':>9} {'TF/s':>8} {'GB/s':>8}  case   [{'generic' if os.environ.get('DU_GEMM_GENERIC') else 'fast'} engine]")
    for name, t, fl, by in rows:
        print(f"{t:9.1f} {fl / t / 1e6:8.1f} {by / t / 1e3:8.1f}  {name}")


if __name__ == "__main__":
    main()
