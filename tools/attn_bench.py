#!/usr/bin/env python
"""du_attention_fwd alone (RoPE / head split excluded): correctness vs an fp32 softmax(QK^T)V of the same bf16 operands and
timing on the ViT shapes (clocks warmed for half a second, the kernels' graphs replayed in turn, median of 5 rounds: a cold chip runs the
first variant 10 % slower than the last), for the round-4 product kernel (32 queries per wave, no running maximum, nearly empty query
tiles dispatched first), the round-2/3 kernel (du_set_option(6, 1): still d_head 128's kernel) and the two kept experiments
(du_set_option(8, 1): 64 queries per wave; (8, 9): its slot-pipelined form).  Also: the cold rescale path (threshold turned down to zero =
every tile takes it; spiked keys far above the first tile's maximum at the shipped threshold).
usage: python tools/attn_bench.py [reps]   (wrap in rocprofv3 --pmc ... for counters)"""
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


def run(q, k, v, out, B, H, N, Npad, Dh):
    _lib.check(L.du_attention_fwd(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(out.data_ptr()),
                                  B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "du_attention_fwd")


def reference(q, k, v, B, H, N, Dh):
    # q already carries Dh^-1/2 * log2(e): softmax in base 2
    s = torch.einsum("bhqd,bhkd->bhqk", q[:, :, :N].float(), k[:, :, :N].float()) * math.log(2.0)
    return torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v[:, :, :N].float()).permute(0, 2, 1, 3).reshape(B * N, H * Dh)


KERNELS = [("product", 0, 0), ("round 3", 1, 0), ("64q", 0, 1), ("pipelined", 0, 9)]


def main():
    import time
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    g = torch.Generator(device="cpu").manual_seed(0)
    ok = True
    shapes = [(8, 16, 1029, 64, "vit_l 512^2 b8"), (8, 12, 1029, 64, "vit_b"), (16, 6, 1029, 64, "vit_s b16"), (2, 16, 261, 64, "vit_l 256^2"),
              (2, 32, 4101, 128, "vit_7b 1024^2 b2"), (4, 16, 1029, 64, "small grid"), (8, 16, 1024, 64, "N 1024")]
    for B, H, N, Dh, name in shapes:
        Npad = (N + 7) // 8 * 8
        scale = Dh ** -0.5 * math.log2(math.e)
        q = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
        k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
        v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
        out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
        ref = reference(q, k, v, B, H, N, Dh)
        fl = 4.0 * B * H * N * N * Dh
        graphs, errs = [], []
        for tag, impl, var in KERNELS:
            L.du_set_option(6, impl); L.du_set_option(8, var)
            out.zero_()
            run(q, k, v, out, B, H, N, Npad, Dh)
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            o0 = out.clone()
            same = True
            for _ in range(3):
                out.zero_()
                run(q, k, v, out, B, H, N, Npad, Dh)
                same &= bool(torch.equal(out, o0))
            if impl == 0 and var == 0 and Dh == 64:
                L.du_set_option(7, -2000)      # every tile through the cold path (a running maximum after all)
                out.zero_()
                run(q, k, v, out, B, H, N, Npad, Dh)
                L.du_set_option(7, 60)
                same &= float((out.float() - ref).abs().max() / ref.abs().max()) < 2e-2
            errs.append((err, same))
            gr = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr):
                for _ in range(20):
                    run(q, k, v, out, B, H, N, Npad, Dh)
            graphs.append(gr)
        L.du_set_option(6, 0); L.du_set_option(8, 0)
        t0 = time.time()
        while time.time() - t0 < 0.5:
            graphs[0].replay()
        torch.cuda.synchronize()
        res = [[] for _ in KERNELS]
        for _ in range(reps):
            for j, gr in enumerate(graphs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                res[j].append(e0.elapsed_time(e1) / 20 * 1e3)
        line = f"{name:>18} B{B} H{H} N{N} Dh{Dh}:"
        for (tag, _, _), r, (err, same) in zip(KERNELS, res, errs):
            t = sorted(r)[len(r) // 2]
            good = err < 2e-2 and same
            ok &= good
            line += f"  [{tag}] {t:6.1f} us {fl / t / 1e6:6.0f} TF/s ({fl / t / 1e6 / 2500 * 100:4.1f} %) err {err:.0e} {'OK' if good else 'FAIL'}"
        print(line, flush=True)
        del ref
    print("CHECK", "PASSED" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
