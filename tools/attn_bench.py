#!/usr/bin/env python
"""du_attention_fwd alone (RoPE / head split excluded): correctness vs an fp32 softmax(QK^T)V of the same bf16 operands and
graph-replayed timing on the ViT shapes.  usage: python tools/attn_bench.py [reps]   (wrap in rocprofv3 --pmc ... for counters)"""
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


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    g = torch.Generator(device="cpu").manual_seed(0)
    ok = True
    for B, H, N, Dh, name in [(8, 16, 1029, 64, "vit_l 512^2 b8"), (8, 12, 1029, 64, "vit_b"), (16, 6, 1029, 64, "vit_s b16"), (2, 16, 261, 64, "vit_l 256^2"),
                              (2, 32, 4101, 128, "vit_7b 1024^2 b2"), (4, 16, 1029, 64, "small grid")]:
        Npad = (N + 127) // 128 * 128
        scale = Dh ** -0.5 * math.log2(math.e)
        q = (torch.randn(B, H, Npad, Dh, generator=g) * scale).to(dev, torch.bfloat16)
        k = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
        v = torch.randn(B, H, Npad, Dh, generator=g).to(dev, torch.bfloat16)
        out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=dev)
        run(q, k, v, out, B, H, N, Npad, Dh)
        # reference: q already carries Dh^-1/2 * log2(e): softmax in base 2
        s = torch.einsum("bhqd,bhkd->bhqk", q[:, :, :N].float(), k[:, :, :N].float()) * math.log(2.0)
        ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v[:, :, :N].float()).permute(0, 2, 1, 3).reshape(B * N, H * Dh)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        same = True
        o0 = out.clone()
        for _ in range(5):
            out.zero_()
            run(q, k, v, out, B, H, N, Npad, Dh)
            same &= bool(torch.equal(out, o0))
        del s, ref
        res = {}
        for w in (0,):
            L.du_set_option(4, w)
            run(q, k, v, out, B, H, N, Npad, Dh)
            if w:
                same &= bool(torch.equal(out, o0)) or True   # W changes only the grouping of query blocks: results identical per row
            gr = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr):
                for _ in range(10):
                    run(q, k, v, out, B, H, N, Npad, Dh)
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            res[w] = sorted(ts)[len(ts) // 2]
            werr = float((out.float() - o0.float()).abs().max())
            assert werr == 0.0, (w, werr)
        L.du_set_option(4, 0)
        t = res[0]
        fl = 4.0 * B * H * N * N * Dh
        good = err < 2e-2 and same
        ok &= good
        print(f"{name:>18} B{B} H{H} N{N} Dh{Dh}: {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s ({fl / t / 1e6 / 2500 * 100:4.1f} % of 2.5 PF)  "
              f"rel err {err:.2e} deterministic {same} -> {'OK' if good else 'FAIL'}", flush=True)
    print("CHECK", "PASSED" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
