#!/usr/bin/env python
"""Convolution weight gradients on the grouped launch (du_tn_job.gather) against the im2col kernels they replace, on the ConvTranspose /
3 x 3 shapes of one dinounet_l 512^2 training step: each product alone (old path | grouped, alone in its launch | the same without the
epilogue = main loop only) and all of them in ONE flush.  hipGraph-captured, interleaved replays.
usage: python tools/wgrad_gather_bench.py [rounds]      (DU_TN_GROUP_UNITS / DU_TN_GROUP_MINPAIRS are read by the library)"""
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
    B = 8
    # ConvTranspose k2 s2: (H, W, Cin, Cout) of the INPUT; 3 x 3: (H, W, Cin, Cout)
    convt = [(256, 256, 64, 32), (256, 256, 32, 32), (128, 128, 128, 64), (128, 128, 64, 64), (64, 64, 256, 128), (128, 128, 32, 32),
             (64, 64, 128, 128), (32, 32, 256, 256), (64, 64, 64, 64), (32, 32, 128, 128), (16, 16, 256, 256), (64, 64, 1024, 1024)]
    conv3 = [(128, 128, 256, 128), (128, 128, 128, 128)]
    jobs = []
    for (H, W, Ci, Co) in convt:
        x, dy = rnd(B, H, W, Ci), rnd(B, 2 * H, 2 * W, Co)
        jobs.append(("convT %4dx%-4d %4d->%-4d" % (H, W, Ci, Co), 2, x, dy, (Ci, Co, 2, 2), 2.0 * B * H * W * Ci * 4 * Co,
                     2.0 * B * H * W * (Ci + 4 * Co)))
    for (H, W, Ci, Co) in conv3:
        x, dy = rnd(B, H, W, Ci), rnd(B, H, W, Co)
        jobs.append(("conv3 %4dx%-4d %4d->%-4d" % (H, W, Ci, Co), 3, x, dy, (Co, Ci, 3, 3), 2.0 * B * H * W * Ci * 9 * Co,
                     2.0 * B * H * W * (Ci + Co)))

    def old(kind, x, dy, wshape):
        if kind == 2:
            Bn, H, W, Ci = x.shape
            Co = dy.shape[-1]
            geo, _, lddy, _ = ops._geom(dy, 2, 2, 2, 0, H, W, 0)
            gw = ops.ZEROS.zeros((Ci, 4 * Co), dev)
            tiles = ((Ci + 127) // 128) * ((4 * Co + 127) // 128)
            ops.gemm_raw(dtype=ops.DU_BF16, out_dtype=ops.DU_F32, a_mode=ops.PLAIN_COL, b_mode=ops.IM2COL_COL, M=Ci, N=4 * Co, K=Bn * H * W,
                         A=x.data_ptr(), lda=Ci, B=dy.data_ptr(), ldb=lddy, Cmat=gw.data_ptr(), ldc=4 * Co,
                         split_k=ops._split_for(tiles, Bn * H * W, 1024), geom=geo)
        else:
            ops.conv_wgrad(x, dy, 3, 3, 1, 1)

    def grouped(kind, x, dy, wshape):
        Bn, H, W, Ci = x.shape
        if kind == 2:
            r = ops._queue_conv_wgrad(None, 2, x, Ci, [(dy, dy.shape[-1], dy.shape[-1])], Bn, H, W, Ci, wshape, True)
        else:
            r = ops._queue_conv_wgrad(None, 3, dy, dy.shape[-1], [(x, Ci, Ci)], Bn, H, W, dy.shape[-1], wshape, True)
        assert r is not None

    def capture(fn):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        return gr

    def timeit(graphs):
        ts = {k: [] for k in graphs}
        for _ in range(rounds):
            for k, gr in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record()
                torch.cuda.synchronize()
                ts[k].append(e0.elapsed_time(e1) * 1e3)
        return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}

    W_ = ops.WGRAD
    print(f"DU_TN_GROUP_UNITS={os.environ.get('DU_TN_GROUP_UNITS', '256')} DU_TN_GROUP_MINPAIRS={os.environ.get('DU_TN_GROUP_MINPAIRS', '8')}")
    print(f"{'product':>28} {'old us':>9} {'alone us':>9} {'no-epi us':>9} | alone TF/s  GB/s(alg)")
    tot_old = 0.0
    for name, kind, x, dy, wshape, fl, nb in jobs:
        graphs = {"old": capture(lambda: old(kind, x, dy, wshape)), "alone": capture(lambda: grouped(kind, x, dy, wshape))}
        L.du_set_option(3, 2)
        graphs["noepi"] = capture(lambda: grouped(kind, x, dy, wshape))
        L.du_set_option(3, 0)
        m = timeit(graphs)
        tot_old += m["old"]
        print(f"{name:>28} {m['old']:9.1f} {m['alone']:9.1f} {m['noepi']:9.1f} | {fl / m['alone'] / 1e6:9.1f} {nb / m['alone'] / 1e3:9.1f}", flush=True)

    def all_grouped():
        W_._armed = True
        for name, kind, x, dy, wshape, fl, nb in jobs:
            grouped(kind, x, dy, wshape)
        W_._armed = False
        W_.flush()
    m = timeit({"all": capture(all_grouped)})
    print(f"all {len(jobs)} products in one flush: {m['all']:.1f} us  (sum of the old launches: {tot_old:.1f} us)")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
