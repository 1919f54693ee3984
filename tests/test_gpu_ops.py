"""GPU parity tests (-m gpu): every HIP kernel, called through the C-ABI (dinounet_amd.ops -> libdinounet_hip.so),
against the CPU oracle / a plain PyTorch fp32 reference of the same op on the same seeded inputs.
Tolerances: fp32 kernels 2e-4 (max-abs error relative to the reference's max-abs; fp32 MFMA accumulation order differs
from the CPU's), bf16 kernels 3e-2; MSDA fp32 vs the reference's fp64 fixture 1e-5."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DTS = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from dinounet_amd import _lib
    assert _lib.lib().du_device_ok() == 1, "libdinounet_hip.so kernels are built for gfx950 only"
    return torch.device("cuda:0")


def rel(a, b):
    """max |a - b| / max |b|, taken over the whole tensor AND per slice of the last dimension (channel / output column; slices floored at
    1 % of the global maximum, weighted 1/4): a channel that is wrong as a whole cannot hide behind a larger one.  Over the 519 comparisons
    of this file the per-slice figure is <= 3.8 x the global one wherever it is not pure fp32 round-off (< 1e-5)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), "non-finite values in kernel output"
    bmax = b.abs().max()
    g = float((a - b).abs().max() / bmax.clamp_min(1e-20))
    if a.dim() >= 2 and a.shape[-1] > 1 and float(bmax) > 0:
        red = tuple(range(a.dim() - 1))
        col = float(((a - b).abs().amax(red) / b.abs().amax(red).clamp_min(1e-2 * bmax)).max())
        g = max(g, col / 4)
    return g


def close(a, b, rtol, atol):
    """element-wise |a - b| <= atol + rtol |b| (torch.allclose's rule, the gate ops/test.py:80 uses): unlike rel(), small-magnitude
    entries cannot hide behind the largest one"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    bad = (a - b).abs() > atol + rtol * b.abs()
    return not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} elements outside rtol {rtol} atol {atol}; worst |diff| {float((a - b).abs().max()):.3e}"


def q(t, dt):
    """quantise a CPU fp32 tensor through dt so the reference sees the same inputs as the kernel"""
    return t.to(dt).float()


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(1029, 1152, 384), (300, 32, 64), (257, 64, 128), (128, 128, 64), (4096, 2, 32), (70, 200, 1024)])
def test_gemm_plain_epilogues(dt, M, N, K):
    from dinounet_amd import ops
    from dinounet_amd._lib import ACT_GELU
    d = dev()
    x, w = q(gen(M, K, seed=1), dt), q(gen(N, K, seed=2, scale=K ** -0.5), dt)
    b, gam, res = gen(N, seed=3), gen(N, seed=4), gen(M, N, seed=5)
    y = ops.mm(x.to(d, dt), w.to(d, dt), bias=b.to(d))
    assert rel(y, x @ w.t() + b) < TOL[dt]
    y = ops.mm(x.to(d, dt), w.to(d, dt), bias=b.to(d), act=ACT_GELU)
    assert rel(y, F.gelu(x @ w.t() + b)) < TOL[dt]
    r = res.to(d)
    y = ops.mm(x.to(d, dt), w.to(d, dt), bias=b.to(d), gamma=gam.to(d), residual=r, out=r)   # in-place fp32 residual stream
    assert y.dtype == torch.float32
    assert rel(y, (x @ w.t() + b) * gam + res) < TOL[dt]


@pytest.mark.parametrize("mode", [1, 2, 4])
@pytest.mark.parametrize("M,N,K,f32out,epi", [(8232, 3072, 1024, False, "bias"), (8232, 4096, 1024, False, "gelu"),
                                              (8232, 1024, 4096, True, "ls_res"), (8232, 1024, 1024, True, "ls_res"),
                                              (1000, 516, 512, True, "bias"), (768, 640, 384, False, "none"), (2048, 384, 1536, False, "rs"),
                                              # the persistent kernel's own corners (mode 4; the other modes run them as ordinary shapes): several
                                              # tiles per workgroup with ragged last tile rows AND columns, K at its lower limit (drain over 4
                                              # K-steps) and above 768 with GELU (drain over 8), no bias, one tile per workgroup
                                              (70000, 264, 512, False, "bias"), (33000, 1000, 896, False, "gelu"), (33000, 1000, 640, False, "gelu"),
                                              (43008, 1024, 512, False, "none"), (5000, 136, 2048, False, "bias"),
                                              # ... K = 256 (the four-K-step form: the drain runs under the whole next tile) and K = 384
                                              (43008, 1024, 256, False, "bias"), (70000, 264, 256, False, "gelu"), (33000, 520, 384, False, "bias"),
                                              # round 6: a bf16 residual as two more K-steps of the persistent kernel's tile (+ DropPath's per-sample
                                              # scale): the adapter's output projection / ConvFFN fc2 shapes, K = 256 (six-K-step tile), 384, 512, 1024,
                                              # ragged last tile row + rows behind the last full tile, one tile per workgroup
                                              (43008, 1024, 256, False, "res_rs"), (43008, 1024, 512, False, "res"), (33000, 1024, 384, False, "res"),
                                              (10752, 256, 256, False, "res_rs"), (21504, 640, 1024, False, "res_rs"), (5000, 128, 512, False, "res"),
                                              (16424, 512, 512, False, "res")])
def test_gemm_multiphase_nt(mode, M, N, K, f32out, epi):
    """256 x 256 (mode 1), 256 x 128 (mode 2) and persistent 256 x 128 (mode 4, round 5; where its epilogue rules do not hold the library
    runs the mode-2 kernel) multi-phase NT kernels (gemm_p8.hip) forced through du_set_option, on the ViT-L products and ragged shapes:
    full-matrix check against the fp32 product of the same bf16 operands, every epilogue the ViT uses, and a repeat-run screen (the
    kernels are deterministic: a run-to-run difference is a pipeline race)."""
    from dinounet_amd import ops, _lib
    from dinounet_amd._lib import ACT_GELU
    d = dev()
    bf = torch.bfloat16
    x, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(N, K, seed=2, scale=K ** -0.5), bf).to(d, bf)
    b, gam, res = gen(N, seed=3).to(d), gen(N, seed=4).to(d), gen(M, N, seed=5).to(d)
    rows = 8
    rs = (torch.arange((M + rows - 1) // rows, device=d) % 3 != 0).float() * 1.5
    ref = x.float() @ w.float().t()
    kw = {}
    if epi in ("bias", "gelu", "ls_res"):
        kw["bias"] = b
        ref = ref + b
    if epi == "gelu":
        kw["act"] = ACT_GELU
        ref = F.gelu(ref)
    if epi == "ls_res":
        kw.update(gamma=gam, residual=res)
        ref = ref * gam + res
    if epi == "rs":
        kw.update(row_scale=rs, rs_rows=rows)
        ref = ref * rs.repeat_interleave(rows)[:M, None]
    if epi in ("res", "res_rs"):                         # y = s * (x w^T + b) + r, r in the result's dtype (dinov3_adapter.py:142-148)
        kw["bias"] = b
        ref = ref + b
        if epi == "res_rs":
            rows = 5376
            rs = (torch.arange((M + rows - 1) // rows, device=d) % 3 != 0).float() / 0.7
            kw.update(row_scale=rs, rs_rows=rows)
            ref = ref * rs.repeat_interleave(rows)[:M, None]
        resb = res.to(bf)
        kw["residual"] = resb
        ref = ref + resb.float()
    od = torch.float32 if f32out else bf
    L = _lib.lib()
    try:
        L.du_set_option(0, mode)
        ops.TRACK_ROUTE = True
        outs = []
        for _ in range(6):
            out = torch.empty((M, N), dtype=od, device=d)
            if epi == "ls_res":
                kw["residual"] = res.clone() if od == torch.float32 else res.to(od)
            outs.append(ops.mm(x, w, out=out, **kw).float())
        if mode == 4 and epi in ("res", "res_rs"):
            assert ops.LAST_GEMM_ROUTE == 6, ops.LAST_GEMM_ROUTE      # the persistent kernel itself, not a fallback
    finally:
        L.du_set_option(0, -1)
        ops.TRACK_ROUTE = False
    assert rel(outs[0], ref) < (2e-4 if f32out and epi != "ls_res" else TOL[bf])
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "run-to-run difference: pipeline race"
    if mode == 4 and not f32out and epi in ("bias", "none"):
        # same K order, same epilogue arithmetic: the persistent kernel's tiles are bit-identical to the 256 x 128 kernel's (the ragged rows
        # behind the last full tile take the same K-parallel tail program in both)
        try:
            L.du_set_option(0, 2)
            ref2 = ops.mm(x, w, out=torch.empty((M, N), dtype=od, device=d), **kw).float()
        finally:
            L.du_set_option(0, -1)
        assert torch.equal(outs[0], ref2)


def test_gemm_persistent_residual_in_place():
    """The residual of the persistent kernel's residual form may BE the output buffer (y = y + f(x), in place): a tile reads its residual
    rows through the staging ring during its own K-steps and stores them in the drain of the NEXT tile, other tiles touch other rows /
    columns.  Same result as the out-of-place call, bit for bit."""
    from dinounet_amd import ops, _lib
    d = dev()
    bf = torch.bfloat16
    M, N, K = 43008, 1024, 256
    x, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(N, K, seed=2, scale=K ** -0.5), bf).to(d, bf)
    b = gen(N, seed=3).to(d)
    res = q(gen(M, N, seed=5), bf).to(d, bf)
    rs = (torch.arange(M // 5376, device=d) % 3 != 0).float() / 0.7
    ops.TRACK_ROUTE = True
    try:
        want = ops.mm(x, w, bias=b, residual=res, row_scale=rs, rs_rows=5376)
        assert ops.LAST_GEMM_ROUTE == 6, ops.LAST_GEMM_ROUTE
        buf = res.clone()
        got = ops.mm(x, w, bias=b, residual=buf, out=buf, row_scale=rs, rs_rows=5376)
    finally:
        ops.TRACK_ROUTE = False
    assert got.data_ptr() == buf.data_ptr() and torch.equal(got, want)


@pytest.mark.parametrize("M,N,K", [(4136, 1024, 1024), (8232, 1024, 4096), (4096, 2048, 1024)])
def test_gemm_k_split_pairs_every_way_through_the_exchange(M, N, K):
    """fp32-result products with 64-128 tiles of 256 x 256 (the ViT's proj / fc2: LayerScale + residual epilogue, layers/block.py:126-198) as
    K-split pairs of workgroups (gemm_nt_p8ks_kernel, du_set_option key 16): against the fp32 product; the three ways through the exchange
    (both halves resident; one leaves first and the other finds its flag in the wait loop / at its first look -- forced by the test aids in
    du_set_option key 3) give the SAME bits, repeats are bit-identical, the scratch state is zero again after every launch and the
    error word (a flag that never came) stays clear; ragged rows ride in the same launch."""
    from dinounet_amd import ops, _lib
    d = dev()
    bf = torch.bfloat16
    L = _lib.lib()
    x, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(N, K, seed=2, scale=K ** -0.5), bf).to(d, bf)
    b, gam, res = gen(N, seed=3).to(d), gen(N, seed=4).to(d), gen(M, N, seed=5).to(d)
    ref = ((x.double() @ w.double().t()) + b.double()) * gam.double() + res.double()
    run = lambda: ops.mm(x, w, bias=b, gamma=gam, residual=res, out=torch.empty((M, N), dtype=torch.float32, device=d))
    outs = []
    ops.TRACK_ROUTE = True
    try:
        L.du_set_option(16, 1)
        for aid in (0, 8, 24, 16):
            L.du_set_option(3, aid)
            y = run()
            assert ops.LAST_GEMM_ROUTE == 8, ops.LAST_GEMM_ROUTE
            for _ in range(10 if aid == 0 else 2):
                assert torch.equal(run(), y)
            torch.cuda.synchronize()
            state = next(iter(ops._KS_SCRATCH.values()))[:131072].view(torch.int32)
            assert int(state.abs().sum().item()) == 0, f"aid {aid}: pair state not restored (error word {int(state[16380].item())})"
            outs.append(y)
        L.du_set_option(16, 0)
        L.du_set_option(3, 0)
        plain = run()
        assert ops.LAST_GEMM_ROUTE != 8, ops.LAST_GEMM_ROUTE
    finally:
        ops.TRACK_ROUTE = False
        L.du_set_option(16, 0)
        L.du_set_option(3, 0)
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    scale = ref.abs().max().item()
    assert (outs[0].double() - ref).abs().max().item() / scale < 2e-5
    assert (outs[0] - plain).abs().max().item() / scale < 2e-5


@pytest.mark.parametrize("od", ["f32", "bf16"])
def test_gemm_ragged_rows_as_k_sliced_units(od):
    """fc2 of the ViT (M = 8 x 1029, K = 4096): the 40 ragged rows run as (32 columns, K slice) units behind the tiles that meet through
    du_gemm_args.ks_ws -- slabs written through, a ticket, the last arriver adds the slices in slice order (du_set_option key 17).  Same rows
    as the one-unit-per-column-block form up to fp32 summation order, bit-identical run to run, tickets back at zero."""
    import ctypes
    from dinounet_amd import ops, _lib
    d = dev()
    bf = torch.bfloat16
    L = _lib.lib()
    M, N, K = 8232, 1024, 4096
    odt = torch.float32 if od == "f32" else bf
    x, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(N, K, seed=2, scale=K ** -0.5), bf).to(d, bf)
    b, gam, res = gen(N, seed=3).to(d), gen(N, seed=4).to(d), gen(M, N, seed=5).to(d).to(odt)
    kw = dict(bias=b, residual=res) if od == "bf16" else dict(bias=b, gamma=gam, residual=res)
    run = lambda: ops.mm(x, w, out=torch.empty((M, N), dtype=odt, device=d), **kw)
    a = _lib.GemmArgs()
    a.dtype, a.out_dtype, a.a_mode, a.b_mode = _lib.DU_BF16, _lib.DU_F32 if od == "f32" else _lib.DU_BF16, ops.PLAIN_ROW, ops.PLAIN_ROW
    a.M, a.N, a.K, a.batch, a.split_k, a.alpha = M, N, K, 1, 1, 1.0
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = x.data_ptr(), K, w.data_ptr(), K, res.data_ptr(), N
    assert int(L.du_gemm_ks_ws_bytes(ctypes.byref(a))) >= 131072 + 32 * 8 * 8192
    try:
        y = run()
        for _ in range(10):
            assert torch.equal(run(), y)
        torch.cuda.synchronize()
        state = next(iter(ops._KS_SCRATCH.values()))[:131072].view(torch.int32)
        assert int(state.abs().sum().item()) == 0
        L.du_set_option(17, 0)
        assert int(L.du_gemm_ks_ws_bytes(ctypes.byref(a))) == 0
        y1 = run()
    finally:
        L.du_set_option(17, 1)
    assert torch.equal(y[:8192], y1[:8192])
    ref = (x[8192:].double() @ w.double().t() + b.double()) * (gam.double() if od == "f32" else 1.0) + res[8192:].double()
    tol = 2e-5 if od == "f32" else 1e-2
    scale = ref.abs().max().item()
    assert (y[8192:].double() - ref).abs().max().item() / scale < tol
    assert (y1[8192:].double() - ref).abs().max().item() / scale < tol


@pytest.mark.parametrize("inline", [0, 1])
@pytest.mark.parametrize("N,K", [(1024, 1024), (3072, 1024), (1024, 4096)])
def test_gemm_ragged_tail_split(N, K, inline):
    """The ViT-L products (M = 8 * 1029 = 64 * 128 + 40): for proj / fc2 the last 40 rows leave the tile grid and run on the K-parallel
    skinny kernels (gemm_skinny.hip); every epilogue, against the fp32 product."""
    import ctypes
    from dinounet_amd import ops, _lib
    from dinounet_amd._lib import ACT_GELU
    d = dev()
    M, bf = 8232, torch.bfloat16
    _lib.lib().du_set_option(15, inline)        # round 6: the tail units inside the tile workgroups (1) or as extra workgroups (0, default)
    try:
        _ragged_tail_body(N, K, M, bf, d, ops, _lib, ACT_GELU, ctypes)
    finally:
        _lib.lib().du_set_option(15, 0)


def _ragged_tail_body(N, K, M, bf, d, ops, _lib, ACT_GELU, ctypes):
    x, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(N, K, seed=2, scale=K ** -0.5), bf).to(d, bf)
    b, gam, res = gen(N, seed=3).to(d), gen(N, seed=4).to(d), gen(M, N, seed=5).to(d)
    rs = (torch.arange(1029, device=d) % 3 != 0).float() * 1.5          # one scale per 8 rows: the split keeps the row blocks aligned
    # the library asks for scratch on this shape
    a = _lib.GemmArgs()
    a.dtype, a.out_dtype, a.a_mode, a.b_mode, a.M, a.N, a.K = _lib.DU_BF16, _lib.DU_BF16, 0, 0, M, N, K
    a.lda, a.ldb, a.ldc, a.batch, a.split_k = K, K, N, 1, 1
    # every ViT-L product sheds its last 40 rows: behind 256 x 256 tiles (gemm_p8.hip) a ragged 33rd tile row would run a full K loop
    # for 40 live rows; behind 128 x 128 tiles one or two exact rounds of full tiles (N = 1024: 512 tiles on 512 slots) are split
    assert int(_lib.lib().du_gemm_ws_elems(ctypes.byref(a))) > 0
    ref = x.float() @ w.float().t()
    y = ops.mm(x, w, bias=b, act=ACT_GELU)
    assert y.dtype == bf and rel(y, F.gelu(ref + b)) < TOL[bf]
    assert rel(y[-40:], F.gelu(ref + b)[-40:]) < TOL[bf]
    r = res.clone()
    y = ops.mm(x, w, bias=b, gamma=gam, residual=r, out=r, row_scale=rs, rs_rows=8)            # in-place fp32 residual stream
    want = (ref + b) * gam * rs.repeat_interleave(8)[:, None] + res
    assert y.dtype == torch.float32 and rel(y, want) < TOL[bf] and rel(y[-40:], want[-40:]) < TOL[bf]
    # the tail rows are produced from fp32 partial sums: at least as close to the fp32 product as the tile kernel's rows
    y32 = ops.mm(x, w, out_dtype=torch.float32)
    assert rel(y32[-40:], ref[-40:]) < 2e-3 and rel(y32[:-40], ref[:-40]) < 2e-3


@pytest.mark.parametrize("dt", DTS)
def test_gemm_dgrad_wgrad(dt):
    from dinounet_amd import ops
    d = dev()
    M, N, K = 5376 * 2, 192, 384
    x, w, dy = q(gen(M, K, seed=1), dt), q(gen(N, K, seed=2, scale=K ** -0.5), dt), q(gen(M, N, seed=3), dt)
    dx = ops.mm_dgrad(dy.to(d, dt), w.to(d, dt))
    assert rel(dx, dy @ w) < TOL[dt]
    dw = ops.mm_wgrad(dy.to(d, dt), x.to(d, dt))
    assert dw.dtype == torch.float32
    assert rel(dw, dy.t() @ x) < TOL[dt]
    assert rel(ops.colsum(dy.to(d, dt)), dy.sum(0)) < TOL[dt]


@pytest.mark.parametrize("rows,N,K", [(2048, 256, 1024), (4224, 192, 520), (43008, 512, 1024), (8192, 1024, 384), (2304, 320, 264)])
def test_gemm_multiphase_wgrad(rows, N, K):
    """Weight gradients on the multi-phase kernel's contraction-major form (gemm_p8.hip, TN: transpose-read fragments, K-tile pairs
    dealt out over <= 256 workgroups, fp32 atomics) vs the fp32 product of the same bf16 operands, and vs the 128 x 128 kernel it
    replaces (du_set_option key 5).  Shapes: the adapter's linears (43008 tokens), a ragged 192-row output, odd pair counts."""
    from dinounet_amd import _lib, ops
    d = dev()
    dt = torch.bfloat16
    dy, x = q(gen(rows, N, seed=1), dt), q(gen(rows, K, seed=2), dt)
    ref = (dy.double().t() @ x.double()).float()
    L = _lib.lib()
    ops.TRACK_ROUTE = True
    try:
        outs = {}
        for flag in (2, 0):
            L.du_set_option(5, flag)
            # with_colsum: the bias gradient sum_rows dy taken inside the weight-gradient kernel (du_gemm_args.a_colsum), both families
            dw, db = ops.mm_wgrad(dy.to(d, dt), x.to(d, dt), with_colsum=True)
            outs[flag] = dw.float().cpu()
            assert (ops.LAST_GEMM_ROUTE == 5) == (flag == 2), ops.LAST_GEMM_ROUTE
            assert rel(db, dy.double().sum(0).float()) < 2e-5, flag
    finally:
        L.du_set_option(5, 1)
        ops.TRACK_ROUTE = False
    assert rel(outs[2], ref) < 2e-5          # bf16 products are exact in fp32: only the summation order differs
    assert rel(outs[0], ref) < 2e-5


def test_gemm_tn_group_many_products_one_launch():
    """du_gemm_tn_group (gemm_p8.hip: gemm_tn_group_kernel): a batch of weight-gradient products of very different shapes -- the adapter's
    43008-token linears, a value projection with a short contraction (one split: plain stores), ragged / tiny M and N (192, 72, 8 columns:
    tiles far from full), a strided dY view, K = 512 (two K-tile pairs, the minimum) -- queued and run by ONE launch, each against the fp32
    product of the same bf16 operands; bias gradients (a_colsum) ride along for some of them.  Then every product alone (one job per
    launch: the workgroups all go to that job, up to 32 splits) must give the same result."""
    from dinounet_amd import ops
    d = dev()
    dt = torch.bfloat16
    shapes = [(43008, 1024, 512, True), (43008, 192, 1024, True), (8192, 512, 1024, False), (5376, 72, 256, True), (512, 256, 384, True),
              (2688, 8, 32, False), (16384, 320, 264, True), (1024, 1024, 1024, False)]
    ins, refs = [], []
    for i, (rows, N, K, cs) in enumerate(shapes):
        dy, x = q(gen(rows, N + (8 if i == 3 else 0), seed=10 + i), dt), q(gen(rows, K, seed=40 + i), dt)
        dyd = dy.to(d, dt)
        if i == 3:
            dy, dyd = dy[:, :N], dyd[:, :N]             # a column slice of a wider tensor: lda = N + 8
        ins.append((dyd, x.to(d, dt), cs))
        refs.append(((dy.double().t() @ x.double()).float(), dy.double().sum(0).float()))
    W = ops.WGRAD
    assert W.enabled
    l0, q0 = W.launches, W.queued
    hold = W._arm
    W._arm = lambda: True                              # hold the queue (outside a backward pass a job would be launched at once)
    try:
        outs = [ops.mm_wgrad(a, b, with_colsum=cs, defer=True) for a, b, cs in ins]
        assert W.queued - q0 == len(shapes) and len(W.jobs) == len(shapes)
    finally:
        W._arm = hold
    W.flush()
    assert W.launches - l0 == 1 and not W.jobs and not W.keep
    for (rows, N, K, cs), o, (rw, rb) in zip(shapes, outs, refs):
        dw, db = (o if cs else (o, None))
        assert dw.shape == (N, K)
        assert rel(dw, rw) < 2e-5, (rows, N, K)
        if cs:
            assert rel(db, rb) < 2e-5, (rows, N, K)
    for (rows, N, K, cs), (a, b, _), (rw, rb) in zip(shapes, ins, refs):
        dw = ops.mm_wgrad(a, b, defer=True)            # not inside a backward pass: launched at once, alone
        assert rel(dw, rw) < 2e-5, ("alone", rows, N, K)
    assert not W.jobs
    # an illegal job (K % 128 != 0) is declined by the queue and served by the immediate path
    dy, x = q(gen(1000, 64, seed=3), dt), q(gen(1000, 96, seed=4), dt)
    n0 = W.queued
    dw = ops.mm_wgrad(dy.to(d, dt), x.to(d, dt), defer=True)
    assert W.queued == n0 and rel(dw, (dy.double().t() @ x.double()).float()) < 2e-5


def test_deferred_weight_gradients_match_immediate_ones_through_autograd():
    """ops.WgradQueue at model level: the same dinounet_s train step (256 x 256, bf16: 2688 query rows per step = 21 K-tile pairs, legal for
    the grouped launch) with the queue on and off -- every gradient equal up to the summation order of the split-K partials; the queue
    is empty after backward(); a gradient that already exists (accumulation over two backward calls) and a tensor hook on a weight keep
    a product out of the queue; two forwards summed into one backward (two contributions per parameter in the pass) stay correct."""
    from dinounet_amd import ops
    from dinounet_amd.dinov3.adapter import DropPath
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.training import dc_and_ce_loss
    from oracle import weights
    from oracle.refshim import PLANS_2D
    dev()
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name="dinounet_s", precision="bf16")
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(weights.make_state_dict(ks, seed=0), strict=True)
    net = net.cuda().train()
    for m in net.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
    x = weights.make_input(2, 3, 256, 256, seed=9).cuda()
    t = weights.make_target(2, 256, 256, 2, seed=9).cuda()
    W = ops.WGRAD

    def grads(enabled, pre=None):
        W.enabled = enabled
        net.zero_grad(set_to_none=True)
        q0, l0 = W.queued, W.launches
        if pre is not None:
            pre()
        dc_and_ce_loss(net(x), t).backward()
        assert not W.jobs and not W.keep and W._armed_task is None and not W.state and not W.groups
        torch.cuda.synchronize()
        return {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}, W.queued - q0, W.launches - l0

    try:
        g_off, nq_off, _ = grads(False)
        g_on, nq_on, nl_on = grads(True)
        assert nq_off == 0 and nq_on >= 20 and 1 <= nl_on <= 4, (nq_off, nq_on, nl_on)
        gmax = max(float(v.norm()) for v in g_off.values())
        assert set(g_on) == set(g_off)
        for k in g_off:
            assert float((g_on[k] - g_off[k]).norm()) <= 1e-4 * max(float(g_off[k].norm()), 1e-3 * gmax), k
        # a hook on one weight: that product is computed at once, the others are still queued
        w = net.encoder.dinov3_adapter.interactions[0].extractor.attn.value_proj.weight
        seen = []
        h = w.register_hook(lambda g: seen.append(float(g.abs().sum())))
        g_h, nq_h, _ = grads(True)
        h.remove()
        assert nq_h == nq_on - 1 and seen and seen[0] > 0
        for k in g_off:
            assert float((g_h[k] - g_off[k]).norm()) <= 1e-4 * max(float(g_off[k].norm()), 1e-3 * gmax), k
        # gradient accumulation over two backward calls: the second call finds p.grad set and must add complete gradients
        W.enabled = True
        net.zero_grad(set_to_none=True)
        dc_and_ce_loss(net(x), t).backward()
        q1 = W.queued
        dc_and_ce_loss(net(x), t).backward()
        assert W.queued == q1                                 # nothing queued by the second call
        torch.cuda.synchronize()
        for k, p in net.named_parameters():
            if p.grad is not None:
                assert float((p.grad.float().cpu() - 2 * g_off[k]).norm()) <= 2e-4 * max(float(2 * g_off[k].norm()), 1e-3 * gmax), k
        # two forwards, one backward: every parameter receives two contributions in the pass -- the queue is flushed before the second
        # one is handed to the engine (which then adds the two buffers)
        net.zero_grad(set_to_none=True)
        (dc_and_ce_loss(net(x), t) + dc_and_ce_loss(net(x), t)).backward()
        assert not W.jobs and not W.state
        torch.cuda.synchronize()
        for k, p in net.named_parameters():
            if p.grad is not None:
                assert float((p.grad.float().cpu() - 2 * g_off[k]).norm()) <= 2e-4 * max(float(2 * g_off[k].norm()), 1e-3 * gmax), k
    finally:
        W.enabled = True


@pytest.mark.parametrize("dt", DTS)
def test_linear_autograd_with_residual_and_droppath(dt):
    from dinounet_amd import ops
    d = dev()
    B, T, K, N = 3, 84, 96, 64
    x, w, b, res = q(gen(B, T, K, seed=1), dt), gen(N, K, seed=2, scale=K ** -0.5), gen(N, seed=3), q(gen(B, T, N, seed=4), dt)
    mask = torch.tensor([0.0, 1 / 0.7, 1 / 0.7])
    go = q(gen(B, T, N, seed=5), dt)
    xr, wr, br, rr = (t.clone().requires_grad_(True) for t in (x, q(w, dt), b, res))
    yr = F.linear(xr, wr, br) * mask.view(-1, 1, 1) + rr
    gr = torch.autograd.grad(yr, (xr, wr, br, rr), go)
    xg, wg, bg, rg = x.to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True), res.to(d, dt).requires_grad_(True)
    y = ops.linear(xg, wg, bg, residual=rg, row_scale=mask.to(d), rs_rows=T)
    gg = torch.autograd.grad(y, (xg, wg, bg, rg), go.to(d, dt))
    assert rel(y, yr) < TOL[dt]
    for a, r_ in zip(gg, gr):
        assert rel(a, r_) < TOL[dt]


def test_linear_droppath_scale_inside_backward_gemms():
    """DropPath in the backward of a linear layer (ConvFFN fc2, dinov3_adapter.py:148): from the second step on (W^T packed) the per-sample
    scale is applied inside the GEMMs -- output rows of the data gradient, contraction rows of the weight gradient with dropped samples'
    K tiles skipped, bias gradient from the scaled fragments -- instead of materialising s . dy.  Same gradients as the reference and as
    the first step, which still builds the scaled copy."""
    from dinounet_amd import ops
    d = dev()
    bf = torch.bfloat16
    B, T, K, N = 4, 1280, 256, 192
    x, res = q(gen(B, T, K, seed=1), bf), q(gen(B, T, N, seed=4), bf)
    w = torch.nn.Parameter(gen(N, K, seed=2, scale=K ** -0.5).to(d))
    b = torch.nn.Parameter(gen(N, seed=3).to(d))
    mask = torch.tensor([0.0, 1 / 0.7, 1 / 0.7, 0.0])
    go = q(gen(B, T, N, seed=5), bf)
    xr, wr, br, rr = x.clone().requires_grad_(True), q(w.detach().cpu(), bf).requires_grad_(True), b.detach().cpu().clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = F.linear(xr, wr, br) * mask.view(-1, 1, 1) + rr
    gr = torch.autograd.grad(yr, (xr, wr, br, rr), go)

    def step():
        xg, rg = x.to(d, bf).requires_grad_(True), res.to(d, bf).requires_grad_(True)
        ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
        try:
            y = ops.linear(xg, w, b, residual=rg, row_scale=mask.to(d), rs_rows=T)
            g = torch.autograd.grad(y, (xg, w, b, rg), go.to(d, bf))
        finally:
            ops.TRACK_ROUTE = False
        return y.float().cpu(), [t.float().cpu() for t in g]

    ops.PACK.refresh()
    _, g0 = step()               # registers W^T: data gradient on the PLAIN_COL kernel, s . dy materialised
    ops.PACK.refresh()
    nq = ops.WGRAD.queued
    y1, g1 = step()
    # the weight gradient went through the grouped launch as one job per sample, scaled by the sample's factor on the device (dropped
    # samples return at once): ops.mm_wgrad(k_scale=..., defer=True)
    assert ops.WGRAD.queued == nq + 1 and not ops.WGRAD.jobs
    assert (ops.PLAIN_ROW, ops.PLAIN_COL) not in [(am, bm) for am, bm, _ in ops.ROUTES]
    assert rel(y1, yr) < TOL[bf]
    for a, b0, r_ in zip(g1, g0, gr):
        assert rel(a, r_) < TOL[bf]
        assert rel(a, b0) < 1e-2
    assert float(g1[0][0].abs().max()) == 0.0 and float(g1[0][3].abs().max()) == 0.0       # dropped samples: exactly zero input gradient


# ------------------------------------------------------------------------------------------------ conv
def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cfg", [dict(B=2, H=24, W=16, Cin=64, Cout=32, s=1), dict(B=1, H=32, W=32, Cin=8, Cout=64, s=2),
                                 dict(B=2, H=17, W=9, Cin=128, Cout=256, s=2), dict(B=1, H=64, W=64, Cin=32, Cout=32, s=1),
                                 # round 5: the SPM's low-resolution stride-2 layers (K = 9 Cin >= 1024 on a small tile grid) take the split-K
                                 # form of the implicit GEMM (fp32 partial sums + a cast) in forward and data gradient
                                 dict(B=2, H=32, W=32, Cin=256, Cout=256, s=2, nobias=True), dict(B=8, H=32, W=32, Cin=128, Cout=256, s=2, nobias=True)])
def test_conv3x3_fwd_bwd(dt, cfg):
    from dinounet_amd import ops
    d = dev()
    B, H, W, Cin, Cout, s = (cfg[k] for k in ("B", "H", "W", "Cin", "Cout", "s"))
    x, w, b = q(gen(B, Cin, H, W, seed=1), dt), gen(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5), gen(Cout, seed=3)
    if cfg.get("nobias"):                         # the SPM's convolutions are bias-free (dinov3_adapter.py:243-270)
        b = torch.zeros_like(b)
    xr, wr, br = x.clone().requires_grad_(True), q(w, dt).requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, s, 1)
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, wr, br), go)
    xg, wg, bg = nhwc(x).to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    if cfg.get("nobias"):
        y = ops.conv2d(xg, wg, None, stride=s, pad=1)
        gg = torch.autograd.grad(y, (xg, wg), nhwc(go).to(d, dt))
    else:
        y = ops.conv2d(xg, wg, bg, stride=s, pad=1)
        gg = torch.autograd.grad(y, (xg, wg, bg), nhwc(go).to(d, dt))
    assert rel(y.permute(0, 3, 1, 2), yr) < TOL[dt]
    assert rel(gg[0].permute(0, 3, 1, 2), gr[0]) < TOL[dt]
    assert rel(gg[1], gr[1]) < TOL[dt]
    if not cfg.get("nobias"):
        assert rel(gg[2], gr[2]) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_conv3x3_fused_concat(dt):
    """decoder stage: conv over cat(up, skip) read through two pointers (dinounet_training.py:614)."""
    from dinounet_amd import ops
    d = dev()
    B, H, W, C1, C2, Cout = 2, 32, 32, 32, 32, 32
    a, s2 = q(gen(B, C1, H, W, seed=1), dt), q(gen(B, C2, H, W, seed=2), dt)
    w, b = gen(Cout, C1 + C2, 3, 3, seed=3, scale=(9 * (C1 + C2)) ** -0.5), gen(Cout, seed=4)
    ar, sr, wr = a.clone().requires_grad_(True), s2.clone().requires_grad_(True), q(w, dt).requires_grad_(True)
    yr = F.conv2d(torch.cat([ar, sr], 1), wr, b, 1, 1)
    go = q(gen(*yr.shape, seed=5), dt)
    gr = torch.autograd.grad(yr, (ar, sr, wr), go)
    ag, sg, wg = nhwc(a).to(d, dt).requires_grad_(True), nhwc(s2).to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True)
    y = ops.conv2d(ag, wg, b.to(d), 1, 1, x2=sg)
    gg = torch.autograd.grad(y, (ag, sg, wg), nhwc(go).to(d, dt))
    assert rel(y.permute(0, 3, 1, 2), yr) < TOL[dt]
    assert rel(gg[0].permute(0, 3, 1, 2), gr[0]) < TOL[dt]
    assert rel(gg[1].permute(0, 3, 1, 2), gr[1]) < TOL[dt]
    assert rel(gg[2], gr[2]) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 256, 128), (1, 8, 12, 32, 32), (2, 4, 4, 384, 384)])
def test_conv_transpose2x2_fwd_bwd(dt, B, H, W, Cin, Cout):
    from dinounet_amd import ops
    d = dev()
    x, w, b = q(gen(B, Cin, H, W, seed=1), dt), gen(Cin, Cout, 2, 2, seed=2, scale=Cin ** -0.5), gen(Cout, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), q(w, dt).requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, wr, br), go)
    xg, wg, bg = nhwc(x).to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    y = ops.conv_transpose2x2(xg, wg, bg)
    gg = torch.autograd.grad(y, (xg, wg, bg), nhwc(go).to(d, dt))
    assert rel(y.permute(0, 3, 1, 2), yr) < TOL[dt]
    assert rel(gg[0].permute(0, 3, 1, 2), gr[0]) < TOL[dt]
    assert rel(gg[1], gr[1]) < TOL[dt]
    assert rel(gg[2], gr[2]) < TOL[dt]


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(8, 16, 16, 256, 256), (4, 16, 32, 384, 512)])
def test_conv_transpose2x2_bwd_on_gather_kernels(B, H, W, Cin, Cout):
    """ConvTranspose2d k2 s2 backward on the multi-phase kernels' gather forms (gemm_p8.hip): the data gradient reads the 2 x 2 patch
    rows of dy in place (NT, A gathered), the weight gradient its columns (TN, B gathered).  Forced on at a size the CPU reference
    finishes quickly (du_set_option key 0 = 1: wherever legal); compared with conv_transpose2d's autograd and with the im2col kernels."""
    from dinounet_amd import _lib, ops
    d = dev()
    dt = torch.bfloat16
    x, w = q(gen(B, Cin, H, W, seed=1), dt), gen(Cin, Cout, 2, 2, seed=2, scale=Cin ** -0.5)
    xr, wr = x.clone().requires_grad_(True), q(w, dt).requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, stride=2)
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, wr), go)
    L = _lib.lib()
    res = {}
    try:
        for mode in (1, 0):
            L.du_set_option(0, mode)
            L.du_set_option(5, 2 if mode else 0)
            xg, wg = nhwc(x).to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True)
            y = ops.conv_transpose2x2(xg, wg, None)
            ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
            res[mode] = [t.float().cpu() for t in torch.autograd.grad(y, (xg, wg), nhwc(go).to(d, dt))]
            ops.TRACK_ROUTE = False
            routes = {(am, bm): r for am, bm, r in ops.ROUTES}
            if mode == 1:
                assert routes[(ops.IM2COL_ROW, ops.PLAIN_ROW)] == 3 and routes[(ops.PLAIN_COL, ops.IM2COL_COL)] == 5, routes
            else:
                assert routes[(ops.IM2COL_ROW, ops.PLAIN_ROW)] == 1 and routes[(ops.PLAIN_COL, ops.IM2COL_COL)] == 1, routes
    finally:
        ops.TRACK_ROUTE = False
        L.du_set_option(0, -1)
        L.du_set_option(5, 1)
    for mode in (1, 0):
        assert rel(res[mode][0].permute(0, 3, 1, 2), gr[0]) < TOL[dt], mode
        assert rel(res[mode][1], gr[1]) < TOL[dt], mode
    assert rel(res[1][0], res[0][0]) < 1e-2         # same bf16 products, fp32 sums in another order, one bf16 rounding
    assert rel(res[1][1], res[0][1]) < 1e-4


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 64, 64, 1024, 1024), (1, 32, 64, 384, 256), (3, 16, 32, 512, 128)])
def test_conv_transpose2x2_residual_on_the_persistent_kernel(B, H, W, Cin, Cout):
    """Round 6: ConvTranspose2d k2 s2 + skip add (dinov3_adapter.py:360,467: `up(c2) + c1`) on the persistent multi-phase kernel -- pixel-shuffle
    store from the drain, the residual read through the same pixel mapping as two more K-steps of the tile (gemm_nt_pp_kernel<.., PS>).
    Forced where the cost model would not choose it (du_set_option key 0 = 4), against conv_transpose2d + add in fp32 of the same bf16
    operands, against the one-shot kernel (key 14 = 0), and repeat-run identical."""
    from dinounet_amd import _lib, ops
    d = dev()
    dt = torch.bfloat16
    x, w, b = q(gen(B, Cin, H, W, seed=1), dt), q(gen(Cin, Cout, 2, 2, seed=2, scale=Cin ** -0.5), dt), gen(Cout, seed=3)
    res = q(gen(B, Cout, 2 * H, 2 * W, seed=5), dt)
    ref = F.conv_transpose2d(x.float(), w.float(), b, stride=2) + res.float()
    xg, rg = nhwc(x).to(d, dt), nhwc(res).to(d, dt)
    L = _lib.lib()
    outs = {}
    try:
        for name, mode, k14 in (("pp", 4, 1), ("pp2", 4, 1), ("oneshot", 1, 0)):
            L.du_set_option(0, mode)
            L.du_set_option(14, k14)
            ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
            with torch.no_grad():
                outs[name] = ops.conv_transpose2x2(xg, w.to(d), b.to(d), residual=rg).float().cpu()
            ops.TRACK_ROUTE = False
            assert ops.LAST_GEMM_ROUTE == (6 if mode == 4 else 3), (name, ops.LAST_GEMM_ROUTE)
    finally:
        ops.TRACK_ROUTE = False
        L.du_set_option(0, -1)
        L.du_set_option(14, 1)
    assert rel(outs["pp"].permute(0, 3, 1, 2), ref) < TOL[dt]
    assert torch.equal(outs["pp"], outs["pp2"]), "run-to-run difference: pipeline race"
    assert rel(outs["pp"], outs["oneshot"]) < 1e-2           # same bf16 products, another summation order, one bf16 rounding


@pytest.mark.parametrize("B,H,W,Cin,Cout,bias", [(2, 16, 16, 256, 128, True), (2, 16, 32, 32, 32, True), (4, 16, 16, 384, 384, False),
                                                 (2, 32, 32, 64, 32, True), (1, 32, 64, 24, 40, True)])
def test_conv_transpose2x2_weight_gradient_on_the_grouped_launch(B, H, W, Cin, Cout, bias):
    """ConvTranspose2d k2 s2 weight (and bias) gradient as a queued job of du_gemm_tn_group (du_tn_job.gather = 2): dy gathered in place
    for ANY channel count (an N tile may span several taps: per-lane tap offsets), result written straight in the parameter's
    (Cin, Cout, 2, 2) layout.  Weights are nn.Parameters (only those are queued) and the gradient is taken inside an autograd pass; the
    fp32 sums of exact bf16 products agree with the fp64 reference to round-off, so a wrong tap / column / layout cannot hide."""
    from dinounet_amd import ops
    d = dev()
    dt = torch.bfloat16
    x, w, b = q(gen(B, Cin, H, W, seed=1), dt), q(gen(Cin, Cout, 2, 2, seed=2, scale=Cin ** -0.5), dt), gen(Cout, seed=3)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br if bias else None, stride=2)
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (wr, br) if bias else (wr,), go.double())
    xg = nhwc(x).to(d, dt).requires_grad_(True)
    wg = torch.nn.Parameter(w.to(d))
    bg = torch.nn.Parameter(b.to(d)) if bias else None
    n0 = ops.WGRAD.queued
    y = ops.conv_transpose2x2(xg, wg, bg)
    gg = torch.autograd.grad(y, (wg, bg) if bias else (wg,), nhwc(go).to(d, dt))
    assert ops.WGRAD.queued == n0 + 1 and not ops.WGRAD.jobs
    assert gg[0].shape == wg.shape and gg[0].is_contiguous()
    assert rel(gg[0], gr[0].float()) < 2e-5
    if bias:
        assert rel(gg[1], gr[1].float()) < 2e-5


@pytest.mark.parametrize("B,H,W,C1,C2,Cout,bias", [(1, 64, 64, 128, 128, 128, True), (2, 32, 64, 128, 0, 128, True), (1, 64, 128, 256, 0, 128, False),
                                                   (2, 16, 64, 40, 24, 128, True)])
def test_conv3x3_weight_gradient_on_the_grouped_launch(B, H, W, C1, C2, Cout, bias):
    """3 x 3 / stride 1 / pad 1 weight gradient of the layers the LDS-tiled kernel does not serve (128 output channels: the first U-Net
    decoder stage, dinounet_training.py:581-592) as queued jobs of du_gemm_tn_group (gather = 3): x gathered in place with per-lane tap
    offsets, zero padding by a border bit mask, one job per source of the fused channel concat, (Cout, Cin, 3, 3) written directly."""
    from dinounet_amd import ops
    d = dev()
    dt = torch.bfloat16
    Cin = C1 + C2
    x = q(gen(B, H, W, C1, seed=31), dt)
    x2 = q(gen(B, H, W, C2, seed=32), dt) if C2 else None
    w = q(gen(Cout, Cin, 3, 3, seed=33, scale=0.1), dt)
    bv = gen(Cout, seed=34, scale=0.1)
    go = q(gen(B, H, W, Cout, seed=35), dt)
    xin = torch.cat([x, x2], -1) if C2 else x
    wr, br = w.double().requires_grad_(True), bv.double().requires_grad_(True)
    yr = F.conv2d(xin.double().permute(0, 3, 1, 2), wr, br if bias else None, 1, 1).permute(0, 2, 3, 1)
    gr = torch.autograd.grad(yr, (wr, br) if bias else (wr,), go.double())
    xg = x.to(d, dt).requires_grad_(True)
    x2g = x2.to(d, dt).requires_grad_(True) if C2 else None
    wg = torch.nn.Parameter(w.to(d))
    bg = torch.nn.Parameter(bv.to(d)) if bias else None
    n0 = ops.WGRAD.queued
    from dinounet_amd import _lib
    y = ops.conv2d(xg, wg, bg, 1, 1, x2g)
    gg = torch.autograd.grad(y, (wg, bg) if bias else (wg,), go.to(d, dt))
    assert ops.WGRAD.queued == n0 + 1 and not ops.WGRAD.jobs
    assert gg[0].shape == wg.shape and gg[0].is_contiguous()
    assert rel(gg[0].flatten(1), gr[0].float().flatten(1)) < 2e-5
    if bias:
        assert rel(gg[1], gr[1].float()) < 2e-5
    # round 6: the same layer through autograd on the rows kernel (du_set_option(13, 2), opt-in): same gradients
    wg2 = torch.nn.Parameter(w.to(d))
    bg2 = torch.nn.Parameter(bv.to(d)) if bias else None
    xg2, x2g2 = x.to(d, dt).requires_grad_(True), (x2.to(d, dt).requires_grad_(True) if C2 else None)
    try:
        _lib.lib().du_set_option(13, 2)
        g2 = torch.autograd.grad(ops.conv2d(xg2, wg2, bg2, 1, 1, x2g2), (wg2, bg2) if bias else (wg2,), go.to(d, dt))
    finally:
        _lib.lib().du_set_option(13, 1)
    if C1 % 32 == 0 and C2 % 32 == 0 and H % 8 == 0 and W % 16 == 0:
        assert ops.WGRAD.queued == n0 + 1                       # not queued: computed at once by du_conv3x3_wgrad_halo
    assert rel(g2[0].flatten(1), gr[0].float().flatten(1)) < 2e-5
    if bias:
        assert rel(g2[1], gr[1].float()) < 2e-5


@pytest.mark.parametrize("dt", DTS)
def test_conv_transpose2x2_fused_residual(dt):
    """ConvTranspose + skip add in the epilogue (dinov3_adapter.py:467): the residual has the OUTPUT (pixel-shuffled) layout."""
    from dinounet_amd import ops
    d = dev()
    B, H, W, Cin, Cout = 2, 8, 12, 64, 64
    x, w, b = q(gen(B, Cin, H, W, seed=1), dt), gen(Cin, Cout, 2, 2, seed=2, scale=Cin ** -0.5), gen(Cout, seed=3)
    res = q(gen(B, Cout, 2 * H, 2 * W, seed=5), dt)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, q(w, dt), b, stride=2) + rr
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, rr), go)
    xg, rg = nhwc(x).to(d, dt).requires_grad_(True), nhwc(res).to(d, dt).requires_grad_(True)
    y = ops.conv_transpose2x2(xg, w.to(d), b.to(d), residual=rg)
    gg = torch.autograd.grad(y, (xg, rg), nhwc(go).to(d, dt))
    assert rel(y.permute(0, 3, 1, 2), yr) < TOL[dt]
    assert rel(gg[0].permute(0, 3, 1, 2), gr[0]) < TOL[dt]
    assert rel(gg[1].permute(0, 3, 1, 2), gr[1]) < 1e-6


def test_weight_pack_matches_torch_packing():
    """du_pack_weights (one launch per step) must reproduce every torch-side packing it replaces, and go stale when a source changes."""
    from dinounet_amd import ops
    d = dev()
    bf = torch.bfloat16
    P = ops.WeightPack()
    lin = torch.nn.Parameter(gen(192, 96, seed=1).to(d))
    lin2 = torch.nn.Parameter(gen(64, 96, seed=2).to(d))
    b1, b2 = torch.nn.Parameter(gen(192, seed=3).to(d)), torch.nn.Parameter(gen(64, seed=4).to(d))
    cw = torch.nn.Parameter(gen(32, 24, 3, 3, seed=5).to(d))
    stem = torch.nn.Parameter(gen(16, 3, 3, 3, seed=6).to(d))
    tw = torch.nn.Parameter(gen(40, 24, 2, 2, seed=7).to(d))
    c11 = torch.nn.Parameter(gen(48, 32, 1, 1, seed=8).to(d))
    reqs = [((lin,), ops.PK_CAST, bf, 0, lambda: lin.to(bf)),
            ((lin, lin2), ops.PK_CAST, bf, 0, lambda: torch.cat([lin, lin2], 0).to(bf)),
            ((b1, b2), ops.PK_CAST, torch.float32, 0, lambda: torch.cat([b1, b2], 0)),
            ((lin,), ops.PK_TRANSPOSE, bf, 0, lambda: lin.t().contiguous().to(bf)),
            ((lin, lin2), ops.PK_TRANSPOSE, bf, 0, lambda: torch.cat([lin, lin2], 0).t().contiguous().to(bf)),
            ((c11.view(48, 32),), ops.PK_CAST, bf, 0, lambda: c11.view(48, 32).to(bf)),
            ((cw,), ops.PK_CONV_FWD, bf, 0, lambda: ops.pack_conv_weight(cw, bf)),
            ((stem,), ops.PK_CONV_FWD, bf, 8, lambda: ops.pack_conv_weight(F.pad(stem, (0, 0, 0, 0, 0, 5)), bf)),
            ((cw,), ops.PK_CONV_DGRAD, bf, 0, lambda: ops.pack_conv_weight_dgrad(cw, bf)),
            ((cw,), ops.PK_CONV_DGRAD_FLIP, bf, 0, lambda: ops.pack_conv_weight_dgrad_flipped(cw, bf)),
            ((tw,), ops.PK_CONVT_FWD, bf, 0, lambda: tw.permute(2, 3, 1, 0).reshape(4 * 24, 40).to(bf)),
            ((tw,), ops.PK_CONVT_DGRAD, bf, 0, lambda: tw.permute(0, 2, 3, 1).reshape(40, 4 * 24).to(bf))]
    for srcs, kind, dt, cp, _ in reqs:
        assert P.get(srcs, kind, dt, cp) is None        # first request only registers
    P.refresh()
    for srcs, kind, dt, cp, ref in reqs:
        got = P.get(srcs, kind, dt, cp)
        assert got is not None and got.dtype == dt
        assert torch.equal(got.float().cpu(), ref().float().cpu()), kind
    with torch.no_grad():
        lin.add_(1.0)                                   # in-place update (optimizer step): packed copy must be declined until refreshed
    assert P.get((lin,), ops.PK_CAST, bf) is None
    P.refresh()
    assert torch.equal(P.get((lin,), ops.PK_CAST, bf).float().cpu(), lin.to(bf).float().cpu())
    assert P.get(torch.cat([lin, lin2], 0), ops.PK_CAST, bf) is None     # temporaries are never packed


def test_packed_transposes_serve_cat_and_fapm_data_gradients():
    """From the second step on the data gradients of linear_cat (MSDeformAttn offsets + weights) and fapm_project (shared + specific
    bases, FiLM generator) read [w1; w2]^T from the weight pack (contraction-contiguous NT products, no PLAIN_COL operand); results
    must match the first, unpacked step's."""
    from dinounet_amd import ops
    d = dev()
    bf = torch.bfloat16
    mk = lambda *s, seed, scale=1.0: torch.nn.Parameter(gen(*s, seed=seed, scale=scale).to(d))
    w1, w2, b1, b2 = mk(128, 256, seed=1, scale=0.06), mk(64, 256, seed=2, scale=0.06), mk(128, seed=3), mk(64, seed=4)
    ws, wp, wf = mk(64, 256, 1, 1, seed=5, scale=0.06), mk(64, 256, 1, 1, seed=6, scale=0.06), mk(128, 64, 1, 1, seed=7, scale=0.12)
    bs, bp, bfm = mk(64, seed=8), mk(64, seed=9), mk(128, seed=10)
    x = gen(2, 16, 24, 256, seed=11).to(d, bf)
    go1, go2 = gen(2 * 16 * 24, 192, seed=12).to(d, bf), gen(2, 16, 24, 64, seed=13).to(d, bf)

    def step():
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
        try:
            y1 = ops.linear_cat(xa.view(-1, 256), w1, w2, b1, b2)
            y2 = ops.fapm_project(xb, ws, wp, bs, bp, wf, bfm)
            g = torch.autograd.grad([y1, y2], [xa, xb, w1, w2, ws, wp, wf], [go1, go2])
        finally:
            ops.TRACK_ROUTE = False
        return [t.float().cpu() for t in g], [(am, bm) for am, bm, _ in ops.ROUTES]

    ops.PACK.refresh()
    g0, r0 = step()                      # registers the packs; runs on torch-side packing + PLAIN_COL data gradients
    ops.PACK.refresh()
    g1, r1 = step()
    assert (ops.PLAIN_ROW, ops.PLAIN_COL) in r0 and (ops.PLAIN_ROW, ops.PLAIN_COL) not in r1, (r0, r1)
    for a, b in zip(g0, g1):
        assert rel(b, a) < 1e-2


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("kind,C,H,W", [("in", 32, 64, 64), ("in", 256, 8, 8), ("bn", 64, 32, 32), ("bn", 384, 16, 16)])
def test_norm_act_fwd_bwd(dt, kind, C, H, W):
    from dinounet_amd import ops
    from dinounet_amd._lib import ACT_LEAKY, ACT_RELU
    d = dev()
    B = 3
    x, w, b = q(gen(B, C, H, W, seed=1) * 1.5 + 0.3, dt), 1 + 0.1 * gen(C, seed=2), 0.1 * gen(C, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    if kind == "in":
        yr = F.leaky_relu(F.instance_norm(xr, None, None, wr, br, True, 0.0, 1e-5), 0.01)
        act = ACT_LEAKY
    else:
        yr = F.relu(F.batch_norm(xr, rm, rv, wr, br, True, 0.1, 1e-5))
        act = ACT_RELU
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, wr, br), go)
    xg, wg, bg = nhwc(x).to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    rmg, rvg = torch.zeros(C, device=d), torch.ones(C, device=d)
    y = ops.norm_act(xg, wg, bg, kind, act, 1e-5, True, rmg, rvg, 0.1, None)
    gg = torch.autograd.grad(y, (xg, wg, bg), nhwc(go).to(d, dt))
    t = TOL[dt] * 2
    assert rel(y.permute(0, 3, 1, 2), yr) < t
    assert rel(gg[0].permute(0, 3, 1, 2), gr[0]) < t
    assert rel(gg[1], gr[1]) < t and rel(gg[2], gr[2]) < t
    if kind == "bn":
        assert rel(rmg, rm) < 1e-3 and rel(rvg, rv) < 1e-3                       # running-stat update (momentum 0.1, unbiased var)
        ye = ops.norm_act(xg, wg, bg, "bn", act, 1e-5, False, rmg, rvg, 0.1, None)   # eval: running statistics
        assert rel(ye.permute(0, 3, 1, 2), F.relu(F.batch_norm(x, rm, rv, w, b, False, 0.0, 1e-5))) < t


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("D", [384, 768, 1024, 4096])
def test_layernorm_fwd_bwd(dt, D):
    from dinounet_amd import ops
    d = dev()
    x, w, b = q(gen(2, 131, D, seed=1) * 2 + 0.5, dt), 1 + 0.1 * gen(D, seed=2), 0.1 * gen(D, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), wr, br, 1e-6)
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, wr, br), go)
    xg, wg, bg = x.to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    y = ops.layer_norm(xg, wg, bg, 1e-6)
    gg = torch.autograd.grad(y, (xg, wg, bg), go.to(d, dt))
    assert rel(y, yr) < TOL[dt]
    for a, r_ in zip(gg, gr):
        assert rel(a, r_) < TOL[dt]
    # fp32 residual stream -> bf16 GEMM input (ViT path)
    y2, _, _ = ops.layernorm_raw(x.to(d).view(-1, D), w.to(d), b.to(d), 1e-5, dt)
    assert rel(y2, F.layer_norm(x, (D,), w, b, 1e-5).view(-1, D)) < TOL[dt]


@pytest.mark.parametrize("rows,D,idt,odt", [(8232, 1024, "f32", "bf16"), (8233, 768, "f32", "bf16"), (9001, 384, "bf16", "bf16"), (8200, 1024, "f32", "f32")])
def test_layernorm_fwd_two_rows_per_wave_is_bit_identical(rows, D, idt, odt):
    """du_layernorm_fwd with two rows per wave (the ViT's 8232 rows overflow one round of waves by forty, `du_set_option(18, .)`): the same
    per-row arithmetic as the one-row kernel -- rows, means and rstds identical bit for bit, odd row counts included; against torch."""
    from dinounet_amd import ops, _lib
    d = dev()
    L = _lib.lib()
    it = torch.float32 if idt == "f32" else torch.bfloat16
    ot = torch.float32 if odt == "f32" else torch.bfloat16
    x = (gen(rows, D, seed=1) * 3.0 + 0.5).to(d).to(it)
    w, b = gen(D, seed=2).to(d), gen(D, seed=3).to(d)
    try:
        L.du_set_option(18, 0)
        y0, m0, r0 = ops.layernorm_raw(x, w, b, 1e-6, ot, want_stats=True)
        L.du_set_option(18, 2)
        y2, m2, r2 = ops.layernorm_raw(x, w, b, 1e-6, ot, want_stats=True)
        L.du_set_option(18, 1)
        y1, _, _ = ops.layernorm_raw(x, w, b, 1e-6, ot)
    finally:
        L.du_set_option(18, 1)
    assert torch.equal(y0, y2) and torch.equal(m0, m2) and torch.equal(r0, r2) and torch.equal(y0, y1)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-6)
    tol = 2e-5 if odt == "f32" else 1.6e-2
    assert (y2.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,D,with_res", [(5376, 1024, True), (4100, 256, False), (2049, 512, True)])
def test_layernorm_bwd_many_rows(dt, rows, D, with_res):
    """Many narrow rows (the adapter's 43008 x 1024 query tokens, scaled down): dx, the two-stage dw / db column reduction over hundreds of
    row strips and the fused residual-branch gradient vs torch autograd.  (A one-pass backward that took the dw / db partials from the dx
    kernel's own reads was measured no faster in the step -- 34.07 vs 34.00 ms -- and removed.)"""
    from dinounet_amd import ops
    d = dev()
    x, w, b = q(gen(rows, D, seed=1) * 1.5 + 0.3, dt), 1 + 0.1 * gen(D, seed=2), 0.1 * gen(D, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), wr, br, 1e-6)
    outr = yr * 2.0 + xr if with_res else yr            # a branch that by-passes the norm: its gradient arrives as dres
    go = q(gen(rows, D, seed=4), dt)
    gr = torch.autograd.grad(outr, (xr, wr, br), go)
    xg, wg, bg = x.to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    if with_res:
        y, xres = ops.layer_norm_res(xg, wg, bg, 1e-6)
        out = y * 2.0 + xres
    else:
        out = ops.layer_norm(xg, wg, bg, 1e-6)
    gg = torch.autograd.grad(out, (xg, wg, bg), go.to(d, dt))
    tol = TOL[dt] if dt == torch.bfloat16 else 5e-4     # fp32: sums over thousands of rows in another order
    for a, r_ in zip(gg, gr):
        assert rel(a, r_) < tol


# ------------------------------------------------------------------------------------------------ MSDA
@pytest.mark.parametrize("tag", ["D2", "D4", "D12", "D24", "D30", "D32", "D64", "D71", "D128", "D1025", "D2048", "D3096", "border"])
def test_msda_reference_fixture(tag):
    """ops/test.py fixture (seed 3) + out-of-range sampling: forward and all three gradients vs the reference's own fp64
    values (generated by oracle/make_golden.py from ms_deform_attn_core_pytorch), through the drop-in extension module."""
    import MultiScaleDeformableAttention as MSDA
    d = dev()
    g = np.load(os.path.join(GOLD, "msda_testpy.npz"))
    shapes, lsi = torch.from_numpy(g["shapes"]).to(d), torch.from_numpy(g["level_start_index"]).to(d)
    v, loc, a, go = (torch.from_numpy(g[f"{tag}_{n}"]).float().to(d) for n in ("value", "loc", "attn", "grad_out"))
    out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, a, 2)
    assert rel(out, torch.from_numpy(g[f"{tag}_out"])) < 1e-5
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, a, go, 2)
    assert rel(gv, torch.from_numpy(g[f"{tag}_grad_value"])) < 1e-5
    assert rel(gl, torch.from_numpy(g[f"{tag}_grad_loc"])) < 2e-4      # fp32 cancellation in the corner differences
    assert rel(ga, torch.from_numpy(g[f"{tag}_grad_attn"])) < 1e-5
    # the reference test's own element-wise criterion (ops/test.py:80: allclose rtol 1e-2 atol 1e-3), 10x tighter, on every output
    for got, name in ((out, "out"), (gv, "grad_value"), (gl, "grad_loc"), (ga, "grad_attn")):
        want = torch.from_numpy(g[f"{tag}_{name}"])
        ok, msg = close(got, want, 1e-3, 1e-4 * max(1.0, float(want.abs().max())))
        assert ok, f"{name}: {msg}"


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Dh", [12, 32])
def test_msda_production_shape(dt, Dh):
    """Lq = 5376 queries over a 32x32 value map, 16 heads, 4 points (dinounet_s / dinounet_l head widths), vs the oracle."""
    from dinounet_amd import ops
    from oracle import dinounet_oracle as O
    d = dev()
    N, S, M, Lq, P = 2, 1024, 16, 5376, 4
    v = q(gen(N, S, M, Dh, seed=1), dt)
    loc = torch.rand(N, Lq, M, 1, P, 2, generator=torch.Generator().manual_seed(2)) * 1.1 - 0.05
    a = torch.softmax(gen(N, Lq, M, 1, P, seed=3), -1)
    go = q(gen(N, Lq, M * Dh, seed=4), dt)
    ref = O.msda_core(v, [(32, 32)], loc, a)
    gvr, glr, gar = O.msda_backward(v, [(32, 32)], loc, a, go)
    shapes, lsi = torch.tensor([[32, 32]], device=d), torch.zeros(1, dtype=torch.long, device=d)
    vg, lg, ag = v.to(d, dt).requires_grad_(True), loc.to(d).requires_grad_(True), a.to(d).requires_grad_(True)
    out = ops.msda(vg, shapes, lsi, lg, ag)
    gv, gl, ga = torch.autograd.grad(out, (vg, lg, ag), go.to(d, dt))
    t = 1e-4 if dt == torch.float32 else 2e-2
    assert rel(out, ref) < t and rel(gv, gvr) < t and rel(ga, gar) < t and rel(gl, glr) < 5 * t


@pytest.mark.parametrize("N,Hs,Ws,Lq,Dh", [(1, 6, 4, 126, 24), (3, 20, 36, 700, 32), (1, 32, 32, 100, 8)])
def test_msda_bf16_mfma_grad_value_odd_planes(N, Hs, Ws, Lq, Dh):
    """bf16 backward = gather kernel + grad_value-as-GEMM on the MFMA pipe: non-square planes, planes larger than one 512-pixel
    tile, ragged query counts, samples outside the plane (zero padding)."""
    from dinounet_amd import ops
    from oracle import dinounet_oracle as O
    d = dev()
    dt = torch.bfloat16
    S, M, P = Hs * Ws, 16, 4
    v = q(gen(N, S, M, Dh, seed=11), dt)
    loc = torch.rand(N, Lq, M, 1, P, 2, generator=torch.Generator().manual_seed(12)) * 1.3 - 0.15
    a = torch.softmax(gen(N, Lq, M, 1, P, seed=13), -1)
    go = q(gen(N, Lq, M * Dh, seed=14), dt)
    gvr, glr, gar = O.msda_backward(v, [(Hs, Ws)], loc, a, go)
    shapes, lsi = torch.tensor([[Hs, Ws]], device=d), torch.zeros(1, dtype=torch.long, device=d)
    vg, lg, ag = v.to(d, dt).requires_grad_(True), loc.to(d).requires_grad_(True), a.to(d).requires_grad_(True)
    out = ops.msda(vg, shapes, lsi, lg, ag)
    assert rel(out, O.msda_core(v, [(Hs, Ws)], loc, a)) < 2e-2        # forward from the LDS-resident plane when D/4 is a power of two
    gv, gl, ga = torch.autograd.grad(out, (vg, lg, ag), go.to(d, dt))
    assert rel(gv, gvr) < 2e-2 and rel(ga, gar) < 2e-2 and rel(gl, glr) < 1e-1


@pytest.mark.parametrize("N,Hs,Ws,Lq,Dh,spread", [(2, 32, 32, 2048, 32, 0.04), (1, 40, 28, 900, 32, 0.10), (2, 64, 64, 1100, 16, 0.02)])
def test_msda_bf16_grad_value_skips_only_blocks_that_cannot_reach_a_tile(N, Hs, Ws, Lq, Dh, spread):
    """Round 3: the grad_value kernel walks only the 32-query blocks whose samples can reach its 512-pixel tile (msda_gv_rows_kernel).
    Localised sampling as in the adapter (a query samples around its own reference point, raster order) makes most blocks skippable;
    some queries are sent far outside the plane (blocks with NO valid sample), some far from their reference point (blocks reaching
    several tiles).  Results must match the oracle as with any other locations."""
    from dinounet_amd import ops
    from oracle import dinounet_oracle as O
    d = dev()
    dt = torch.bfloat16
    S, M, P = Hs * Ws, 16, 4
    g = torch.Generator().manual_seed(21)
    v = q(gen(N, S, M, Dh, seed=22), dt)
    ref_y = ((torch.arange(Lq) * Hs * Ws // Lq) // Ws).float().add(0.5) / Hs          # raster-ordered reference points
    ref_x = ((torch.arange(Lq) * Hs * Ws // Lq) % Ws).float().add(0.5) / Ws
    loc = torch.stack([ref_x, ref_y], -1).view(1, Lq, 1, 1, 1, 2) + (torch.rand(N, Lq, M, 1, P, 2, generator=g) - 0.5) * 2 * spread
    loc[:, 64:128] = 3.0                                                               # two whole blocks outside the plane
    loc[:, 300:305] = torch.rand(N, 5, M, 1, P, 2, generator=g)                        # a few queries that sample anywhere
    a = torch.softmax(gen(N, Lq, M, 1, P, seed=23), -1)
    go = q(gen(N, Lq, M * Dh, seed=24), dt)
    gvr, glr, gar = O.msda_backward(v, [(Hs, Ws)], loc, a, go)
    shapes, lsi = torch.tensor([[Hs, Ws]], device=d), torch.zeros(1, dtype=torch.long, device=d)
    vg, lg, ag = v.to(d, dt).requires_grad_(True), loc.to(d).requires_grad_(True), a.to(d).requires_grad_(True)
    out = ops.msda(vg, shapes, lsi, lg, ag)
    gv, gl, ga = torch.autograd.grad(out, (vg, lg, ag), go.to(d, dt))
    assert rel(gv, gvr) < 2e-2 and rel(ga, gar) < 2e-2 and rel(gl, glr) < 1e-1
    # every plane pixel no sample reaches keeps an exactly zero gradient (a skipped block must not leave stale products behind)
    assert bool((gv.float().cpu()[gvr == 0] == 0).all())


def test_offsets_and_prep_as_one_autograd_node_match_the_two_node_form():
    """Round 6: ops.offsets_prep (the offsets | weights product + msda_prep in ONE node, the fp32 matrix between them internal, its gradient
    written in bf16 by du_msda_prep_bwd) against ops.linear_cat(out_dtype=fp32) + ops.msda_prep: same forward bits, gradients equal to the
    rounding of one bf16 cast (the two-node form casts the fp32 gradient, the fused one rounds inside the kernel: the same values)."""
    from dinounet_amd import ops
    d = dev()
    bf = torch.bfloat16
    N, Lq, Cc, M, P = 2, 672, 256, 16, 4
    x = q(gen(N, Lq, Cc, seed=1), bf).to(d, bf)
    w1, w2 = gen(M * P * 2, Cc, seed=2, scale=0.05).to(d), gen(M * P, Cc, seed=3, scale=0.05).to(d)
    b1, b2 = gen(M * P * 2, seed=4).to(d), gen(M * P, seed=5).to(d)
    ref = torch.rand(Lq, 2, generator=torch.Generator().manual_seed(6)).to(d)
    gl, ga = gen(N * Lq, M, P, 2, seed=7).to(d), gen(N * Lq, M, P, seed=8).to(d)
    outs = []
    for fused in (True, False):
        xs = x.clone().requires_grad_(True)
        ps = [t.clone().requires_grad_(True) for t in (w1, w2, b1, b2)]
        if fused:
            loc, at = ops.offsets_prep(xs, ps[0], ps[1], ps[2], ps[3], ref, Lq, M, P, 4, 8)
        else:
            raw = ops.linear_cat(xs, ps[0], ps[1], ps[2], ps[3], out_dtype=torch.float32).view(N * Lq, M * P * 3)
            loc, at = ops.msda_prep(raw, ref, Lq, M, P, 4, 8)
        g = torch.autograd.grad((loc * gl).sum() + (at * ga).sum(), [xs] + ps)
        outs.append((loc.detach(), at.detach(), [t.float() for t in g]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert rel(a, b) < 1e-5, (a.shape, rel(a, b))


def test_msda_prep_fwd_bwd():
    from dinounet_amd import ops
    d = dev()
    rows, Lq, M, P = 2 * 84, 84, 16, 4
    raw = gen(rows, M * P * 3, seed=1)
    ref = torch.rand(Lq, 2, generator=torch.Generator().manual_seed(2))
    rr = raw.clone().requires_grad_(True)
    off = rr[:, :M * P * 2].view(rows, M, P, 2)
    loc_r = ref.repeat(2, 1).view(rows, 1, 1, 2) + off / torch.tensor([8.0, 4.0])
    at_r = F.softmax(rr[:, M * P * 2:].view(rows, M, P), -1)
    g1, g2 = gen(rows, M, P, 2, seed=3), gen(rows, M, P, seed=4)
    gr = torch.autograd.grad([loc_r, at_r], rr, [g1, g2])[0]
    rg = raw.to(d).requires_grad_(True)
    loc, at = ops.msda_prep(rg, ref.to(d), Lq, M, P, 4, 8)
    gg = torch.autograd.grad([loc, at], rg, [g1.to(d), g2.to(d)])[0]
    assert rel(loc, loc_r) < 1e-6 and rel(at, at_r) < 1e-5 and rel(gg, gr) < 1e-5


# ------------------------------------------------------------------------------------------------ small NHWC ops
@pytest.mark.parametrize("dt", DTS)
def test_dwconv_tokens_and_nhwc(dt):
    from dinounet_amd import ops
    from dinounet_amd._lib import ACT_GELU, ACT_NONE
    d = dev()
    B, H, W, C = 2, 4, 6, 96
    n = H * W // 4
    N = 21 * n
    x, w, b = q(gen(B, N, C, seed=1), dt), gen(C, 1, 3, 3, seed=2, scale=1 / 3), gen(C, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    outs = []
    for (lo, hi, h, ww) in ((0, 16 * n, 2 * H, 2 * W), (16 * n, 20 * n, H, W), (20 * n, N, H // 2, W // 2)):
        t = xr[:, lo:hi].transpose(1, 2).reshape(B, C, h, ww)
        outs.append(F.conv2d(t, wr, br, 1, 1, groups=C).flatten(2).transpose(1, 2))
    yr = F.gelu(torch.cat(outs, 1))
    go = q(gen(*yr.shape, seed=4), dt)
    gr = torch.autograd.grad(yr, (xr, wr, br), go)
    xg, wg, bg = x.to(d, dt).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    y = ops.dwconv_tokens(xg, wg, bg, H, W, ACT_GELU)
    gg = torch.autograd.grad(y, (xg, wg, bg), go.to(d, dt))
    assert rel(y, yr) < TOL[dt]
    for a, r_ in zip(gg, gr):
        assert rel(a, r_) < TOL[dt]
    x4 = q(gen(B, C, 9, 7, seed=5), dt)
    y4 = ops.dwconv3x3(nhwc(x4).to(d, dt), w.to(d), b.to(d), ACT_NONE)
    assert rel(y4.permute(0, 3, 1, 2), F.conv2d(x4, w, b, 1, 1, groups=C)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_maxpool_and_bilinear(dt):
    from dinounet_amd import ops
    d = dev()
    x = q(F.relu(gen(2, 64, 17, 22, seed=1)), dt)          # ReLU output: many exact ties at 0, like the SPM stem
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    go = q(gen(*yr.shape, seed=2), dt)
    gr = torch.autograd.grad(yr, xr, go)[0]
    xg = nhwc(x).to(d, dt).requires_grad_(True)
    y = ops.maxpool3x3s2(xg)
    gg = torch.autograd.grad(y, xg, nhwc(go).to(d, dt))[0]
    assert rel(y.permute(0, 3, 1, 2), yr) < 1e-6
    assert rel(gg.permute(0, 3, 1, 2), gr) < TOL[dt]
    for (hs, ws, ho, wo) in ((4, 4, 16, 16), (4, 6, 8, 12), (4, 4, 4, 4), (8, 8, 4, 4)):
        src, base = q(gen(2, 32, hs, ws, seed=3), dt), q(gen(2, 32, ho, wo, seed=4), dt)
        out = ops.bilinear_add(nhwc(src).to(d, dt), nhwc(base).to(d, dt))
        ref = base + F.interpolate(src, size=(ho, wo), mode="bilinear", align_corners=False)
        assert rel(out.permute(0, 3, 1, 2), ref) < TOL[dt]
    # plain resize with its data gradient (tail of LearnableUpsampleBlock, dinounet_training.py:262-263): between 1x and 2x, exact 2x,
    # non-square, a mild downscale
    for (hs, ws, ho, wo) in ((16, 16, 25, 25), (16, 24, 31, 40), (8, 8, 16, 16), (12, 10, 12, 17), (16, 16, 13, 11)):
        src = q(gen(2, 32, hs, ws, seed=5), dt)
        sr = src.clone().requires_grad_(True)
        yr = F.interpolate(sr, size=(ho, wo), mode="bilinear", align_corners=False)
        go = q(gen(*yr.shape, seed=6), dt)
        gr = torch.autograd.grad(yr, sr, go)[0]
        sg = nhwc(src).to(d, dt).requires_grad_(True)
        y = ops.bilinear_resize(sg, (ho, wo))
        gg = torch.autograd.grad(y, sg, nhwc(go).to(d, dt))[0]
        assert rel(y.permute(0, 3, 1, 2), yr) < TOL[dt], (hs, ws, ho, wo)
        assert rel(gg.permute(0, 3, 1, 2), gr) < TOL[dt], (hs, ws, ho, wo)


def test_learnable_upsample_block_bilinear_tail():
    """LearnableUpsampleBlock (dinounet_training.py:249-264) with a target that is not a power-of-two multiple of the input: transpose
    convolutions while a doubling still fits, then the bilinear resize -- forward and gradients vs the torch composition."""
    from dinounet_amd.network_architecture.dinounet import LearnableUpsampleBlock
    d = dev()
    blk = LearnableUpsampleBlock(32).to(d)
    with torch.no_grad():
        blk.up2.weight.copy_(gen(32, 32, 2, 2, seed=1, scale=0.2)); blk.up2.bias.copy_(gen(32, seed=2))
    x = gen(2, 32, 6, 5, seed=3)
    xr = x.clone().requires_grad_(True)
    w, b = blk.up2.weight.detach().cpu().clone().requires_grad_(True), blk.up2.bias.detach().cpu().clone().requires_grad_(True)
    t = F.conv_transpose2d(F.conv_transpose2d(xr, w, b, stride=2), w, b, stride=2)          # 24 x 20: one more doubling would exceed 29 x 33
    yr = F.interpolate(t, size=(29, 33), mode="bilinear", align_corners=False)
    go = gen(*yr.shape, seed=4)
    gr = torch.autograd.grad(yr, (xr, w, b), go)
    xg = nhwc(x).to(d).requires_grad_(True)
    y = blk(xg, (29, 33))
    gg = torch.autograd.grad(y, (xg, blk.up2.weight, blk.up2.bias), nhwc(go).to(d))
    assert rel(y.permute(0, 3, 1, 2), yr) < 2e-4
    assert rel(gg[0].permute(0, 3, 1, 2), gr[0]) < 2e-4
    assert rel(gg[1], gr[1]) < 2e-4 and rel(gg[2], gr[2]) < 2e-4


def test_layout_helpers():
    from dinounet_amd import ops
    d = dev()
    x = gen(2, 3, 32, 48, seed=1)
    y = ops.nchw_to_nhwc(x.to(d), torch.float32, 8)
    assert rel(y[..., :3], x.permute(0, 2, 3, 1)) < 1e-7 and float(y[..., 3:].abs().max()) == 0.0
    assert rel(ops.nhwc_to_nchw_f32(y[..., :3]), x) < 1e-7
    p = ops.patchify16(x.to(d), torch.float32)
    assert rel(p, F.unfold(x, 16, stride=16).transpose(1, 2).reshape(-1, 768)) < 1e-7


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,H,N,Dh", [(2, 6, 1029, 64), (1, 3, 21, 64), (1, 2, 260, 128)])
def test_vit_attention(dt, B, H, N, Dh):
    """RoPE + head split + softmax(QK^T)V vs torch SDPA in fp32 (layers/attention.py:66-118), 5 prefix tokens un-rotated."""
    from dinounet_amd import ops
    d = dev()
    prefix = 5
    qkv = q(gen(B * N, 3 * H * Dh, seed=1), dt)
    ang = gen(N - prefix, Dh, seed=2)
    sin, cos = torch.sin(ang), torch.cos(ang)
    qq, kk, vv = [t.transpose(1, 2) for t in qkv.view(B, N, 3, H, Dh).unbind(2)]

    def rope(t):
        a = t[:, :, prefix:]
        x1, x2 = a.chunk(2, -1)
        return torch.cat([t[:, :, :prefix], a * cos + torch.cat([-x2, x1], -1) * sin], 2)

    ref = F.scaled_dot_product_attention(rope(qq), rope(kk), vv).transpose(1, 2).reshape(B * N, H * Dh)
    out = ops.attention(qkv.to(d, dt), sin.to(d), cos.to(d), B, N, H, Dh, prefix, {})
    assert rel(out, ref) < (2e-4 if dt == torch.float32 else 3e-2)


def _attn_raw(q_, k_, v_, B, H, N, Dh):
    import ctypes as C
    from dinounet_amd import _lib
    Npad = q_.shape[2]
    out = torch.zeros(B * N, H * Dh, dtype=torch.bfloat16, device=q_.device)
    _lib.check(_lib.lib().du_attention_fwd(C.c_void_p(q_.data_ptr()), C.c_void_p(k_.data_ptr()), C.c_void_p(v_.data_ptr()), C.c_void_p(out.data_ptr()),
                                          B, H, N, Npad, Dh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "du_attention_fwd")
    return out


def _attn_ref(q_, k_, v_, B, H, N, Dh):
    import math
    s = torch.einsum("bhqd,bhkd->bhqk", q_[:, :, :N].double().cpu(), k_[:, :, :N].double().cpu()) * math.log(2.0)      # q carries log2(e): base-2 softmax
    return torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v_[:, :, :N].double().cpu()).permute(0, 2, 1, 3).reshape(B * N, H * Dh).float()


# every attention kernel of the library: (du_set_option(6, impl), du_set_option(8, variant)): the product kernel (32 queries per wave, no
# running maximum, nearly empty query tiles first), the round-3 kernel (d_head 128's product kernel), and the measured-and-kept experiments
# (64 queries per wave; its slot-pipelined form)
ATTN_KERNELS = [(0, 0, "product"), (1, 0, "round 3"), (0, 1, "64 q per wave"), (0, 9, "slot-pipelined")]


@pytest.mark.parametrize("impl,var,name", ATTN_KERNELS)
@pytest.mark.parametrize("Dh", [64, 128])
def test_attention_spiked_keys_and_forced_rescale(impl, var, name, Dh):
    """A rare data-dependent branch needs its own test: keys whose score lies 100 - 600 (log2 units) above everything the kernel has seen
    so far -- in either half-wave, in the first (ragged) tile and later ones, one and several per query block.  The kernels without a running
    maximum must take their cold rescale path (the probabilities overflow fp32 otherwise); the round-2 / 3 kernel once took its running
    maximum from the lower half-wave's keys only (hipcc folded the two results of permlane32_swap) and returned NaN on exactly these inputs.
    Reference: fp64 softmax of the same bf16 operands.  Also: with the threshold of the cold path turned down to zero (every tile takes
    it = a true running maximum) the result stays within rounding."""
    from dinounet_amd import _lib
    L = _lib.lib()
    d = dev()
    B, H, N = 1, 2, 1029
    Npad = (N + 7) // 8 * 8
    g = torch.Generator().manual_seed(5)
    cases = [[(5, 100, 300.0)], [(5, 104, 300.0)], [(5, 100, 300.0), (9, 700, 300.0)], [(5, 100, 140.0), (40, 708, 600.0), (70, 1026, 250.0)],
             [(5, 10, 100.0), (5, 500, 300.0)], [(1027, 64, 200.0), (300, 1028, 200.0)]]
    L.du_set_option(6, impl); L.du_set_option(8, var)
    try:
        for spikes in cases:
            q_ = (torch.randn(B, H, Npad, Dh, generator=g) * 0.05).to(d, torch.bfloat16)
            k_ = (torch.randn(B, H, Npad, Dh, generator=g) * 0.05).to(d, torch.bfloat16)
            v_ = torch.randn(B, H, Npad, Dh, generator=g).to(d, torch.bfloat16)
            for h in range(H):
                for qrow, key, val in spikes:
                    q_[0, h, qrow] = 0; q_[0, h, qrow, (qrow + h) % Dh] = 1.0
                for qrow, key, val in spikes:
                    k_[0, h, key, (qrow + h) % Dh] = val
            ref = _attn_ref(q_, k_, v_, B, H, N, Dh)
            out = _attn_raw(q_, k_, v_, B, H, N, Dh)
            assert bool(torch.isfinite(out.float()).all()), (name, spikes)
            assert rel(out, ref) < 2e-2, (name, spikes)
            if impl == 0 and Dh == 64:
                L.du_set_option(7, -2000)
                try:
                    out0 = _attn_raw(q_, k_, v_, B, H, N, Dh)
                finally:
                    L.du_set_option(7, 60)
                assert rel(out0, ref) < 2e-2, (name, spikes)
    finally:
        L.du_set_option(6, 0); L.du_set_option(8, 0)


@pytest.mark.parametrize("impl,var,name", ATTN_KERNELS)
@pytest.mark.parametrize("B,H,N,Dh", [(2, 8, 1029, 64), (1, 8, 64, 64), (1, 8, 65, 64), (1, 8, 7, 64), (2, 4, 1024, 64), (1, 8, 300, 128), (1, 3, 129, 64), (2, 16, 261, 64)])
def test_attention_kernels_vs_fp64_softmax(impl, var, name, B, H, N, Dh):
    """softmax(Q K^T) V of bf16 operands vs fp64, for every kernel and ragged / tiny / exact token counts; deterministic across runs."""
    import math
    from dinounet_amd import _lib
    L = _lib.lib()
    d = dev()
    Npad = (N + 7) // 8 * 8
    g = torch.Generator().manual_seed(N + Dh)
    q_ = (torch.randn(B, H, Npad, Dh, generator=g) * Dh ** -0.5 * math.log2(math.e)).to(d, torch.bfloat16)
    k_ = torch.randn(B, H, Npad, Dh, generator=g).to(d, torch.bfloat16)
    v_ = torch.randn(B, H, Npad, Dh, generator=g).to(d, torch.bfloat16)
    ref = _attn_ref(q_, k_, v_, B, H, N, Dh)
    L.du_set_option(6, impl); L.du_set_option(8, var)
    try:
        out = _attn_raw(q_, k_, v_, B, H, N, Dh)
        again = _attn_raw(q_, k_, v_, B, H, N, Dh)
    finally:
        L.du_set_option(6, 0); L.du_set_option(8, 0)
    assert rel(out, ref) < 2e-2, name
    assert torch.equal(out, again), name


@pytest.mark.parametrize("B,H,N,D", [(8, 16, 1029, 1024), (2, 6, 1029, 384), (3, 12, 261, 768)])
def test_vit_qkv_rope_in_gemm_epilogue(B, H, N, D):
    """qkv projection with RoPE + q scale + head-major store in the multi-phase GEMM's epilogue (DU_STORE_QKV_ROPE, rows past the last
    256-row tile through du_qkv_rope_split_rows) vs the separate product + du_qkv_rope_split: identical q / k / v planes (same fp32 values,
    same single bf16 rounding ... of the rotated value instead of rounding the projection first: compared at bf16 resolution), and the
    attention output vs torch SDPA on the fp32 projection."""
    from dinounet_amd import ops
    d = dev()
    bf = torch.bfloat16
    Dh, prefix = 64, 5
    h = q(gen(B * N, D, seed=1), bf)
    w = q(gen(3 * H * Dh, D, seed=2, scale=D ** -0.5), bf)
    bias = gen(3 * H * Dh, seed=3, scale=0.1)
    ang = gen(N - prefix, Dh, seed=4)
    sin, cos = torch.sin(ang), torch.cos(ang)
    hd, wd, bd, sd, cd = h.to(d, bf), w.to(d, bf), bias.to(d), sin.to(d).contiguous(), cos.to(d).contiguous()
    ws1, ws2 = {}, {}
    ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
    fused_default = ops._QKV_FUSED
    ops._QKV_FUSED = True                  # opt-in path (measured neutral in the step, see ops.py)
    try:
        out_f = ops.qkv_attention(hd, wd, bd, sd, cd, B, N, H, Dh, prefix, ws1)
    finally:
        ops.TRACK_ROUTE = False
        ops._QKV_FUSED = fused_default
    assert 4 in [r for _, _, r in ops.ROUTES], ops.ROUTES                 # the 256 x 128 multi-phase kernel took the fused store
    out_u = ops.attention(ops.mm(hd, wd, bias=bd), sd, cd, B, N, H, Dh, prefix, ws2)
    (k1,), (k2,) = ws1.keys(), ws2.keys()
    for name, a, b in zip("qkv", ws1[k1], ws2[k2]):
        assert rel(a, b) < 1.2e-2, name          # <= 1 bf16 ulp apart: the unfused path rounds the projection to bf16 before rotating it
    assert rel(out_f, out_u) < 2e-2          # the 1-ulp q / k / v differences through the softmax
    qkv = h @ w.t() + bias
    qq, kk, vv = [t.transpose(1, 2) for t in qkv.view(B, N, 3, H, Dh).unbind(2)]

    def rope(t):
        a = t[:, :, prefix:]
        x1, x2 = a.chunk(2, -1)
        return torch.cat([t[:, :, :prefix], a * cos + torch.cat([-x2, x1], -1) * sin], 2)

    ref = F.scaled_dot_product_attention(rope(qq), rope(kk), vv).transpose(1, 2).reshape(B * N, H * Dh)
    assert rel(out_f, ref) < 3e-2
    assert rel(out_u, ref) < 3e-2


@pytest.mark.parametrize("B,H,N,D", [(8, 16, 1029, 1024), (2, 6, 1029, 384), (3, 12, 261, 768), (9, 16, 1029, 1024), (1, 16, 1024, 1024)])
def test_vit_qkv_head_major_store_and_in_place_rope(B, H, N, D):
    """Round 6 (the shipping ViT path, x1.0041 in the step): the persistent kernel's drain stores q / k / v head-major (DU_STORE_QKV_HEADS: the
    plain drain with another row offset) and du_qkv_rope_inplace rotates q and k where they lie -- against the plain product +
    du_qkv_rope_split: v bit-identical (no arithmetic), q / k equal to the contraction order of one fused multiply-add (same bf16 rounding
    of the projection, the same fp32 rotation), prefix rows, the rows behind the last full 256-row tile (du_qkv_rope_split_rows), tiles that
    straddle samples (B = 9), no ragged rows at all (N = 1024)."""
    from dinounet_amd import ops, _lib
    d = dev()
    bf = torch.bfloat16
    Dh, prefix = 64, 5
    h = q(gen(B * N, D, seed=1), bf)
    w = q(gen(3 * H * Dh, D, seed=2, scale=D ** -0.5), bf)
    bias = gen(3 * H * Dh, seed=3, scale=0.1)
    ang = gen(N - prefix, Dh, seed=4)
    sin, cos = torch.sin(ang), torch.cos(ang)
    hd, wd, bd, sd, cd = h.to(d, bf), w.to(d, bf), bias.to(d), sin.to(d).contiguous(), cos.to(d).contiguous()
    ws1, ws2 = {}, {}
    heads_default = ops._QKV_HEADS
    ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
    try:
        ops._QKV_HEADS = True
        _lib.lib().du_set_option(0, 4)       # (small shapes: the cost model is not asked -- the store mode exists on the persistent kernel only)
        out_f = ops.qkv_attention(hd, wd, bd, sd, cd, B, N, H, Dh, prefix, ws1)
    finally:
        ops.TRACK_ROUTE = False
        ops._QKV_HEADS = heads_default
        _lib.lib().du_set_option(0, -1)
    assert 6 in [r for _, _, r in ops.ROUTES], ops.ROUTES
    out_u = ops.attention(ops.mm(hd, wd, bias=bd), sd, cd, B, N, H, Dh, prefix, ws2)
    (k1,), (k2,) = ws1.keys(), ws2.keys()
    if (B, N, D) == (8, 1029, 1024):       # both forms run this product on the persistent kernel (same K order): v is stored, not computed
        assert torch.equal(ws1[k1][2][:, :, :N], ws2[k2][2][:, :, :N])
    for name, a, b in zip("qkv", ws1[k1], ws2[k2]):
        assert rel(a[:, :, :N], b[:, :, :N]) < 4e-3, name                    # <= 1 bf16 ulp on a few elements (fma contraction / K order)
    assert rel(out_f, out_u) < 1e-2


@pytest.mark.parametrize("B,H,hp,wp,D", [(8, 16, 32, 32, 1024), (2, 6, 32, 32, 384), (3, 12, 16, 24, 768), (9, 16, 32, 32, 1024)])
def test_vit_qkv_rope_in_the_persistent_kernels_drain(B, H, hp, wp, D):
    """Round 6: RoPE + q scale + head-major store in the DRAIN of the persistent kernel (gemm_nt_pp_kernel<.., ROPE>): the rotation of
    (token, d) comes from a factorised table in LDS -- dimensions 0..15 of a head follow the token's row, 16..31 its column, 32..63 repeat
    them (rope_position_encoding.py:98-104) -- so the tables here are built the reference's way (coords / periods, tiled twice, with
    per-axis jitter so that the two axes differ).  Against the separate product + du_qkv_rope_split (planes equal to one bf16 ulp: the
    unfused path rounds the projection first) and torch SDPA on the fp32 projection; prefix tokens unrotated; v untouched; rows behind
    the last full 256-row tile through du_qkv_rope_split_rows; B = 9: tiles that straddle samples."""
    import math
    from dinounet_amd import ops
    d = dev()
    bf = torch.bfloat16
    Dh, prefix = 64, 5
    N = prefix + hp * wp
    h = q(gen(B * N, D, seed=1), bf)
    w = q(gen(3 * H * Dh, D, seed=2, scale=D ** -0.5), bf)
    bias = gen(3 * H * Dh, seed=3, scale=0.1)
    periods = 100.0 ** (2 * torch.arange(Dh // 4, dtype=torch.float32) / (Dh // 2))
    ch = (torch.arange(0.5, hp) / hp * 2 - 1) * 1.37 + 0.11           # (a jitter / shift per axis, as in training)
    cw = (torch.arange(0.5, wp) / wp * 2 - 1) * 0.71 - 0.23
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), -1).flatten(0, 1)          # [HW, 2]
    ang = (2 * math.pi * coords[:, :, None] / periods[None, None, :]).flatten(1, 2).tile(2)   # [HW, 64]
    sin, cos = torch.sin(ang), torch.cos(ang)
    hd, wd, bd, sd, cd = h.to(d, bf), w.to(d, bf), bias.to(d), sin.to(d).contiguous(), cos.to(d).contiguous()
    ws1, ws2 = {}, {}
    from dinounet_amd import _lib
    ops.TRACK_ROUTE, ops.ROUTES[:] = True, []
    drain_default = ops._QKV_ROPE_DRAIN
    try:
        ops._QKV_ROPE_DRAIN = True           # opt-in path (measured x0.995 in the step, see ops.py)
        _lib.lib().du_set_option(0, 4)       # (the small shapes have fewer than two tiles per CU: the cost model would keep them off the persistent kernel)
        out_f = ops.qkv_attention(hd, wd, bd, sd, cd, B, N, H, Dh, prefix, ws1, grid=(hp, wp))
        ops.TRACK_ROUTE = False
        out_f2 = ops.qkv_attention(hd, wd, bd, sd, cd, B, N, H, Dh, prefix, {}, grid=(hp, wp))
    finally:
        ops.TRACK_ROUTE = False
        ops._QKV_ROPE_DRAIN = drain_default
        _lib.lib().du_set_option(0, -1)
    assert 6 in [r for _, _, r in ops.ROUTES], ops.ROUTES                 # the persistent kernel took the fused store
    assert torch.equal(out_f, out_f2), "run-to-run difference"
    out_u = ops.attention(ops.mm(hd, wd, bias=bd), sd, cd, B, N, H, Dh, prefix, ws2)
    (k1,), (k2,) = ws1.keys(), ws2.keys()
    for name, a, b in zip("qkv", ws1[k1], ws2[k2]):
        assert rel(a[:, :, :N], b[:, :, :N]) < 1.2e-2, name          # <= 1 bf16 ulp apart
    assert rel(out_f, out_u) < 2e-2
    qkv = h @ w.t() + bias
    qq, kk, vv = [t.transpose(1, 2) for t in qkv.view(B, N, 3, H, Dh).unbind(2)]

    def rope(t):
        a = t[:, :, prefix:]
        x1, x2 = a.chunk(2, -1)
        return torch.cat([t[:, :, :prefix], a * cos + torch.cat([-x2, x1], -1) * sin], 2)

    ref = F.scaled_dot_product_attention(rope(qq), rope(kk), vv).transpose(1, 2).reshape(B * N, H * Dh)
    assert rel(out_f, ref) < 3e-2
    # the planes themselves against the fp32 rotation (q carries the softmax scale in log2 units, ops.qkv_attention)
    qs = Dh ** -0.5 * math.log2(math.e)
    for name, plane, want in (("q", ws1[k1][0], rope(qq) * qs), ("k", ws1[k1][1], rope(kk)), ("v", ws1[k1][2], vv)):
        assert rel(plane[:, :, :N].float().cpu(), want) < 6e-3, name


@pytest.mark.parametrize("B,H,W,K,ld", [(2, 24, 40, 2, 32), (1, 17, 13, 1, 32), (3, 8, 16, 3, 32), (2, 33, 9, 4, 32), (2, 16, 16, 2, 64)])
def test_seg_head_streaming_kernels_match_conv1x1(B, H, W, K, ld):
    """The decoder's last layer (32 channels -> K classes, dinounet_training.py:603-629) as one streaming pass: fp32 NCHW logits, and dx /
    dw / db from ONE backward pass, vs torch's conv2d on the same bf16 features with the weights rounded to bf16 (autocast).  Pixel counts
    that are no multiple of the workgroup size, 1..4 classes, features that are the first 32 channels of a wider tensor (ld = 64)."""
    from dinounet_amd import ops
    d = dev()
    dt = torch.bfloat16
    xfull = q(gen(B, H, W, ld, seed=41), dt)
    w, b = gen(K, 32, 1, 1, seed=42, scale=0.3), gen(K, seed=43, scale=0.3)
    go = gen(B, K, H, W, seed=44)
    xr = xfull[..., :32].float().clone().requires_grad_(True)
    wr, br = q(w, dt).clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br)
    yr.backward(go)
    xg_full = xfull.to(d, dt)
    xg = xg_full[..., :32].detach().requires_grad_(True) if ld == 32 else xg_full[..., :32]
    if ld != 32:
        xg_full.requires_grad_(True)
        xg = xg_full[..., :32]
    assert ops.seg_head_ok(xg, K)
    wg, bg = w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    y = ops.seg_head(xg, wg, bg)
    assert y.shape == (B, K, H, W) and y.dtype == torch.float32
    assert rel(y, yr) < 1e-5
    y.backward(go.to(d))
    gx = xg.grad if ld == 32 else xg_full.grad[..., :32]
    assert rel(gx, xr.grad) < TOL[dt]                      # dx is stored in bf16
    assert rel(wg.grad, wr.grad) < 1e-4 and rel(bg.grad, br.grad) < 1e-5
    assert not ops.seg_head_ok(xg.float(), K) and not ops.seg_head_ok(xg, 5)


@pytest.mark.parametrize("B,H,W,C1,C2,Cout", [(2, 16, 32, 64, 0, 32), (1, 24, 16, 32, 32, 32), (2, 8, 16, 64, 64, 64), (1, 16, 16, 128, 128, 128),
                                              (1, 32, 48, 32, 0, 64), (3, 8, 16, 128, 0, 64),
                                              # served by the streaming strip kernel (conv_strip.hip): W % 128 == 0, 32 input channels; segments
                                              # of 8 / 12 / 10 rows, image borders on every side of a strip, 64 outputs = wave pairs
                                              (1, 8, 128, 32, 0, 32), (2, 24, 256, 32, 0, 32), (2, 40, 384, 32, 0, 64), (8, 64, 128, 32, 0, 32),
                                              # ... 64 input channels = two ring planes (one tensor, or the two tensors of a concat)
                                              (1, 16, 128, 64, 0, 32), (2, 24, 256, 32, 32, 32), (2, 40, 128, 64, 0, 64), (3, 64, 128, 32, 32, 64),
                                              # ... and at PRODUCTION size: the start-up hazard of DESIGN 6.49 only showed for
                                              # workgroups dispatched onto a busy chip (second round), i.e. never in the small cases above
                                              (8, 512, 512, 32, 0, 32), (8, 512, 512, 32, 32, 32), (8, 512, 512, 32, 0, 64), (8, 256, 256, 64, 0, 64)])
def test_conv3x3_halo_kernel_fwd_bwd_stats(B, H, W, C1, C2, Cout):
    """LDS-tiled direct conv (bf16): forward (+fused concat), flipped-weight data gradient, weight gradient, and the per-tile channel
    statistics it emits, vs torch conv2d / the separate statistics kernel."""
    from dinounet_amd import ops
    d = dev()
    dt = torch.bfloat16
    Cin = C1 + C2
    x = q(gen(B, H, W, C1, seed=31), dt)
    x2 = q(gen(B, H, W, C2, seed=32), dt) if C2 else None
    w = gen(Cout, Cin, 3, 3, seed=33, scale=0.1)
    bias = gen(Cout, seed=34, scale=0.1)
    go = q(gen(B, H, W, Cout, seed=35), dt)
    xin = torch.cat([x, x2], -1) if C2 else x
    xr = xin.clone().requires_grad_(True)
    wr = q(w, dt).clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, 1, 1).permute(0, 2, 3, 1)
    yr.backward(go)
    xg = x.to(d, dt).requires_grad_(True)
    x2g = x2.to(d, dt).requires_grad_(True) if C2 else None
    wg, bg = w.to(d).requires_grad_(True), bias.to(d).requires_grad_(True)
    y, part = ops.conv2d_stats(xg, wg, bg, 1, 1, x2g)
    assert part is not None, "shape should be served by the halo kernel, statistics included (32-channel outputs too since round 3)"
    y.backward(go.to(d, dt))
    tol = TOL[dt]
    assert rel(y, yr) < tol
    assert rel(wg.grad, wr.grad) < tol and rel(bg.grad, br.grad) < tol
    gx = torch.cat([xg.grad, x2g.grad], -1) if C2 else xg.grad
    assert rel(gx, xr.grad) < tol
    # statistics: finalize the partials and compare with the stand-alone statistics kernel on the same output
    sums_ref, _ = ops.chan_stats(y.detach(), B)
    sums = torch.empty_like(sums_ref)
    from dinounet_amd import _lib
    import ctypes as C
    _lib.check(_lib.lib().du_strip_finalize(C.c_void_p(part.data_ptr()), C.c_void_p(sums.data_ptr()), B, part.shape[0] // B, Cout,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "du_strip_finalize")
    assert rel(sums, sums_ref) < 1e-4


@pytest.mark.parametrize("B,H,W,C1,C2,Cout", [(1, 8, 16, 32, 0, 32), (1, 8, 32, 64, 0, 64), (3, 40, 80, 96, 0, 32), (2, 24, 16, 32, 32, 64),
                                              (5, 64, 64, 64, 64, 64), (2, 16, 48, 64, 0, 32), (7, 8, 16, 32, 0, 64), (2, 128, 256, 64, 0, 64),
                                              (4, 264, 272, 32, 0, 32),
                                              # round 6: 128 output channels (the first decoder stage): one 32-output block per wave, 4 and 8 input chunks,
                                              # the fused concat, more tiles than the 128 workgroups
                                              (1, 8, 16, 128, 0, 128), (2, 64, 64, 128, 128, 128), (3, 128, 128, 128, 0, 128), (1, 16, 32, 96, 32, 128)])
@pytest.mark.parametrize("with_db", [False, True])
def test_conv3x3_weight_gradient_rows_kernel(B, H, W, C1, C2, Cout, with_db):
    """The round-5 3 x 3 weight-gradient kernel (conv3x3_wgrad_rows_kernel: a wave owns all 9 taps of a 32 x 32 channel block pair, the
    waves split the tile's rows when there are fewer than 4 pairs and are added through LDS, two register sets of prefetch) against the
    fp32 weight gradient of the same bf16 operands and against the round-3 kernel (du_set_option(13, 0)): one tile, fewer tiles than
    workgroups, tile counts that are no multiple of the grid, 1-3 channel chunks, the fused concat, the bias-gradient sums."""
    from dinounet_amd import _lib, ops
    d = dev()
    dt = torch.bfloat16
    Cin = C1 + C2
    x = q(gen(B, H, W, C1, seed=41), dt)
    x2 = q(gen(B, H, W, C2, seed=42), dt) if C2 else None
    go = q(gen(B, H, W, Cout, seed=43), dt)
    xin = (torch.cat([x, x2], -1) if C2 else x).float().permute(0, 3, 1, 2)
    wr = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    (F.conv2d(xin, wr, None, 1, 1) * go.float().permute(0, 3, 1, 2)).sum().backward()
    ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    ref_db = go.float().sum((0, 1, 2))
    L = _lib.lib()
    outs = {}
    modes = (2,) if Cout == 128 else (2, 0)        # (the round-3 kernel does not serve 128 outputs)
    try:
        for mode in modes:
            L.du_set_option(13, mode)
            r = ops.conv3x3_wgrad_halo(x.to(d, dt), go.to(d, dt), x2.to(d, dt) if C2 else None, with_db=with_db)
            assert r is not None
            outs[mode] = r if with_db else (r, None)
            r2 = ops.conv3x3_wgrad_halo(x.to(d, dt), go.to(d, dt), x2.to(d, dt) if C2 else None, with_db=with_db)
            assert torch.equal(r2[0] if with_db else r2, outs[mode][0]), "two runs differ"
    finally:
        L.du_set_option(13, 1)
    scale = float(ref.abs().max())
    for mode in modes:
        assert float((outs[mode][0].cpu() - ref).abs().max()) / scale < 2e-5, mode       # fp32 accumulation of exact bf16 products
        if with_db:
            assert float((outs[mode][1].cpu() - ref_db).abs().max()) / float(ref_db.abs().max()) < 1e-5, mode


# ------------------------------------------------------------------------------------------------ squeeze-excitation
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,H,W,C,with_sc", [(2, 16, 16, 32, True), (3, 8, 8, 256, False), (8, 32, 32, 128, True)])
def test_squeeze_excite_fwd_bwd(dt, B, H, W, C, with_sc):
    """SqueezeExcitation + fused residual (dinounet_training.py:210-225,438) vs the torch formula, forward and all gradients."""
    from dinounet_amd import ops
    d = dev()
    R = max(1, C // 16)
    x = q(gen(B, H, W, C, seed=1), dt)
    sc = q(gen(B, H, W, C, seed=2), dt) if with_sc else None
    w1, b1 = gen(R, C, 1, 1, seed=3, scale=0.3), gen(R, seed=4, scale=0.3)
    w2, b2 = gen(C, R, 1, 1, seed=5, scale=0.3), gen(C, seed=6, scale=0.3)
    go = q(gen(B, H, W, C, seed=7), dt)
    ref_in = [t.clone().requires_grad_(True) for t in ([x, w1, b1, w2, b2] + ([sc] if with_sc else []))]
    s = ref_in[0].mean((1, 2))
    s = torch.sigmoid(F.linear(F.relu(F.linear(s, ref_in[1].flatten(1), ref_in[2])), ref_in[3].flatten(1), ref_in[4]))
    yr = ref_in[0] * s[:, None, None, :]
    if with_sc:
        yr = yr + ref_in[5]
    yr.backward(go)
    ins = [t.to(d).requires_grad_(True) for t in (x.to(dt), w1, b1, w2, b2)]
    scd = sc.to(d).to(dt).requires_grad_(True) if with_sc else None
    y = ops.squeeze_excite(*ins, scd)
    y.backward(go.to(d).to(dt))
    tol = TOL[dt]
    assert rel(y, yr) < tol
    for got, want in zip(ins + ([scd] if with_sc else []), ref_in):
        assert rel(got.grad, want.grad) < tol * 2, (got.shape, rel(got.grad, want.grad))


@pytest.mark.parametrize("B,K,H,W", [(2, 2, 64, 64), (3, 4, 48, 40), (8, 2, 512, 512), (1, 8, 32, 32)])
def test_fused_dice_ce_loss(B, K, H, W):
    """Fused DC+CE (loss + d loss / d logits) vs the reference formula (oracle restatement of compound_losses.py / dice.py)."""
    from dinounet_amd import ops
    from oracle import dinounet_oracle as O
    d = dev()
    logits = gen(B, K, H, W, seed=21, scale=2.0)
    tgt = torch.randint(0, K, (B, 1, H, W), generator=torch.Generator().manual_seed(22))
    lr = logits.clone().requires_grad_(True)
    ref = O.dc_and_ce_loss(lr, tgt)
    (ref * 1.7).backward()
    lg = logits.to(d).requires_grad_(True)
    out = ops.dice_ce_loss(lg, tgt.to(d))
    (out * 1.7).backward()
    assert abs(float(out) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    assert rel(lg.grad, lr.grad) < 2e-4


def test_train_step_hipgraph_matches_eager():
    """A hipGraph-captured TrainStep (training.TrainStep, what bench.py times) must reproduce eager steps: same losses over several
    replays, finite gradients (regression test: library reductions/GEMMs inside the captured region went non-finite on the 2nd replay)."""
    import copy
    from oracle import weights
    from oracle.refshim import PLANS_2D
    from dinounet_amd.dinov3.adapter import DropPath
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.training import TrainStep
    d = dev()

    def build():
        net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name="dinounet_s", precision="bf16")
        ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        net.load_state_dict(weights.make_state_dict(ks, seed=0), strict=True)
        net = net.to(d).train()
        for m in net.modules():
            if isinstance(m, DropPath):
                m.drop_prob = 0.0
        net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
        return net

    x = weights.make_input(4, 3, 128, 128, seed=3).to(d)
    tgt = weights.make_target(4, 128, 128, 2, seed=3).to(d)
    losses = {}
    for mode in (False, True):
        net = build()
        params = [p for p in net.parameters() if p.requires_grad]
        if mode:      # the captured step also runs the fused clip + SGD (csrc/optim.hip); the eager one torch's clip_grad_norm_ + SGD
            from dinounet_amd.optim import FusedClipSGD
            opt = FusedClipSGD(params, 1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5)
        else:
            opt = torch.optim.SGD(params, 1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5)
        ts = TrainStep(net, opt, params, x.shape, tgt.shape, d, graph=mode, warmup=2)
        ts(x, tgt)
        ls = [float(ts()) for _ in range(7)]
        torch.cuda.synchronize()
        assert all(np.isfinite(ls)), (mode, ls)
        assert all(torch.isfinite(p.grad).all() for p in params if p.grad is not None)
        losses[mode] = ls
    print("eager", losses[False], "graph", losses[True])
    assert max(abs(a - b) for a, b in zip(losses[False], losses[True])) < 5e-3


# ------------------------------------------------------------------------------------------------ fused clip + SGD
@pytest.mark.parametrize("scale,max_norm", [(1.0, 12.0), (50.0, 12.0), (1.0, None)])
def test_fused_clip_sgd_matches_torch(scale, max_norm):
    """FusedClipSGD == clip_grad_norm_(max_norm) + torch.optim.SGD(momentum 0.99, nesterov, weight decay) over several steps
    (clip active / inactive / disabled), odd tensor sizes, a parameter without gradient, a learning-rate change between steps."""
    from dinounet_amd.optim import FusedClipSGD
    d = dev()
    shapes = [(1024, 256), (37,), (3, 5, 7), (4097,), (256, 64, 3, 3), (1,), (8192,)]
    pa = [torch.nn.Parameter(gen(*s_, seed=10 + i).to(d)) for i, s_ in enumerate(shapes)]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    pa.append(torch.nn.Parameter(torch.ones(5, device=d)))       # never receives a gradient
    pb.append(torch.nn.Parameter(torch.ones(5, device=d)))
    oa = FusedClipSGD(pa, lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True, max_norm=max_norm)
    ob = torch.optim.SGD(pb, lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True)
    for it in range(4):
        for i in range(len(shapes)):
            g = gen(*shapes[i], seed=100 * it + i).to(d) * scale
            pa[i].grad, pb[i].grad = g.clone(), g.clone()
        if it == 2:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 3e-3
        oa.step()
        tn = torch.nn.utils.clip_grad_norm_(pb, max_norm) if max_norm is not None else torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in pb[:-1]]))
        ob.step()
        assert abs(float(oa.total_norm) - float(tn)) < 1e-5 * float(tn)
        for x, y in zip(pa, pb):
            assert rel(x, y) < 1e-6
            if x.grad is not None:
                assert rel(x.grad, y.grad) < 1e-6                     # gradients are left clipped, like clip_grad_norm_
    for x, y in zip(pa[:-1], pb[:-1]):
        assert rel(oa.state[x]["momentum_buffer"], ob.state[y]["momentum_buffer"]) < 1e-6
    assert torch.equal(pa[-1].detach().cpu(), torch.ones(5))


# ------------------------------------------------------------------------------------------------ gradient reducer on the GPU
def test_grad_reducer_single_rank_gather_matches_plain_grads():
    """GradAllReducer on one rank (RCCL world of 1): the per-bucket gather kernel + all-reduce + 1/world must hand back exactly the
    gradients autograd produced, as views of the flat buckets, over two steps (addresses of the incoming gradients change)."""
    import os
    import torch.distributed as dist
    from dinounet_amd.parallel import GradAllReducer
    d = dev()
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
        created = True
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(300, 257), torch.nn.GELU(), torch.nn.Linear(257, 64), torch.nn.Linear(64, 5)).to(d)
        red = GradAllReducer(net, 1, bucket_elems=10000)
        assert len(red.buckets) >= 2
        for step in range(2):
            x = gen(32, 300, seed=step).to(d)
            net.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
            want = [p.grad.clone() for p in net.parameters()]
            # the hooks already gathered + reduced; finish() installs the flat views
            red.finish()
            for p, w in zip(net.parameters(), want):
                assert p.grad.data_ptr() != w.data_ptr() and torch.equal(p.grad, w)
        red.remove()
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ sliding-window inference
@pytest.mark.parametrize("shape,patch,bs", [((3, 2, 160, 200), (128, 128), 5), ((3, 1, 100, 128), (128, 128), 8), ((3, 3, 128, 128), (128, 128), 2)])
def test_sliding_window_inference_matches_oracle(shape, patch, bs):
    """inference.predict_sliding_window_logits (batched windows, HIP accumulate / normalise kernels) against the CPU restatement of the
    reference predictor (oracle/sliding_window_oracle.py: one window per call, torch indexing) driving the SAME network: image larger
    than / smaller than / equal to the patch, ragged last batch of windows."""
    from oracle import sliding_window_oracle as SW
    from oracle import weights
    from dinounet_amd.plans import PLANS_2D
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd import inference as INF
    d = dev()
    net = DinoUNet.from_config(PLANS_2D, 3, 3, dinov3_pretrained_path=None, dinov3_model_name="dinounet_s", precision="fp32")
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(weights.make_state_dict(ks, seed=0), strict=True)
    net = net.to(d).eval()
    data = gen(*shape, seed=7)
    got = INF.predict_sliding_window_logits(net, data, patch, tile_step_size=0.5, use_gaussian=True, batch_size=bs)
    with torch.no_grad():
        want = SW.predict_sliding_window_logits(lambda w: net(w.to(d)).float().cpu(), data, patch, 0.5, True)
    assert got.shape == want.shape == (3, shape[1], shape[2], shape[3])
    assert rel(got, want) < 2e-5
    assert torch.equal(got.argmax(0).cpu(), want.argmax(0)) or float((got.argmax(0).cpu() != want.argmax(0)).float().mean()) < 1e-4
    flat = INF.predict_sliding_window_logits(net, data, patch, tile_step_size=1.0, use_gaussian=False, batch_size=bs)
    with torch.no_grad():
        want2 = SW.predict_sliding_window_logits(lambda w: net(w.to(d)).float().cpu(), data, patch, 1.0, False)
    assert rel(flat, want2) < 2e-5
    captured = INF.predict_sliding_window_logits(net, data, patch, tile_step_size=0.5, use_gaussian=True, batch_size=bs, graph=True)
    assert rel(captured, want) < 2e-5                    # hipGraph-replayed window forward, zero-padded ragged batch
    assert net.training is False
    # test-time mirroring (predict_from_raw_data.py:537-552) over both in-plane axes: 4 forwards per window batch, eager and replayed
    with torch.no_grad():
        want_m = SW.predict_sliding_window_logits(lambda w: net(w.to(d)).float().cpu(), data, patch, 0.5, True, mirror_axes=(0, 1))
    assert rel(want_m, want) > 1e-4                      # the mirrored average is a different prediction
    for g in (False, True):
        got_m = INF.predict_sliding_window_logits(net, data, patch, tile_step_size=0.5, use_gaussian=True, batch_size=bs, graph=g,
                                                  mirror_axes=(0, 1))
        assert rel(got_m, want_m) < 2e-5, g


# ------------------------------------------------------------------------------------------------ GPU augmentation (8(f) rank 4)
def test_augmentation_kernels_vs_numpy_restatement():
    """Every du_aug_* kernel against oracle/augment_oracle.py with explicit parameters: rotation + scale + mirror resampling of image and
    labels, contrast, gamma (plain and inverted, retain_stats), Gaussian blur, low-resolution simulation, brightness; noise statistics."""
    from dinounet_amd.augment import GPUAugment2D
    from oracle import augment_oracle as AO
    d = dev()
    B, Cc, Hi, Wi, H, W = 3, 2, 80, 72, 64, 64
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(Hi, dtype=torch.float32), torch.arange(Wi, dtype=torch.float32), indexing="ij")
    smooth = torch.stack([torch.sin(yy * (0.11 + 0.03 * k)) * torch.cos(xx * (0.07 + 0.02 * k)) for k in range(B * Cc)]).view(B, Cc, Hi, Wi)
    data = (smooth + 0.05 * torch.randn(B, Cc, Hi, Wi, generator=g)).contiguous()
    seg = ((yy // 9 + xx // 11) % 4).float().expand(B, 1, Hi, Wi).contiguous()
    aug = GPUAugment2D((H, W), seed=1)
    ang, sc = [0.6, -2.3, 0.0], [0.8, 1.3, 1.0]
    prm = np.zeros((B, 6), np.float32)
    for b in range(B):
        c, s_ = np.cos(ang[b]), np.sin(ang[b])
        prm[b] = [c * sc[b], -s_ * sc[b], s_ * sc[b], c * sc[b], b % 2, (b + 1) % 2]
    off = dict(noise_sigma=np.zeros((B, Cc), np.float32), blur_sigma=np.zeros((B, Cc), np.float32), mult=np.ones((B, Cc), np.float32),
               contrast=np.ones((B, Cc), np.float32), zoom=np.zeros((B, Cc), np.float32), gamma_inv=np.zeros((B, Cc), np.float32),
               gamma=np.zeros((B, Cc), np.float32), seed=7)
    # spatial + mirror
    out, sout = aug.apply(data.to(d), seg.to(d), dict(off, spatial=prm))
    ro, rs = AO.spatial(data.numpy(), seg.numpy(), prm, (H, W))
    assert rel(out, torch.from_numpy(ro).float()) < 1e-4
    mism = float((sout.cpu() != torch.from_numpy(rs).float()).float().mean())
    print(f"label resampling: {mism:.2e} of the pixels differ from the scipy restatement (fp32 vs fp64 coordinates at exact 0.5 ties)")
    assert mism < 2e-3
    # what batchgenerators itself calls is scipy's cubic B-spline (order 3, prefiltered): report how far the Keys kernel is from it on the
    # smooth image, away from the zero-padded border (there the spline's prefilter and the plain zero padding differ by construction)
    ks, _ = AO.spatial(smooth.numpy(), None, prm, (H, W))
    sp3 = AO.spatial_scipy_order3(smooth.numpy(), prm, (H, W))
    inside = np.zeros((B, 1, H, W), bool)
    for b in range(B):
        yy_, xx_ = AO._coords(prm[b], Hi, Wi, H, W)
        inside[b, 0] = (yy_ > 4) & (yy_ < Hi - 5) & (xx_ > 4) & (xx_ < Wi - 5)
    dev3 = float(np.abs(ks - sp3)[np.broadcast_to(inside, ks.shape)].max())
    print(f"Keys bicubic vs scipy order-3 spline on the smooth test image (amplitude 1), interior pixels: max abs diff {dev3:.4f}")
    assert dev3 < 0.05
    ident = np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), (B, 1))
    base = data[:, :, 8:72, 4:68].contiguous()                                             # identity transform = centre crop
    o0, _ = aug.apply(data.to(d), None, dict(off, spatial=ident))
    assert rel(o0, base) < 1e-6
    x0 = base.numpy().astype(np.float64)
    # brightness, contrast
    mult = np.array([[1.2, 0.8], [1.0, 1.0], [0.9, 1.1]], np.float32)
    o, _ = aug.apply(data.to(d), None, dict(off, spatial=ident, mult=mult))
    assert rel(o, torch.from_numpy(x0 * mult[:, :, None, None]).float()) < 1e-6
    con = np.array([[0.8, 1.0], [1.2, 0.77], [1.0, 1.24]], np.float32)
    o, _ = aug.apply(data.to(d), None, dict(off, spatial=ident, contrast=con))
    assert rel(o, torch.from_numpy(AO.contrast(x0, con)).float()) < 1e-5
    # gamma, plain and inverted, with retain_stats
    gm = np.array([[0.7, 0.0], [1.4, 1.1], [0.0, 0.9]], np.float32)
    for key, inv in (("gamma", False), ("gamma_inv", True)):
        o, _ = aug.apply(data.to(d), None, dict(off, spatial=ident, **{key: gm}))
        r = AO.gamma(x0, gm, inv)
        assert rel(o, torch.from_numpy(r).float()) < 2e-4, key
    # blur, low resolution
    sg = np.array([[0.5, 0.0], [1.0, 0.73], [0.0, 0.9]], np.float32)
    o, _ = aug.apply(data.to(d), None, dict(off, spatial=ident, blur_sigma=sg))
    assert rel(o, torch.from_numpy(AO.blur(x0, sg)).float()) < 1e-5
    zm = np.array([[0.5, 0.0], [0.77, 0.93], [0.0, 0.61]], np.float32)
    o, _ = aug.apply(data.to(d), None, dict(off, spatial=ident, zoom=zm))
    assert rel(o, torch.from_numpy(AO.lowres(x0, zm)).float()) < 1e-5
    # noise: zero-mean, the requested variance, deterministic per seed, off where sigma is 0
    ns = np.array([[0.3, 0.3], [0.0, 0.0], [0.1, 0.1]], np.float32)
    big = torch.zeros(B, Cc, 256, 256)
    aug2 = GPUAugment2D((256, 256), seed=2)
    i256 = dict(off, spatial=ident, noise_sigma=ns)
    n1, _ = aug2.apply(big.to(d), None, i256)
    n2, _ = aug2.apply(big.to(d), None, i256)
    assert torch.equal(n1, n2)
    assert float(n1[1].abs().max()) == 0.0
    for b in (0, 2):
        assert abs(float(n1[b].mean())) < 5e-3 and abs(float(n1[b].std()) - ns[b, 0]) < 0.02 * ns[b, 0]
    assert abs(float((n1[0, 0].flatten()[:-1] * n1[0, 0].flatten()[1:]).mean())) < 2e-3          # neighbouring pixels uncorrelated
    # the full random pipeline runs, keeps shapes / label set, and is reproducible from the seed
    a1, a2 = GPUAugment2D((H, W), seed=5), GPUAugment2D((H, W), seed=5)
    for _ in range(4):
        o1, s1 = a1(data.to(d), seg.to(d))
        o2, s2 = a2(data.to(d), seg.to(d))
        assert torch.equal(o1, o2) and torch.equal(s1, s2)
        assert o1.shape == (B, Cc, H, W) and torch.isfinite(o1).all()
        assert set(s1.unique().tolist()) <= {0.0, 1.0, 2.0, 3.0}


@pytest.mark.parametrize("Cin,Cout,off,wide", [(32, 32, 32, 96), (64, 64, 16, 96), (64, 32, 0, 72), (32, 64, 8, 40)])
def test_conv3x3_strip_kernel_on_channel_slices_of_wider_tensors(Cin, Cout, off, wide):
    """The strip kernel addresses its input through a pixel pitch (ldx) like every conv kernel here: a channel slice of a wider NHWC
    tensor (FAPM's shared / specific halves, the decoder's concat halves) is read in place -- 16-byte aligned slices at pitches that are
    no multiple of the slice width, one tensor feeding both 32-channel ring planes."""
    from dinounet_amd import ops
    d = dev()
    dt = torch.bfloat16
    B, H, W = 2, 24, 256
    xw = q(gen(B, H, W, wide, seed=51), dt).to(d, dt)
    x = xw[..., off:off + Cin]
    assert not x.is_contiguous()
    w = gen(Cout, Cin, 3, 3, seed=52, scale=0.1)
    bias = gen(Cout, seed=53, scale=0.1)
    y, part = ops.conv2d_stats(x, w.to(d), bias.to(d), 1, 1)
    assert part is not None and part.shape[0] == int(__import__("dinounet_amd._lib", fromlist=["lib"]).lib().du_conv3x3_halo_parts(Cin, Cin, Cout, B, H, W))
    yr = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), q(w, dt), bias, 1, 1).permute(0, 2, 3, 1)
    assert rel(y, yr) < TOL[dt]


def test_vit_half_batch_chains_on_side_streams_match_one_chain():
    """Round 5: the frozen ViT runs as two half-batch chains on two side HIP streams (vision_transformer.py:
    begin_intermediate_layers), the caller's stream is free until the join.  Same kernels, same arithmetic per sample: every tap output must
    agree with the one-chain forward up to the rounding of the few ragged rows that take the K-parallel tail kernel in one split and a tile
    kernel in the other -- eagerly, with the caller's stream doing unrelated work between launch and join, AND when the whole thing is
    captured into a hipGraph and replayed (fork / join by stream events inside the capture)."""
    from dinounet_amd.dinov3.vision_transformer import DinoVisionTransformer
    d = dev()
    torch.manual_seed(0)
    vit = DinoVisionTransformer(embed_dim=384, depth=4, num_heads=6).to(d).eval()
    for blk in vit.blocks:                       # LayerScale 1e-5 would hide a wrong branch behind the residual
        blk.ls1.gamma.data.fill_(1.0)
        blk.ls2.gamma.data.fill_(0.5)
    x = gen(4, 3, 256, 256, seed=5).to(d)
    taps = [1, 3]
    vit.chains = 1
    ref = vit.get_intermediate_layers(x, n=taps, dtype=torch.bfloat16)
    vit.chains = 2
    h = vit.begin_intermediate_layers(x, n=taps, dtype=torch.bfloat16)
    junk = torch.randn(4096, 4096, device=d) @ torch.randn(4096, 4096, device=d)      # the caller's stream is busy meanwhile
    got = h()
    torch.cuda.synchronize()
    assert len(vit._chain_ws) == 2                                                    # the chains ran (and own separate q/k/v workspaces)
    for (pr, cr), (pg, cg) in zip(ref, got):
        assert pg.shape == pr.shape and cg.shape == cr.shape
        assert rel(pg, pr) < 2e-2 and rel(cg, cr) < 2e-2
        # the bulk of the rows go through the same tile kernels in the same accumulation order: identical bits for most elements
        assert float((pg == pr).float().mean()) > 0.9
    # captured and replayed
    xs = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        vit.get_intermediate_layers(xs, n=taps, dtype=torch.bfloat16)                # warm-up on a side stream, as TrainStep does
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = vit.get_intermediate_layers(xs, n=taps, dtype=torch.bfloat16)
    x2 = gen(4, 3, 256, 256, seed=6).to(d)
    xs.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    vit.chains = 1
    ref2 = vit.get_intermediate_layers(x2, n=taps, dtype=torch.bfloat16)
    for (pr, cr), (pg, cg) in zip(ref2, cap):
        assert rel(pg, pr) < 2e-2 and rel(cg, cr) < 2e-2
    del junk


@pytest.mark.parametrize("M,N,K,form", [(5000, 96, 64, "nt"), (4100, 288, 32, "nt_relu"), (9000, 1024, 192, "nt"), (4096, 512, 256, "nt"), (6000, 160, 128, "nt_nobias"),
                                        (7000, 256, 32, "dgrad"), (4500, 64, 256, "dgrad"), (131072, 256, 64, "nt")])
def test_gemm_resident_weights_streaming_kernel(M, N, K, form):
    """gemm_nt_rk_kernel (csrc/gemm_rk.hip, round 5): tall bf16 products with K <= 256 -- the weights' slice resident in LDS, A fragments
    from global memory into registers, the result stored from registers.  Forced wherever legal (du_set_option(12, 3); the default rule
    takes products of >= 2^24 output elements): ragged M, column chunks of unequal size, every K, W as [N][K] and as [K][N], bias /
    activation, against the fp32 product of the same bf16 operands and bit for bit against the kernel it replaces."""
    from dinounet_amd import ops, _lib
    from dinounet_amd._lib import ACT_RELU
    d = dev()
    bf = torch.bfloat16
    L = _lib.lib()
    if form == "dgrad":
        dy, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(K, N, seed=2, scale=K ** -0.5), bf).to(d, bf)
        fn = lambda: ops.mm_dgrad(dy, w)
        ref = dy.float() @ w.float()
    else:
        x, w = q(gen(M, K, seed=1), bf).to(d, bf), q(gen(N, K, seed=2, scale=K ** -0.5), bf).to(d, bf)
        b = None if form == "nt_nobias" else gen(N, seed=3).to(d)
        act = ACT_RELU if form == "nt_relu" else 0
        fn = lambda: ops.mm(x, w, bias=b, act=act)
        ref = x.float() @ w.float().t()
        if b is not None:
            ref = ref + b
        if act:
            ref = F.relu(ref)
    try:
        L.du_set_option(12, 0)
        ops.TRACK_ROUTE = True
        y_old = fn().float()
        L.du_set_option(12, 3)
        y_new = fn().float()
        assert ops.LAST_GEMM_ROUTE == 7, ops.LAST_GEMM_ROUTE           # the streaming kernel ran
        again = [fn().float() for _ in range(4)]
    finally:
        ops.TRACK_ROUTE = False
        L.du_set_option(12, 1)
    assert rel(y_new, ref) < TOL[bf]
    assert torch.equal(y_new, y_old)            # same fp32 accumulation order over k, same epilogue arithmetic
    for o in again:
        assert torch.equal(o, y_new)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 64, 64, 32, 32), (1, 96, 64, 64, 32), (4, 32, 32, 128, 64), (5, 40, 24, 64, 64)])
def test_conv_transpose2x2_on_the_streaming_kernel(B, H, W, Ci, Co):
    """ConvTranspose2d k2 s2 (dinounet_training.py:255-264,558) through gemm_nt_rk_kernel: the forward's pixel-shuffle store (a 32-column
    block inside one tap) and the data gradient's 2 x 2 patch gather as the A operand (a k-step inside one tap), forced with
    du_set_option(12, 3), against torch's conv_transpose2d / its gradient in fp32."""
    from dinounet_amd import ops, _lib
    d = dev()
    bf = torch.bfloat16
    L = _lib.lib()
    x = q(gen(B, H, W, Ci, seed=1), bf)
    w, b = gen(Ci, Co, 2, 2, seed=2, scale=Ci ** -0.5), gen(Co, seed=3)
    go = q(gen(B, 2 * H, 2 * W, Co, seed=4), bf)
    xr, wr = x.clone().requires_grad_(True), q(w, bf).clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr.permute(0, 3, 1, 2), wr, b, stride=2).permute(0, 2, 3, 1)
    yr.backward(go)
    xg, wg, bg = x.to(d, bf).requires_grad_(True), w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    try:
        L.du_set_option(12, 3)
        ops.TRACK_ROUTE = True
        ops.ROUTES.clear()
        y = ops.conv_transpose2x2(xg, wg, bg)
        y.backward(go.to(d, bf))
        routes = [r for _, _, r in ops.ROUTES]
    finally:
        ops.TRACK_ROUTE = False
        L.du_set_option(12, 1)
    assert routes.count(7) >= (2 if 4 * Co <= 256 else 1), routes          # forward and data gradient both took the streaming kernel
    assert rel(y, yr) < TOL[bf]
    assert rel(xg.grad, xr.grad) < TOL[bf]
    assert rel(wg.grad, wr.grad) < TOL[bf] and rel(bg.grad, b * 0 + go.float().sum((0, 1, 2))) < TOL[bf]
