"""Host-side operator layer: thin Python wrappers that hand raw device pointers to libdinounet_hip.so
(include/dinounet_hip.h) on the current HIP stream, plus the torch.autograd.Function classes that give the
trainable part of Dino U-Net its backward pass.  PyTorch is used for device memory, streams and autograd
bookkeeping only; there is no CPU path -- a CPU tensor raises.

Layouts: conv-side activations are NHWC tensors (B, H, W, C) (channel-contiguous; a channel slice of a wider
buffer is allowed), token tensors are (B, N, D).  `dt` is the activation dtype (bf16 throughput mode, fp32
parity mode); statistics, MSDA locations/weights, weight gradients and the ViT residual stream are fp32.
"""
import ctypes as C
import math
import os
import weakref

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SWIGLU, DU_BF16, DU_F32, IM2COL_COL, IM2COL_ROW, PLAIN_COL,
                   PLAIN_ROW, STORE_MSDA_PREP, STORE_PIXEL_SHUFFLE2, STORE_QKV_HEADS, STORE_QKV_ROPE, STORE_SLABS, ConvGeom, GemmArgs)

__all__ = ["mm", "linear", "conv2d", "conv_transpose2x2", "norm_act", "layer_norm", "msda", "dwconv3x3",
           "maxpool3x3s2", "bilinear_add", "bilinear_resize", "squeeze_excite", "dice_ce_loss"]


# ----------------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------------
def _code(dt):
    if dt == torch.bfloat16:
        return DU_BF16
    if dt == torch.float32:
        return DU_F32
    raise RuntimeError(f"dinounet_amd: unsupported dtype {dt} (bf16 / fp32 only)")


def _req(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dinounet_amd ops run on the GPU through libdinounet_hip.so only (no CPU fallback); "
                               "got a CPU tensor")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    if t is None:
        return None
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _rows2d(t):
    """(rows, cols) view info of a matrix whose last dim is contiguous: returns (rows, cols, ld)."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.shape[0], t.shape[1], t.stride(0)


def _nhwc(t):
    """NHWC tensor (B,H,W,C) with dense pixels -> (B,H,W,C,ld)."""
    assert t.dim() == 4 and t.stride(3) == 1, (t.shape, t.stride())
    B, H, W, Cc = t.shape
    ld = t.stride(2)
    assert (H == 1 or t.stride(1) == W * ld) and (B == 1 or t.stride(0) == H * W * ld), (t.shape, t.stride())
    return B, H, W, Cc, ld


class KernelProfile:
    """HIP-event timing of individual kernel launches on the stream they are launched on (torch's current stream).
    Enabled by bench.py inside its timed region: ops.PROFILE = KernelProfile()."""

    def __init__(self, detail=False):
        self.rec = []
        self.detail = detail      # key GEMM launches by full shape signature (tools/debug_step.py --shapes)

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, e0, flops, nbytes):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append((name, e0, e1, flops, nbytes))

    def roofline(self, peak_tflops, peak_gbs, steps):
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, fl, nb in self.rec:
            a = agg.setdefault(name, [0.0, 0, 0.0, 0.0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += 1
            a[2] += fl
            a[3] += nb
        rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
        brk = [{"kernel": k, "launches_per_step": round(v[1] / steps, 1), "ms_per_step": round(v[0] / steps * 1e3, 3),
                "tflops": round(v[2] / v[0] / 1e12, 1), "gbs": round(v[3] / v[0] / 1e9, 1)} for k, v in rows[:12]]
        k, v = rows[0]
        mfma_frac = v[2] / v[0] / 1e12 / peak_tflops
        hbm_frac = v[3] / v[0] / 1e9 / peak_gbs
        if mfma_frac >= hbm_frac:
            roof = {"kernel": k, "bound": "mfma", "achieved": round(v[2] / v[0] / 1e12, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                    "frac": round(mfma_frac, 4), "traffic": None, "avg_launch_us": round(v[0] / v[1] * 1e6, 2)}
        else:
            roof = {"kernel": k, "bound": "hbm", "achieved": round(v[3] / v[0] / 1e9, 1), "peak": peak_gbs, "unit": "GB/s",
                    "frac": round(hbm_frac, 4), "traffic": None, "avg_launch_us": round(v[0] / v[1] * 1e6, 2)}
        return roof, brk


PROFILE = None


CAPTURE_CUT = None           # training.TrainStep while it captures a step in segments: callable(eager_fn, what) ending the current graph
SMALL_COLLECTIVES = None     # bench.py (N > 1 / forced reducer): list -> every small all-reduce of an eager step is bracketed by HIP events


# DINOUNET_FORCE_SMALL_COLLECTIVES=1 (test configuration, like DINOUNET_FORCE_REDUCER): the SyncBatchNorm statistics and batch-Dice
# all-reduces are issued on a ONE-rank process group too, where they are identities -- so a single MI355X executes the N > 1 step with every
# collective in it (the whole-step capture records them from autograd's device thread), bit-identical to the ungated step.
_FORCE_SMALL = os.environ.get("DINOUNET_FORCE_SMALL_COLLECTIVES") == "1"


def sync_active(group=None):
    """True when the step's statistics collectives are issued over `group`: more than one rank (nnUNetTrainer.py:216-218,
    convert_sync_batchnorm + DDP; dice.py:58-119 with ddp=True), or forced on a single rank."""
    d = torch.distributed
    if not (d.is_available() and d.is_initialized()):
        return False
    return d.get_world_size(group) > 1 or _FORCE_SMALL


def _small_all_reduce(t, group, what):
    """the step's latency-bound collectives (SyncBatchNorm statistics forward / backward, the batch-Dice sums): <= 2 KB each, on the
    critical path.  With SMALL_COLLECTIVES set (eager steps only) each one is bracketed by events on the compute stream so the bench
    line can say what they cost at N ranks."""
    capturing = t.is_cuda and torch.cuda.is_current_stream_capturing()
    if capturing and CAPTURE_CUT is not None:
        # a step captured in segments (training.TrainStep, DINOUNET_COMM_OUTSIDE_GRAPH=1): the graph ends here, the collective is issued
        # from the host on every replay, the next graph begins
        CAPTURE_CUT(lambda: torch.distributed.all_reduce(t, group=group), what)
        return
    rec = SMALL_COLLECTIVES is not None and t.is_cuda and not capturing
    if rec:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    torch.distributed.all_reduce(t, group=group)
    if rec:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        SMALL_COLLECTIVES.append((what, e0, e1))
_MODE_NAMES = {(0, 0): "linear", (0, 1): "linear_dgrad", (1, 1): "linear_wgrad", (2, 0): "conv_im2col", (1, 3): "conv_wgrad"}


_ROUTE_NAMES = {0: "gemm_kernel", 1: "gemm_bf16_kernel", 2: "gemm_nt_glds_kernel", 3: "gemm_nt_p8_kernel", 4: "gemm_nt_p8n_kernel", 5: "gemm_tn_p8_kernel",
                6: "gemm_nt_pp_kernel", 7: "gemm_nt_rk_kernel", 8: "gemm_nt_p8ks_kernel"}


_AB_KNOBS = os.environ.get("DINOUNET_AB_KNOBS") == "1"


def _ab_env(name, default):
    """A/B switches of single code paths (tools, bisecting): read only under DINOUNET_AB_KNOBS=1.  The release configuration is the
    defaults -- the one configuration `pytest -m gpu` exercises (every ungated knob is an untested configuration).
    What stays readable without the gate is configuration, not an A/B aid: DINOUNET_PRECISION, DINOUNET_WGRAD_DEFER / _KEEP_MB,
    DINOUNET_VIT_CHAINS, DINOUNET_COMM_OUTSIDE_GRAPH, DINOUNET_LIB_OPTIONS."""
    return os.environ.get(name, default) if _AB_KNOBS else default


_WGRAD_COLSUM = _ab_env("DINOUNET_WGRAD_COLSUM", "1") == "1"     # bias gradients inside the weight-gradient kernels (A-B aid)
_KSCALE = _ab_env("DINOUNET_WGRAD_KSCALE", "1") == "1"           # DropPath scale inside the backward GEMMs (A-B aid)


def gemm_route(**kw):
    """kernel family du_gemm would run for these gemm_raw arguments (du_gemm_route)"""
    return gemm_raw(route_only=True, **kw)


TRACK_ROUTE = False          # tests: record the kernel family of the last product (du_gemm_route) in LAST_GEMM_ROUTE
LAST_GEMM_ROUTE = -1
ROUTES = []                  # (a_mode, b_mode, route) of every product since TRACK_ROUTE was switched on


_KS_PAIRS = _ab_env("DINOUNET_KS_PAIRS", "1") == "1"      # lend du_gemm the pair-exchange scratch (the library decides: du_set_option key 16)
_KS_SCRATCH = {}
_KS_RETIRED = []


def _ks_scratch(nbytes):
    """du_gemm_args.ks_ws: per (device, stream), persistent; the state words at its head are zeroed once (every launch leaves them zero)"""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    buf = _KS_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _KS_RETIRED.append(buf)      # a captured graph may still hold its address: never handed back to the allocator
        buf = torch.empty(max(nbytes, 4 << 20), dtype=torch.uint8, device=torch.device("cuda", key[0]))
        buf[:131072].zero_()
        _KS_SCRATCH[key] = buf
    return buf


def gemm_raw(*, dtype, out_dtype, a_mode, b_mode, M, N, K, A, lda, B, ldb, Cmat, ldc, batch=1, abs_=0, bbs=0, cbs=0,
             split_k=1, alpha=1.0, bias=None, act=ACT_NONE, gamma=None, row_scale=None, rs_rows=0, residual=None, ldr=0,
             store_mode=0, ps=(0, 0, 0), geom=None, a_colsum=None, b_colsum=None, rope=None, c2=None, route_only=False):
    a = GemmArgs()
    a.dtype, a.out_dtype, a.a_mode, a.b_mode = dtype, out_dtype, a_mode, b_mode
    a.M, a.N, a.K = M, N, K
    a.A, a.lda, a.a_batch_stride = A, lda, abs_
    a.B, a.ldb, a.b_batch_stride = B, ldb, bbs
    a.C, a.ldc, a.c_batch_stride = Cmat, ldc, cbs
    a.batch, a.split_k, a.alpha = batch, split_k, alpha
    a.bias = bias
    a.act = act
    a.gamma = gamma
    a.row_scale, a.rs_rows = row_scale, rs_rows
    a.residual, a.ldr = residual, ldr
    a.store_mode = store_mode
    a.ps_H, a.ps_W, a.ps_C = ps
    if geom is not None:
        a.geom = geom
    a.a_colsum = a_colsum
    a.b_colsum = b_colsum
    a.C2 = c2
    if rope is not None:
        a.rope_sin, a.rope_cos, a.rope_prefix, a.rope_qscale = rope
    if route_only:
        return int(_lib.lib().du_gemm_route(C.byref(a)))
    ws = None
    if dtype == DU_BF16 and a_mode == PLAIN_ROW and b_mode == PLAIN_ROW and M >= 1024 and 0 < M % 128 <= 64:
        # tall product with a short ragged last tile row: lend the library the scratch of its K-parallel tail kernels (gemm_skinny.hip)
        n_ws = int(_lib.lib().du_gemm_ws_elems(C.byref(a)))
        if n_ws > 0:
            ws = torch.empty(n_ws, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
            a.ws, a.ws_elems = ws.data_ptr(), n_ws
    if _KS_PAIRS and dtype == DU_BF16 and a_mode == PLAIN_ROW and b_mode == PLAIN_ROW and store_mode == 0 and M >= 1024 and K >= 1024 and (
            out_dtype == DU_F32 or 0 < M % 256 <= 64):
        n_ks = int(_lib.lib().du_gemm_ks_ws_bytes(C.byref(a)))
        if n_ks > 0:
            a.ks_ws, a.ks_ws_bytes = _ks_scratch(n_ks).data_ptr(), n_ks
    global LAST_GEMM_ROUTE
    if TRACK_ROUTE:
        LAST_GEMM_ROUTE = int(_lib.lib().du_gemm_route(C.byref(a)))
        ROUTES.append((a_mode, b_mode, LAST_GEMM_ROUTE))
    if PROFILE is not None:
        e0 = PROFILE.start()
        _lib.check(_lib.lib().du_gemm(C.byref(a), _st()), "du_gemm")
        es = 2 if dtype == DU_BF16 else 4
        eo = 2 if out_dtype == DU_BF16 else 4
        kin = K if geom is None else K // max(geom.KH * geom.KW, 1) if a_mode == IM2COL_ROW else K
        # the library names the kernel family it ran for this product (du_gemm_route): no mirror of the C-side dispatch here
        kname = _ROUTE_NAMES[int(_lib.lib().du_gemm_route(C.byref(a)))]
        mode = _MODE_NAMES.get((a_mode, b_mode), 'other')
        if residual is not None and out_dtype == DU_BF16:
            # products that add a bf16 residual (the adapter's output projection / ConvFFN fc2 / `up` + c1) are a family of their own: with
            # the residual read they are HBM-bound (K = 256: 198 MB for 22.5 GFLOP), a different roofline than the plain NT products
            mode += "+res"
        tag = f"{kname}<{'bf16' if dtype == DU_BF16 else 'f32'},{mode}>"
        if PROFILE.detail:
            tag += f" M{M} N{N} K{K} b{batch} sk{split_k}" + (f" k{geom.KH}s{geom.stride}t{geom.transposed}" if geom is not None else "")
        PROFILE.stop(tag, e0,
                     2.0 * M * N * K * batch, float(batch) * (M * kin * es + N * K * es + M * N * eo * (2 if residual is not None else 1)))
        return
    _lib.check(_lib.lib().du_gemm(C.byref(a), _st()), "du_gemm")


def set_corun(n):
    """tell du_gemm's tile choice how many independent products the caller keeps in flight on different streams (du_set_option key 9)"""
    _lib.check(_lib.lib().du_set_option(9, int(n)), "du_set_option(9)")


def _dp(t):
    return None if t is None else t.data_ptr()



# ----------------------------------------------------------------------------------------------------
# per-step weight packing
# ----------------------------------------------------------------------------------------------------
PK_CAST, PK_CONV_FWD, PK_CONV_DGRAD, PK_CONV_DGRAD_FLIP, PK_CONVT_FWD, PK_CONVT_DGRAD, PK_TRANSPOSE = 0, 1, 2, 3, 4, 5, 6


class WeightPack:
    """Kernel-ready copies of the trainable weights (bf16 cast, im2col column order, flipped data-gradient order, ConvTranspose
    layouts, concatenations), refreshed by ONE du_pack_weights launch at the start of every DinoUNet forward instead of ~300 tiny
    cast / permute / cat launches spread over the step.

    get() registers a (weight, kind) pair the first time it is asked for and returns None (the caller packs with torch ops, as before);
    from the next refresh() on it returns the packed buffer, valid as long as the source's autograd version counter is unchanged
    (i.e. through the forward and backward of the step; an optimizer step invalidates it until the next refresh).  The device-side
    table is rebuilt only outside stream capture, so the eager warm-up steps of TrainStep settle it before the hipGraph is recorded."""

    def __init__(self):
        self.enabled = _ab_env("DINOUNET_WEIGHT_PACK", "1") != "0"
        self.entries = {}
        self.order = []
        self.dirty = False
        self.table = None
        self.prefix = None
        self.total = 0
        self.nbuilt = 0

    @staticmethod
    def _shape(srcs, kind, cp):
        w = srcs[0]
        if kind == PK_CAST:
            rows = sum(t.shape[0] for t in srcs)
            return (rows,) + tuple(w.flatten(1).shape[1:]) if w.dim() > 1 else (rows,)
        if kind == PK_CONV_FWD:
            return (w.shape[0], w.shape[2] * w.shape[3] * cp)
        if kind in (PK_CONV_DGRAD, PK_CONV_DGRAD_FLIP):
            return (w.shape[1], w.shape[2] * w.shape[3] * w.shape[0])
        if kind == PK_CONVT_FWD:
            return (4 * w.shape[1], w.shape[0])
        if kind == PK_CONVT_DGRAD:
            return (w.shape[0], 4 * w.shape[1])
        if kind == PK_TRANSPOSE:                       # (rows of all sources, K) -> (K, rows)
            return (w.flatten(1).shape[1], sum(t.shape[0] for t in srcs))
        raise ValueError(kind)

    @staticmethod
    def _base(t):
        return t._base if t._base is not None else t

    def get(self, srcs, kind, dt=torch.bfloat16, cp=0):
        """srcs: weight tensor or tuple of tensors to concatenate along dim 0 (PK_CAST, PK_TRANSPOSE).  Only (views of) leaf Parameters
        are packed; temporaries (padded / concatenated copies made by the caller) are declined."""
        if not self.enabled:
            return None
        if torch.is_tensor(srcs):
            srcs = (srcs,)
        w = srcs[0]
        if not w.is_cuda or w.dtype != torch.float32 or dt not in (torch.bfloat16, torch.float32):
            return None
        for t in srcs:
            bt = self._base(t)
            if not (t.is_contiguous() and bt.is_leaf and isinstance(bt, torch.nn.Parameter)):
                return None
        if kind == PK_CONV_FWD and cp == 0:
            cp = w.shape[1]
        key = (tuple(t.data_ptr() for t in srcs), tuple(w.shape), kind, dt, cp)
        e = self.entries.get(key)
        if e is not None and any(r() is None for r in e["refs"]):     # a source parameter died and its address was recycled
            self._drop(key)
            e = None
        if e is None:
            dst = torch.zeros(self._shape(srcs, kind, cp), dtype=dt, device=w.device)
            e = {"key": key, "refs": [weakref.ref(self._base(t)) for t in srcs], "ptrs": [t.data_ptr() for t in srcs],
                 "nels": [t.numel() for t in srcs], "wshape": tuple(w.shape), "dst": dst, "kind": kind, "cp": cp, "vers": None,
                 "built": False}
            self.entries[key] = e
            self.order.append(e)
            self.dirty = True
            return None
        if not e["built"] or e["vers"] != tuple(r()._version for r in e["refs"]):
            return None
        return e["dst"]

    def invalidate(self):
        """Every packed copy is stale until the next refresh() (called by FusedClipSGD.step(): HIP kernels that write parameters
        through raw pointers do not bump the autograd version counters get() compares).  get() then declines and callers pack with
        torch ops, so sub-modules called outside DinoUNet.forward after an optimizer step still see the current weights."""
        for e in self.order:
            e["vers"] = None

    def _drop(self, key):
        e = self.entries.pop(key, None)
        if e is not None:
            self.order = [x for x in self.order if x is not e]
            self.dirty = True

    def refresh(self):
        if not self.enabled or not self.order:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            for e in [x for x in self.order if any(r() is None for r in x["refs"])]:
                self._drop(e["key"])
        if self.dirty and not capturing:
            if not self.order:
                self.table = None
                return
            rows, prefix, tot = [], [0], 0
            dev = self.order[0]["dst"].device
            for e in self.order:
                off = 0
                k = e["kind"]
                ws = e["wshape"]
                for ptr, nel in zip(e["ptrs"], e["nels"]):
                    if k == PK_CAST:
                        A, B, T, Cp, n = 0, 0, 0, 0, nel
                    elif k == PK_TRANSPOSE:
                        # one source: (A, B) -> (B, A); several: every source fills its column block of the (B, sum A) result
                        rows_i = nel // e["dst"].shape[0]
                        A, B, T, Cp, n = rows_i, e["dst"].shape[0], (e["dst"].shape[1] if len(e["ptrs"]) > 1 else 0), 0, nel
                    elif k in (PK_CONV_FWD, PK_CONV_DGRAD, PK_CONV_DGRAD_FLIP):
                        A, B, T, Cp, n = ws[0], ws[1], ws[2] * ws[3], e["cp"], e["dst"].numel()
                    else:
                        A, B, T, Cp, n = ws[0], ws[1], 4, 0, e["dst"].numel()
                    rows.append([ptr, e["dst"].data_ptr() + off * e["dst"].element_size(),
                                 k | ((1 if e["dst"].dtype == torch.float32 else 0) << 8), A, B, T, Cp, n])
                    off += A if (k == PK_TRANSPOSE and T) else n
                    # workgroups of this row: PACK_CHUNK outputs each; transposes: one 64 x 64 tile each
                    tot += ((A + 63) // 64) * ((B + 63) // 64) if k == PK_TRANSPOSE else (n + 4095) // 4096
                    prefix.append(tot)
                e["built"] = True
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self.prefix = torch.tensor(prefix, dtype=torch.int64).to(dev)
            self.total, self.nbuilt, self.dirty = tot, len(rows), False
        if self.table is None:
            return
        _lib.check(_lib.lib().du_pack_weights(_p(self.table), _p(self.prefix), self.nbuilt, self.total, _st()), "du_pack_weights")
        for e in self.order:
            if e["built"]:
                e["vers"] = tuple((r()._version if r() is not None else -1) for r in e["refs"])


PACK = WeightPack()

# ----------------------------------------------------------------------------------------------------
# dense products
# ----------------------------------------------------------------------------------------------------
def mm(x, w, *, out=None, out_dtype=None, bias=None, act=ACT_NONE, gamma=None, residual=None, row_scale=None,
       rs_rows=0, alpha=1.0):
    """out[m][n] = epi(sum_k x[m][k] * w[n][k]).  x (M,K), w (N,K) same dtype; bias/gamma/row_scale fp32."""
    _req(x, w)
    M, K, lda = _rows2d(x)
    N, K2, ldb = _rows2d(w)
    assert K == K2 and x.dtype == w.dtype, (x.shape, w.shape, x.dtype, w.dtype)
    od = out_dtype or (out.dtype if out is not None else x.dtype)
    if out is None:
        out = torch.empty((M, N), dtype=od, device=x.device)
    _, _, ldc = _rows2d(out)
    ldr = 0
    if residual is not None:
        assert residual.dtype == out.dtype
        ldr = residual.stride(0)
    gemm_raw(dtype=_code(x.dtype), out_dtype=_code(out.dtype), a_mode=PLAIN_ROW, b_mode=PLAIN_ROW, M=M, N=N, K=K,
             A=x.data_ptr(), lda=lda, B=w.data_ptr(), ldb=ldb, Cmat=out.data_ptr(), ldc=ldc, alpha=alpha,
             bias=_dp(bias), act=act, gamma=_dp(gamma), row_scale=_dp(row_scale), rs_rows=rs_rows,
             residual=_dp(residual), ldr=ldr)
    return out


def interleave_pairs(w1, w2):
    """(h, K) + (h, K) -> (2h, K) with rows (2j, 2j+1) = (w1[j], w2[j]) (also for 1-D biases): the operand layout of mm_swiglu."""
    return torch.stack((w1, w2), dim=1).reshape((2 * w1.shape[0],) + tuple(w1.shape[1:])).contiguous()


def mm_swiglu(x, w12, b12=None):
    """SwiGLU hidden activation silu(x w1^T + b1) * (x w2^T + b2) (layers/ffn_layers.py:73-77) from the INTERLEAVED projection
    w12 = interleave_pairs(w1, w2): one product; on the bf16 multi-phase NT kernels the gate runs in the GEMM epilogue (the (M, 2h)
    projection never reaches HBM), otherwise the product is followed by the du_swiglu_pairs kernel."""
    _req(x, w12)
    M, K, lda = _rows2d(x)
    N2, K2, ldb = _rows2d(w12)
    assert K == K2 and N2 % 2 == 0 and x.dtype == w12.dtype
    h = N2 // 2
    out = torch.empty((M, h), dtype=x.dtype, device=x.device)
    if x.dtype == torch.bfloat16:
        a = GemmArgs()
        a.dtype = a.out_dtype = DU_BF16
        a.a_mode = a.b_mode = PLAIN_ROW
        a.M, a.N, a.K = M, N2, K
        a.A, a.lda, a.B, a.ldb, a.C, a.ldc = x.data_ptr(), lda, w12.data_ptr(), ldb, out.data_ptr(), h
        a.batch, a.split_k, a.alpha, a.act = 1, 1, 1.0, ACT_SWIGLU
        a.bias = _dp(b12)
        e0 = PROFILE.start() if PROFILE is not None else None
        rc = _lib.lib().du_gemm(C.byref(a), _st())
        if rc == 0:
            if PROFILE is not None:
                PROFILE.stop("gemm_nt_p8_kernel<bf16,swiglu>", e0, 2.0 * M * N2 * K, 2.0 * (M * K + N2 * K + M * h))
            return out
        if rc != -2:
            _lib.check(rc, "du_gemm (swiglu)")
    u = mm(x, w12, bias=b12)
    _lib.check(_lib.lib().du_swiglu_pairs(_code(x.dtype), _p(u), _p(out), M, h, _st()), "du_swiglu_pairs")
    return out


def sample_gather(x, idx):
    """x (B, n) fp32 contiguous, idx (k,) int64 on the device -> (k, n) = x[idx]"""
    _req(x, idx)
    assert x.dtype == torch.float32 and x.is_contiguous() and idx.dtype == torch.int64
    n = x[0].numel()
    out = torch.empty((idx.numel(),) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().du_sample_copy(_p(x), _p(out), _p(idx), idx.numel(), n, 0, _st()), "du_sample_copy")
    return out


def sample_scatter_(x, src, idx):
    """x[idx] = src, in place (rows of idx are distinct)"""
    _req(x, src, idx)
    assert x.dtype == torch.float32 and src.dtype == torch.float32 and x.is_contiguous() and src.is_contiguous()
    _lib.check(_lib.lib().du_sample_copy(_p(src), _p(x), _p(idx), idx.numel(), x[0].numel(), 1, _st()), "du_sample_copy")
    return x


def mm_dgrad(dy, w, out=None):
    """dx[m][k] = sum_n dy[m][n] * w[n][k]   (w (N,K) read column-wise, no transposed copy)."""
    _req(dy, w)
    M, N, lda = _rows2d(dy)
    N2, K, ldb = _rows2d(w)
    assert N == N2 and dy.dtype == w.dtype
    if out is None:
        out = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    gemm_raw(dtype=_code(dy.dtype), out_dtype=_code(out.dtype), a_mode=PLAIN_ROW, b_mode=PLAIN_COL, M=M, N=K, K=N,
             A=dy.data_ptr(), lda=lda, B=w.data_ptr(), ldb=ldb, Cmat=out.data_ptr(), ldc=out.stride(0))
    return out


# split-K of the weight gradients: workgroups aimed for / minimum contraction rows per split.  Swept on the dinounet_l shapes
# (tools/gemm_bench.py, profiles/r01_splitk_sweep.txt): 512 x 1024 for the linear layers (2 workgroups per CU, fewer fp32 atomics),
# 1024 x 1024 for the convolutions (their M = Cout <= 128 tiles are small).
class ZeroPool:
    """fp32 accumulators of the split-K weight-gradient GEMMs (the kernel adds into them atomically, so they start at zero).
    A train step asks for ~100 of them; each torch.zeros is its own >= 5 us fill launch even inside a hipGraph.  The pool records
    the request sizes of one step and, from the next step on, serves the same sequence as views of ONE freshly allocated, once
    filled buffer per step (a new buffer every step: gradients handed to autograd never alias a later step's).  Any deviation
    from the recorded sequence falls back to plain torch.zeros for the rest of that step.  `new_step()` is called at the top of
    DinoUNet.forward; DINOUNET_ZERO_POOL=0 disables."""
    ALIGN = 64     # floats (256 B)

    def __init__(self):
        self.enabled = _ab_env("DINOUNET_ZERO_POOL", "1") != "0"
        self.plan = None
        self.rec = []
        self._reset()

    def _reset(self):
        self.buf = None
        self.idx = 0
        self.off = 0
        self.ok = True

    def new_step(self):
        if self.rec:
            self.plan = self.rec
        self.rec = []
        self._reset()

    def zeros(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        if not self.enabled:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        self.rec.append(n)
        if self.ok and self.plan is not None and self.idx < len(self.plan) and self.plan[self.idx] == n and \
                (self.buf is None or self.buf.device == device):
            if self.buf is None:
                if self.idx != 0:
                    self.ok = False
                    return torch.zeros(shape, dtype=torch.float32, device=device)
                total = sum((m + self.ALIGN - 1) // self.ALIGN * self.ALIGN for m in self.plan)
                self.buf = torch.zeros(total, dtype=torch.float32, device=device)
            v = self.buf[self.off:self.off + n].view(shape)
            self.off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.idx += 1
            return v
        self.ok = False
        return torch.zeros(shape, dtype=torch.float32, device=device)


ZEROS = ZeroPool()

_SPLIT_TARGET = int(_ab_env("DU_SPLIT_TARGET", "0"))
_SPLIT_MINK = int(_ab_env("DU_SPLIT_MINK", "1024"))


def _split_for(tiles, kdim, target=512):
    s = max(1, min((_SPLIT_TARGET or target) // max(tiles, 1), kdim // _SPLIT_MINK))
    return max(1, s)


class WgradQueue:
    """Deferred weight gradients (du_gemm_tn_group: several dW = dY^T X products in ONE launch).

    Nothing reads a weight gradient before clip_grad_norm_ / the optimizer (nnUNetTrainer.py:919-928), while one product alone has to be
    cut into 16-64 K splits to fill 256 CUs and pays a full fp32 read-modify-write of its result per split (more than its MFMA loop for
    most adapter layers, tools/gemm_tn_bench.py).  A backward node therefore only allocates the (zeroed) result, records the job and
    keeps dY / X alive; flush() launches everything queued -- the workgroups are dealt out over all jobs, 2-4 splits per product.

    flush() runs (a) from an autograd-engine callback at the end of the backward pass that queued the job (`p.grad` is complete when
    backward() / autograd.grad() returns, whatever the caller does next), (b) from GradAllReducer just before a bucket of gradients is
    gathered for its all-reduce, (c) at once when a job is queued outside a backward pass (op tests, tools).

    Protocol of a backward node: grp = begin(weak refs of its weight / bias parameters) -> None: compute now, as before; else queue the
    products with mm_wgrad(defer=grp) / _queue_conv_wgrad(grp, ...) and hand the buffers to autograd iff grp["ret"].  What the queue
    guarantees: the autograd engine adds the contributions of a parameter that is used several times in a pass (the ConvTranspose of
    LearnableUpsampleBlock is applied twice, dinounet_training.py:255-264; FAPM's shared basis four times, :423) AS SOON AS the second one
    is returned -- so every contribution of a pass goes into ONE buffer (later jobs accumulate into the first one's result and the
    node returns None), and once a contribution of a parameter has been handed over complete (flushed, or computed at once because a
    job was not legal) all its later ones are computed at once too.  Kept OUT of the queue: parameters that already have a gradient
    (AccumulateGrad would add the unfinished buffer immediately), carry tensor hooks or foreign post-accumulate hooks (GradAllReducer's
    flush before they read), or whose forward ran inside a torch DistributedDataParallel wrapper (its reducer copies gradients out of
    the accumulators as they arrive).  DINOUNET_WGRAD_DEFER=0 disables the queue."""

    trace = None       # tools/wgrad_jobs.py: list -> every flush appends its job table

    def __init__(self):
        self.enabled = os.environ.get("DINOUNET_WGRAD_DEFER", "1") != "0"
        self.jobs = []          # du_tn_job records
        self.keep = []          # dY / X tensors of the queued jobs
        self.flops = 0.0
        self.nbytes = 0.0
        self._armed_task = None # id of the autograd graph task (backward pass) whose end-of-pass callback is queued
        self.keep_bytes = 0     # bytes of dY / X pinned by the queued jobs
        self.keep_limit = int(os.environ.get("DINOUNET_WGRAD_KEEP_MB", "16384")) << 20
        self.state = {}         # id(parameter) -> group key (contributions pending in that group's buffers) | "done" (handed over complete)
        self.groups = {}        # group key (ids of a node's parameters) -> {"bufs", "jobs", "first", "ret", "key"}
        self.flush_aware_hooks = 0   # > 0: the post-accumulate hooks on the parameters belong to GradAllReducer
        self.launches = 0       # statistics (tests)
        self.queued = 0

    @staticmethod
    def _param(w):
        """the leaf Parameter behind `w`: w itself, or a view of it whose backward hands the gradient back as a VIEW of the same memory
        (same element count, contiguous: conv1x1's w.view(Cout, -1)).  A slice / expand view would have autograd copy or sum the still
        unfinished buffer (SliceBackward, ExpandBackward): those weights are not deferred."""
        b = w._base if w._base is not None else w
        if not (isinstance(b, torch.nn.Parameter) and b.is_leaf):
            return None
        if w is not b and not (w.numel() == b.numel() and w.is_contiguous() and b.is_contiguous()):
            return None
        return b

    def note_use(self, *ws):
        """forward side -> weak references to the leaf Parameters behind the weights for begin(), or None when they are not
        (views of) leaf Parameters or the forward runs inside a torch DDP wrapper."""
        if not self.enabled:              # (grad mode is always off inside autograd.Function.forward: ctx.needs_input_grad decides later)
            return None
        if getattr(torch.nn.parallel.DistributedDataParallel, "_active_ddp_module", None) is not None:
            return None
        ps = [self._param(w) for w in ws]
        if any(p is None for p in ps):
            return None
        return [weakref.ref(p) for p in ps]

    def _arm(self):
        """make sure THIS backward pass ends with _end_of_pass().  Keyed on the engine's graph-task id, not on a flag: a pass that raised
        (out of memory, KeyboardInterrupt, a failed hipGraph capture) never runs its final callbacks -- a flag would stay set and every later
        backward() would return with unfinished gradients.  A different task id while one is armed means either that the armed pass died
        or that this is a NESTED pass (reentrant torch.utils.checkpoint, autograd.grad inside a hook or a custom Function) and the armed one
        is suspended, alive, with its zero-filled buffers already handed to autograd: the two cannot be told apart from here,
        so whatever is queued is LAUNCHED now, never dropped -- for a live outer pass that completes its gradients early, for a dead one it
        is a wasted launch into buffers the queue itself keeps alive (dY / X in `keep`, results in the groups) and nobody reads."""
        tid = torch._C._current_graph_task_id()
        if tid == -1:                      # not inside a backward pass: nothing will call back
            return False
        if self._armed_task != tid:
            if self._armed_task is not None:
                self._armed_task = None
                self.flush()
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
            except RuntimeError:
                return False
            self._armed_task = tid
        return True

    def _drop(self):
        self.jobs, self.keep, self.flops, self.nbytes, self.keep_bytes = [], [], 0.0, 0.0, 0
        self.groups.clear()
        self.state.clear()
        self._armed_task = None

    def reset(self):
        """Forget everything queued WITHOUT launching it.  Call after a backward pass that raised (out of memory, an interrupted step): its
        final callback never ran, and the queue cannot tell that dead pass from a suspended outer pass of a nested backward -- so the next
        pass would LAUNCH the dead jobs.  That is harmless when the parameters' gradients were dropped since (zero_grad(set_to_none=True),
        the default), but with zero_grad(set_to_none=False) or gradient accumulation autograd may still hold a dead job's result buffer
        as p.grad, and the stale product would be added into the new gradient.  TrainStep does this itself after a failed
        capture (training._forget_failed_pass)."""
        self._drop()

    def _hand_over(self, ids):
        """the parameters `ids` are about to receive a COMPLETE contribution: whatever is queued for them must be complete first"""
        for i in ids:
            key = self.state.get(i)
            if key not in (None, "done") and self.groups.get(key, {}).get("jobs"):
                self.flush()
                break
        for i in ids:
            key = self.state.get(i)
            if key not in (None, "done"):
                self.groups.pop(key, None)
            self.state[i] = "done"

    def begin(self, refs):
        """backward side, once per node before its weight-gradient products -> group dict or None (see the class docstring)."""
        if not self.enabled or refs is None:
            return None
        ps = [r() for r in refs]
        if any(p is None for p in ps) or not self._arm():
            return None
        ids = tuple(id(p) for p in ps)
        ok = True
        for p in ps:
            if p.grad is not None or p._backward_hooks:
                ok = False
            if getattr(p, "_post_accumulate_grad_hooks", None) and not self.flush_aware_hooks:
                ok = False
        if not ok or any(self.state.get(i) not in (None, ids) for i in ids):
            self._hand_over(ids)
            return None
        grp = self.groups.get(ids)
        if grp is None:
            grp = {"key": ids, "bufs": {}, "jobs": [], "first": True, "ret": True, "refs": list(refs)}
            self.groups[ids] = grp
            for i in ids:
                self.state[i] = ids
        else:
            grp["first"] = grp["ret"] = False
            for j in grp["jobs"]:          # the earlier products now share their result: no plain stores
                j.accumulate = 1
        return grp

    def abort(self, grp):
        """a node that was given a group computes its products at once after all (job not legal)"""
        self._hand_over(grp["key"])
        self.groups.pop(grp["key"], None)
        grp["ret"] = True

    def buffer(self, grp, name, shape, device):
        """the group's result buffer `name` (zeroed on first use; later contributions of the pass add into it)"""
        if grp is None:
            return ZEROS.zeros(shape, device)
        t = grp["bufs"].get(name)
        if t is None:
            t = grp["bufs"][name] = ZEROS.zeros(shape, device)
        # a fresh alias, never the object kept in the group: AccumulateGrad adopts an incoming gradient by reference only when nobody
        # else holds that tensor object (use_count), otherwise it CLONES it on the spot -- i.e. copies the still-unfinished buffer
        return t.view(t.shape)

    def legal(self, job):
        return bool(_lib.lib().du_gemm_tn_group_legal(C.byref(job)))

    def add(self, job, keep, grp=None):
        if grp is not None:
            if not grp["first"]:
                job.accumulate = 1
            grp["jobs"].append(job)
        self.jobs.append(job)
        self.keep.extend(keep)
        self.keep_bytes += sum(t.numel() * t.element_size() for t in keep)
        self.queued += 1
        self.flops += 2.0 * job.M * job.N * job.K
        self.nbytes += 2.0 * job.K * (job.M + job.N) + 4.0 * job.M * job.N
        if not self._arm():
            self.flush()
        elif self.keep_bytes > self.keep_limit:       # bound what the queue pins (dY and X of every deferred layer) by launching early
            self.flush()

    def _end_of_pass(self):
        self._armed_task = None
        self.flush()
        self.state.clear()
        self.groups.clear()

    def flush(self):
        done = []
        for ids, grp in list(self.groups.items()):
            if not grp["jobs"]:            # begun (FAPM begins three groups up front) but nothing queued yet: stays open
                continue
            for i in ids:
                self.state[i] = "done"
            done.append(self.groups.pop(ids))
        if not self.jobs:
            return
        n = len(self.jobs)
        arr = (_lib.TnJob * n)(*self.jobs)
        if self.trace is not None:           # tools/wgrad_jobs.py: the job table of a step (M, N, K, gather kind, per-sample scale?)
            self.trace.append([(int(j.gather), int(j.M), int(j.N), int(j.K), bool(j.alpha), int(j.accumulate)) for j in self.jobs])
        e0 = PROFILE.start() if PROFILE is not None else None
        rc = _lib.lib().du_gemm_tn_group(arr, n, _st())
        fl, nb = self.flops, self.nbytes
        self.jobs, self.keep, self.flops, self.nbytes, self.keep_bytes = [], [], 0.0, 0.0, 0
        _lib.check(rc, "du_gemm_tn_group")
        self.launches += 1
        # Did autograd adopt the buffers by reference?  Where it did not (AccumulateGrad clones when the layout contract fails or grad
        # mode is on; the engine's input buffer adds out of place when the parameter has another, non-queued use in the pass) the
        # parameter holds a copy made while the buffer was still ZERO: the product is added to it here, behind the launch.
        for grp in done:
            refs = grp.get("refs", ())
            if len(refs) > 2:              # one product for several parameters (the MSDA offsets + weights pack): the node hands out slices
                continue
            for name, r in zip(("dw", "db"), refs):
                buf, p = grp["bufs"].get(name), r()
                if buf is None or p is None or p.grad is None or p.grad.numel() != buf.numel():
                    continue
                if p.grad.untyped_storage().data_ptr() != buf.untyped_storage().data_ptr():
                    p.grad.add_(buf.view(p.grad.shape).to(p.grad.dtype))
        if PROFILE is not None:
            PROFILE.stop("gemm_tn_group_kernel<bf16,wgrad>" + (f" jobs{n}" if PROFILE.detail else ""), e0, fl, nb)


WGRAD = WgradQueue()


def _pow2(v):
    return v > 0 and (v & (v - 1)) == 0


def _queue_conv_wgrad(grp, kind, a, lda, srcs, dyB, Hs, Ws, M, w_shape, want_bias):
    """Queue a convolution weight gradient on the grouped launch (du_tn_job.gather) -> (dw in torch layout, db or None), or None when a
    job is not legal.  kind 3: 3 x 3 / s1 / p1, A = dy (pixels, Cout), srcs = [(x, ld, C), ...] the sources of the (fused concat) input, bias
    gradient = column sums of A.  kind 2: ConvTranspose2d k2 s2, A = x (pixels, Cin), srcs = [(dy, ld, Cout)], bias gradient = sum of dy."""
    K = dyB * Hs * Ws
    taps = 9 if kind == 3 else 4
    ctot = sum(c for _, _, c in srcs)
    jobs = []
    for i, (t, ld, cc) in enumerate(srcs):
        j = _lib.TnJob(A=a.data_ptr(), lda=lda, B=t.data_ptr(), ldb=ld, C=1, ldc=0, a_colsum=None, alpha=None, M=M, N=taps * cc, K=K,
                       accumulate=0, b_colsum=None, gather=kind, Hs=Hs, Ws=Ws, Cb=cc, taps=taps, inner=cc, inner_total=ctot,
                       c_off=sum(c for _, _, c in srcs[:i]))
        if (kind == 3 and Ws % 64) or not WGRAD.legal(j):
            return None
        jobs.append(j)
    dw = WGRAD.buffer(grp, "dw", w_shape, a.device)
    db = None
    if want_bias:
        db = WGRAD.buffer(grp, "db", (M if kind == 3 else ctot,), a.device)
        if kind == 3:
            jobs[0].a_colsum = db.data_ptr()
        else:
            jobs[0].b_colsum = db.data_ptr()
    for j, (t, _, _) in zip(jobs, srcs):
        j.C = dw.data_ptr()
        WGRAD.add(j, (a, t), grp)
    WGRAD.queued -= len(jobs) - 1
    return dw, db


def mm_wgrad(dy, x, with_colsum=False, k_scale=None, defer=False):
    """dw[n][k] = sum_m dy[m][n] * x[m][k]  -> fp32 (N,K); split-K over the rows with fp32 atomics.
    with_colsum: also return the bias gradient db[n] = sum_m dy[m][n] -- taken inside the weight-gradient kernel from the dY fragments
    it streams anyway (du_gemm_args.a_colsum) where the kernel family can, by du_colsum (two more launches, one more pass over dY)
    otherwise.
    defer: a group from WGRAD.begin() (a backward node: the product is queued and computed later in the pass together with others; the
    returned tensors are the group's buffers, complete after WGRAD.flush()) or True (stand-alone job: tests / tools).  A product that
    cannot be queued is computed at once and the group is told (WGRAD.abort)."""
    _req(dy, x)
    Mr, N, lda = _rows2d(dy)
    Mr2, K, ldb = _rows2d(x)
    assert Mr == Mr2 and dy.dtype == x.dtype
    grp = defer if isinstance(defer, dict) else None
    if defer and WGRAD.enabled and dy.dtype == torch.bfloat16 and (_WGRAD_COLSUM or not with_colsum):
        # k_scale (DropPath in backward): one job per sample, scaled by that sample's factor read on the device -- dropped samples (factor 0)
        # cost nothing, and no scaled copy of dy is ever built
        nj, rows_j = (1, Mr) if k_scale is None else (k_scale[0].numel(), k_scale[1])
        job = _lib.TnJob(A=dy.data_ptr(), lda=lda, B=x.data_ptr(), ldb=ldb, C=1, ldc=K, a_colsum=None, alpha=None, M=N, N=K, K=rows_j,
                         accumulate=0)
        if nj * rows_j == Mr and (k_scale is None or (k_scale[0].dtype == torch.float32 and k_scale[0].is_contiguous())) and \
                (nj == 1 or (rows_j * lda * 2) % 16 == 0 and (rows_j * ldb * 2) % 16 == 0) and WGRAD.legal(job):
            out = WGRAD.buffer(grp, "dw", (N, K), dy.device)
            db = WGRAD.buffer(grp, "db", (N,), dy.device) if with_colsum else None
            for b in range(nj):
                jb = _lib.TnJob(A=dy.data_ptr() + b * rows_j * lda * 2, lda=lda, B=x.data_ptr() + b * rows_j * ldb * 2, ldb=ldb,
                                C=out.data_ptr(), ldc=K, a_colsum=db.data_ptr() if with_colsum else None,
                                alpha=(k_scale[0].data_ptr() + 4 * b) if k_scale is not None else None, M=N, N=K, K=rows_j,
                                accumulate=1 if nj > 1 else 0)
                WGRAD.add(jb, (dy, x, k_scale[0]) if k_scale is not None else (dy, x), grp)
            WGRAD.queued -= nj - 1                     # statistics count products, not per-sample jobs
            return (out, db) if with_colsum else out
    if grp is not None:
        WGRAD.abort(grp)
    out = ZEROS.zeros((N, K), dy.device)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    kw = dict(dtype=_code(dy.dtype), out_dtype=DU_F32, a_mode=PLAIN_COL, b_mode=PLAIN_COL, M=N, N=K, K=Mr,
              A=dy.data_ptr(), lda=lda, B=x.data_ptr(), ldb=ldb, Cmat=out.data_ptr(), ldc=K, split_k=_split_for(tiles, Mr))
    scaled_in_kernel = False
    if k_scale is not None:
        # k_scale = (per-sample scale (B,) fp32, rows per sample): scales the contraction rows of dy inside the kernel (du_gemm: row_scale
        # with a contraction-major A).  Only the bf16 tile engine does it; otherwise the scaled copy is built here
        kws = dict(kw, row_scale=k_scale[0].data_ptr(), rs_rows=k_scale[1])
        if gemm_route(**kws) == 1:
            kw = kws
            scaled_in_kernel = True
        else:
            dy = (dy.view(k_scale[0].numel(), k_scale[1], -1) * k_scale[0].view(-1, 1, 1).to(dy.dtype)).view(dy.shape)
            kw = dict(kw, A=dy.data_ptr(), lda=dy.stride(0))
    if not with_colsum:
        gemm_raw(**kw)
        return out
    if _WGRAD_COLSUM and dy.dtype == torch.bfloat16 and gemm_route(**kw) in (1, 5):
        db = ZEROS.zeros((N,), dy.device)
        gemm_raw(a_colsum=db.data_ptr(), **kw)
        return out, db
    gemm_raw(**kw)
    if scaled_in_kernel:        # the bias gradient is the column sum of the SCALED dy: the kernel scaled its own copy only
        dy = (dy.view(k_scale[0].numel(), k_scale[1], -1) * k_scale[0].view(-1, 1, 1).to(dy.dtype)).view(dy.shape)
    return out, colsum(dy)


def _reduce_ws(dt, G, P, Cc, device):
    """scratch for the two-stage column reductions (du_reduce_ws_elems); a plain torch.empty: the caching allocator makes it free
    and it is captured like any other temporary under hipGraph."""
    n = int(_lib.lib().du_reduce_ws_elems(_code(dt), G, P, Cc))
    return torch.empty(max(n, 1), dtype=torch.float32, device=device), n


def colsum(x2d):
    """sum over rows of a (rows, C) matrix -> fp32 (C,)   (bias gradients)."""
    rows, Cc, ld = _rows2d(x2d)
    out = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
    ws, n = _reduce_ws(x2d.dtype, 1, rows, Cc, x2d.device)
    _lib.check(_lib.lib().du_colsum(_code(x2d.dtype), _p(x2d), ld, _p(out), rows, Cc, _p(ws), n, _st()), "du_colsum")
    return out


def _geom(src, KH, KW, stride, pad, Ho, Wo, transposed, src2=None):
    B, Hi, Wi, C1, ld = _nhwc(src)
    g = ConvGeom()
    Ct = C1
    if src2 is not None:
        B2, H2, W2, C2, ld2 = _nhwc(src2)
        assert (B2, H2, W2) == (B, Hi, Wi) and src2.dtype == src.dtype
        g.p2, g.ld2 = src2.data_ptr(), ld2
        Ct = C1 + C2
    g.C1 = C1
    g.Hi, g.Wi, g.C = Hi, Wi, Ct
    g.KH, g.KW, g.stride, g.pad = KH, KW, stride, pad
    g.Ho, g.Wo, g.transposed = Ho, Wo, transposed
    return g, B, ld, Ct


def pack_conv_weight(w, dt):
    """(Cout,Cin,KH,KW) -> (Cout, KH*KW*Cin) in the activation dtype, (tap, ci) column order."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dt).contiguous()


def pack_conv_weight_dgrad(w, dt):
    """(Cout,Cin,KH,KW) -> (Cin, KH*KW*Cout): [ci][(dy,dx,co)]."""
    return w.permute(1, 2, 3, 0).reshape(w.shape[1], -1).to(dt).contiguous()


def pack_conv_weight_dgrad_flipped(w, dt):
    """(Cout,Cin,3,3) -> (Cin, 9*Cout): the stride-1 data gradient is the 3x3 convolution of dY with the spatially flipped,
    transposed filter, [ci][(ky,kx,co)] = w[co][ci][2-ky][2-kx]."""
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).to(dt).contiguous()


def conv3x3_halo(x, wp, bias, x2=None, want_stats=False):
    """LDS-tiled direct 3x3 / stride 1 / pad 1 convolution (du_conv3x3_halo).  Returns (y, stats_partial or None), or None when the
    kernel does not serve the shape (caller falls back to the implicit GEMM)."""
    if x.dtype != torch.bfloat16:
        return None
    B, H, W, C1, ld = _nhwc(x)
    Cout, Kc, ldb = _rows2d(wp)
    Cin = Kc // 9
    if H % 8 or W % 16 or Cout not in (32, 64, 128) or ldb != Kc or Cin % 32:
        return None
    ld2, p2 = 0, None
    if x2 is not None:
        B2, H2, W2, C2, ld2 = _nhwc(x2)
        if (B2, H2, W2) != (B, H, W) or C1 + C2 != Cin or C1 % 32:
            return None
        p2 = _p(x2)
    elif C1 != Cin:
        return None
    y = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    nparts = int(_lib.lib().du_conv3x3_halo_parts(C1, Cin, Cout, B, H, W)) if want_stats else 0
    part = torch.empty((nparts, Cout, 2), dtype=torch.float32, device=x.device) if want_stats else None
    e0 = PROFILE.start() if PROFILE is not None else None
    rc = _lib.lib().du_conv3x3_halo(_p(x), ld, p2, ld2, C1, Cin, Cout, B, H, W, _p(wp), _p(bias), _p(y), Cout, _p(part), _st())
    if rc == -2 and part is not None:
        # the kernel the partial-statistics buffer was sized for declined on the tensor's byte size: same convolution without
        # epilogue statistics (the norm layer then runs its own statistics pass)
        part = None
        rc = _lib.lib().du_conv3x3_halo(_p(x), ld, p2, ld2, C1, Cin, Cout, B, H, W, _p(wp), _p(bias), _p(y), Cout, None, _st())
    if rc == -2:                                  # DU_ERR_UNSUPPORTED
        return None
    _lib.check(rc, "du_conv3x3_halo")
    if PROFILE is not None:
        # <= 64 output channels: HBM-bound layers (512^2 / 256^2); 128: above the ridge (bench.py reports them against the MFMA peak)
        # named by the kernel that ran: the streaming strip kernel writes one partial row per (image, segment, strip),
        # the LDS-tiled kernel one per 8 x 16 tile -- du_conv3x3_halo_parts tells them apart without mirroring the C-side choice
        strip = int(_lib.lib().du_conv3x3_halo_parts(C1, Cin, Cout, B, H, W)) != B * (H // 8) * (W // 16)
        kname = "conv3x3_strip_kernel<bf16>" if strip else ("conv3x3_halo_kernel<bf16>" if Cout <= 64 else "conv3x3_halo_c128_kernel<bf16>")
        PROFILE.stop(kname + (f" {H}x{W} {Cin}->{Cout}" if PROFILE.detail else ""), e0,
                     2.0 * B * H * W * Cin * Cout * 9, 2.0 * B * H * W * (Cin + Cout))
    return y, part


_CONV_SPLITK = _ab_env("DINOUNET_CONV_SPLITK", "1") == "1"


def _im2col_split(M, N, K):
    """K splits for an implicit-GEMM convolution whose tile grid leaves most of the chip idle (the SPM's low-resolution stride-2 layers,
    dinov3_adapter.py:259-277: 2048 x 256 x 2304 is 32 tiles of 128 x 128 with 36 serial K-steps each -- 79 us for 2.4 GFLOP).  The generic
    engine's split-K form adds fp32 partial tiles into a zeroed buffer (no bias / activation: those layers have none); the caller converts
    to bf16.  0 = leave the product alone."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if not _CONV_SPLITK or tiles > 128 or K < 1024 or N % 4:
        return 0
    return max(2, min(8, 256 // tiles, K // 256))


def _conv_splitk(sk, M, N, K, a, lda, w, ldb, g, out):
    """implicit GEMM cut into sk K ranges: every range writes its fp32 partial product to its own slab (DU_STORE_SLABS: plain stores, no
    zero fill, no atomics) and du_splitk_reduce_bf16 adds the slabs in order -- the result does not depend on the order in which the
    workgroups finish (a forward pass whose bits changed from run to run would make every repeat-run test a tolerance test)."""
    slabs = torch.empty((sk, M, N), dtype=torch.float32, device=a.device)
    kw = dict(dtype=DU_BF16, out_dtype=DU_F32, a_mode=IM2COL_ROW, b_mode=PLAIN_ROW, M=M, N=N, K=K, A=a.data_ptr(), lda=lda,
              B=w.data_ptr(), ldb=ldb, Cmat=slabs.data_ptr(), ldc=N, split_k=sk, store_mode=STORE_SLABS, geom=g)
    if gemm_route(**kw) != 1:          # only the bf16 tile engine writes slabs (du_gemm would return DU_ERR_UNSUPPORTED): the caller runs
        return False                   # the product unsplit
    # (the library rounds the K range per split up to whole K tiles and may run fewer splits than asked: slabs it does not write must not
    #  be read -- it says how many it writes)
    used = int(_lib.lib().du_gemm_slab_count(K, sk))
    gemm_raw(**kw)
    _lib.check(_lib.lib().du_splitk_reduce_bf16(_p(slabs), _p(out), used, M * N, _st()), "du_splitk_reduce_bf16")
    return True


def conv_fwd(x, wp, bias, KH, KW, stride, pad, x2=None, out=None, act=ACT_NONE):
    _req(x, wp)
    if KH == 3 and KW == 3 and stride == 1 and pad == 1 and out is None and act == ACT_NONE:
        r = conv3x3_halo(x, wp, bias, x2)
        if r is not None:
            return r[0]
    B, Hi, Wi, _, _ = _nhwc(x)
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    g, B, ld, Ct = _geom(x, KH, KW, stride, pad, Ho, Wo, 0, x2)
    Cout, Kc, ldb = _rows2d(wp)
    assert Kc == KH * KW * Ct and wp.dtype == x.dtype, (wp.shape, KH, KW, Ct)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    _, _, _, _, ldc = _nhwc(out)
    sk = _im2col_split(B * Ho * Wo, Cout, Kc) if (x.dtype == torch.bfloat16 and bias is None and act == ACT_NONE and out.is_contiguous()) else 0
    if sk and _conv_splitk(sk, B * Ho * Wo, Cout, Kc, x, ld, wp, ldb, g, out):
        return out
    gemm_raw(dtype=_code(x.dtype), out_dtype=_code(out.dtype), a_mode=IM2COL_ROW, b_mode=PLAIN_ROW, M=B * Ho * Wo, N=Cout,
             K=Kc, A=x.data_ptr(), lda=ld, B=wp.data_ptr(), ldb=ldb, Cmat=out.data_ptr(), ldc=ldc, bias=_dp(bias), act=act,
             geom=g)
    return out


def conv_dgrad(dy, wd, KH, KW, stride, pad, Hin, Win, out=None):
    """dX (B,Hin,Win,Cin) from dY (B,Ho,Wo,Cout) and wd = pack_conv_weight_dgrad(w)."""
    _req(dy, wd)
    g, B, ld, Cout = _geom(dy, KH, KW, stride, pad, Hin, Win, 1)
    Cin, Kc, ldb = _rows2d(wd)
    assert Kc == KH * KW * Cout
    if out is None:
        out = torch.empty((B, Hin, Win, Cin), dtype=dy.dtype, device=dy.device)
    _, _, _, _, ldc = _nhwc(out)
    sk = _im2col_split(B * Hin * Win, Cin, Kc) if (dy.dtype == torch.bfloat16 and out.is_contiguous()) else 0
    if sk and _conv_splitk(sk, B * Hin * Win, Cin, Kc, dy, ld, wd, ldb, g, out):
        return out
    gemm_raw(dtype=_code(dy.dtype), out_dtype=_code(out.dtype), a_mode=IM2COL_ROW, b_mode=PLAIN_ROW, M=B * Hin * Win, N=Cin,
             K=Kc, A=dy.data_ptr(), lda=ld, B=wd.data_ptr(), ldb=ldb, Cmat=out.data_ptr(), ldc=ldc, geom=g)
    return out


def conv3x3_wgrad_halo(x, dy, x2=None, with_db=False):
    """LDS-tiled 3x3 / stride 1 / pad 1 weight gradient (du_conv3x3_wgrad_halo) -> fp32 (Cout, 9*Cin), or None if not served.
    with_db: -> (dw, db): the bias gradient (column sums of dy) rides behind the weight gradient in the same partial slabs."""
    if x.dtype != torch.bfloat16:
        return None
    B, H, W, C1, ld = _nhwc(x)
    Bo, Ho, Wo, Cout, lddy = _nhwc(dy)
    Cin, ld2, p2 = C1, 0, None
    if x2 is not None:
        _, _, _, C2, ld2 = _nhwc(x2)
        Cin, p2 = C1 + C2, _p(x2)
    L = _lib.lib()
    blocks = int(L.du_conv3x3_wgrad_halo_blocks(C1, Cin, Cout, B, H, W))
    if blocks <= 0 or (Ho, Wo) != (H, W):
        return None
    nel = Cout * 9 * Cin + (Cout if with_db else 0)
    part = torch.empty((blocks, nel), dtype=torch.float32, device=x.device)
    out = torch.empty(nel, dtype=torch.float32, device=x.device)
    dw = out[:Cout * 9 * Cin].view(Cout, 9 * Cin)
    e0 = PROFILE.start() if PROFILE is not None else None
    rc = L.du_conv3x3_wgrad_halo(_p(x), ld, p2, ld2, C1, Cin, Cout, B, H, W, _p(dy), lddy, _p(part), _p(out), 1 if with_db else 0, _st())
    if rc == -2:
        return None
    _lib.check(rc, "du_conv3x3_wgrad_halo")
    if PROFILE is not None:
        # (named by the kernel the library runs for this shape: the round-5 rows kernel serves 32 / 64 output channels, conv_halo.hip)
        wk = "conv3x3_wgrad_rows_kernel<bf16>" if Cout in (32, 64) else "conv3x3_wgrad_halo_kernel<bf16>"
        PROFILE.stop(wk + (f" {H}x{W} {Cin}->{Cout}" if PROFILE.detail else ""), e0,
                     2.0 * B * H * W * Cin * Cout * 9, 2.0 * B * H * W * (Cin + Cout))
    return (dw, out[Cout * 9 * Cin:]) if with_db else dw


def conv_wgrad(x, dy, KH, KW, stride, pad, x2=None, with_db=False):
    """-> fp32 (Cout, KH*KW*C) in (tap, ci) column order; with_db: -> (dw, db), the bias gradient sum_pixels dy taken inside the
    weight-gradient kernel where it can be (halo kernel: extra slab columns; implicit GEMM: du_gemm_args.a_colsum), else by du_colsum."""
    _req(x, dy)
    if KH == 3 and KW == 3 and stride == 1 and pad == 1:
        r = conv3x3_wgrad_halo(x, dy, x2, with_db and _WGRAD_COLSUM)
        if r is not None:
            if with_db and not _WGRAD_COLSUM:
                return r, colsum(dy.view(-1, dy.shape[-1]))
            return r
    Bo, Ho, Wo, Cout, lddy = _nhwc(dy)
    g, B, ld, Ct = _geom(x, KH, KW, stride, pad, Ho, Wo, 0, x2)
    Ncol = KH * KW * Ct
    out = ZEROS.zeros((Cout, Ncol), x.device)
    npix = B * Ho * Wo
    tiles = ((Cout + 127) // 128) * ((Ncol + 127) // 128)
    kw = dict(dtype=_code(x.dtype), out_dtype=DU_F32, a_mode=PLAIN_COL, b_mode=IM2COL_COL, M=Cout, N=Ncol, K=npix,
              A=dy.data_ptr(), lda=lddy, B=x.data_ptr(), ldb=ld, Cmat=out.data_ptr(), ldc=Ncol,
              split_k=_split_for(tiles, npix, 1024), geom=g)
    if not with_db:
        gemm_raw(**kw)
        return out
    if _WGRAD_COLSUM and x.dtype == torch.bfloat16 and gemm_route(**kw) == 1:
        db = ZEROS.zeros((Cout,), x.device)
        gemm_raw(a_colsum=db.data_ptr(), **kw)
        return out, db
    gemm_raw(**kw)
    return out, colsum(dy.view(-1, dy.shape[-1]))


# statistics of 32-channel outputs from the convolution's own epilogue too (round 3: the epilogue reduction is DPP row shifts now, cheap
# enough that the separate statistics pass over the 134 MB tensors of the 512^2 stages costs more); DINOUNET_CONV_STATS32=0: A-B aid
_STATS32 = _ab_env("DINOUNET_CONV_STATS32", "1") != "0"


class _Conv2d(torch.autograd.Function):
    """Conv2d on NHWC, optional second input = fused channel concat (dinounet_training.py:614).  With want_stats the forward also
    returns the per-tile partial channel statistics of the output (emitted by the LDS-tiled kernel's epilogue) so the following
    InstanceNorm / BatchNorm needs no extra pass; None when the shape went through the implicit-GEMM path."""

    @staticmethod
    def forward(ctx, x, x2, w, bias, stride, pad, want_stats):
        KH, KW = w.shape[2], w.shape[3]
        wp = PACK.get(w, PK_CONV_FWD, x.dtype)
        if wp is None:
            wp = pack_conv_weight(w, x.dtype)
        part = None
        r = None
        if KH == 3 and KW == 3 and stride == 1 and pad == 1:
            # the epilogue statistics pay off from 64 output channels up (measured: +35 us on the 32-channel 512^2 layers, where the
            # separate statistics pass costs ~30 us; -10 us on the 64/128-channel ones)
            r = conv3x3_halo(x, wp, _f32(bias), x2, want_stats and (w.shape[0] >= 64 or _STATS32))
        if r is not None:
            y, part = r
        else:
            y = conv_fwd(x, wp, _f32(bias), KH, KW, stride, pad, x2)
        ctx.save_for_backward(x, x2, w)
        ctx.conf = (KH, KW, stride, pad, bias is not None)
        ctx.wrefs = WGRAD.note_use(*([w] if bias is None else [w, bias]))
        if part is None:
            part = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)      # no zero tensor is built for the (never differentiated) statistics output
        return y, part

    @staticmethod
    def backward(ctx, dy, _dpart):
        x, x2, w = ctx.saved_tensors
        KH, KW, stride, pad, has_bias = ctx.conf
        dy = dy.contiguous()
        B, Hi, Wi, C1, _ = _nhwc(x)
        dx = dx2 = dw = db = None
        need_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
        if need_dx:
            done = False
            if KH == 3 and KW == 3 and stride == 1 and pad == 1 and dy.dtype == torch.bfloat16:
                wf = PACK.get(w, PK_CONV_DGRAD_FLIP, dy.dtype)            # (Cin_total, 9*Cout)
                if wf is None:
                    wf = pack_conv_weight_dgrad_flipped(w, dy.dtype)
                if x2 is None:
                    r = conv3x3_halo(dy, wf, None)
                    if r is not None:
                        dx, done = r[0], True
                else:
                    r1 = conv3x3_halo(dy, wf[:C1], None)
                    r2 = conv3x3_halo(dy, wf[C1:], None) if r1 is not None else None
                    if r1 is not None and r2 is not None:
                        dx, dx2, done = r1[0], r2[0], True
            if not done:
                wd = PACK.get(w, PK_CONV_DGRAD, dy.dtype)
                if wd is None:
                    wd = pack_conv_weight_dgrad(w, dy.dtype)
                dfull = conv_dgrad(dy, wd, KH, KW, stride, pad, Hi, Wi)
                if x2 is None:
                    dx = dfull
                else:
                    dx, dx2 = dfull[..., :C1], dfull[..., C1:]
        want_db = has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2]:
            r = None
            Bo, Ho, Wo, Cout, lddy = _nhwc(dy)
            # 3 x 3 layers the LDS-tiled weight-gradient kernel does not serve (128 output channels): in-place gather on the grouped launch
            if (KH, KW, stride, pad) == (3, 3, 1, 1) and dy.dtype == torch.bfloat16 and _lib.lib().du_conv3x3_wgrad_halo_blocks(
                    C1, C1 + (x2.shape[-1] if x2 is not None else 0), Cout, B, Hi, Wi) <= 0 and _pow2(Hi) and _pow2(Wi) and \
                    (_WGRAD_COLSUM or not want_db):
                grp = WGRAD.begin(ctx.wrefs)
                if grp is not None:
                    srcs = [(x, _nhwc(x)[4], C1)] + ([(x2, _nhwc(x2)[4], x2.shape[-1])] if x2 is not None else [])
                    r = _queue_conv_wgrad(grp, 3, dy, lddy, srcs, B, Hi, Wi, Cout, tuple(w.shape), want_db)
                    if r is None:
                        WGRAD.abort(grp)
                    elif not grp["ret"]:
                        r = (None, None)
            if r is not None:
                dw, db = r
            else:
                g = conv_wgrad(x, dy, KH, KW, stride, pad, x2, with_db=want_db)
                if want_db:
                    g, db = g
                dw = g.view(w.shape[0], KH, KW, w.shape[1]).permute(0, 3, 1, 2).contiguous()
        elif want_db:
            db = colsum(dy.view(-1, dy.shape[-1]))
        return dx, dx2, dw, db, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=1, x2=None):
    return _Conv2d.apply(x, x2, w, bias, stride, pad, False)[0]


def conv2d_stats(x, w, bias=None, stride=1, pad=1, x2=None):
    """conv2d that also hands back the output's partial channel statistics (or None) for norm_act(..., stats_part=...)."""
    y, part = _Conv2d.apply(x, x2, w, bias, stride, pad, True)
    return y, (part if part.numel() else None)


_DGRAD_NT = _ab_env("DINOUNET_DGRAD_NT", "1") == "1"


class _Linear(torch.autograd.Function):
    """y = x w^T + b [* row_scale per sample] [+ residual]   on (rows, K) matrices."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, row_scale, rs_rows, out_dtype):
        wq = w
        if w.dtype != x.dtype:
            wq = PACK.get(w, PK_CAST, x.dtype)
            if wq is None:
                wq = w.to(x.dtype)
            elif wq.dim() != 2:
                wq = wq.view(w.shape[0], -1)
        y = mm(x, wq, bias=_f32(bias), residual=residual, row_scale=row_scale, rs_rows=rs_rows, out_dtype=out_dtype)
        # W^T (K, N) from the weight pack turns the data gradient into a contraction-contiguous product, i.e. one the 256-wide
        # multi-phase NT kernels (gemm_p8.hip) serve: 39 of the 58 linear data gradients of a dinounet_l step move there, -0.75 ms per
        # step (against the 128 x 128 kernel of round 1 it was a wash; DINOUNET_DGRAD_NT=0 restores the transpose-read ROW x COL kernel)
        ctx.wT = PACK.get(w, PK_TRANSPOSE, x.dtype) if (_DGRAD_NT and w.dtype != x.dtype and w.shape[0] % 64 == 0) else None
        ctx.save_for_backward(x, wq, row_scale)
        # the weight gradient may be computed late in the backward pass (WgradQueue) when this is the only use of w in the step
        ctx.wrefs = WGRAD.note_use(*([w] if bias is None else [w, bias]))
        ctx.rs_rows = rs_rows
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wq, row_scale = ctx.saved_tensors
        dres = dy if ctx.has_res else None
        dyc = dy if dy.stride(1) == 1 else dy.contiguous()
        if dyc.dtype != x.dtype:
            dyc = cast(dyc, x.dtype)
        ks = None
        if row_scale is not None:
            # d(s . y) = s . dy per sample (DropPath).  Where the kernels can, s is applied inside them -- on the OUTPUT rows of the data
            # gradient (epilogue) and on the CONTRACTION rows of the weight gradient (fragment scale, zero-scale tiles skipped) -- instead of
            # materialising s . dy (a read + write of dy and two launches per DropPath layer)
            if _KSCALE and dyc.dtype == torch.bfloat16 and ctx.rs_rows % 64 == 0 and ctx.wT is not None:
                ks = (row_scale, ctx.rs_rows)
            else:
                dyc = (dyc.view(row_scale.numel(), ctx.rs_rows, -1) * row_scale.view(-1, 1, 1).to(dyc.dtype)).view(dyc.shape)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.wT is not None:
                dx = mm(dyc, ctx.wT, row_scale=ks[0], rs_rows=ks[1]) if ks is not None else mm(dyc, ctx.wT)
            else:
                dx = mm_dgrad(dyc, wq)
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            grp = WGRAD.begin(ctx.wrefs)
            if want_db:
                dw, db = mm_wgrad(dyc, x, with_colsum=True, k_scale=ks, defer=grp)
            else:
                dw = mm_wgrad(dyc, x, k_scale=ks, defer=grp)
            if grp is not None and not grp["ret"]:     # a later use of the same weight in this pass: added into the first one's result
                dw = db = None
        elif want_db:
            dys = dyc if ks is None else (dyc.view(row_scale.numel(), ctx.rs_rows, -1) * row_scale.view(-1, 1, 1).to(dyc.dtype)).view(dyc.shape)
            db = colsum(dys)
        return dx, dw, db, dres, None, None, None


def linear(x, w, bias=None, residual=None, row_scale=None, rs_rows=0, out_dtype=None):
    """x (..., K) -> (..., N); leading dims are flattened (last dim contiguous, uniform row stride)."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    r2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
    y = _Linear.apply(x2, w, bias, r2, row_scale, rs_rows, out_dtype)
    return y.view(*shp[:-1], w.shape[0])


class _LinearCat(torch.autograd.Function):
    """y = x [w1; w2]^T + [b1; b2]: two linear layers reading the same input as ONE product (MSDeformAttn's sampling_offsets +
    attention_weights, ms_deform_attn.py:188-189; FAPM's shared + specific bases, dinounet_training.py:423-424).  The concatenated
    bf16 weight comes from the per-step weight pack; the gradients are split back to the four parameters."""

    @staticmethod
    def forward(ctx, x, w1, w2, b1, b2, out_dtype):
        y = _linear_cat_fwd(ctx, x, w1, w2, b1, b2, out_dtype)
        ctx.save_for_backward(*ctx.lc_saved)
        ctx.lc_saved = None
        return y

    @staticmethod
    def backward(ctx, dy):
        x = ctx.saved_tensors[0]
        dyc = dy if dy.stride(1) == 1 else dy.contiguous()
        if dyc.dtype != x.dtype:
            dyc = cast(dyc, x.dtype)
        return (*_linear_cat_bwd(ctx, dyc), None)


def _linear_cat_fwd(ctx, x, w1, w2, b1, b2, out_dtype, product=None):
    """product: callable(x, wq, bq) -> result instead of the plain mm (ops._OffsetsPrep: the product with the msda_prep epilogue)"""
    n1 = w1.shape[0]
    wq = PACK.get((w1.reshape(n1, -1), w2.reshape(w2.shape[0], -1)), PK_CAST, x.dtype)
    if wq is None:
        wq = torch.cat([w1.reshape(n1, -1), w2.reshape(w2.shape[0], -1)], 0).to(x.dtype)
    bq = None
    if b1 is not None:
        bq = PACK.get((b1, b2), PK_CAST, torch.float32)
        if bq is None:
            bq = torch.cat([b1, b2], 0).float()
    y = mm(x, wq, bias=bq, out_dtype=out_dtype) if product is None else product(x, wq, bq)
    # [w1; w2]^T from the weight pack: the data gradient becomes a contraction-contiguous product (see _Linear)
    ctx.wT = None
    if _DGRAD_NT and w1.dtype != x.dtype and (n1 + w2.shape[0]) % 64 == 0:
        ctx.wT = PACK.get((w1.reshape(n1, -1), w2.reshape(w2.shape[0], -1)), PK_TRANSPOSE, x.dtype)
    ctx.lc_saved = (x, wq)
    ctx.conf = (n1, tuple(w1.shape), tuple(w2.shape), b1 is not None)
    ctx.wrefs = WGRAD.note_use(*([w1, w2] if b1 is None else [w1, w2, b1, b2]))
    return y


def _linear_cat_bwd(ctx, dyc, x=None, wq=None):
    """-> (dx, dw1, dw2, db1, db2) for dyc in the GEMM dtype"""
    if x is None:
        x, wq = ctx.saved_tensors[:2]
    n1, s1, s2, has_bias = ctx.conf
    dx = None
    if ctx.needs_input_grad[0]:
        dx = mm(dyc, ctx.wT) if ctx.wT is not None else mm_dgrad(dyc, wq)
    db1 = db2 = None
    grp = WGRAD.begin(ctx.wrefs)
    if has_bias:
        dw, db = mm_wgrad(dyc, x, with_colsum=True, defer=grp)
        db1, db2 = db[:n1], db[n1:]
    else:
        dw = mm_wgrad(dyc, x, defer=grp)
    if grp is not None and not grp["ret"]:
        return dx, None, None, None, None
    return dx, dw[:n1].view(s1), dw[n1:].view(s2), db1, db2


def linear_cat(x, w1, w2, b1=None, b2=None, out_dtype=None):
    """x (..., K) -> (..., N1 + N2)."""
    shp = x.shape
    y = _LinearCat.apply(x.reshape(-1, shp[-1]), w1, w2, b1, b2, out_dtype)
    return y.view(*shp[:-1], y.shape[-1])


def _pixel_rows(x, rows, Cc, ld):
    """NHWC tensor (pixel stride ld) as a (rows, C) matrix for the linear ops.  A plain view when x is dense: as_strided on a tensor
    that requires grad records AsStridedBackward, whose backward zero-fills a flat buffer of the whole storage and copies the
    gradient into it (two extra passes per 1x1 convolution, seen in the glue trace); only channel slices of wider tensors need it."""
    if ld == Cc and x.is_contiguous():
        return x.view(rows, Cc)
    return x.as_strided((rows, Cc), (ld, 1), x.storage_offset())


def conv1x1_cat(x, w1, w2, b1=None, b2=None, out_dtype=None):
    """two 1x1 convolutions of the same NHWC input as one product; w (Cout_i, Cin, 1, 1)."""
    B, H, W, Cc, ld = _nhwc(x)
    xm = _pixel_rows(x, B * H * W, Cc, ld)
    y = _LinearCat.apply(xm, w1, w2, b1, b2, out_dtype)
    return y.view(B, H, W, y.shape[-1])


class _FAPMProject(torch.autograd.Function):
    """FAPM projection of one scale (dinounet_training.py:423-429) as ONE autograd node:
        z2 = x [W_shared; W_specific]^T + b        (one product, 2R columns)
        gb = z2[:, :R] W_film^T + b_film           ([gamma | beta])
        z  = gamma * z2[:, R:] + beta
    The backward fills the two halves of dz2 in place (FiLM backward kernel + the film generator's data gradient written through a
    strided output), so the channel slices cost no zero-fill / copy / add launches of the autograd engine."""

    @staticmethod
    def forward(ctx, x, ws, wp, bs, bp, wf, bf):
        B, H, W, Cc, ld = _nhwc(x)
        rows = B * H * W
        R = ws.shape[0]
        xm = x.as_strided((rows, Cc), (ld, 1), x.storage_offset())
        dt = x.dtype
        wq = PACK.get((ws.reshape(R, -1), wp.reshape(R, -1)), PK_CAST, dt)
        if wq is None:
            wq = torch.cat([ws.reshape(R, -1), wp.reshape(R, -1)], 0).to(dt)
        bq = None
        if bs is not None:
            bq = PACK.get((bs, bp), PK_CAST, torch.float32)
            if bq is None:
                bq = torch.cat([bs, bp], 0).float()
        wfq = PACK.get(wf.reshape(2 * R, -1), PK_CAST, dt)
        if wfq is None:
            wfq = wf.reshape(2 * R, -1).to(dt)
        ctx.wqT = ctx.wfT = None
        if _DGRAD_NT and ws.dtype != dt and (2 * R) % 64 == 0:
            ctx.wqT = PACK.get((ws.reshape(R, -1), wp.reshape(R, -1)), PK_TRANSPOSE, dt)
            ctx.wfT = PACK.get(wf.reshape(2 * R, -1), PK_TRANSPOSE, dt)
        ctx.wrefs = (WGRAD.note_use(*([ws] if bs is None else [ws, bs])), WGRAD.note_use(*([wp] if bp is None else [wp, bp])),
                     WGRAD.note_use(*([wf] if bf is None else [wf, bf])))
        z2 = mm(xm, wq, bias=bq)
        gb = mm(z2[:, :R], wfq, bias=_f32(bf))
        z = torch.empty((rows, R), dtype=dt, device=x.device)
        _lib.check(_lib.lib().du_film_fwd(_code(dt), _p(gb), _p(z2), _p(z), rows, R, _st()), "du_film_fwd")
        ctx.save_for_backward(xm, wq, wfq, z2, gb)
        ctx.conf = (R, tuple(ws.shape), tuple(wp.shape), tuple(wf.shape), bs is not None, bf is not None, (B, H, W, Cc))
        return z.view(B, H, W, R)

    @staticmethod
    def backward(ctx, dz):
        xm, wq, wfq, z2, gb = ctx.saved_tensors
        R, s_ws, s_wp, s_wf, has_b, has_bf, (B, H, W, Cc) = ctx.conf
        rows = xm.shape[0]
        dt = xm.dtype
        dz = dz.contiguous().view(rows, R)
        dgb = torch.empty((rows, 2 * R), dtype=dt, device=dz.device)
        dz2 = torch.empty((rows, 2 * R), dtype=dt, device=dz.device)
        _lib.check(_lib.lib().du_film_bwd(_code(dt), _p(dz), _p(gb), _p(z2), _p(dgb), _p(dz2), rows, R, _st()), "du_film_bwd")
        if ctx.wfT is not None:                                  # d z_shared, written into the left half of dz2
            mm(dgb, ctx.wfT, out=dz2[:, :R])
        else:
            mm_dgrad(dgb, wfq, out=dz2[:, :R])
        # weight gradients: queued (WgradQueue) in three groups -- the shared basis (used by all four scales, dinounet_training.py:423: its
        # contributions add into one buffer), this scale's basis, the film generator
        g_s, g_p, g_f = (WGRAD.begin(r) for r in ctx.wrefs)
        if has_bf:
            dwf, dbf = mm_wgrad(dgb, z2[:, :R], with_colsum=True, defer=g_f if g_f is not None else False)
        else:
            dwf, dbf = mm_wgrad(dgb, z2[:, :R], defer=g_f if g_f is not None else False), None
        if g_f is not None and not g_f["ret"]:
            dwf = dbf = None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (mm(dz2, ctx.wqT) if ctx.wqT is not None else mm_dgrad(dz2, wq)).view(B, H, W, Cc)
        dbs = dbp = None
        if g_s is None and g_p is None:
            if has_b:
                dw, db = mm_wgrad(dz2, xm, with_colsum=True)
                dbs, dbp = db[:R], db[R:]
            else:
                dw = mm_wgrad(dz2, xm)
            dws, dwp = dw[:R], dw[R:]
        else:
            rs = mm_wgrad(dz2[:, :R], xm, with_colsum=has_b, defer=g_s if g_s is not None else False)
            rp = mm_wgrad(dz2[:, R:], xm, with_colsum=has_b, defer=g_p if g_p is not None else False)
            (dws, dbs), (dwp, dbp) = (rs, rp) if has_b else ((rs, None), (rp, None))
            if g_s is not None and not g_s["ret"]:
                dws = dbs = None
            if g_p is not None and not g_p["ret"]:
                dwp = dbp = None
        return (dx, None if dws is None else dws.view(s_ws), None if dwp is None else dwp.view(s_wp), dbs, dbp,
                None if dwf is None else dwf.view(s_wf), dbf)


def fapm_project(x, ws, wp, bs, bp, wf, bf):
    """x NHWC (B,H,W,D) -> FiLM-modulated z (B,H,W,R); ws / wp: shared / scale-specific basis (R,D,1,1), wf: film generator (2R,R,1,1)."""
    return _FAPMProject.apply(x, ws, wp, bs, bp, wf, bf)


def conv1x1(x, w, bias=None, out_dtype=None):
    """1x1 conv on NHWC = linear over pixels.  w (Cout, Cin, 1, 1)."""
    B, H, W, Cc, ld = _nhwc(x)
    xm = _pixel_rows(x, B * H * W, Cc, ld)
    y = _Linear.apply(xm, w.view(w.shape[0], -1), bias, None, None, 0, out_dtype)
    return y.view(B, H, W, w.shape[0])


class _ConvT2x2(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) on NHWC as GEMM [pixels x Cin] . [Cin x 4 Cout] + pixel-shuffle store."""

    @staticmethod
    def forward(ctx, x, w, bias, residual=None):
        B, H, W, Cin, ld = _nhwc(x)
        Cout = w.shape[1]
        wp = PACK.get(w, PK_CONVT_FWD, x.dtype)
        if wp is None:
            wp = w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).to(x.dtype).contiguous()
        out = torch.empty((B, 2 * H, 2 * W, Cout), dtype=x.dtype, device=x.device)
        _, _, _, _, ldc = _nhwc(out)
        b4 = None
        if bias is not None:                               # bias per (tap, co) column: [b; b; b; b] from the weight pack (no repeat launch)
            b4 = PACK.get((bias, bias, bias, bias), PK_CAST, torch.float32)
            if b4 is None:
                b4 = _f32(bias).repeat(4)
        ldr = 0
        if residual is not None:                       # tensor of the output shape added in the epilogue (dinov3_adapter.py:467)
            Br, Hr, Wr, Cr, ldr = _nhwc(residual)
            assert (Br, Hr, Wr, Cr) == (B, 2 * H, 2 * W, Cout) and residual.dtype == x.dtype
        gemm_raw(dtype=_code(x.dtype), out_dtype=_code(out.dtype), a_mode=PLAIN_ROW, b_mode=PLAIN_ROW, M=B * H * W,
                 N=4 * Cout, K=Cin, A=x.data_ptr(), lda=ld, B=wp.data_ptr(), ldb=Cin, Cmat=out.data_ptr(), ldc=ldc,
                 bias=_dp(b4), residual=_dp(residual), ldr=ldr, store_mode=STORE_PIXEL_SHUFFLE2, ps=(H, W, Cout))
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.wrefs = WGRAD.note_use(*([w] if bias is None else [w, bias]))
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, Cin, ld = _nhwc(x)
        Cout = w.shape[1]
        if not (dy.stride(3) == 1 and dy.stride(1) == dy.shape[2] * dy.stride(2)):
            dy = dy.contiguous()
        dx = dw = db = None
        g, _, lddy, _ = _geom(dy, 2, 2, 2, 0, H, W, 0)
        if ctx.needs_input_grad[0]:
            wd = PACK.get(w, PK_CONVT_DGRAD, dy.dtype)
            if wd is None:
                wd = w.permute(0, 2, 3, 1).reshape(Cin, 4 * Cout).to(dy.dtype).contiguous()
            dx = torch.empty((B, H, W, Cin), dtype=dy.dtype, device=dy.device)
            gemm_raw(dtype=_code(dy.dtype), out_dtype=_code(dx.dtype), a_mode=IM2COL_ROW, b_mode=PLAIN_ROW, M=B * H * W,
                     N=Cin, K=4 * Cout, A=dy.data_ptr(), lda=lddy, B=wd.data_ptr(), ldb=4 * Cout, Cmat=dx.data_ptr(),
                     ldc=Cin, geom=g)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        queued = None
        if ctx.needs_input_grad[1] and dy.dtype == torch.bfloat16 and _pow2(W) and (_WGRAD_COLSUM or not want_db):
            # grouped launch, dy gathered in place, result written in the parameter's (Cin, Cout, 2, 2) layout; the same ConvTranspose
            # applied twice in a forward (LearnableUpsampleBlock's loop, dinounet_training.py:259-261) adds into one buffer
            grp = WGRAD.begin(ctx.wrefs)
            if grp is not None:
                queued = _queue_conv_wgrad(grp, 2, x, ld, [(dy, lddy, Cout)], B, H, W, Cin, tuple(w.shape), want_db)
                if queued is None:
                    WGRAD.abort(grp)
                elif not grp["ret"]:
                    queued = (None, None)
        if queued is not None:
            dw, db = queued
            want_db = False
        elif ctx.needs_input_grad[1]:
            gw = ZEROS.zeros((Cin, 4 * Cout), dy.device)
            npix = B * H * W
            tiles = ((Cin + 127) // 128) * ((4 * Cout + 127) // 128)
            kw = dict(dtype=_code(dy.dtype), out_dtype=DU_F32, a_mode=PLAIN_COL, b_mode=IM2COL_COL, M=Cin, N=4 * Cout,
                      K=npix, A=x.data_ptr(), lda=ld, B=dy.data_ptr(), ldb=lddy, Cmat=gw.data_ptr(), ldc=4 * Cout,
                      split_k=_split_for(tiles, npix, 1024), geom=g)
            if want_db and _WGRAD_COLSUM and dy.dtype == torch.bfloat16 and gemm_route(**kw) in (1, 5):
                # bias gradient = sum over all dY pixels = sum over (input pixel, tap) of the gathered operand: taken inside the kernel
                db = ZEROS.zeros((Cout,), dy.device)
                gemm_raw(b_colsum=db.data_ptr(), **kw)
                want_db = False
            else:
                gemm_raw(**kw)
            dw = gw.view(Cin, 2, 2, Cout).permute(0, 3, 1, 2).contiguous()
        if want_db:
            Bo, Ho, Wo, Co, ldo = _nhwc(dy)
            db = colsum(dy.as_strided((Bo * Ho * Wo, Co), (ldo, 1), dy.storage_offset()))
        return dx, dw, db, (dy if ctx.has_res else None)


def conv_transpose2x2(x, w, bias=None, residual=None):
    """ConvTranspose2d(k=2, s=2) on NHWC [+ residual of the output shape, fused into the epilogue]."""
    return _ConvT2x2.apply(x, w, bias, residual)


# ----------------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------------
def chan_stats(x, G):
    """NHWC x -> (G, C, 2) fp32 sums (sum, sum of squares) over each group's pixels (G=B: InstanceNorm, G=1: BatchNorm)."""
    B, H, W, Cc, ld = _nhwc(x)
    P = (B // G) * H * W
    sums = torch.empty((G, Cc, 2), dtype=torch.float32, device=x.device)
    ws, n = _reduce_ws(x.dtype, G, P, Cc, x.device)
    _lib.check(_lib.lib().du_chan_stats(_code(x.dtype), _p(x), ld, _p(sums), G, P, Cc, _p(ws), n, _st()), "du_chan_stats")
    return sums, P


def _norm_fwd(x, mean, rstd, w, b, G, P, act):
    B, H, W, Cc, ld = _nhwc(x)
    y = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().du_norm_act_fwd(_code(x.dtype), _p(x), ld, _p(y), Cc, _p(mean), _p(rstd), _p(w), _p(b), G, P, Cc, act,
                                          _st()), "du_norm_act_fwd")
    return y


class _NormAct(torch.autograd.Function):
    """InstanceNorm2d / (Sync)BatchNorm2d + activation on NHWC.

    kind 'in': statistics per (sample, channel) (torch InstanceNorm2d, biased var, eps).
    kind 'bn': training -> batch statistics (all-reduced over the process group when one is given = SyncBatchNorm,
               dinov3_adapter.py:242,361) and running-stat update (momentum 0.1, unbiased var); eval -> running stats."""

    @staticmethod
    def forward(ctx, x, w, b, kind, act, eps, training, running_mean, running_var, momentum, group, stats_part=None):
        B, H, W, Cc, ld = _nhwc(x)
        wf, bf = _f32(w), _f32(b)
        use_batch = (kind == "in") or training
        count = None
        if use_batch:
            G = B if kind == "in" else 1
            P = (B // G) * H * W
            count = float(P)
            mean = torch.empty((G, Cc), dtype=torch.float32, device=x.device)
            rstd = torch.empty((G, Cc), dtype=torch.float32, device=x.device)
            upd = kind == "bn" and running_mean is not None
            rm, rv = (_p(running_mean), _p(running_var)) if upd else (None, None)
            synced = kind == "bn" and group is not None and sync_active(group)
            L = _lib.lib()
            if not synced:
                # totals, mean / rstd and the running statistics in ONE launch behind the partials (du_*_norm)
                if stats_part is not None:
                    # per-tile partials from the producing convolution's epilogue (tiles of one image are contiguous)
                    _lib.check(L.du_strip_finalize_norm(_p(stats_part), None, G, stats_part.shape[0] // G, Cc, count, eps, _p(mean), _p(rstd),
                                                        rm, rv, float(momentum), _st()), "du_strip_finalize_norm")
                else:
                    ws, n = _reduce_ws(x.dtype, G, P, Cc, x.device)
                    _lib.check(L.du_chan_stats_norm(_code(x.dtype), _p(x), ld, None, G, P, Cc, _p(ws), n, count, eps, _p(mean), _p(rstd),
                                                    rm, rv, float(momentum), _st()), "du_chan_stats_norm")
            else:
                if stats_part is not None:
                    sums = torch.empty((G, Cc, 2), dtype=torch.float32, device=x.device)
                    _lib.check(L.du_strip_finalize(_p(stats_part), _p(sums), G, stats_part.shape[0] // G, Cc, _st()), "du_strip_finalize")
                else:
                    sums, P = chan_stats(x, G)
                sums = sums.clone()
                _small_all_reduce(sums, group, "syncbn_fwd")   # equal per-rank batch (TRN:322-327 splits evenly)
                count = count * torch.distributed.get_world_size(group)
                _lib.check(L.du_norm_stats_finalize(_p(sums), count, eps, _p(mean), _p(rstd), G, Cc, rm, rv, float(momentum), _st()),
                           "du_norm_stats_finalize")
        else:
            G, P = 1, B * H * W
            mean = running_mean.float().view(1, Cc)
            rstd = torch.rsqrt(running_var.float() + eps).view(1, Cc)
        mean, rstd = mean.contiguous(), rstd.contiguous()
        y = _norm_fwd(x, mean, rstd, wf, bf, G, P, act)
        ctx.save_for_backward(x, mean, rstd, wf, bf)
        ctx.conf = (G, P, act, use_batch, count, kind, group)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, wf, bf = ctx.saved_tensors
        G, P, act, use_batch, count, kind, group = ctx.conf
        B, H, W, Cc, ld = _nhwc(x)
        dy = dy.contiguous()
        L = _lib.lib()
        bs = torch.empty((G, Cc, 2), dtype=torch.float32, device=x.device)
        ws, n = _reduce_ws(x.dtype, G, P, Cc, x.device)
        dw = torch.empty(Cc, dtype=torch.float32, device=x.device)
        db = torch.empty(Cc, dtype=torch.float32, device=x.device)
        _lib.check(L.du_norm_act_bwd_stats_grads(_code(x.dtype), _p(x), ld, _p(dy), Cc, _p(mean), _p(rstd), _p(wf), _p(bf), _p(bs), _p(dw),
                                                 _p(db), G, P, Cc, act, _p(ws), n, _st()), "du_norm_act_bwd_stats_grads")
        bsr = bs
        if kind == "bn" and use_batch and group is not None and sync_active(group):
            bsr = bs.clone()
            _small_all_reduce(bsr, group, "syncbn_bwd")
        dx = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
        _lib.check(L.du_norm_act_bwd_dx(_code(x.dtype), _p(x), ld, _p(dy), Cc, _p(dx), Cc, _p(mean), _p(rstd), _p(wf), _p(bf),
                                        _p(bsr), G, P, Cc, act, float(count or 0.0), 1 if use_batch else 0, _st()),
                   "du_norm_act_bwd_dx")
        return dx, dw, db, None, None, None, None, None, None, None, None, None


def norm_act(x, w, b, kind, act=ACT_NONE, eps=1e-5, training=True, running_mean=None, running_var=None, momentum=0.1,
             group=None, stats_part=None):
    return _NormAct.apply(x, w, b, kind, act, eps, training, running_mean, running_var, momentum, group, stats_part)


class _SyncBNMulti(torch.autograd.Function):
    """Several independent SyncBatchNorm layers (the adapter's four output norms, dinov3_adapter.py:361-364,479-482) with ONE
    statistics all-reduce in the forward and ONE in the backward instead of one per layer: the per-layer (sum, sum of squares) /
    (sum dy, sum dy*xhat) vectors are produced straight into slices of a single flat buffer.  (The six BatchNorms of the SPM stem
    cannot be packed: each one normalises the input of the next convolution.)"""

    @staticmethod
    def forward(ctx, group, act, n, *args):
        xs, ws, bs = args[:n], args[n:2 * n], args[2 * n:3 * n]
        mods = args[3 * n]                       # [(eps, running_mean, running_var, momentum)] * n
        L = _lib.lib()
        world = torch.distributed.get_world_size(group)
        geo = [_nhwc(x) for x in xs]
        offs, tot = [], 0
        for (_, _, _, Cc, _) in geo:
            offs.append(tot)
            tot += 2 * Cc
        flat = torch.empty(tot, dtype=torch.float32, device=xs[0].device)
        for x, (B, H, W, Cc, ld), o in zip(xs, geo, offs):
            wsb, nws = _reduce_ws(x.dtype, 1, B * H * W, Cc, x.device)
            _lib.check(L.du_chan_stats(_code(x.dtype), _p(x), ld, C.c_void_p(flat.data_ptr() + 4 * o), 1, B * H * W, Cc, _p(wsb), nws, _st()),
                       "du_chan_stats")
        _small_all_reduce(flat, group, "syncbn_fwd")
        ys, saved = [], []
        for j, (x, (B, H, W, Cc, ld), o) in enumerate(zip(xs, geo, offs)):
            eps, rm, rv, mom = mods[j]
            count = float(B * H * W * world)
            mean = torch.empty((1, Cc), dtype=torch.float32, device=x.device)
            rstd = torch.empty((1, Cc), dtype=torch.float32, device=x.device)
            _lib.check(L.du_norm_stats_finalize(C.c_void_p(flat.data_ptr() + 4 * o), count, eps, _p(mean), _p(rstd), 1, Cc,
                                                _p(rm) if rm is not None else None, _p(rv) if rm is not None else None, float(mom), _st()),
                       "du_norm_stats_finalize")
            wf, bf = _f32(ws[j]), _f32(bs[j])
            ys.append(_norm_fwd(x, mean, rstd, wf, bf, 1, B * H * W, act))
            saved += [x, mean, rstd, wf, bf]
        ctx.save_for_backward(*saved)
        ctx.conf = (group, act, n, world, geo, offs, tot)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        group, act, n, world, geo, offs, tot = ctx.conf
        sv = ctx.saved_tensors
        L = _lib.lib()
        dev = sv[0].device
        flat = torch.empty(tot, dtype=torch.float32, device=dev)
        dws, dbs, dyc = [], [], []
        for j in range(n):
            x, mean, rstd, wf, bf = sv[5 * j:5 * j + 5]
            B, H, W, Cc, ld = geo[j]
            dy = dys[j].contiguous()
            dyc.append(dy)
            wsb, nws = _reduce_ws(x.dtype, 1, B * H * W, Cc, dev)
            bsj = C.c_void_p(flat.data_ptr() + 4 * offs[j])
            _lib.check(L.du_norm_act_bwd_stats(_code(x.dtype), _p(x), ld, _p(dy), Cc, _p(mean), _p(rstd), _p(wf), _p(bf), bsj, 1, B * H * W,
                                               Cc, act, _p(wsb), nws, _st()), "du_norm_act_bwd_stats")
            dw = torch.empty(Cc, dtype=torch.float32, device=dev)
            db = torch.empty(Cc, dtype=torch.float32, device=dev)
            _lib.check(L.du_norm_param_grads(bsj, _p(dw), _p(db), 1, Cc, _st()), "du_norm_param_grads")    # local sums: DDP averages them
            dws.append(dw); dbs.append(db)
        _small_all_reduce(flat, group, "syncbn_bwd")
        dxs = []
        for j in range(n):
            x, mean, rstd, wf, bf = sv[5 * j:5 * j + 5]
            B, H, W, Cc, ld = geo[j]
            dx = torch.empty((B, H, W, Cc), dtype=x.dtype, device=dev)
            _lib.check(L.du_norm_act_bwd_dx(_code(x.dtype), _p(x), ld, _p(dyc[j]), Cc, _p(dx), Cc, _p(mean), _p(rstd), _p(wf), _p(bf),
                                            C.c_void_p(flat.data_ptr() + 4 * offs[j]), 1, B * H * W, Cc, act, float(B * H * W * world), 1, _st()),
                       "du_norm_act_bwd_dx")
            dxs.append(dx)
        return (None, None, None, *dxs, *dws, *dbs, None)


def sync_bn_multi(xs, bns, act, group):
    """Training-mode SyncBatchNorm (+ activation) of several independent NHWC tensors with packed statistics collectives."""
    n = len(xs)
    mods = [(bn.eps, bn.running_mean, bn.running_var, bn.momentum if bn.momentum is not None else 0.1) for bn in bns]
    return list(_SyncBNMulti.apply(group, act, n, *xs, *[bn.weight for bn in bns], *[bn.bias for bn in bns], mods))


def layernorm_raw(x2d, w, b, eps, out_dtype, want_stats=False, out=None):
    rows, D, ld = _rows2d(x2d)
    if out is None:
        y = torch.empty((rows, D), dtype=out_dtype, device=x2d.device)
    else:                                    # caller-owned rows (the ViT's tap outputs, written by two half-batch chains)
        y = out
        assert y.shape == (rows, D) and y.dtype == out_dtype and y.is_contiguous()
    mean = rstd = None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.lib().du_layernorm_fwd(_code(x2d.dtype), _code(out_dtype), _p(x2d), ld, _p(w), _p(b), _p(y), D, _p(mean),
                                           _p(rstd), rows, D, eps, _st()), "du_layernorm_fwd")
    return y, mean, rstd


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last dim.  With `with_res` the input is also handed through as a second output (identity) to be used as the
    residual of the block (y = x + f(LN(x)), dinov3_adapter.py:142-148): the gradient that comes back on that branch is then added
    inside the LayerNorm backward kernel instead of by a separate full-size add of the autograd engine."""

    @staticmethod
    def forward(ctx, x, w, b, eps, with_res):
        xc = x.contiguous()
        wf, bf = _f32(w), _f32(b)
        y, mean, rstd = layernorm_raw(xc.view(-1, xc.shape[-1]), wf, bf, eps, xc.dtype, True)
        ctx.save_for_backward(xc, wf, mean, rstd)
        ctx.with_res = with_res
        ctx.set_materialize_grads(False)      # an unused residual output hands None (not a zero tensor) to backward
        if with_res:
            return y.view(x.shape), xc.view(x.shape).view_as(xc)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dres=None):
        xc, wf, mean, rstd = ctx.saved_tensors
        D = xc.shape[-1]
        if dy is None:
            return dres, None, None, None, None
        dyc = dy.contiguous()
        dx = torch.empty_like(xc)
        dwdb = torch.empty((2, D), dtype=torch.float32, device=xc.device)      # planar: dw, db (two-stage path)
        ws, n = _reduce_ws(xc.dtype, 1, xc.numel() // D, D, xc.device)
        dr = None
        if dres is not None:
            dr = dres.contiguous()
            if dr.dtype != xc.dtype:
                dr = cast(dr, xc.dtype)
        _lib.check(_lib.lib().du_layernorm_bwd(_code(xc.dtype), _p(xc), _p(dyc), _p(wf), _p(mean), _p(rstd), _p(dx), _p(dwdb),
                                               xc.numel() // D, D, _p(ws), n, _p(dr), _st()), "du_layernorm_bwd")
        return dx, dwdb[0], dwdb[1], None, None


def layer_norm(x, w, b, eps):
    return _LayerNorm.apply(x, w, b, eps, False)


def layer_norm_res(x, w, b, eps):
    """-> (LN(x), x): use the second output as the block's residual so its gradient is fused into the LayerNorm backward."""
    return _LayerNorm.apply(x, w, b, eps, True)


# ----------------------------------------------------------------------------------------------------
# multi-scale deformable attention
# ----------------------------------------------------------------------------------------------------
def msda_forward_raw(value, shapes, lsi, loc, attn):
    _req(value, shapes, lsi, loc, attn)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    e0 = PROFILE.start() if PROFILE is not None else None
    _lib.check(_lib.lib().du_msda_forward(_code(value.dtype), _p(value), _p(shapes), _p(lsi), _p(loc), _p(attn), _p(out), N, S, M, D,
                                          L, Lq, P, _st()), "du_msda_forward")
    if PROFILE is not None:       # algorithmic bytes: value + sampling locations + attention weights read once, the output written once
        PROFILE.stop("msda_fwd<gather>", e0, 8.0 * N * Lq * M * L * P * D,
                     float(value.numel() * value.element_size() + loc.numel() * 4 + attn.numel() * 4 + out.numel() * out.element_size()))
    return out


def msda_backward_raw(value, shapes, lsi, loc, attn, grad_out, gv_like_value=False):
    """-> (grad_value, grad_loc, grad_attn), fp32; gv_like_value: grad_value in value's dtype where the library can write it directly
    (du_msda_backward_bf16gv: no cast pass), else fp32 cast here."""
    _req(value, shapes, lsi, loc, attn, grad_out)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gl = torch.empty(loc.shape, dtype=torch.float32, device=value.device)
    ga = torch.empty(attn.shape, dtype=torch.float32, device=value.device)
    n = int(_lib.lib().du_msda_bwd_ws_elems(N, S, M, D, L, Lq, P))
    ws = torch.empty(max(n, 1), dtype=torch.float32, device=value.device)
    e0 = PROFILE.start() if PROFILE is not None else None

    def _prof(gv):                # algorithmic bytes: value, locations, weights, grad_out read once; the three gradients written once
        if PROFILE is not None:
            PROFILE.stop("msda_bwd<gather>", e0, 24.0 * N * Lq * M * L * P * D,
                         float(value.numel() * value.element_size() + loc.numel() * 8 + attn.numel() * 8
                               + grad_out.numel() * grad_out.element_size() + gv.numel() * gv.element_size()))
    if gv_like_value and value.dtype == torch.bfloat16:
        gvb = torch.empty((N, S, M, D), dtype=torch.bfloat16, device=value.device)
        rc = _lib.lib().du_msda_backward_bf16gv(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attn), _p(grad_out), _p(gvb), _p(gl), _p(ga),
                                                N, S, M, D, L, Lq, P, _p(ws), n, _st())
        if rc == 0:
            _prof(gvb)
            return gvb, gl, ga
        if rc != -2:
            _lib.check(rc, "du_msda_backward_bf16gv")
    gv = torch.empty((N, S, M, D), dtype=torch.float32, device=value.device)      # all three are fully written by the library
    _lib.check(_lib.lib().du_msda_backward(_code(value.dtype), _p(value), _p(shapes), _p(lsi), _p(loc), _p(attn), _p(grad_out), _p(gv),
                                           _p(gl), _p(ga), N, S, M, D, L, Lq, P, _p(ws), n, _st()), "du_msda_backward")
    _prof(gv)
    return gv, gl, ga


class _MSDA(torch.autograd.Function):
    """MSDeformAttnFunction (ms_deform_attn.py:28-68): value (N,S,M,D) in the activation dtype, sampling locations
    and attention weights fp32 (the reference casts everything to fp32, :30)."""

    @staticmethod
    def forward(ctx, value, shapes, lsi, loc, attn):
        value, loc, attn = value.contiguous(), loc.contiguous(), attn.contiguous()
        out = msda_forward_raw(value, shapes, lsi, loc, attn)
        ctx.save_for_backward(value, shapes, lsi, loc, attn)
        return out

    @staticmethod
    def backward(ctx, go):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        gv, gl, ga = msda_backward_raw(value, shapes, lsi, loc, attn, go.contiguous(), gv_like_value=True)
        return (gv if gv.dtype == value.dtype else gv.to(value.dtype)), None, None, gl, ga


def msda(value, shapes, lsi, loc, attn):
    return _MSDA.apply(value, shapes, lsi, loc, attn)


class _MSDAPrep(torch.autograd.Function):
    """offsets|logits (rows, M*P*3) -> sampling locations (rows,M,P,2) and softmaxed weights (rows,M,P), fp32."""

    @staticmethod
    def forward(ctx, raw, ref, Lq, M, P, Hs, Ws):
        rows, ncol, ld = _rows2d(raw)
        assert ncol == M * P * 3
        loc = torch.empty((rows, M, P, 2), dtype=torch.float32, device=raw.device)
        attn = torch.empty((rows, M, P), dtype=torch.float32, device=raw.device)
        _lib.check(_lib.lib().du_msda_prep(_code(raw.dtype), _p(raw), ld, _p(ref), _p(loc), _p(attn), rows, Lq, M, P, Hs, Ws, _st()),
                   "du_msda_prep")
        ctx.save_for_backward(attn)
        ctx.conf = (M, P, Hs, Ws, raw.dtype, ncol)
        return loc, attn

    @staticmethod
    def backward(ctx, gloc, gattn):
        (attn,) = ctx.saved_tensors
        M, P, Hs, Ws, dt, ncol = ctx.conf
        rows = attn.shape[0]
        graw = torch.empty((rows, ncol), dtype=dt, device=attn.device)
        _lib.check(_lib.lib().du_msda_prep_bwd(_code(dt), _p(attn), _p(gloc.contiguous()), _p(gattn.contiguous()), _p(graw), ncol, rows,
                                               M, P, Hs, Ws, _st()), "du_msda_prep_bwd")
        return graw, None, None, None, None, None, None


def msda_prep(raw, ref, Lq, M, P, Hs, Ws):
    return _MSDAPrep.apply(raw, ref, Lq, M, P, Hs, Ws)


_MSDA_PREP_FUSED = _ab_env("DINOUNET_MSDA_PREP_FUSED", "1") == "1"


class _OffsetsPrep(torch.autograd.Function):
    """MSDeformAttn's sampling_offsets + attention_weights product (ms_deform_attn.py:188-189, fp32 result as the reference's
    custom_fwd(cast_inputs=fp32), :30) and msda_prep (:190-197) as ONE autograd node: the fp32 (rows, M*P*3) matrix is an internal buffer,
    so its gradient can be written by du_msda_prep_bwd directly in the GEMM dtype -- as two nodes autograd demands an fp32 gradient for
    the fp32 tensor, i.e. a 33 MB fp32 write plus a cast pass per extractor and step (6 x 17 us, the `cast_kernel<float, bf16>` line of the
    round-5 trace)."""

    @staticmethod
    def forward(ctx, x, w1, w2, b1, b2, ref, Lq, M, P, Hs, Ws):
        rows = x.shape[0]
        ncol = M * P * 3
        loc = torch.empty((rows, M, P, 2), dtype=torch.float32, device=x.device)
        attn = torch.empty((rows, M, P), dtype=torch.float32, device=x.device)

        def product(xm, wq, bq):
            # round 6: the reference-point / softmax step in the product's epilogue (DU_STORE_MSDA_PREP): the (rows, M*P*3) fp32 matrix is
            # never written; where the library declines (another kernel family for this shape), the plain product + du_msda_prep
            _, K, lda = _rows2d(xm)
            Nw, _, ldb = _rows2d(wq)
            if _MSDA_PREP_FUSED and P == 4 and xm.dtype == torch.bfloat16 and Nw == ncol:
                kw = dict(dtype=DU_BF16, out_dtype=DU_F32, a_mode=PLAIN_ROW, b_mode=PLAIN_ROW, M=rows, N=Nw, K=K, A=xm.data_ptr(), lda=lda,
                          B=wq.data_ptr(), ldb=ldb, Cmat=loc.data_ptr(), ldc=Nw, bias=_dp(bq), store_mode=STORE_MSDA_PREP, ps=(Hs, Ws, Lq),
                          rope=(ref.data_ptr(), None, 0, 1.0), c2=attn.data_ptr())
                if gemm_route(**kw) == 3:
                    gemm_raw(**kw)
                    return None
            raw = mm(xm, wq, bias=bq, out_dtype=torch.float32)
            _, _, ld = _rows2d(raw)
            _lib.check(_lib.lib().du_msda_prep(_code(raw.dtype), _p(raw), ld, _p(ref), _p(loc), _p(attn), rows, Lq, M, P, Hs, Ws, _st()),
                       "du_msda_prep")
            return None

        assert w1.shape[0] + w2.shape[0] == ncol
        _linear_cat_fwd(ctx, x, w1, w2, b1, b2, torch.float32, product=product)
        xs, wq = ctx.lc_saved
        ctx.lc_saved = None
        ctx.save_for_backward(xs, wq, attn)
        ctx.prep = (M, P, Hs, Ws, ncol)
        return loc, attn

    @staticmethod
    def backward(ctx, gloc, gattn):
        x, wq, attn = ctx.saved_tensors
        M, P, Hs, Ws, ncol = ctx.prep
        rows = attn.shape[0]
        graw = torch.empty((rows, ncol), dtype=x.dtype, device=attn.device)
        _lib.check(_lib.lib().du_msda_prep_bwd(_code(x.dtype), _p(attn), _p(gloc.contiguous()), _p(gattn.contiguous()), _p(graw), ncol, rows,
                                               M, P, Hs, Ws, _st()), "du_msda_prep_bwd")
        return (*_linear_cat_bwd(ctx, graw, x, wq), None, None, None, None, None, None)


def offsets_prep(x, w1, w2, b1, b2, ref, Lq, M, P, Hs, Ws):
    """(N, Lq, C) queries -> sampling locations (N*Lq, M, P, 2) and softmaxed attention weights (N*Lq, M, P), fp32."""
    return _OffsetsPrep.apply(x.reshape(-1, x.shape[-1]), w1, w2, b1, b2, ref, Lq, M, P, Hs, Ws)


# ----------------------------------------------------------------------------------------------------
# small NHWC ops
# ----------------------------------------------------------------------------------------------------
class _DWConvSegs(torch.autograd.Function):
    """Depthwise 3x3 (+bias, +activation) over one or more image segments of a flat channel-last tensor.
    seg = (element offset, B, H, W, pixel stride ld, per-image stride bs).  Serves plain NHWC tensors
    (dinounet_training.py:235) and ConvFFN's DWConv (dinov3_adapter.py:99-109): one depthwise kernel applied to the
    three token ranges of a (B, N, C) tensor viewed as (2H x 2W), (H x W), (H/2 x W/2) grids, then GELU (:87)."""

    @staticmethod
    def forward(ctx, x, w, bias, segs, act):
        x = x.contiguous()
        Cc = w.shape[0]
        wf = _f32(w).view(Cc, 9)
        bf = _f32(bias)
        y = torch.empty_like(x)
        z = torch.empty_like(x) if act != ACT_NONE else None
        es = x.element_size()
        L = _lib.lib()
        for (off, B, h, ww, ld, bs) in segs:
            _lib.check(L.du_dwconv3x3_fwd(_code(x.dtype), C.c_void_p(x.data_ptr() + off * es), ld, bs, _p(wf), _p(bf),
                                          C.c_void_p(y.data_ptr() + off * es), ld, bs,
                                          None if z is None else C.c_void_p(z.data_ptr() + off * es), B, h, ww, Cc, act, _st()),
                       "du_dwconv3x3_fwd")
        ctx.save_for_backward(x, wf, z)
        ctx.conf = (segs, act, bias is not None, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wf, z = ctx.saved_tensors
        segs, act, has_bias, Cc = ctx.conf
        L = _lib.lib()
        es = x.element_size()
        code = _code(x.dtype)
        dy = dy.contiguous()
        if act != ACT_NONE:
            dz = torch.empty_like(dy)
            _lib.check(L.du_act_bwd(code, _p(z), _p(dy), _p(dz), dy.numel(), act, _st()), "du_act_bwd")
        else:
            dz = dy
        dx = torch.empty_like(x)
        dw = torch.empty((Cc, 9), dtype=torch.float32, device=x.device)      # written by the first segment, accumulated by the rest
        db = torch.empty(Cc, dtype=torch.float32, device=x.device) if has_bias else None
        for si, (off, B, h, ww, ld, bs) in enumerate(segs):
            _lib.check(L.du_dwconv3x3_bwd_data(code, C.c_void_p(dz.data_ptr() + off * es), ld, bs, _p(wf),
                                               C.c_void_p(dx.data_ptr() + off * es), ld, bs, B, h, ww, Cc, _st()),
                       "du_dwconv3x3_bwd_data")
            n = int(L.du_dwconv_wgrad_ws_elems(code, B, h, ww, Cc))
            ws = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
            _lib.check(L.du_dwconv3x3_bwd_weight(code, C.c_void_p(x.data_ptr() + off * es), ld, bs,
                                                 C.c_void_p(dz.data_ptr() + off * es), ld, bs, _p(dw), _p(db), B, h, ww, Cc,
                                                 _p(ws), n, 1 if si > 0 else 0, _st()), "du_dwconv3x3_bwd_weight")
        return dx, dw.view(Cc, 1, 3, 3), db, None, None


def dwconv3x3(x, w, bias=None, act=ACT_NONE):
    """x NHWC (B,H,W,C)."""
    x = x.contiguous()
    B, H, W, Cc = x.shape
    return _DWConvSegs.apply(x, w, bias, ((0, B, H, W, Cc, H * W * Cc),), act)


class _DWConvTokens(torch.autograd.Function):
    """ConvFFN's DWConv (dinov3_adapter.py:99-109) + GELU (:87) over the (B, 21 n, C) token pyramid: the three grids of every image in
    ONE launch per pass (forward, data gradient, weight gradient) instead of one per grid."""

    @staticmethod
    def forward(ctx, x, w, bias, H, W, act):
        x = x.contiguous()
        B, N, Cc = x.shape
        wf = _f32(w).view(Cc, 9)
        y = torch.empty_like(x)
        z = torch.empty_like(x) if act != ACT_NONE else None
        _lib.check(_lib.lib().du_dwconv3x3_tokens_fwd(_code(x.dtype), _p(x), _p(wf), _p(_f32(bias)), _p(y), _p(z), B, H, W, Cc, act, _st()),
                   "du_dwconv3x3_tokens_fwd")
        ctx.save_for_backward(x, wf, z)
        ctx.conf = (H, W, act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wf, z = ctx.saved_tensors
        H, W, act, has_bias = ctx.conf
        B, N, Cc = x.shape
        L = _lib.lib()
        code = _code(x.dtype)
        dy = dy.contiguous()
        if act != ACT_NONE:
            dz = torch.empty_like(dy)
            _lib.check(L.du_act_bwd(code, _p(z), _p(dy), _p(dz), dy.numel(), act, _st()), "du_act_bwd")
        else:
            dz = dy
        dx = torch.empty_like(x)
        _lib.check(L.du_dwconv3x3_tokens_bwd_data(code, _p(dz), _p(wf), _p(dx), B, H, W, Cc, _st()), "du_dwconv3x3_tokens_bwd_data")
        dw = torch.empty((Cc, 9), dtype=torch.float32, device=x.device)
        db = torch.empty(Cc, dtype=torch.float32, device=x.device) if has_bias else None
        n = int(L.du_dwconv_wgrad_ws_elems(code, B, N, 1, Cc))
        ws = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
        _lib.check(L.du_dwconv3x3_tokens_bwd_weight(code, _p(x), _p(dz), _p(dw), _p(db), B, H, W, Cc, _p(ws), n, _st()),
                   "du_dwconv3x3_tokens_bwd_weight")
        return dx, dw.view(Cc, 1, 3, 3), db, None, None, None


def dwconv_tokens(x, w, bias, H, W, act=ACT_GELU):
    """x (B, N, C) with N = 21 * (H*W/4) tokens (ConvFFN)."""
    B, N, Cc = x.shape
    assert N == 21 * ((H * W) // 4) and H % 2 == 0 and W % 2 == 0, (N, H, W)
    return _DWConvTokens.apply(x, w, bias, H, W, act)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, Cc = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
        idx = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().du_maxpool3x3s2_fwd(_code(x.dtype), _p(x), _p(y), _p(idx), B, H, W, Cc, _st()), "du_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, H, W, Cc, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, H, W, Cc, dt = ctx.shape
        dx = torch.empty((B, H, W, Cc), dtype=dt, device=dy.device)
        _lib.check(_lib.lib().du_maxpool3x3s2_bwd(_code(dt), _p(idx), _p(dy.contiguous()), _p(dx), B, H, W, Cc, _st()),
                   "du_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x):
    return _MaxPool.apply(x)


class _BilinearAdd(torch.autograd.Function):
    """out = base + bilinear_upsample(src), src carries no gradient (frozen ViT features, dinov3_adapter.py:472-476)."""

    @staticmethod
    def forward(ctx, src, base):
        B, Hs, Ws, Cc, lds = _nhwc(src)
        Bo, Ho, Wo, Co, ldb = _nhwc(base)
        out = torch.empty((Bo, Ho, Wo, Co), dtype=base.dtype, device=base.device)
        _lib.check(_lib.lib().du_bilinear_add_fwd(_code(src.dtype), _code(base.dtype), _p(src), lds, _p(base), ldb, _p(out), Co, B, Hs,
                                                  Ws, Ho, Wo, Cc, _st()), "du_bilinear_add_fwd")
        return out

    @staticmethod
    def backward(ctx, dy):
        return None, dy


def bilinear_add(src, base):
    return _BilinearAdd.apply(src, base)


class _BilinearResize(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) on NHWC with its data gradient (the tail of
    LearnableUpsampleBlock, dinounet_training.py:262-263: only reached when the target is not a power-of-two multiple of the input)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        B, Hs, Ws, Cc, lds = _nhwc(x)
        out = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().du_bilinear_add_fwd(_code(x.dtype), _code(x.dtype), _p(x), lds, None, 0, _p(out), Cc, B, Hs, Ws, Ho, Wo,
                                                  Cc, _st()), "du_bilinear_add_fwd")
        ctx.geo = (B, Hs, Ws, Ho, Wo, Cc)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, Hs, Ws, Ho, Wo, Cc = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty((B, Hs, Ws, Cc), dtype=dy.dtype, device=dy.device)
        _lib.check(_lib.lib().du_bilinear_resize_bwd(_code(dy.dtype), _p(dy), Cc, _p(dx), Cc, B, Hs, Ws, Ho, Wo, Cc, _st()),
                   "du_bilinear_resize_bwd")
        return dx, None, None


def bilinear_resize(x, size):
    """NHWC x -> (B, size[0], size[1], C), bilinear, align_corners=False."""
    return _BilinearResize.apply(x, int(size[0]), int(size[1]))


class _SqueezeExcite(torch.autograd.Function):
    """SqueezeExcitation (dinounet_training.py:210-225) with the residual add of :438 fused:
    y = x * sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2) [+ shortcut].  x, shortcut NHWC; weights fp32 (R,C) / (C,R)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, shortcut):
        B, H, W, Cc, ld = _nhwc(x)
        P = H * W
        R = w1.shape[0]
        L = _lib.lib()
        w1f, b1f, w2f, b2f = _f32(w1.reshape(R, Cc)), _f32(b1), _f32(w2.reshape(Cc, R)), _f32(b2)
        sums, _ = chan_stats(x, B)
        hidden = torch.empty((B, R), dtype=torch.float32, device=x.device)
        gate = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        _lib.check(L.du_se_gate_fwd(_p(sums), 1.0 / P, _p(w1f), _p(b1f), _p(w2f), _p(b2f), _p(hidden), _p(gate), B, Cc, R, _st()),
                   "du_se_gate_fwd")
        y = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
        lds = 0
        if shortcut is not None:
            Bs, Hs, Ws, Cs, lds = _nhwc(shortcut)
            assert (Bs, Hs, Ws, Cs) == (B, H, W, Cc) and shortcut.dtype == x.dtype
        _lib.check(L.du_se_scale_fwd(_code(x.dtype), _p(x), ld, _p(gate), _p(shortcut), lds, _p(y), Cc, B, P, Cc, _st()), "du_se_scale_fwd")
        ctx.save_for_backward(x, sums, hidden, gate, w1f, w2f)
        ctx.shapes = (w1.shape, w2.shape, shortcut is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sums, hidden, gate, w1f, w2f = ctx.saved_tensors
        w1s, w2s, has_sc = ctx.shapes
        B, H, W, Cc, ld = _nhwc(x)
        P = H * W
        R = w1f.shape[0]
        L = _lib.lib()
        dy = dy.contiguous()
        dsum = torch.empty((B, Cc, 2), dtype=torch.float32, device=x.device)
        ws, n = _reduce_ws(x.dtype, B, P, Cc, x.device)
        _lib.check(L.du_chan_dot(_code(x.dtype), _p(dy), Cc, _p(x), ld, _p(dsum), B, P, Cc, _p(ws), n, _st()), "du_chan_dot")
        dpool = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        dw1 = torch.empty((R, Cc), dtype=torch.float32, device=x.device)
        db1 = torch.empty(R, dtype=torch.float32, device=x.device)
        dw2 = torch.empty((Cc, R), dtype=torch.float32, device=x.device)
        db2 = torch.empty(Cc, dtype=torch.float32, device=x.device)
        _lib.check(L.du_se_gate_bwd(_p(dsum), _p(sums), 1.0 / P, _p(gate), _p(hidden), _p(w1f), _p(w2f), _p(dpool), _p(dw1), _p(db1),
                                    _p(dw2), _p(db2), B, Cc, R, _st()), "du_se_gate_bwd")
        dx = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device)
        _lib.check(L.du_se_scale_bwd(_code(x.dtype), _p(dy), Cc, _p(gate), _p(dpool), _p(dx), Cc, B, P, Cc, _st()), "du_se_scale_bwd")
        return dx, dw1.view(w1s), db1, dw2.view(w2s), db2, (dy if has_sc else None)


def squeeze_excite(x, w1, b1, w2, b2, shortcut=None):
    return _SqueezeExcite.apply(x, w1, b1, w2, b2, shortcut)


class _DiceCE(torch.autograd.Function):
    """DC_and_CE_loss of the reference trainer (compound_losses.py:8-56; MemoryEfficientSoftDiceLoss batch_dice, do_bg=False,
    smooth, dice.py:58-119) on fp32 NCHW logits, fused: one pass for the softmax sums, one pass for d loss / d logits.
    With a process group the three dice sums are all-reduced (AllGatherGrad + sum, utilities/ddp_allgather.py:25-48)."""

    @staticmethod
    def forward(ctx, logits, target, smooth, group):
        logits = logits.float().contiguous()
        B, K, H, W = logits.shape
        HW = H * W
        tgt = target.reshape(B, HW)
        if tgt.dtype != torch.int64:
            tgt = tgt.long()
        tgt = tgt.contiguous()
        L = _lib.lib()
        n = int(L.du_dice_ce_ws_elems(B, K, HW))
        if n <= 0:
            raise RuntimeError(f"dinounet_hip: fused Dice+CE supports 2..8 classes, got {K}")
        ws = torch.empty(n, dtype=torch.float32, device=logits.device)
        sums = torch.empty(1 + 3 * (K - 1), dtype=torch.float32, device=logits.device)
        _lib.check(L.du_dice_ce_sums(_p(logits), _p(tgt), _p(sums), B, K, HW, _p(ws), n, _st()), "du_dice_ce_sums")
        mult = 1.0
        if group is not None and torch.distributed.is_initialized():   # (dc_and_ce_loss passes a group only for a DDP loss)
            _small_all_reduce(sums[1:], group, "dice_sums")
            mult = float(torch.distributed.get_world_size(group))
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        coef = torch.empty(2 * (K - 1), dtype=torch.float32, device=logits.device)
        _lib.check(L.du_dice_ce_finish(_p(sums), _p(loss), _p(coef), K, B * HW, float(smooth), mult, _st()), "du_dice_ce_finish")
        ctx.save_for_backward(logits, tgt, coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        logits, tgt, coef = ctx.saved_tensors
        B, K, H, W = logits.shape
        dl = torch.empty_like(logits)
        gof = go.float().contiguous()
        _lib.check(_lib.lib().du_dice_ce_bwd(_p(logits), _p(tgt), _p(coef), _p(gof), _p(dl), B, K, H * W, _st()), "du_dice_ce_bwd")
        return dl, None, None, None


def dice_ce_loss(logits, target, smooth=1e-5, group=None):
    """logits (B,K,H,W) fp32, target (B,1,H,W) integer labels -> scalar loss (CE - mean soft dice)."""
    _req(logits, target)
    return _DiceCE.apply(logits, target, smooth, group)


# ----------------------------------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, dt, cpad=None):
    """fp32 NCHW image -> NHWC `dt` with channels zero-padded to cpad (>= C, multiple of 8)."""
    _req(x)
    x = x.float().contiguous()
    B, Cc, H, W = x.shape
    cp = cpad or Cc
    y = torch.empty((B, H, W, cp), dtype=dt, device=x.device)
    _lib.check(_lib.lib().du_nchw_to_nhwc_pad(_code(dt), _p(x), _p(y), B, Cc, H, W, cp, _st()), "du_nchw_to_nhwc_pad")
    return y


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, Cc, ld = _nhwc(x)
        y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().du_nhwc_to_nchw_f32(_code(x.dtype), _p(x), ld, _p(y), B, Cc, H, W, _st()), "du_nhwc_to_nchw_f32")
        ctx.dt = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.float().contiguous()
        B, Cc, H, W = dy.shape
        return nchw_to_nhwc(dy, ctx.dt)


def nhwc_to_nchw_f32(x):
    return _ToNCHW.apply(x)


class _SegHead(torch.autograd.Function):
    """The decoder's last 1x1 convolution (32 channels -> K <= 4 classes, seg_layers[-1]: dinounet_training.py:603-629) as ONE streaming
    pass: bf16 NHWC features -> fp32 NCHW logits; backward = ONE pass over the features for dx, dw and db (du_seg_head_*)."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, H, W, Cc, ld = _nhwc(x)
        K = w.shape[0]
        wf = _f32(w).reshape(K, Cc).contiguous()
        out = torch.empty((B, K, H, W), dtype=torch.float32, device=x.device)
        e0 = PROFILE.start() if PROFILE is not None else None
        _lib.check(_lib.lib().du_seg_head_fwd(_p(x), ld, _p(wf), _p(_f32(b)) if b is not None else None, _p(out), B, H * W, Cc, K, _st()),
                   "du_seg_head_fwd")
        if PROFILE is not None:
            PROFILE.stop("seg_head_fwd_kernel<bf16>", e0, 2.0 * B * H * W * Cc * K, 2.0 * B * H * W * Cc + 4.0 * B * H * W * K)
        ctx.save_for_backward(x, wf)
        ctx.meta = (w.shape, b is not None)
        return out

    @staticmethod
    def backward(ctx, dl):
        x, wf = ctx.saved_tensors
        w_shape, has_b = ctx.meta
        B, H, W, Cc, ld = _nhwc(x)
        K = wf.shape[0]
        dl = dl.float().contiguous()
        L = _lib.lib()
        n = (K * Cc + K + 1) // 2 * 2
        part = torch.empty((int(L.du_seg_head_bwd_blocks(B, H * W)), n), dtype=torch.float32, device=x.device)
        dwb = torch.empty(n, dtype=torch.float32, device=x.device)
        dx = torch.empty((B, H, W, Cc), dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        e0 = PROFILE.start() if PROFILE is not None else None
        _lib.check(L.du_seg_head_bwd(_p(x), ld, _p(wf), _p(dl), _p(dx), Cc, _p(part), _p(dwb), B, H * W, Cc, K, _st()), "du_seg_head_bwd")
        if PROFILE is not None:
            PROFILE.stop("seg_head_bwd_kernel<bf16>", e0, 4.0 * B * H * W * Cc * K, 4.0 * B * H * W * Cc + 4.0 * B * H * W * K)
        dw = dwb[:K * Cc].view(w_shape) if ctx.needs_input_grad[1] else None
        db = dwb[K * Cc:K * Cc + K] if (has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db


_SEG_HEAD = _ab_env("DINOUNET_SEG_HEAD", "1") != "0"          # A-B aid: 0 = the padded GEMM + layout passes of round 2


def seg_head_ok(x, K):
    """True when seg_head() serves this head: bf16 NHWC features with 32 channels (16-byte aligned pixels), 1..4 classes, on the GPU."""
    return (_SEG_HEAD and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[-1] == 32 and 1 <= K <= 4
            and x.stride(3) == 1 and x.stride(2) % 8 == 0 and x.stride(1) == x.shape[2] * x.stride(2) and x.stride(0) == x.shape[1] * x.stride(1)
            and x.data_ptr() % 16 == 0)


def seg_head(x, w, b):
    """logits (B, K, H, W) fp32 = 1x1 conv of x (B, H, W, 32) bf16 with w (K, 32, 1, 1), b (K)."""
    return _SegHead.apply(x, w, b)


def patchify16(x, dt):
    _req(x)
    x = x.float().contiguous()
    B, Cc, H, W = x.shape
    y = torch.empty((B * (H // 16) * (W // 16), Cc * 256), dtype=dt, device=x.device)
    _lib.check(_lib.lib().du_patchify16(_code(dt), _p(x), _p(y), B, Cc, H, W, _st()), "du_patchify16")
    return y


def cast(x, dt):
    _req(x)
    if x.dtype == dt:
        return x
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=dt, device=x.device)
    _lib.check(_lib.lib().du_cast(_code(xc.dtype), _code(dt), _p(xc), _p(y), xc.numel(), _st()), "du_cast")
    return y


# ----------------------------------------------------------------------------------------------------
# ViT attention (forward only: the backbone is frozen, dinov3_adapter.py:326,423)
# ----------------------------------------------------------------------------------------------------
# RoPE + head split in the qkv GEMM's epilogue (DU_STORE_QKV_ROPE).  Parity-tested, but measured NEUTRAL in the dinounet_l step (33.64 /
# 33.71 ms fused vs 33.56 ms with the separate du_qkv_rope_split pass: the epilogue's table loads, the per-row division and the 128-byte
# head-major store segments cost what the 100 MB pass saved), so it stays opt-in.
_QKV_FUSED = _ab_env("DINOUNET_QKV_FUSED", "0") == "1"
# round 6: RoPE + head split in the persistent kernel's drain (gemm_nt_pp_kernel<.., ROPE>, needs `grid`).  Built, parity-tested, and measured
# SLOWER than the plain persistent product + du_qkv_rope_split: x0.995 in the step (profiles/r06_ab_rope_drain_v1.txt) -- the drain has no
# registers for the 16-byte store form (it spills), and sixteen 8-byte stores per lane and tile cost more than the 21 us pass saves.  Opt-in.
_QKV_ROPE_DRAIN = _ab_env("DINOUNET_QKV_ROPE_DRAIN", "0") == "1"
# round 6: head-major store from the persistent kernel's drain (DU_STORE_QKV_HEADS, the ragged rows in the same launch) + du_qkv_rope_inplace:
# 67 MB of RoPE traffic instead of 101 MB.  x1.0041 in the step (profiles/r06_ab_qkv_heads_v2.txt); a first form that ran the 40 ragged
# rows as a product of their own + du_qkv_rope_split_rows measured x0.993 (r06_ab_qkv_heads_v1.txt: two more launches per block).
_QKV_HEADS = _ab_env("DINOUNET_QKV_HEADS", "1") == "1"


def qkv_attention(h, w, bias, sin, cos, B, N, H, Dh, prefix, workspace, grid=None):
    """attention(h w^T + bias) of one ViT block (layers/attention.py:88-118): h (B*N, D) normalised tokens -> (B*N, H*Dh).
    bf16, d_head 64: the qkv product's epilogue applies RoPE and the q scale and stores q / k / v head-major (DU_STORE_QKV_ROPE), so the
    (B*N, 3*H*Dh) matrix and the pass that re-read it (du_qkv_rope_split) do not exist; the few rows past the last full 256-row tile
    (M = 8 * 1029 = 32 * 256 + 40) go through a small plain product + du_qkv_rope_split_rows.  Otherwise: mm + attention().
    grid = (H_t, W_t): the caller states that sin / cos are RoPE tables of an H_t x W_t token grid (separable: dimensions 0..15 of a head
    depend on the token's row, 16..31 on its column, 32..63 repeat them, rope_position_encoding.py:98-104) -- round 6: the persistent
    kernel then applies the rotation in its drain from an 8 KB factorised table (gemm_nt_pp_kernel<.., ROPE>)."""
    dt = h.dtype
    M, D = h.shape
    if _QKV_HEADS and dt == torch.bfloat16 and Dh == 64 and M == B * N and M >= 256 and not _QKV_FUSED and not (_QKV_ROPE_DRAIN and grid is not None):
        # round 6: the persistent kernel's drain stores q / k / v head-major (DU_STORE_QKV_HEADS: the plain drain with another row offset),
        # du_qkv_rope_inplace rotates q and k where they lie: 67 MB instead of the 101 MB pass of du_qkv_rope_split; the same bits
        Npad = (N + 7) // 8 * 8
        key = ("qkv", dt, B, H, Npad, Dh)
        if key not in workspace:
            workspace[key] = torch.zeros((3, B, H, Npad, Dh), dtype=dt, device=h.device)
        qkv3 = workspace[key]
        q, k, v = qkv3
        qscale = Dh ** -0.5 * math.log2(math.e)
        _, _, lda = _rows2d(h)
        Nw, _, ldb = _rows2d(w)
        # (the <= 64 rows behind the last full 256-row tile ride in the same launch: du_gemm's tail units store head-major too)
        kw = dict(dtype=DU_BF16, out_dtype=DU_BF16, a_mode=PLAIN_ROW, b_mode=PLAIN_ROW, M=M, N=Nw, K=D, A=h.data_ptr(), lda=lda,
                  B=w.data_ptr(), ldb=ldb, Cmat=qkv3.data_ptr(), ldc=B * H * Npad * Dh, bias=_dp(bias), store_mode=STORE_QKV_HEADS,
                  ps=(N, Npad, H))
        if Nw == 3 * H * Dh and gemm_route(**kw) == 6:
            gemm_raw(**kw)
            L = _lib.lib()
            _lib.check(L.du_qkv_rope_inplace(DU_BF16, _p(q), _p(k), _p(sin), _p(cos), B, N, Npad, H, Dh, prefix, qscale, M, _st()),
                       "du_qkv_rope_inplace")
            out = torch.empty((M, H * Dh), dtype=dt, device=h.device)
            e0 = PROFILE.start() if PROFILE is not None else None
            _lib.check(L.du_attention_fwd(_p(q), _p(k), _p(v), _p(out), B, H, N, Npad, Dh, _st()), "du_attention_fwd")
            if PROFILE is not None:
                PROFILE.stop("attn_fwd_w64_kernel<bf16>", e0, 4.0 * B * H * N * N * Dh, 4.0 * B * H * N * Dh * 2)
            return out
    if (_QKV_FUSED or (_QKV_ROPE_DRAIN and grid is not None)) and dt == torch.bfloat16 and Dh == 64 and M == B * N and M >= 256:
        Npad = (N + 7) // 8 * 8
        key = ("qkv", dt, B, H, Npad, Dh)
        if key not in workspace:
            workspace[key] = torch.zeros((3, B, H, Npad, Dh), dtype=dt, device=h.device)
        qkv3 = workspace[key]
        q, k, v = qkv3
        r = M % 256
        M0 = M - r
        qscale = Dh ** -0.5 * math.log2(math.e)
        _, _, lda = _rows2d(h)
        Nw, _, ldb = _rows2d(w)
        kw = dict(dtype=DU_BF16, out_dtype=DU_BF16, a_mode=PLAIN_ROW, b_mode=PLAIN_ROW, M=M0, N=Nw, K=D, A=h.data_ptr(), lda=lda,
                  B=w.data_ptr(), ldb=ldb, Cmat=qkv3.data_ptr(), ldc=B * H * Npad * Dh, bias=_dp(bias), store_mode=STORE_QKV_ROPE,
                  ps=(N, Npad, H), rope=(sin.data_ptr(), cos.data_ptr(), prefix, qscale))
        if grid is not None and _QKV_ROPE_DRAIN:
            g = ConvGeom()
            g.Hi, g.Wi = int(grid[0]), int(grid[1])
            kw["geom"] = g
        route = gemm_route(**kw) if Nw == 3 * H * Dh else 0
        if route == 6 or (route == 4 and _QKV_FUSED):
            gemm_raw(**kw)
            L = _lib.lib()
            if r:
                tail = mm(h[M0:], w, bias=bias)
                _lib.check(L.du_qkv_rope_split_rows(DU_BF16, _p(tail), _p(q), _p(k), _p(v), _p(sin), _p(cos), B, N, Npad, H, Dh, prefix,
                                                    qscale, M0, r, _st()), "du_qkv_rope_split_rows")
            out = torch.empty((M, H * Dh), dtype=dt, device=h.device)
            e0 = PROFILE.start() if PROFILE is not None else None
            _lib.check(L.du_attention_fwd(_p(q), _p(k), _p(v), _p(out), B, H, N, Npad, Dh, _st()), "du_attention_fwd")
            if PROFILE is not None:
                PROFILE.stop(("attn_fwd_w64_kernel<bf16>" if Dh == 64 else "attn_fwd_kernel<bf16>"), e0, 4.0 * B * H * N * N * Dh, 4.0 * B * H * N * Dh * 2)
            return out
    return attention(mm(h, w, bias=bias), sin, cos, B, N, H, Dh, prefix, workspace)


def attention(qkv, sin, cos, B, N, H, Dh, prefix, workspace):
    """qkv (B*N, 3*H*Dh) -> (B*N, H*Dh).  bf16: fused flash kernel; fp32 (parity mode): QK^T / softmax / PV as batched
    MFMA GEMMs with materialised scores."""
    dt = qkv.dtype
    Npad = (N + 7) // 8 * 8
    key = ("qkv", dt, B, H, Npad, Dh)
    if key not in workspace:
        workspace[key] = torch.zeros((3, B, H, Npad, Dh), dtype=dt, device=qkv.device)
    q, k, v = workspace[key]
    L = _lib.lib()
    scale = Dh ** -0.5
    qscale = scale * math.log2(math.e) if dt == torch.bfloat16 else scale
    _lib.check(L.du_qkv_rope_split(_code(dt), _p(qkv), _p(q), _p(k), _p(v), _p(sin), _p(cos), B, N, Npad, H, Dh, prefix, qscale, _st()),
               "du_qkv_rope_split")
    out = torch.empty((B * N, H * Dh), dtype=dt, device=qkv.device)
    if dt == torch.bfloat16:
        e0 = PROFILE.start() if PROFILE is not None else None
        _lib.check(L.du_attention_fwd(_p(q), _p(k), _p(v), _p(out), B, H, N, Npad, Dh, _st()), "du_attention_fwd")
        if PROFILE is not None:
            PROFILE.stop(("attn_fwd_w64_kernel<bf16>" if Dh == 64 else "attn_fwd_kernel<bf16>"), e0, 4.0 * B * H * N * N * Dh, 4.0 * B * H * N * Dh * 2)
        return out
    skey = ("scores", B, H, Npad)
    if skey not in workspace:
        workspace[skey] = torch.empty((B * H, Npad, Npad), dtype=torch.float32, device=qkv.device)
    S = workspace[skey]
    gemm_raw(dtype=DU_F32, out_dtype=DU_F32, a_mode=PLAIN_ROW, b_mode=PLAIN_ROW, M=N, N=N, K=Dh, A=q.data_ptr(), lda=Dh,
             B=k.data_ptr(), ldb=Dh, Cmat=S.data_ptr(), ldc=Npad, batch=B * H, abs_=Npad * Dh, bbs=Npad * Dh, cbs=Npad * Npad)
    _lib.check(L.du_softmax_rows_f32(_p(S), B * H * Npad, N, Npad, _st()), "du_softmax_rows_f32")
    # O[b,n,h,:] = P[b,h] V[b,h]: one batched GEMM per batch element so the (B,N,H,Dh) output strides are expressible
    for b in range(B):
        gemm_raw(dtype=DU_F32, out_dtype=DU_F32, a_mode=PLAIN_ROW, b_mode=PLAIN_COL, M=N, N=Dh, K=Npad,
                 A=S[b * H].data_ptr(), lda=Npad, B=v[b].data_ptr(), ldb=Dh, Cmat=out[b * N].data_ptr(), ldc=H * Dh,
                 batch=H, abs_=Npad * Npad, bbs=Npad * Dh, cbs=Dh)
    return out
