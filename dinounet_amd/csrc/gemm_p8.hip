// bf16 "NT" GEMM for the large contraction-contiguous products (the frozen ViT's qkv / proj / fc1 / fc2, layers/attention.py:88-90,
// layers/ffn_layers.py:43-49): 256 x 256 x 64 tiles, 8 waves (2 x 4), v_mfma_f32_32x32x16_bf16, multi-phase LDS-DMA pipeline.
//
//   C[m][n] = epilogue( alpha * sum_k A[m][k] * B[n][k] )
//
// Why a second NT kernel: the 128 x 128 two-barrier kernel (gemm_glds.hip) moves 32 KB L2->LDS per 2.1 MFLOP and drains its DMA
// queue at every K step (vmcnt(0) + barrier); it tops out at ~650-700 TF/s on the ViT-L shapes.  Here
//   * a 256 x 256 tile halves the operand bytes per flop (64 KB per 8.4 MFLOP),
//   * the K step is cut into 4 phases, one 64 x 32 "quadrant" pair of the wave's 128 x 64 output each (8 MFMAs); every phase
//     prefetches the NEXT phase's fragments into a second register set and issues ONE 16 KB half-tile of LDS-DMA
//     (buffer_load_dwordx4 ... lds), so loads, LDS reads and MFMAs of different phases overlap,
//   * the DMA queue is never drained in the main loop: each phase ends with a COUNTED s_waitcnt vmcnt(12) (six half-tiles stay
//     in flight, ~1.5 K steps of latency cover) + lgkmcnt(0) + one raw s_barrier.
//
// LDS map (128 KB): buffer b = K-tile parity, 64 KB each = { A-half0, A-half1, B-half0, B-half1 }, a half = 128 tile rows x 64 k
// (16 KB, 128-B rows).  A-half i holds tile rows [128 i, 128 i + 128); wave (wm, wn) owns rows wm*64 + [0,64) of EACH A half and
// rows wn*32 + [0,32) of EACH B half, so quadrant (i, j) of every wave reads only A-half i and B-half j: once all waves have read a
// half (one barrier later) the DMA for tile t+2 may overwrite it.  The 16-byte chunks of a row are XOR-swizzled by ((row >> 1) & 7)
// on the SOURCE address (the DMA destination is lane-linear) and un-swizzled in the ds_read_b128 fragment address.
//
// Phase program of K-tile t (p = t & 1; fragment sets A0f/A1f and Bx/By, By <-> Bx swap roles every tile):
//   q0: MFMA (A0f, B0)  | ds_read  B1(t)   -> free B set | DMA B0(t+2) -> B-half0[p]   (last read in q3(t-1))
//   q1: MFMA (A0f, B1)  | ds_read  A1(t)   -> A1f        | DMA B1(t+2) -> B-half1[p]   (last read in q0(t))
//   q2: MFMA (A1f, B1)  | ds_read  A0(t+1) -> A0f        | DMA A1(t+2) -> A-half1[p]   (last read in q1(t))
//   q3: MFMA (A1f, B0)  | ds_read  B0(t+1) -> free B set | DMA A0(t+3) -> A-half0[1-p] (last read in q2(t))
// A half-tile is therefore issued seven phases before its first read.  Issue order: A0(t) B0(t) B1(t) A1(t) A0(t+1) ...; at the end
// of phase q the half-tile read in phase q+1 is the 7th youngest of the 13 outstanding => vmcnt(2 * 6).  The last two K-tile pairs
// use exact smaller counts (nothing is issued past the end of K).
//
// Rows past M / N read as zeros through the buffer descriptor's bounds check (their products are never stored).  K % 128 == 0, K >= 256.
//
// gemm_nt_p8n_kernel is the 256 x 128 sibling for products whose 256 x 256 tile count quantises badly on 256 CUs (ViT-L proj / fc2:
// 128 tiles, qkv: 384 = 1.5 rounds): 8 waves as 4 x 2 (64 x 64 per wave), two phases per K step (A-half 0 / 1 against the whole
// 128-row B tile), THREE 48 KB LDS buffers rotated at run time (a K-tile is issued five phases before its first read, counted
// vmcnt(6)).  Both kernels feed the MFMA with (B fragment, A fragment): the accumulator then holds, per lane, one output row and
// 4 consecutive columns per register quad, so the epilogue stages 8-byte (bf16, after bias / activation / LayerScale in registers)
// or 16-byte (fp32) pieces into a row-major LDS tile and writes whole 512-byte output rows (a lane-per-row store would touch 32
// cache lines per instruction).
#include <algorithm>
#include <vector>
#include "common.h"
#include "gemm_params.h"
#include "gemm_skinny_body.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int HALF_B = 128 * PBK * 2;          // 16 KB
constexpr int BUF_B = 4 * HALF_B;              // 64 KB
constexpr int STG_LD = PBN + 4;                // fp32 staging row (epilogue)

template <typename TC> struct Out4p;
template <> struct Out4p<float> {
  static __device__ __forceinline__ void load(const float* p, float* v) { float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Out4p<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
    bf16x4 t = __builtin_bit_cast(bf16x4, *(const uint2*)p);
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (float)t[j];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
    bf16x4 t;
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = (bf16_t)v[j];
    *(uint2*)p = __builtin_bit_cast(uint2, t);
  }
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int N> struct IC { static constexpr int value = N; };

__device__ __forceinline__ void wait_vm_count(int n) {      // n = LDS-DMA instructions that may stay in flight (even, <= 12)
  switch (n) {
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// XCD-aware tile order shared by both kernels: an XCD owns a run of tiles; inside it bands of group_m tile rows are walked column
// by column so the ~32 resident tiles of an XCD share few A / B panels through its L2.
__device__ __forceinline__ void tile_coords_of(const GemmParams& P, int nwg, int bid, int& tm, int& tn);
__device__ __forceinline__ void tile_coords(const GemmParams& P, int& tm, int& tn) {
  tile_coords_of(P, P.main_wgs ? P.main_wgs : (int)gridDim.x, blockIdx.x, tm, tn);
}
__device__ __forceinline__ void tile_coords_of(const GemmParams& P, int nwg, int bid, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;     // bijective
  if (P.group_m > 1) {
    const int band = P.group_m * P.tiles_n;
    const int g = tile / band, l = tile - g * band;
    const int first = g * P.group_m;
    const int gsz = min(P.tiles_m - first, P.group_m);
    tn = l / gsz; tm = first + (l - tn * gsz);
  } else {
    tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n;
  }
}

// ---- epilogue pieces.  Accumulator layout (MFMA fed with (B fragment, A fragment)): lane l holds output row (l & 31) of the 32 x 32
// block and columns 8 g + 4 (l >> 5) + e of it in registers 4 g + e. ----

// erf GELU of the bf16-output path (fc1 of the frozen ViT), two elements per call on the PACKED fp32 pipe.  The epilogue of a 256 x 256
// tile is 128 activations per lane with the matrix pipe idle (one workgroup per CU), so its VALU cycles are exposed: ~22 us of a 90 us
// fc1 product with the previous form (Abramowitz-Stegun 7.1.26: v_rcp + v_exp + ~12 full-rate VALU per element = ~80 cycles / wave).
//   gelu(v) = max(v, 0) - u * E(u),   u = min(|v|, 5.6),   E(u) = 0.5 * erfc(u / sqrt 2) = 2 ^ q(u)
// q = degree-7 least-squares (Chebyshev-node) fit of log2(0.5 erfc(u / sqrt 2)) on [0, 5.6] (tools: numpy.polynomial.chebyshev.fit, fp32
// Horner checked): RELATIVE error of E < 1.1e-5 everywhere (so the tiny negative tail keeps its relative accuracy), |gelu error| < 5e-7,
// both far inside the bf16 rounding of the stored value.  Past u = 5.6, E < 1.1e-8 and u * E is held there (|error| < 1e-7 |v|).
// Cost: 2 v_med3 + 4 v_pk_fma (8 packed FMAs / 2) + 1 v_exp per element = ~40 cycles / wave.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { f32x2 r; r.x = c; r.y = c; return r; }
__device__ __forceinline__ f32x2 gelu_pk(f32x2 v) {
  f32x2 u, m, e;
  u.x = __builtin_amdgcn_fmed3f(fabsf(v.x), 0.f, 5.6f); u.y = __builtin_amdgcn_fmed3f(fabsf(v.y), 0.f, 5.6f);
  m.x = __builtin_amdgcn_fmed3f(v.x, 0.f, __builtin_inff()); m.y = __builtin_amdgcn_fmed3f(v.y, 0.f, __builtin_inff());
  f32x2 q = splat2(-1.99582904e-06f);
  q = q * u + splat2(6.48327432e-05f);
  q = q * u + splat2(-9.56180914e-04f);
  q = q * u + splat2(8.60978469e-03f);
  q = q * u + splat2(-5.41717858e-02f);
  q = q * u + splat2(-4.58246213e-01f);
  q = q * u + splat2(-1.15134465e+00f);
  q = q * u + splat2(-9.99985061e-01f);
  e.x = __builtin_amdgcn_exp2f(q.x); e.y = __builtin_amdgcn_exp2f(q.y);
  return m - u * e;
}

// bf16 staging of one 32 x 32 block: bias (+ GELU) in registers, 4 columns packed into one ds_write_b64.  ACT == SWIGLU: the 4 columns
// are two (w1, w2) pairs of an interleaved projection -> 2 gated outputs, one ds_write_b32 at half the column offset
template <int ACT>
__device__ __forceinline__ void stage_block_bf16(const f32x16& a, const float4 (&bv)[4], bf16_t* stg, int ldb, int srow, int scol, int hi) {
#pragma unroll
  for (int g = 0; g < 4; g++) {
    float o[4] = {a[4 * g] + bv[g].x, a[4 * g + 1] + bv[g].y, a[4 * g + 2] + bv[g].z, a[4 * g + 3] + bv[g].w};
    if constexpr (ACT == DU_ACT_SWIGLU) {
      bf16x2 t;
      t[0] = (bf16_t)(o[0] * __builtin_amdgcn_rcpf(1.0f + __expf(-o[0])) * o[1]);
      t[1] = (bf16_t)(o[2] * __builtin_amdgcn_rcpf(1.0f + __expf(-o[2])) * o[3]);
      *(unsigned*)(stg + srow * ldb + ((scol + 8 * g + 4 * hi) >> 1)) = __builtin_bit_cast(unsigned, t);
    } else {
      if constexpr (ACT == DU_ACT_GELU) {
        f32x2 p0 = {o[0], o[1]}, p1 = {o[2], o[3]};
        p0 = gelu_pk(p0); p1 = gelu_pk(p1);
        o[0] = p0.x; o[1] = p0.y; o[2] = p1.x; o[3] = p1.y;
      }
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (bf16_t)o[e];
      *(uint2*)(stg + srow * ldb + scol + 8 * g + 4 * hi) = __builtin_bit_cast(uint2, t);
    }
  }
}
// DU_STORE_QKV_ROPE staging of one head's two 32-column halves (accumulator blocks lo = dims 0..31, hi = dims 32..63 of the SAME lane
// and register: rotate-half pairs never leave the lane): bias, RoPE for q / k rows past the prefix tokens
// (layers/attention.py:66-85: x * cos + rotate_half(x) * sin in fp32), q scale, bf16 pack
__device__ __forceinline__ void stage_head_rope(const GemmParams& P, const f32x16& lo, const f32x16& hi_, const float4 (&blo)[4],
                                                const float4 (&bhi)[4], bf16_t* stg, int ldb, int srow, int scol, int hi, int m, int which) {
  const int b = m / P.ps_H, t = m - b * P.ps_H;
  const bool rot = which < 2 && t >= P.rope_prefix && m < P.M;
  const float qs = which == 0 ? P.rope_qscale : 1.0f;
  const long trow = rot ? (long)(t - P.rope_prefix) * 64 : 0;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int d = 8 * g + 4 * hi;
    float x1[4] = {lo[4 * g] + blo[g].x, lo[4 * g + 1] + blo[g].y, lo[4 * g + 2] + blo[g].z, lo[4 * g + 3] + blo[g].w};
    float x2[4] = {hi_[4 * g] + bhi[g].x, hi_[4 * g + 1] + bhi[g].y, hi_[4 * g + 2] + bhi[g].z, hi_[4 * g + 3] + bhi[g].w};
    float o1[4], o2[4];
    if (rot) {
      const float4 c1 = *(const float4*)(P.rope_cos + trow + d), s1 = *(const float4*)(P.rope_sin + trow + d);
      const float4 c2 = *(const float4*)(P.rope_cos + trow + 32 + d), s2 = *(const float4*)(P.rope_sin + trow + 32 + d);
      const float cc1[4] = {c1.x, c1.y, c1.z, c1.w}, ss1[4] = {s1.x, s1.y, s1.z, s1.w};
      const float cc2[4] = {c2.x, c2.y, c2.z, c2.w}, ss2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
      for (int e = 0; e < 4; e++) { o1[e] = x1[e] * cc1[e] - x2[e] * ss1[e]; o2[e] = x2[e] * cc2[e] + x1[e] * ss2[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) { o1[e] = x1[e]; o2[e] = x2[e]; }
    }
    bf16x4 t1, t2;
#pragma unroll
    for (int e = 0; e < 4; e++) { t1[e] = (bf16_t)(o1[e] * qs); t2[e] = (bf16_t)(o2[e] * qs); }
    *(uint2*)(stg + srow * ldb + scol + d) = __builtin_bit_cast(uint2, t1);
    *(uint2*)(stg + srow * ldb + scol + 32 + d) = __builtin_bit_cast(uint2, t2);
  }
}
// this lane's bias values for the block whose first column is ncol0 (zeros without a bias); columns past N are clamped (never stored)
__device__ __forceinline__ void load_bias4(const GemmParams& P, int ncol0, int hi, float4 (&bv)[4]) {
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int n = ncol0 + 8 * g + 4 * hi;
    bv[g] = P.bias ? *(const float4*)(P.bias + (n < P.N ? n : 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void stage_block_f32(const f32x16& a, float* stg, int ldf, int srow, int scol, int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; g++)
    *(float4*)(stg + srow * ldf + scol + 8 * g + 4 * hi) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
}
// rows [0, nrows) of the bf16 staging tile -> C rows mrow0 + ...: 16 bytes per lane, whole rows per wave.  TBN = staged columns,
// n0 / ncols = first column / column count of C they map to (half the GEMM's for the SwiGLU gate)
template <int TBN, bool QKV = false, int NTHR = 512>
__device__ __forceinline__ void readout_bf16(const GemmParams& P, const bf16_t* stg, int ldb, int nrows, int mrow0, int n0, int ncols,
                                             bf16_t* Cb, int tid) {
  constexpr int C8 = TBN / 8;
  // QKV: 8 consecutive d of one head are 16 contiguous bytes whatever the plane stride
  const bool wide = ((((uintptr_t)Cb) & 15) == 0) && P.ldc % 8 == 0 && (QKV || P.store_mode == DU_STORE_PLAIN || P.ps_C % 8 == 0);
#pragma unroll 4
  for (int v = tid; v < nrows * C8; v += NTHR) {
    const int row = v / C8, c8 = v % C8;
    const int m = mrow0 + row, n = n0 + c8 * 8;
    if (m >= P.M || n >= ncols) continue;
    const uint4 t = *(const uint4*)(stg + row * ldb + c8 * 8);
    if ((P.dbg & 1) && t.x != 0x12345678u) continue;      // measurement aid (du_set_option key 3): staging without the global stores
    bf16_t* dst = Cb + (QKV ? qkv_heads_offset(P, m, n, P.ldc) : out_offset(P, m, n, P.ldc));
    if (wide && n + 8 <= ncols) *(uint4*)dst = t;
    else {
      *(uint2*)dst = make_uint2(t.x, t.y);
      if (n + 4 < ncols) *(uint2*)(dst + 4) = make_uint2(t.z, t.w);
    }
  }
}
// rows of the fp32 staging tile -> C with the full du_gemm epilogue (alpha, bias, act, gamma, row_scale, residual).  ACT: the
// activation when it is known at compile time (NONE / GELU), -1 = read P.act per element.  An item = W consecutive columns of one row
// (16 bytes of C: 4 fp32, or 8 bf16 when the row strides allow 16-byte accesses); four items per thread and trip, the residual loads of
// a trip all issued before the first is consumed.
template <typename TC, int W> struct RowVec;
template <> struct RowVec<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float* v) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct RowVec<bf16_t, 4> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* v) { Out4p<bf16_t>::load(p, v); }
  static __device__ __forceinline__ void store(bf16_t* p, const float* v) { Out4p<bf16_t>::store(p, v); }
};
template <> struct RowVec<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
    const bf16x8 t = __builtin_bit_cast(bf16x8, *(const uint4*)p);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (float)t[j];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = (bf16_t)v[j];
    *(uint4*)p = __builtin_bit_cast(uint4, t);
  }
};

template <typename TC, int TBN, int ACT, int W, int NTHR = 512>
__device__ __forceinline__ void readout_f32(const GemmParams& P, const float* stg, int ldf, int nrows, int mrow0, int n0, TC* Cb, const TC* Rb,
                                            int tid) {
  constexpr int CW = TBN / W, U = W == 8 ? 2 : 4;        // 16 residual floats in flight either way (the 256 x 256 kernel still holds
                                                          // half of its accumulators during the first pass)
  const int total = nrows * CW;
  for (int v0 = tid; v0 < total; v0 += NTHR * U) {
    float rr[U][W];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int v = v0 + u * NTHR;
      const int row = v / CW, cw = v % CW;
      const int m = mrow0 + row, n = n0 + cw * W;
      live[u] = v < total && m < P.M && n < P.N;
#pragma unroll
      for (int e = 0; e < W; e++) rr[u][e] = 0.f;
      if (Rb && live[u]) RowVec<TC, W>::load(Rb + out_offset(P, m, n, P.ldr), rr[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!live[u]) continue;
      const int v = v0 + u * NTHR;
      const int row = v / CW, cw = v % CW;
      const int m = mrow0 + row, n = n0 + cw * W;
      float o[W];
#pragma unroll
      for (int h = 0; h < W / 4; h++) {
        const float4 tt = *(const float4*)(stg + row * ldf + cw * W + 4 * h);
        o[4 * h] = tt.x * P.alpha; o[4 * h + 1] = tt.y * P.alpha; o[4 * h + 2] = tt.z * P.alpha; o[4 * h + 3] = tt.w * P.alpha;
        if (P.bias) {
          const float4 bb = *(const float4*)(P.bias + n + 4 * h);
          o[4 * h] += bb.x; o[4 * h + 1] += bb.y; o[4 * h + 2] += bb.z; o[4 * h + 3] += bb.w;
        }
      }
      if constexpr (ACT == DU_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < W; e++) o[e] = apply_act(o[e], DU_ACT_GELU);
      } else if constexpr (ACT < 0) {
#pragma unroll
        for (int e = 0; e < W; e++) o[e] = apply_act(o[e], P.act);
      }
      if (P.gamma) {
#pragma unroll
        for (int h = 0; h < W / 4; h++) {
          const float4 gg = *(const float4*)(P.gamma + n + 4 * h);
          o[4 * h] *= gg.x; o[4 * h + 1] *= gg.y; o[4 * h + 2] *= gg.z; o[4 * h + 3] *= gg.w;
        }
      }
      if (P.row_scale) {
        const float rs = P.row_scale[m / P.rs_rows];
#pragma unroll
        for (int e = 0; e < W; e++) o[e] *= rs;
      }
#pragma unroll
      for (int e = 0; e < W; e++) o[e] += rr[u][e];
      RowVec<TC, W>::store(Cb + out_offset(P, m, n, P.ldc), o);
    }
  }
}
// fp32 result WITH a residual, plain store, 256 x 128 tile (the ViT's proj / fc2: x + LayerScale(...)): the whole residual tile of this
// thread (16 x 16 bytes) is requested BEFORE the accumulators are staged, so its HBM latency is paid once and under the staging pass;
// readout_f32 keeps 4 loads in flight and pays it four times in a row (the epilogue was ~15 us of a 40 us proj product, r02 table).
struct ResidualTile { float4 r[16]; };
template <int TBN, int NTHR = 512>
__device__ __forceinline__ void residual_prefetch(const GemmParams& P, const float* Rb, int mrow0, int n0, int tid, ResidualTile& R) {
  constexpr int CW = TBN / 4;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int v = tid + u * NTHR;
    const int m = mrow0 + v / CW, n = n0 + (v % CW) * 4;
    R.r[u] = (m < P.M && n < P.N) ? *(const float4*)(Rb + (long)m * P.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int TBN, int NTHR = 512>
__device__ __forceinline__ void readout_f32_prefetched(const GemmParams& P, const float* stg, int ldf, int mrow0, int n0, float* Cb, int tid,
                                                       const ResidualTile& R) {
  constexpr int CW = TBN / 4;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int v = tid + u * NTHR;
    const int row = v / CW, cw = v % CW;
    const int m = mrow0 + row, n = n0 + cw * 4;
    if (m >= P.M || n >= P.N) continue;
    const float4 tt = *(const float4*)(stg + row * ldf + cw * 4);
    float o[4] = {tt.x * P.alpha, tt.y * P.alpha, tt.z * P.alpha, tt.w * P.alpha};
    if (P.bias) {
      const float4 bb = *(const float4*)(P.bias + n);
      o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
    }
    if (P.act != DU_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = apply_act(o[e], P.act);
    }
    if (P.gamma) {
      const float4 gg = *(const float4*)(P.gamma + n);
      o[0] *= gg.x; o[1] *= gg.y; o[2] *= gg.z; o[3] *= gg.w;
    }
    if (P.row_scale) {
      const float rs = P.row_scale[m / P.rs_rows];
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] *= rs;
    }
    *(float4*)(Cb + (long)m * P.ldc + n) = make_float4(o[0] + R.r[u].x, o[1] + R.r[u].y, o[2] + R.r[u].z, o[3] + R.r[u].w);
  }
}
template <typename TC, int TBN, int NTHR = 512>
__device__ __forceinline__ void readout_f32_any(const GemmParams& P, const float* stg, int ldf, int nrows, int mrow0, int n0, TC* Cb,
                                                const TC* Rb, int tid) {
  if constexpr (sizeof(TC) == 2) {
    // 8 bf16 per item when every access is 16-byte aligned and an item cannot straddle N or a pixel-shuffle segment
    const bool w8 = P.N % 8 == 0 && P.ldc % 8 == 0 && ((((uintptr_t)Cb) & 15) == 0) && (!Rb || (P.ldr % 8 == 0 && ((((uintptr_t)Rb) & 15) == 0))) &&
                    (P.store_mode == DU_STORE_PLAIN || P.ps_C % 8 == 0);
    if (w8) {
      if (P.act == DU_ACT_NONE) readout_f32<TC, TBN, DU_ACT_NONE, 8, NTHR>(P, stg, ldf, nrows, mrow0, n0, Cb, Rb, tid);
      else readout_f32<TC, TBN, -1, 8, NTHR>(P, stg, ldf, nrows, mrow0, n0, Cb, Rb, tid);
      return;
    }
  }
  if (P.act == DU_ACT_NONE) readout_f32<TC, TBN, DU_ACT_NONE, 4, NTHR>(P, stg, ldf, nrows, mrow0, n0, Cb, Rb, tid);
  else if (P.act == DU_ACT_GELU) readout_f32<TC, TBN, DU_ACT_GELU, 4, NTHR>(P, stg, ldf, nrows, mrow0, n0, Cb, Rb, tid);
  else readout_f32<TC, TBN, -1, 4, NTHR>(P, stg, ldf, nrows, mrow0, n0, Cb, Rb, tid);
}
// DU_STORE_MSDA_PREP: rows of the fp32 staging tile = [heads x 8 offsets | heads x 4 logits] of MSDeformAttn's offsets | weights product
// (ms_deform_attn.py:188-197): + bias, loc = ref[m % Lq] + offset / (W, H), attn = softmax over the 4 points -- the arithmetic of
// msda_prep_kernel (elementwise.hip), one (row, head) per thread and trip; the matrix itself is never written
template <int NTHR = 512>
__device__ __forceinline__ void readout_msda_prep(const GemmParams& P, const float* stg, int ldf, int nrows, int mrow0, int tid) {
  const int Mh = P.N / 12, Lq = P.ps_C;
  const float invW = 1.0f / (float)P.ps_W, invH = 1.0f / (float)P.ps_H;
  float* loc = (float*)P.C;
  for (int v = tid; v < nrows * Mh; v += NTHR) {
    const int row = v / Mh, h = v - row * Mh;
    const int m = mrow0 + row;
    if (m >= P.M) continue;
    const int q = m % Lq;
    const float rx = P.rope_sin[q * 2], ry = P.rope_sin[q * 2 + 1];
    const float* so = stg + row * ldf + h * 8;
    const float* sl = stg + row * ldf + Mh * 8 + h * 4;
    float off[8], lg[4];
#pragma unroll
    for (int e = 0; e < 8; e++) off[e] = so[e] + (P.bias ? P.bias[h * 8 + e] : 0.f);
#pragma unroll
    for (int e = 0; e < 4; e++) lg[e] = sl[e] + (P.bias ? P.bias[Mh * 8 + h * 4 + e] : 0.f);
    const float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    float ex[4], sm = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) { ex[e] = expf(lg[e] - mx); sm += ex[e]; }
    const float inv = 1.f / sm;
    float* lo = loc + ((long)m * Mh + h) * 8;
    *(float4*)lo = make_float4(rx + off[0] * invW, ry + off[1] * invH, rx + off[2] * invW, ry + off[3] * invH);
    *(float4*)(lo + 4) = make_float4(rx + off[4] * invW, ry + off[5] * invH, rx + off[6] * invW, ry + off[7] * invH);
    *(float4*)(P.C2 + ((long)m * Mh + h) * 4) = make_float4(ex[0] * inv, ex[1] * inv, ex[2] * inv, ex[3] * inv);
  }
}
// bf16 result whose epilogue is bias (+ GELU) only: staged as bf16 (the ViT's qkv and fc1)
__device__ __forceinline__ bool bf16_simple(const GemmParams& P, const void* Rb) {
  return !Rb && !P.gamma && !P.row_scale && P.alpha == 1.0f && (P.act == DU_ACT_NONE || P.act == DU_ACT_GELU || P.act == DU_ACT_SWIGLU);
}

constexpr int P8_STG_LDB = PBN + 8;            // bf16 staging row, elements (528 B: 16-byte aligned rows, 2-way ds_write_b64 conflicts)
constexpr int P8_STG_LDF = PBN + 4;            // fp32 staging row, floats (1040 B: conflict-free ds_write_b128)
constexpr int P8_LDS = 256 * P8_STG_LDB * 2;   // 135 168 B >= 2 * BUF_B and >= 128 * P8_STG_LDF * 4

// GA ("gather"): the ConvTranspose2d k2 s2 backward products read their im2col operand in place -- NT: A(m = input pixel, k = (tap, co)) =
// dy[out pixel (2y + tap/2, 2x + tap%2)][co] (data gradient); TN: B(k = input pixel, n = (tap, co)) likewise (weight gradient).
// The ragged tail of a tall product inside the tile kernel's own launch (du_gemm_bf16_fast: M = head + r, r <= 64 -- the ViT's 8 x 1029 =
// 32 x 256 + 40 rows): workgroups behind the main tiles run the K-parallel skinny program on rows [P.M, P.M + tail_rows).  They are
// dispatched when the first tiles retire and overlap the stragglers; as launches of their own the 40-row tails cost ~8 us each, 72 of
// them per dinounet_l step (profiles/r02_launch_counts_v5.txt).
template <typename TC, int NW = 8, bool SL = false>
__device__ __forceinline__ void p8_tail(const GemmParams& P, unsigned char* smem, int unit = -1) {
  SkinnyEpi E;
  E.qkv_H = 0; E.qkv_N = 0; E.qkv_Npad = 0;
  E.C = (TC*)P.C + (long)P.M * P.ldc; E.ldc = P.ldc;
  if (P.store_mode == DU_STORE_QKV_HEADS) { E.C = P.C; E.qkv_H = P.ps_C; E.qkv_N = P.ps_H; E.qkv_Npad = P.ps_W; }
  E.residual = P.residual ? (const void*)((const TC*)P.residual + (long)P.M * P.ldr) : nullptr; E.ldr = P.ldr;
  E.bias = P.bias; E.gamma = P.gamma; E.row_scale = P.row_scale;
  E.alpha = P.alpha; E.act = P.act; E.rs_rows = P.rs_rows; E.out_bf16 = sizeof(TC) == 2; E.row0 = P.M;
  E.slices = 1; E.slice = 0; E.slab = nullptr; E.cnt = nullptr;
  int u = unit >= 0 ? unit : (int)blockIdx.x - P.main_wgs;
  int k0 = 0, klen = P.K;
  if (SL && P.tail_slices > 1) {       // (column block, K slice): a long contraction on 32 CUs is a chain of TA-bound fragment loads (fc2: 18 us for 40 rows)
    const int cb = u / P.tail_slices;
    E.slices = P.tail_slices; E.slice = u - cb * P.tail_slices;
    klen = P.K / P.tail_slices; k0 = E.slice * klen;
    E.cnt = (int*)((char*)P.ks_ws + 65536) + cb;
    E.slab = (float*)((char*)P.ks_ws + 131072) + (long)cb * P.tail_slices * (64 * SK_BN);
    u = cb;
  }
  skinny_fused_body<NW, 2, SL, (NW == 8)>((const bf16_t*)P.a.p + (long)P.M * P.a.ld + k0, P.a.ld, (const bf16_t*)P.b.p + k0, P.b.ld, P.tail_rows, P.N, klen, E,
                       (float*)smem, u);
}

// Round 6: the ragged rows INSIDE the tile workgroups.  As extra workgroups behind the tiles (round 3) the tail units share the launch's
// dynamic LDS size (135-149 KB), so none of them is resident beside a tile workgroup: they are dispatched when the tiles retire and the
// launch ends with a round of its own -- 6.4-6.6 us per ViT product for ~2 us of work.  Here the workgroups that have
// finished their tiles run the units themselves (unit = 32 output columns x the whole contraction, workgroup w takes units w, w + G, ...):
// no dispatch, no second round.  The grid then has no extra workgroups: tail_rows > 0 and gridDim.x == main_wgs.
template <typename TC, int NW = 8>
__device__ __forceinline__ void p8_tail_inline(const GemmParams& P, unsigned char* smem, int w, int G) {
  const int nunits = (P.N + SK_BN - 1) / SK_BN;
  if (w >= nunits) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing of the tile program may still land in LDS (out-of-range ring requests write zeros)
  __syncthreads();
  for (int u = w; u < nunits; u += G) {
    p8_tail<TC, NW>(P, smem, u);
    __syncthreads();
  }
}

// workgroup -> XCD-contiguous linear index: the hardware places workgroup b on XCD b % 8; an XCD then owns a contiguous run of indices
__device__ __forceinline__ int xcd_linear_index() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// The tile program of both launch forms: the plain kernel below (one product per launch) and gemm_tn_group_kernel (several weight-gradient
// products per launch).  TN: `lin` = this workgroup's (split, tile) index inside the product, tiles of one split adjacent.
// GG (TN only, grouped launches): the B operand is gathered in place with a per-lane source offset -- ConvTranspose2d k2 s2 weight gradients
// of ANY channel count (B(k = input pixel, n = (tap, co)) = dy[output pixel (2y + tap / 2, 2x + tap % 2)][co]) and 3 x 3 / stride 1 / pad 1
// convolution weight gradients (B(k = pixel, n = (tap, ci)) = x[pixel + tap shift][ci], zero outside the image; two concatenated sources).
// GG = 2: the ConvTranspose form, GG = 3: the 3 x 3 convolution form (compile-time: the kernel is at its register limit, scalar ones included).
// KS (NT, fp32 result): the tile's contraction is cut in two halves run by the workgroups (16 g + x, 16 g + 8 + x) -- the same XCD -- so that
// a product with <= 128 tiles still occupies 256 CUs; the halves meet through fp32 slabs in P.ks_ws (ks_exchange below).
template <typename TC, int SCHED, bool TN, bool GA, int GG = 0, bool KS = false>
__device__ __forceinline__ void p8_tile_body(const GemmParams& P, unsigned char* smem, const int lin) {
  static_assert(GG == 0 || ((GG == 2 || GG == 3) && TN && !GA), "general gather: weight-gradient form only");
  static_assert(!KS || (!TN && !GA && sizeof(TC) == 4), "K-split pairs: plain NT products with an fp32 result");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  int nk = P.K / PBK;
  long kbeg = 0;                      // TN: first contraction row of this workgroup's split
  int my_split = 0;
  int ks_split = 0, ks_tile = 0, ks_kt0 = 0;
  if constexpr (TN) {
    // (split, tile): the linear index walks the tiles of one split before the next split, so the workgroups that share a K range (and
    // with it the A / B panels) sit on one XCD's L2
    const int ntiles = P.tiles_m * P.tiles_n;
    const int split = lin / ntiles, tile = lin - split * ntiles;
    tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n;
    // K-tile PAIRS (128 contraction rows) are dealt out evenly: the first `extra` splits take one more
    const int npairs = P.K / (2 * PBK), base = npairs / P.split_k, extra = npairs - base * P.split_k;
    const int first = split * base + min(split, extra);
    kbeg = (long)first * (2 * PBK);
    nk = 2 * (base + (split < extra ? 1 : 0));
    my_split = split;
  } else if constexpr (KS) {
    const int bid = blockIdx.x;
    ks_split = (bid >> 3) & 1;
    ks_tile = ((bid >> 4) << 3) | (bid & 7);
    tile_coords_of(P, P.main_wgs >> 1, ks_tile, tm, tn);
    nk >>= 1;
    ks_kt0 = ks_split * nk;
  } else {
    tile_coords(P, tm, tn);
  }
  const int m0 = tm * PBM, n0 = tn * PBN;
  const int batch = TN ? 0 : blockIdx.y;

  // ---- buffer descriptors based at the tile's first row (TN: first contraction row + first tile column); reads past the matrix fail
  // the bounds check and load zeros ----
  const bf16_t *Ab, *Bb;
  long abytes, bbytes;
  if constexpr (TN) {
    Ab = (const bf16_t*)P.a.p + (long)batch * P.a.bstride + kbeg * P.a.ld + m0;
    abytes = ((P.K - kbeg - 1) * P.a.ld + (P.M - m0)) * 2;
    if constexpr (GG != 0) {       // the whole source tensor; every lane carries its own (pixel shift, channel) offset
      Bb = (const bf16_t*)P.b.p;
      bbytes = P.b.bstride;   // (bytes of the source, set by the group kernel)
      if constexpr (GG == 3) {     // based one image row + one pixel before the tensor (see g_pv below)
        Bb -= (long)(P.b.Wi + 1) * P.b.ld;
        bbytes += (long)(P.b.Wi + 1) * P.b.ld * 2;
      }
    } else if constexpr (GA) {       // the whole tile lies in one tap (C % 256 == 0): base = the tap's pixel offset + first channel
      const int tapn = n0 / P.b.C, co0 = n0 - tapn * P.b.C;
      const long off = ((long)(tapn >> 1) * P.b.Wi + (tapn & 1)) * P.b.ld + co0;
      Bb = (const bf16_t*)P.b.p + off;
      bbytes = ((long)(P.K / (P.b.Ho * P.b.Wo)) * P.b.Hi * P.b.Wi * P.b.ld - off) * 2;
    } else {
      Bb = (const bf16_t*)P.b.p + (long)batch * P.b.bstride + kbeg * P.b.ld + n0;
      bbytes = ((P.K - kbeg - 1) * P.b.ld + (P.N - n0)) * 2;
    }
  } else {
    if constexpr (GA) {
      Ab = (const bf16_t*)P.a.p;
      abytes = (long)(P.M / (P.a.Ho * P.a.Wo)) * P.a.Hi * P.a.Wi * P.a.ld * 2;
    } else {
      Ab = (const bf16_t*)P.a.p + (long)batch * P.a.bstride + (long)m0 * P.a.ld;
      abytes = ((long)(P.M - m0) * P.a.ld - (P.a.ld - P.K)) * 2;
    }
    Bb = (const bf16_t*)P.b.p + (long)batch * P.b.bstride + (long)n0 * P.b.ld;
    bbytes = ((long)(P.N - n0) * P.b.ld - (P.b.ld - P.K)) * 2;
  }
  if (abytes > 0x7fffffffL) abytes = 0x7fffffffL;
  if (bbytes > 0x7fffff00L) bbytes = 0x7fffff00L;
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)abytes, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bbytes, 0x00020000);

  // staging: piece (round r, wave w) of a half is 1 KB of its LDS image.
  //   NT: tile rows (r*8 + w)*8 .. +8, lane l -> row + l/8, physical 16-byte chunk l%8 of the 128-byte row
  //   TN: the half is [64 contraction rows][128 outer elements] (256-byte rows); the piece = contraction rows (r*8 + w)*4 .. +4,
  //       lane l -> row + l/16, physical chunk l%16.  Logical chunk = physical ^ 4*(row & 3): the four rows a transpose read touches
  //       then sit on four different 64-byte bank groups (256-byte rows would otherwise all start on bank 0)
  unsigned va[2][2], vb[2][2];
  unsigned kstep_a = PBK * 2, kstep_b = PBK * 2;       // bytes from one K-tile to the next (buffer soffset)
  if constexpr (TN) {
    kstep_a = (unsigned)(P.a.ld * 2 * PBK); kstep_b = (unsigned)(P.b.ld * 2 * PBK);
    const int r4 = lane >> 4, lc = (lane & 15) ^ (4 * r4);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int krow = (r * 8 + wave) * 4 + r4;
        va[h][r] = (unsigned)krow * (unsigned)(P.a.ld * 2) + (h * 128 + lc * 8) * 2;
        vb[h][r] = (unsigned)krow * (unsigned)(P.b.ld * 2) + (h * 128 + lc * 8) * 2;
      }
  } else {
    const int sw = ((wave & 1) << 2) | (lane >> 4);          // ((row >> 1) & 7) of this lane's row
    const int lc = (lane & 7) ^ sw;                          // logical k chunk stored at this lane's physical chunk
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int row = h * 128 + (r * 8 + wave) * 8 + (lane >> 3);
        if constexpr (GA) {      // row = input pixel (b, y, x) -> byte offset of output pixel (b, 2y, 2x); rows past M re-read row M - 1
          const int m = min(m0 + row, P.M - 1);
          const int xo = m % P.a.Wo, t2 = m / P.a.Wo, yo = t2 % P.a.Ho, bb = t2 / P.a.Ho;
          va[h][r] = (unsigned)(((bb * P.a.Hi + 2 * yo) * P.a.Wi + 2 * xo)) * (unsigned)(P.a.ld * 2) + lc * 16;
        } else {
          va[h][r] = (unsigned)row * (unsigned)(P.a.ld * 2) + lc * 16;
        }
        vb[h][r] = (unsigned)row * (unsigned)(P.b.ld * 2) + lc * 16;
      }
  }
  // TN gather: contraction row = input pixel; with Wo a power of two, pixel p = (yy, x) reads output pixel (2 yy, 2 x) (+ the tap offset
  // folded into the descriptor base), yy = b * Ho + y
  int logW = 0;
  if constexpr ((GA || GG != 0) && TN) logW = __builtin_ctz(P.b.Wo);
  // GG: this lane's 16-byte chunk of half h is column n0 + h * 128 + lc * 8 = (tap, channel).
  //   GG == 2 (ConvT): g_off[h] = byte offset of (tap shift, channel) relative to the base pixel (1 = the column does not exist: byte
  //     offsets are even); the base pixel of a contraction row is computed per piece (a K-tile may span several image rows).
  //   GG == 3 (3 x 3): Ws % 64 == 0, so a K-tile (64 consecutive pixels) lies inside ONE image row: the source offset of a piece is
  //     a per-lane CONSTANT g_pv[h][r] = ((local pixel + tap shift) * row pitch + channel) * 2 plus the tile's scalar base offset (the
  //     buffer soffset) -- no address arithmetic in the loop.  The descriptor is based one image row + one pixel BEFORE the tensor so the
  //     (dy, dx) = (-1, -1) shifts stay non-negative (the voffset is range-checked as an unsigned number).  Zero padding: g_mask holds, per
  //     (h, r), 4 bits = "this lane's tap reaches above / below / left of / right of the image when the tile sits on that border and the
  //     lane's pixel on that edge"; the tile's own 4 border bits are scalar.  A hit sends the load past the descriptor's range (zeros).
  int g_off[2] = {1, 1};
  unsigned g_pv[2][2] = {{0u, 0u}, {0u, 0u}};
  unsigned g_mask = 0u;
  const unsigned g_ldb = (unsigned)(P.b.ld * 2);      // row pitch of the source in bytes
  if constexpr (GG != 0) {
    const int r4 = lane >> 4, lc = (lane & 15) ^ (4 * r4);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int col0 = n0 + h * 128, tap0 = col0 / P.b.C, ci0 = col0 - tap0 * P.b.C;        // wave-uniform: first column of the half
      const int ld = (int)P.b.ld;
      int ci = ci0 + lc * 8, tap = tap0;                     // a half spans at most 128 / 8 = 16 taps' worth of 8-channel chunks
      while (ci >= P.b.C) { ci -= P.b.C; tap++; }
      const bool exists = col0 + lc * 8 < P.N;
      if constexpr (GG == 2) {
        g_off[h] = exists ? (((tap >> 1) * P.b.Wi + (tap & 1)) * ld + ci) * 2 : 1;
      } else {
        const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int lr = (r * 8 + wave) * 4 + r4;            // this lane's pixel inside the K-tile
          g_pv[h][r] = exists ? (unsigned)(((lr + ty * P.b.Wi + tx) * ld + ci) * 2) : 0x7fffff00u;
          const unsigned bits = (ty == 0 ? 1u : 0u) | (ty == 2 ? 2u : 0u) | ((tx == 0 && lr == 0) ? 4u : 0u) | ((tx == 2 && lr == 63) ? 8u : 0u);
          g_mask |= bits << (4 * (2 * h + r));
        }
      }
    }
  }
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  // which: 0 = A-half0, 1 = A-half1, 2 = B-half0, 3 = B-half1 of K-tile kt, into buffer buf
  auto stage = [&](auto which_c, auto buf_c, int kt) {
    constexpr int which = decltype(which_c)::value, buf = decltype(buf_c)::value;
    constexpr int h = which & 1;
    // buffer soffset of K-tile kt.  NT gather: K runs over (tap, co); a K-tile lies inside one tap (C % 64 == 0, C a power of two).
    // (plain statements, not a helper lambda: a lambda call inside the LDS-DMA builtin's argument list makes the HOST pass drop the
    // kernel's stub without a diagnostic -- the library then fails to load with an undefined symbol)
    unsigned soff = (KS ? kt + ks_kt0 : kt) * (which < 2 ? kstep_a : kstep_b);
    if constexpr (GA && !TN && which < 2) {
      const int k = kt * PBK, tap = k >> P.a.logC, within = k - (tap << P.a.logC);
      soff = (unsigned)((((tap >> 1) * P.a.Wi + (tap & 1)) * (int)P.a.ld + within) * 2);
    }
    if constexpr ((GA || GG == 2) && TN && which >= 2) soff = 0;
    unsigned tile_bits = 0u;
    if constexpr (GG == 3 && which >= 2) {
      const int p0 = (int)kbeg + kt * PBK;                                  // first pixel of the K-tile (scalar): (image row y, x0)
      const int x0 = p0 & (P.b.Wo - 1), y = (p0 >> logW) & (P.b.Ho - 1);
      soff = (unsigned)p0 * g_ldb;
      tile_bits = (y == 0 ? 1u : 0u) | (y == P.b.Ho - 1 ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + PBK == P.b.Wo ? 8u : 0u);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      unsigned voff = which < 2 ? va[h][r] : vb[h][r];
      if constexpr (GG == 3 && which >= 2) {
        voff = ((g_mask >> (4 * (2 * h + r))) & tile_bits) ? 0x7fffff00u : g_pv[h][r];
      }
      if constexpr (GG == 2 && which >= 2) {
        const int krow = (int)kbeg + kt * PBK + (r * 8 + wave) * 4 + (lane >> 4);
        const int x = krow & (P.b.Wo - 1), yy = krow >> logW;
        const int bp = 4 * (yy << logW) + 2 * x;                  // 2 yy * Wi + 2 x with Wi = 2 Wo
        voff = g_off[h] != 1 ? (unsigned)bp * g_ldb + (unsigned)g_off[h] : 0x7fffff00u;       // past the descriptor's range: loads zeros
      }
      if constexpr (GA && TN && which >= 2) {
        const int r4 = lane >> 4, lc = (lane & 15) ^ (4 * r4);
        const int krow = (int)kbeg + kt * PBK + (r * 8 + wave) * 4 + r4;
        const int yy = krow >> logW, x = krow & (P.b.Wo - 1);
        voff = (unsigned)(2 * yy * P.b.Wi + 2 * x) * (unsigned)(P.b.ld * 2) + (h * 128 + lc * 8) * 2;
      }
      unsigned char* dst = smem + buf * BUF_B + which * HALF_B + (r * 8 + wave) * 1024;
      if constexpr (TN) {
        // inline asm: the compiler orders every ds_read_b64_tr_b16 behind ALL LDS-DMA it knows of (s_waitcnt vmcnt(0) in front of each
        // phase's reads, which drains the ring); unseen, the counted waits below are the only ordering.  M0 saved / restored inside.
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_base + buf * BUF_B + which * HALF_B + (r * 8 + wave) * 1024);
        unsigned keep;
        const auto srd = which < 2 ? ra : rb;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst) : "memory");
      } else {
        if constexpr (which < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, voff, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)dst, 16, voff, soff, 0, 0);
      }
    }
  };

  // fragment addresses.
  //   NT: row (lane & 31) of a 32-row block, logical chunk kk*2 + (lane >> 5), physical = logical ^ ((row >> 1) & 7)
  //   TN: ds_read_b64_tr_b16 pairs.  Lane (g = lane >> 4, p = lane & 15) reads 8 bytes of contraction row kk*16 + 8 (g >> 1) + (p >> 2)
  //       (+ 4 for the second read) at outer element i0 + 16 (g & 1) + 4 (p & 3) and receives outer i0 + (lane & 31), contraction
  //       kk*16 + 8 (lane >> 5) + 0..7 -- the same fragment as the NT ds_read_b128.  LT[x] = the lane's offset inside a half for the
  //       32-outer block x (0..3) at kk = 0, first read; + kk * 4096 + 1024 * second
  int L[4];
  int LT[3];                                               // TN: A row blocks b = 0, 1 and the B block
  const int aoff = wm * 64 * 128, boff = wn * 32 * 128;    // NT only
  if constexpr (TN) {
    const int g = lane >> 4, p = lane & 15, r4 = p >> 2;
    const int common = (g >> 1) * 2048 + r4 * 256 + (2 * (g & 1) + ((p >> 1) & 1)) * 16 + (p & 1) * 8;
    LT[0] = common + (((wm * 2 + 0) ^ r4) << 6);
    LT[1] = common + (((wm * 2 + 1) ^ r4) << 6);
    LT[2] = common + ((wn ^ r4) << 6);
  } else {
    const int x = (lane >> 5) ^ ((lane >> 1) & 7);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) L[kk] = (lane & 31) * 128 + ((x ^ (2 * kk)) << 4);
  }
  auto tr_frag = [&](const unsigned char* q) -> bf16x8 {
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4_t lds_v4;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)q);
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(q + 1024));
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  bf16x8 Af[2][2][4];     // [set = A half][row block][kk]
  bf16x8 Bf[2][4];        // [set][kk]
  f32x16 acc[2][2][2];    // [i][j][row block]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][b][r] = 0.f;

  auto readA = [&](auto set_c, auto buf_c) {      // A-half `set` of buffer `buf` -> Af[set]
    constexpr int set = decltype(set_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        if constexpr (TN) Af[set][b][kk] = tr_frag(smem + buf * BUF_B + set * HALF_B + kk * 4096 + LT[b]);
        else Af[set][b][kk] = *(const bf16x8*)(smem + buf * BUF_B + set * HALF_B + aoff + b * 4096 + L[kk]);
      }
  };
  auto readB = [&](auto set_c, auto half_c, auto buf_c) {    // B-half `half` of buffer `buf` -> Bf[set]
    constexpr int set = decltype(set_c)::value, half = decltype(half_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      if constexpr (TN) Bf[set][kk] = tr_frag(smem + buf * BUF_B + (2 + half) * HALF_B + kk * 4096 + LT[2]);
      else Bf[set][kk] = *(const bf16x8*)(smem + buf * BUF_B + (2 + half) * HALF_B + boff + L[kk]);
    }
  };
  auto mma = [&](auto i_c, auto j_c, auto bset_c) {
    constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value, bset = decltype(bset_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        // NT: (B, A) operand order = transposed accumulators for the row-contiguous staged epilogue; TN: (A, B), lanes 0..31 = 32
        // consecutive output columns for the split-K atomics
        if constexpr (TN) acc[i][j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[i][b][kk], Bf[bset][kk], acc[i][j][b], 0, 0, 0);
        else acc[i][j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[bset][kk], Af[i][b][kk], acc[i][j][b], 0, 0, 0);
      }
  };
  // bias-gradient side sum of the weight-gradient form (du_gemm_args.a_colsum): sum_k A(m, k) from the A fragments, once per fragment
  // set (phases q0 / q3 = the j == 0 products), by the two waves with wn == 0 of ONE tile column per (tile row, split)
  const bool colsum_on = TN && P.a_colsum != nullptr && wn == 0 && tn == my_split % P.tiles_n;
  float csum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // [A half i][row block b]
  auto colsum_acc = [&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (TN) {
      if (colsum_on) {
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int kk = 0; kk < 4; kk++)
            csum[i][b] = frag_sum8(Af[i][b][kk], csum[i][b]);
      }
    }
  };
  // ConvT weight gradient (GA): bias gradient from the gathered dY operand (du_gemm_args.b_colsum): sum_k B(n, k) from the B fragments,
  // by the two waves with wm == 0 of ONE tile row per (tile column, split); B-half 0 fragments are complete in phase q3, B-half 1 in q0
  const bool bsum_on = TN && (GA || GG == 2) && P.b_colsum != nullptr && wm == 0 && tm == my_split % P.tiles_m;
  float bsum[2] = {0.f, 0.f};                       // [B half j]
  auto bsum_acc = [&](auto j_c, auto set_c) {
    constexpr int j = decltype(j_c)::value, set = decltype(set_c)::value;
    if constexpr (TN && (GA || GG == 2)) {
      if (bsum_on) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
          bsum[j] = frag_sum8(Bf[set][kk], bsum[j]);
      }
    }
  };
  // pin the issue order of a phase.  The compiler orders every ds_read of the phase before its LDS-DMA issues (it must assume they
  // alias), so the reads ride behind the first four MFMAs and the two DMA issues behind the next two.
  auto pin = [&](auto nrd_c) {
    constexpr int nrd = decltype(nrd_c)::value;
    if constexpr (SCHED == 1) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
        if (m < 4) __builtin_amdgcn_sched_group_barrier(0x100, (TN ? 2 : 1) * nrd / 4, 0);   // 1 or 2 ds_read_b128 (TN: tr_b64 pairs)
        if (m == 4 || m == 5) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 VMEM (LDS-DMA)
      }
    }
  };

  // one K-tile: PAR = t & 1 (compile time), TAIL = runtime guards + exact vmcnt for the last four tiles
  auto ktile = [&](auto par_c, auto tail_c, int t) {
    constexpr int p = decltype(par_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr int b0set = p, b1set = 1 - p;    // fragment set holding B-half0 / B-half1 of this tile
    auto finish = [&](int q) {
      __builtin_amdgcn_sched_barrier(0);      // phase boundary: register-only MFMAs must not drift across it either
      if constexpr (TAIL) {
        int h = 4 * (nk - t - 1) - q;
        h = h < 0 ? 0 : (h > 6 ? 6 : h);
        wait_vm_count(2 * h);
      } else {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // q0
    readB(IC<b1set>{}, IC<1>{}, IC<p>{});
    if (!TAIL || t + 2 < nk) stage(IC<2>{}, IC<p>{}, t + 2);
    mma(IC<0>{}, IC<0>{}, IC<b0set>{});
    pin(IC<4>{});
    colsum_acc(IC<0>{});
    bsum_acc(IC<0>{}, IC<b0set>{});
    finish(0);
    // q1
    readA(IC<1>{}, IC<p>{});
    if (!TAIL || t + 2 < nk) stage(IC<3>{}, IC<p>{}, t + 2);
    mma(IC<0>{}, IC<1>{}, IC<b1set>{});
    pin(IC<8>{});
    bsum_acc(IC<1>{}, IC<b1set>{});
    finish(1);
    // q2
    if (!TAIL || t + 1 < nk) readA(IC<0>{}, IC<1 - p>{});
    if (!TAIL || t + 2 < nk) stage(IC<1>{}, IC<p>{}, t + 2);
    mma(IC<1>{}, IC<1>{}, IC<b1set>{});
    pin(IC<8>{});
    finish(2);
    // q3  (the next tile's B-half0 goes to the set that held this tile's B-half1)
    if (!TAIL || t + 1 < nk) readB(IC<b1set>{}, IC<0>{}, IC<1 - p>{});
    if (!TAIL || t + 3 < nk) stage(IC<0>{}, IC<1 - p>{}, t + 3);
    mma(IC<1>{}, IC<0>{}, IC<b0set>{});
    pin(IC<4>{});
    colsum_acc(IC<1>{});
    finish(3);
  };

  int* ks_st = nullptr;          // KS: {alive, flag of split 0, flag of split 1, done} of this tile (all zero between launches)
  if constexpr (KS) {
    ks_st = (int*)P.ks_ws + ks_tile * 4;
    if (tid == 0) __hip_atomic_fetch_add(ks_st, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // "resident": the peer may wait for this workgroup from here on
  }
  // ---- prologue: K-tiles 0 and 1 entirely, then A0(2) once A-half0[0] has been read ----
  stage(IC<0>{}, IC<0>{}, 0); stage(IC<2>{}, IC<0>{}, 0); stage(IC<3>{}, IC<0>{}, 0); stage(IC<1>{}, IC<0>{}, 0);
  stage(IC<0>{}, IC<1>{}, 1); stage(IC<2>{}, IC<1>{}, 1); stage(IC<3>{}, IC<1>{}, 1); stage(IC<1>{}, IC<1>{}, 1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // A0(0), B0(0) and B1(0) (read in q0 of tile 0: no earlier phase waits for it) landed
  __builtin_amdgcn_s_barrier();
  readA(IC<0>{}, IC<0>{});
  readB(IC<0>{}, IC<0>{}, IC<0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stage(IC<0>{}, IC<0>{}, 2);

  int t = 0;
  for (; t + 6 <= nk; t += 2) {
    ktile(IC<0>{}, IC<false>{}, t);
    ktile(IC<1>{}, IC<false>{}, t + 1);
  }
  for (; t < nk; t += 2) {
    ktile(IC<0>{}, IC<true>{}, t);
    ktile(IC<1>{}, IC<true>{}, t + 1);
  }

  // ---- epilogue (every LDS read of the main loop is behind the last phase's lgkmcnt(0) + barrier) ----
  TC* Cb = (TC*)P.C + (long)batch * P.cbs;
  const TC* Rb = (const TC*)P.residual;
  if (Rb) Rb += (long)batch * P.cbs;
  // ---- KS: the two K halves of the tile meet.  Slab of (tile, split): 32 chunks of 512 lanes x 16 bytes = the 128 accumulator registers in
  // register order (chunk ((i*2+j)*2+b)*4+q, lane tid): private layout, both sides use the same lane <-> element map.  Stores and loads are
  // sc1 (write-through / L2-bypassing: the per-XCD L2s are not coherent and a pair may sit anywhere), flags relaxed agent-scope atomics
  // behind a vmcnt(0) drain + barrier.  Three ways through, decided per tile at run time so that NO workgroup ever waits for one that is
  // not resident (any occupancy, any dispatch order):
  //   both resident (alive == 2): each publishes the quadrant row the OTHER finishes (128 KB), raises its flag (1), waits for the peer's
  //     flag (the peer is running: bounded), adds the peer's half to its own quadrant row and runs the epilogue on those 128 rows;
  //   peer not started yet (alive == 1): publish everything (256 KB), flag = 2, leave; the peer finds the flag whenever it gets there,
  //     adds the whole slab and finishes both quadrant rows;
  //   (a peer that already left with flag 2 is found by the first look, or by the wait loop if the two decisions raced.)
  // a + b == b + a in fp32: the result does not depend on which way a tile went (bit-reproducible run to run).
  // The second of the pair to leave restores the four words to zero.
  int ks_rows = 3;                  // bit i: this workgroup finishes quadrant row i
  if constexpr (KS) {
    volatile int* bc = (volatile int*)smem;
    const int peer = 1 - ks_split;
    const auto rmine = __builtin_amdgcn_make_buffer_rsrc((void*)((float*)((char*)P.ks_ws + 131072) + ((long)ks_tile * 2 + ks_split) * (PBM * PBN)), 0,
                                                         PBM * PBN * 4, 0x00020000);
    const auto rpeer = __builtin_amdgcn_make_buffer_rsrc((void*)((float*)((char*)P.ks_ws + 131072) + ((long)ks_tile * 2 + peer) * (PBM * PBN)), 0,
                                                         PBM * PBN * 4, 0x00020000);
    auto leave = [&]() {
      if (tid == 0) {
        const int d = __hip_atomic_fetch_add(ks_st + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == 1) {
#pragma unroll
          for (int w = 0; w < 4; w++) __hip_atomic_store(ks_st + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    };
    auto publish = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = acc[i][j][b][4 * q + e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rmine, (((((i * 2 + j) * 2 + b) * 4 + q) * 512) + tid) * 16, 0, 16);
          }
    };
    auto take = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
#pragma unroll
      for (int g = 0; g < 2; g++) {       // eight 16-byte loads in flight per lane
        u32x4_t v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = __builtin_amdgcn_raw_buffer_load_b128(rpeer, (((i * 16 + g * 8 + c) * 512) + tid) * 16, 0, 16);
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const f32x4 f = __builtin_bit_cast(f32x4, v[c]);
#pragma unroll
          for (int e = 0; e < 4; e++) acc[i][g][(c >> 2) & 1][4 * (c & 3) + e] += f[e];
        }
      }
    };
    if (tid == 0) {
      if ((P.dbg & 16) && ks_split == 1) {       // test aid: this half looks late (finds a peer that left)
        for (int w = 0; w < 2000; w++) __builtin_amdgcn_s_sleep(32);
      }
      const int f = __hip_atomic_load(ks_st + 1 + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int a = __hip_atomic_load(ks_st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bc[0] = f == 2 ? 2 : (a >= 2 ? 1 : 0);
    }
    __syncthreads();
    int mode = bc[0];
    __syncthreads();
    if (P.dbg & 2) mode = 0;        // measurement aid: publish and leave
    if ((P.dbg & 8) && ks_split == 0 && mode == 1) mode = 0;      // test aid: this half behaves as if its peer had not started
    if (mode == 0) {                // the peer has not started: hand everything over
      publish(IC<0>{}); publish(IC<1>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(ks_st + 1 + ks_split, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      leave();
      return;
    }
    if (mode == 1) {
      if (ks_split == 0) publish(IC<1>{}); else publish(IC<0>{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(ks_st + 1 + ks_split, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int f = 0;
        for (long spin = 0; spin < (1L << 24); spin++) {      // the peer is resident: it raises its flag after at most its main loop + publish
          f = __hip_atomic_load(ks_st + 1 + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (f) break;
          __builtin_amdgcn_s_sleep(4);
        }
        if (!f) __hip_atomic_store((int*)P.ks_ws + 16380, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // never seen; tests read this word
        bc[0] = f;
      }
      __syncthreads();
      mode = bc[0] == 2 ? 2 : (bc[0] == 1 ? 1 : 3);
      __syncthreads();
    }
    // (one site per quadrant row: two copies of take<i> on different paths made the allocator give the sums new registers and spill at the merge)
    if (mode != 2) ks_rows = 1 << ks_split;      // (mode 3, the error path -- the flag never came: the own rows without the peer's half)
    if (mode == 2 || (mode == 1 && ks_split == 0)) take(IC<0>{});
    if (mode == 2 || (mode == 1 && ks_split == 1)) take(IC<1>{});
  }
  if (P.dbg & 2) return;            // measurement aid (du_set_option key 3): no epilogue at all
  if constexpr (TN) {
    if (colsum_on) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const int m = m0 + i * 128 + wm * 64 + b * 32 + (lane & 31);
          if (m < P.M) atomic_add_f32(P.a_colsum + m, csum[i][b] * P.alpha);
        }
    }
    if constexpr (GA || GG == 2) {
      if (bsum_on) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int n = n0 + j * 128 + wn * 32 + (lane & 31);
          if (n < P.N) atomic_add_f32(P.b_colsum + n % P.b.C, bsum[j] * P.alpha);
        }
      }
    }
    // split-K partial: fp32 atomics into the (zeroed) result.  Register r of a 32 x 32 block = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
    // lanes 0..31 = 32 consecutive columns (one 128-byte segment per row).  A product that is NOT split (grouped launches give short
    // products one workgroup per tile) owns its tile: plain stores, no read-modify-write at the memory side
    const int hi = lane >> 5;
    const bool plain = P.split_k == 1 && !P.k_scale;     // (grouped launches set k_scale when several jobs accumulate into one C)
    // DU_STORE_TAPS (grouped convolution jobs): column n = (tap, c) of the product goes to element (m, c_off + c, tap) of a torch-layout
    // weight (Cout, Cin, KH, KW) / (Cin, Cout, 2, 2): ps_C channels per tap in this job, ps_W channels in the parameter (a job may cover one
    // source of a channel concat: c_off = rope_prefix), ps_H taps; no permute copy afterwards
    const bool taps = P.store_mode == DU_STORE_TAPS;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n = n0 + j * 128 + wn * 32 + (lane & 31);
      if (n >= P.N) continue;
      long noff = n, mld = P.ldc;
      if (taps) { const int t = n / P.ps_C; noff = (long)(P.rope_prefix + n - t * P.ps_C) * P.ps_H + t; mld = (long)P.ps_W * P.ps_H; }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int m = m0 + i * 128 + wm * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < P.M) {
              float* dst = (float*)Cb + (long)m * mld + noff;
              if (plain) *dst = acc[i][j][b][r] * P.alpha;
              else atomic_add_f32(dst, acc[i][j][b][r] * P.alpha);
            }
          }
    }
    return;
  }
  bool done = false;
  if constexpr (sizeof(TC) == 2) {
    if (bf16_simple(P, Rb)) {        // the whole 256 x 256 tile staged as bf16, one pass
      bf16_t* stg = (bf16_t*)smem;
      const int hi = lane >> 5;
      float4 bv[2][4];
#pragma unroll
      for (int j = 0; j < 2; j++) load_bias4(P, n0 + j * 128 + wn * 32, hi, bv[j]);
      auto stage_all = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int b = 0; b < 2; b++)
              stage_block_bf16<ACT>(acc[i][j][b], bv[j], stg, P8_STG_LDB, i * 128 + wm * 64 + b * 32 + (lane & 31), j * 128 + wn * 32, hi);
      };
      if (P.act == DU_ACT_SWIGLU) {
        stage_all(IC<DU_ACT_SWIGLU>{});
        __syncthreads();
        readout_bf16<PBN / 2>(P, stg, P8_STG_LDB, 256, m0, n0 / 2, P.N / 2, (bf16_t*)Cb, tid);
        return;
      }
      if (P.act == DU_ACT_GELU) stage_all(IC<DU_ACT_GELU>{}); else stage_all(IC<DU_ACT_NONE>{});
      __syncthreads();
      readout_bf16<PBN>(P, stg, P8_STG_LDB, 256, m0, n0, P.N, (bf16_t*)Cb, tid);
      done = true;
    }
  }
  if (!done) {          // fp32 staging, two passes of 128 rows (quadrant row i)
    float* stg = (float*)smem;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if constexpr (KS) { if (!((ks_rows >> i) & 1)) continue; }
      if (i > 0) __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int b = 0; b < 2; b++)
          stage_block_f32(acc[i][j][b], stg, P8_STG_LDF, wm * 64 + b * 32 + (lane & 31), j * 128 + wn * 32, lane);
      __syncthreads();
      if constexpr (sizeof(TC) == 4 && !TN && !GA) {
        if (P.store_mode == DU_STORE_MSDA_PREP) { readout_msda_prep(P, stg, P8_STG_LDF, 128, m0 + i * 128, tid); continue; }
      }
      readout_f32_any<TC, PBN>(P, stg, P8_STG_LDF, 128, m0 + i * 128, n0, Cb, Rb, tid);
    }
  }
  if constexpr (KS) {              // the second of the pair to get here restores the tile's four words
    if (tid == 0) {
      const int d = __hip_atomic_fetch_add(ks_st + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 1) {
#pragma unroll
        for (int w = 0; w < 4; w++) __hip_atomic_store(ks_st + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <typename TC, int SCHED, bool TN, bool GA>
__global__ __launch_bounds__(512) void gemm_nt_p8_kernel(GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if constexpr (!TN && !GA) {
    if (P.tail_rows && (int)blockIdx.x >= P.main_wgs) { p8_tail<TC, 8, true>(P, smem); return; }
  }
  p8_tile_body<TC, SCHED, TN, GA>(P, smem, TN ? xcd_linear_index() : 0);
  if constexpr (!TN && !GA) {
    if (P.tail_rows && (int)gridDim.x == P.main_wgs && gridDim.y == 1) p8_tail_inline<TC>(P, smem, (int)blockIdx.x, P.main_wgs);
  }
}

// the K-split pair form (fp32 result, <= 128 tiles): see KS in p8_tile_body
template <int SCHED>
__global__ __launch_bounds__(512) void gemm_nt_p8ks_kernel(GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (P.tail_rows && (int)blockIdx.x >= P.main_wgs) { p8_tail<float>(P, smem); return; }
  p8_tile_body<float, SCHED, false, false, 0, true>(P, smem, 0);
}

// ---- grouped weight gradients: several dW = dY^T X products in ONE launch (du_gemm_tn_group) ----------------------------------------
// Why: a weight gradient has a huge contraction (rows = B x pixels / tokens) and a small result, so one product alone must be cut into
// ~256 / tiles K splits to fill the chip -- and every split ends with a 256 x 256 fp32 read-modify-write of the result (35-60 us of
// memory-side atomics per 256-workgroup launch whatever K: more than the MFMA loop of most of these products, tools/gemm_tn_bench.py).
// Nothing reads a weight gradient before the optimizer, so the host defers them (ops.WgradQueue) and launches many together: the 256
// workgroups are then dealt out over ALL queued products in proportion to their contraction length -- 2-4 splits per product instead
// of 16-64, an order of magnitude fewer partial tiles, unsplit products written with plain stores -- with the tile program above
// unchanged.  The job table travels in the kernel arguments (no device-side table to keep alive across hipGraph replays).
struct TnJob {
  const void* A; const void* B; float* C; float* a_colsum;
  const float* alpha;                             // nullable: the product is scaled by *alpha (device memory); 0 = nothing to do
  float* b_colsum;                                // ConvT jobs: bias gradient
  int lda, ldb, ldc, M, N, K, splits, unit0;      // unit0: first (split, tile) unit of this job in the launch
  int accumulate;                                 // other jobs add into the same C -> atomics even when unsplit
  int gather, Hs, Ws, Cb;                         // gather: 0 plain, 2 ConvT k2 s2, 3 conv 3 x 3 s1 p1; (Hs, Ws) = pixel grid of the contraction
  int taps, inner, inner_total, c_off;            // taps > 1: torch-layout store C[m][c_off + c][tap] of a parameter with inner_total channels
  int pad_;
};
constexpr int TN_GROUP_MAX = 32;
struct TnGroupArgs { int njobs, nunits, dbg, pad_; TnJob jobs[TN_GROUP_MAX]; };
static_assert(sizeof(TnGroupArgs) <= 4096, "kernel arguments");

template <int SCHED, int GG>
__global__ __launch_bounds__(512) void gemm_tn_group_kernel(TnGroupArgs G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lin = xcd_linear_index();
  if (lin >= G.nunits) return;
  int j = 0;
  while (j + 1 < G.njobs && lin >= G.jobs[j + 1].unit0) j++;
  GemmParams P{};
  P.a.p = G.jobs[j].A; P.a.ld = G.jobs[j].lda;
  P.b.p = G.jobs[j].B; P.b.ld = G.jobs[j].ldb;
  P.C = G.jobs[j].C; P.ldc = G.jobs[j].ldc;
  P.M = G.jobs[j].M; P.N = G.jobs[j].N; P.K = G.jobs[j].K;
  P.split_k = G.jobs[j].splits;
  P.alpha = 1.0f;
  if (G.jobs[j].alpha) {          // DropPath's per-sample scale (ops.mm_wgrad k_scale: one job per sample): dropped samples cost nothing
    P.alpha = *G.jobs[j].alpha;
    if (P.alpha == 0.f) return;
  }
  P.k_scale = G.jobs[j].accumulate;      // (re-used as the "several jobs share C" flag of the tile program's epilogue)
  P.a_colsum = G.jobs[j].a_colsum;
  P.tiles_m = (P.M + PBM - 1) / PBM; P.tiles_n = (P.N + PBN - 1) / PBN;
  P.dbg = G.dbg;
  if (G.jobs[j].taps > 1) {
    P.store_mode = DU_STORE_TAPS; P.ps_H = G.jobs[j].taps; P.ps_C = G.jobs[j].inner; P.ps_W = G.jobs[j].inner_total; P.rope_prefix = G.jobs[j].c_off;
  }
  if constexpr (GG != 0) {
    P.b.KH = GG;
    P.b.C = G.jobs[j].Cb;
    P.b.Ho = G.jobs[j].Hs; P.b.Wo = G.jobs[j].Ws;
    P.b.Wi = GG == 2 ? 2 * G.jobs[j].Ws : G.jobs[j].Ws;             // row pitch (pixels) of the SOURCE tensor
    // bytes of source 1: ConvT gathers from dy (4 pixels per contraction row), the convolution from x (one)
    P.b.bstride = (long)P.K * (GG == 2 ? 4 : 1) * P.b.ld * 2;
    if constexpr (GG == 2) P.b_colsum = G.jobs[j].b_colsum;
  }
  p8_tile_body<float, SCHED, true, false, GG>(P, smem, lin - G.jobs[j].unit0);
}

// ================================================================================================================================
// 256 x 128 tiles: 8 waves as 4 (M) x 2 (N), 64 x 64 per wave; three 48 KB LDS buffers { A-half0, A-half1, B } rotated at run time.
//   q0(t): MFMA (A0f, B) | ds_read A1(t) -> A1f                       | DMA A0(t+2), B(t+2) -> buffer (t+2) % 3
//   q1(t): MFMA (A1f, B) | ds_read A0(t+1) -> A0f, B(t+1) -> other set | DMA A1(t+2)
// Buffer (t+2) % 3 held tile t-1, last read in q1(t-2) (A0, B) and q0(t-1) (A1): free when q0(t) starts.  Issue order per tile:
// A0 B | A1; at the end of either phase the data read next is followed by exactly 6 younger DMA instructions => vmcnt(6).
// ================================================================================================================================
constexpr int NBN = 128;
constexpr int NBUF_B = 3 * HALF_B;             // 48 KB
constexpr int N_STG_LDB = NBN + 8;             // 272 B rows
constexpr int N_STG_LDF = NBN + 4;             // 528 B rows
constexpr int P8N_LDS = 3 * NBUF_B;            // 147 456 B >= 256 * N_STG_LDF * 4 = 135 168

template <typename TC, int SCHED>
__device__ __forceinline__ void p8n_tile_body(const GemmParams& P, unsigned char* smem);

template <typename TC, int SCHED>
__global__ __launch_bounds__(512) void gemm_nt_p8n_kernel(GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (P.tail_rows && (int)blockIdx.x >= P.main_wgs) { p8_tail<TC, 8, true>(P, smem); return; }
  p8n_tile_body<TC, SCHED>(P, smem);
  if (P.tail_rows && (int)gridDim.x == P.main_wgs && gridDim.y == 1) p8_tail_inline<TC>(P, smem, (int)blockIdx.x, P.main_wgs);
}

template <typename TC, int SCHED>
__device__ __forceinline__ void p8n_tile_body(const GemmParams& P, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_coords(P, tm, tn);
  const int m0 = tm * PBM, n0 = tn * NBN;
  const int batch = blockIdx.y;

  const bf16_t* Ab = (const bf16_t*)P.a.p + (long)batch * P.a.bstride + (long)m0 * P.a.ld;
  const bf16_t* Bb = (const bf16_t*)P.b.p + (long)batch * P.b.bstride + (long)n0 * P.b.ld;
  long abytes = ((long)(P.M - m0) * P.a.ld - (P.a.ld - P.K)) * 2, bbytes = ((long)(P.N - n0) * P.b.ld - (P.b.ld - P.K)) * 2;
  if (abytes > 0x7fffffffL) abytes = 0x7fffffffL;
  if (bbytes > 0x7fffff00L) bbytes = 0x7fffff00L;
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)abytes, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bbytes, 0x00020000);

  unsigned va[2][2], vb[2];
  {
    const int sw = ((wave & 1) << 2) | (lane >> 4);
    const int lc = (lane & 7) ^ sw;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int row = (r * 8 + wave) * 8 + (lane >> 3);
      va[0][r] = (unsigned)row * (unsigned)(P.a.ld * 2) + lc * 16;
      va[1][r] = (unsigned)(128 + row) * (unsigned)(P.a.ld * 2) + lc * 16;
      vb[r] = (unsigned)row * (unsigned)(P.b.ld * 2) + lc * 16;
    }
  }
  // which: 0 = A-half0, 1 = A-half1, 2 = B of K-tile kt, into the buffer at byte offset bo
  auto stage = [&](auto which_c, int bo, int kt) {
    constexpr int which = decltype(which_c)::value;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      unsigned char* dst = smem + bo + which * HALF_B + (r * 8 + wave) * 1024;
      if constexpr (which < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, va[which][r], kt * (PBK * 2), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)dst, 16, vb[r], kt * (PBK * 2), 0, 0);
    }
  };
  int L[4];
  {
    const int x = (lane >> 5) ^ ((lane >> 1) & 7);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) L[kk] = (lane & 31) * 128 + ((x ^ (2 * kk)) << 4);
  }
  const int aoff = wm * 32 * 128, boff = 2 * HALF_B + wn * 64 * 128;

  bf16x8 Af[2][4];        // [set = A half][kk]
  bf16x8 Bf[2][2][4];     // [set][column block][kk]
  f32x16 acc[2][2];       // [i][column block]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][c][r] = 0.f;

  auto readA = [&](auto set_c, int bo) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) Af[set][kk] = *(const bf16x8*)(smem + bo + set * HALF_B + aoff + L[kk]);
  };
  auto readB = [&](auto set_c, int bo) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int kk = 0; kk < 4; kk++) Bf[set][c][kk] = *(const bf16x8*)(smem + bo + boff + c * 4096 + L[kk]);
  };
  auto mma = [&](auto i_c, auto bset_c) {
    constexpr int i = decltype(i_c)::value, bset = decltype(bset_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int c = 0; c < 2; c++)
        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[bset][c][kk], Af[i][kk], acc[i][c], 0, 0, 0);
  };
  auto pin = [&](auto nrd_c, auto ndma_c) {
    constexpr int nrd = decltype(nrd_c)::value, ndma = decltype(ndma_c)::value;
    if constexpr (SCHED == 1) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < 4) __builtin_amdgcn_sched_group_barrier(0x100, nrd / 4, 0);
        else if (m - 4 < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
  };

  const int nk = P.K / PBK;
  auto finish = [&](int vm) {
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_count(vm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto finish6 = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one K-tile; bc / bn / bnn = byte offsets of the buffers of tiles t, t+1, t+2
  auto ktile = [&](auto par_c, auto tail_c, int t, int bc, int bn, int bnn) {
    constexpr int p = decltype(par_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
    // q0
    readA(IC<1>{}, bc);
    if (!TAIL || t + 2 < nk) { stage(IC<0>{}, bnn, t + 2); stage(IC<2>{}, bnn, t + 2); }
    mma(IC<0>{}, IC<p>{});
    pin(IC<4>{}, IC<4>{});
    if constexpr (TAIL) finish((t + 1 < nk ? 2 : 0) + (t + 2 < nk ? 4 : 0)); else finish6();
    // q1
    if (!TAIL || t + 1 < nk) { readA(IC<0>{}, bn); readB(IC<1 - p>{}, bn); }
    if (!TAIL || t + 2 < nk) stage(IC<1>{}, bnn, t + 2);
    mma(IC<1>{}, IC<p>{});
    pin(IC<12>{}, IC<2>{});
    if constexpr (TAIL) finish(t + 2 < nk ? 6 : 0); else finish6();
  };

  // ---- prologue: K-tiles 0 and 1 ----
  stage(IC<0>{}, 0, 0); stage(IC<2>{}, 0, 0); stage(IC<1>{}, 0, 0);
  stage(IC<0>{}, NBUF_B, 1); stage(IC<2>{}, NBUF_B, 1); stage(IC<1>{}, NBUF_B, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // tile 0 landed (A-half1 is read in q0(0))
  __builtin_amdgcn_s_barrier();
  readA(IC<0>{}, 0);
  readB(IC<0>{}, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  int bc = 0, bn = NBUF_B, bnn = 2 * NBUF_B;
  int t = 0;
  for (; t + 4 <= nk; t += 2) {
    ktile(IC<0>{}, IC<false>{}, t, bc, bn, bnn);
    ktile(IC<1>{}, IC<false>{}, t + 1, bn, bnn, bc);
    const int o = bc; bc = bnn; bnn = bn; bn = o;        // advance two tiles: (bc, bn, bnn) <- (bnn, bc, bn)
  }
  for (; t < nk; t += 2) {
    ktile(IC<0>{}, IC<true>{}, t, bc, bn, bnn);
    ktile(IC<1>{}, IC<true>{}, t + 1, bn, bnn, bc);
    const int o = bc; bc = bnn; bnn = bn; bn = o;
  }

  // ---- epilogue: one pass, 256 rows x 128 columns ----
  TC* Cb = (TC*)P.C + (long)batch * P.cbs;
  const TC* Rb = (const TC*)P.residual;
  if (Rb) Rb += (long)batch * P.cbs;
  if (P.dbg & 2) return;
  bool done = false;
  if constexpr (sizeof(TC) == 2) {
    if (bf16_simple(P, Rb)) {
      bf16_t* stg = (bf16_t*)smem;
      const int hi = lane >> 5;
      float4 bv[2][4];
#pragma unroll
      for (int c = 0; c < 2; c++) load_bias4(P, n0 + wn * 64 + c * 32, hi, bv[c]);
      auto stage_all = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int c = 0; c < 2; c++)
            stage_block_bf16<ACT>(acc[i][c], bv[c], stg, N_STG_LDB, i * 128 + wm * 32 + (lane & 31), wn * 64 + c * 32, hi);
      };
      if (P.store_mode == DU_STORE_QKV_ROPE) {        // a wave's 64 columns = one (which, head): RoPE in registers, head-major store
        const int which = (n0 + wn * 64) / (P.ps_C * 64);
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const int srow = i * 128 + wm * 32 + (lane & 31);
          stage_head_rope(P, acc[i][0], acc[i][1], bv[0], bv[1], stg, N_STG_LDB, srow, wn * 64, hi, m0 + srow, which);
        }
        __syncthreads();
        readout_bf16<NBN, true>(P, stg, N_STG_LDB, 256, m0, n0, P.N, (bf16_t*)Cb, tid);
        return;
      }
      if (P.act == DU_ACT_SWIGLU) {
        stage_all(IC<DU_ACT_SWIGLU>{});
        __syncthreads();
        readout_bf16<NBN / 2>(P, stg, N_STG_LDB, 256, m0, n0 / 2, P.N / 2, (bf16_t*)Cb, tid);
        return;
      }
      if (P.act == DU_ACT_GELU) stage_all(IC<DU_ACT_GELU>{}); else stage_all(IC<DU_ACT_NONE>{});
      __syncthreads();
      readout_bf16<NBN>(P, stg, N_STG_LDB, 256, m0, n0, P.N, (bf16_t*)Cb, tid);
      done = true;
    }
  }
  if (!done) {
    float* stg = (float*)smem;
    bool pre = false;
    ResidualTile R;
    if constexpr (sizeof(TC) == 4) {
      pre = Rb && P.store_mode == DU_STORE_PLAIN && !(P.dbg & 4) && P.ldr % 4 == 0 && P.ldc % 4 == 0 && !((((uintptr_t)Rb) | ((uintptr_t)Cb)) & 15);
      if (pre) residual_prefetch<NBN>(P, (const float*)Rb, m0, n0, tid, R);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int c = 0; c < 2; c++)
        stage_block_f32(acc[i][c], stg, N_STG_LDF, i * 128 + wm * 32 + (lane & 31), wn * 64 + c * 32, lane);
    __syncthreads();
    if constexpr (sizeof(TC) == 4) {
      if (pre) { readout_f32_prefetched<NBN>(P, stg, N_STG_LDF, m0, n0, (float*)Cb, tid, R); return; }
    }
    readout_f32_any<TC, NBN>(P, stg, N_STG_LDF, 256, m0, n0, Cb, Rb, tid);
  }
}

// ================================================================================================================================
// 256 x 128 tiles on FOUR waves (2 x 2, 128 x 64 per wave), TWO workgroups per CU (round 4).
//
// Why: at K = 1024 (the ViT's qkv / fc1) a tile of the kernels above is 16 K-steps between a cold prologue and an exposed epilogue, and
// with ONE workgroup per CU (135 - 147 KB of LDS, 8 waves) the matrix pipe idles through both: 14 - 22 us of a 65 - 84 us product
// (profiles/r03_gemm_p8_table_v2.txt).  Two smaller workgroups per CU overlap them: the SIMD arbitrates oldest-first, so the older
// workgroup runs its main loop at full speed while the younger one trails, and when the older one leaves the matrix pipe for its epilogue
// (bias / GELU / LayerScale + residual in registers, LDS-staged row stores) the younger one takes the pipe -- measured on this chip a pure
// MFMA stream and a pure VALU stream of DIFFERENT waves share a SIMD at full speed each (tools/scratch/corun_bench.hip: 32.5 cycles per
// MFMA beside 6.1 cycles per v_add against 32.1 / 4.6 alone).  The dispatcher then keeps the pairs de-phased by itself.
//
//   * LDS: three 24 KB stages of K = 32 { A: 256 rows x 64 B, B: 128 rows x 64 B } = 72 KB per workgroup.  64-byte rows put four rows on one
//     256-byte bank row: 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3) (source-side swizzle of the DMA, undone in the
//     ds_read_b128 address): the 16 lanes of a read group then touch 16 different 16-byte slots.
//   * a wave's 128 x 64 tile reads 0.75 LDS fragments per MFMA (the 64 x 64 tile of the 8-wave 256 x 128 kernel: 1.0).
//   * a stage is cut into two k-halves of 8 MFMAs; every half reads the NEXT half's fragments into the other register set; the DMA of
//     stage s + 3 goes into the buffer of stage s right behind the barrier that ends its last fragment read; counted vmcnt(6) (one stage
//     stays in flight), one raw barrier per stage.  Steps past K are requested beyond the descriptor's range (zeros, no traffic, never
//     read), so the wait counts hold for any K.
//   * epilogues: the helpers of the 8-wave kernels with 256 threads; the bf16 tile is staged whole (68 KB), the fp32 tile in two passes of
//     128 rows (the ring is 72 KB).
// ================================================================================================================================
constexpr int P4_STAGE_B = (256 + 128) * 32 * 2;       // 24 KB
constexpr int P4_LDS = 3 * P4_STAGE_B;                 // 73 728 B >= 256 * N_STG_LDB * 2 = 69 632 and >= 128 * N_STG_LDF * 4 = 67 584

template <typename TC>
__global__ __launch_bounds__(256, 2) void gemm_nt_p4_kernel(GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (P.tail_rows && (int)blockIdx.x >= P.main_wgs) { p8_tail<TC, 4>(P, smem); return; }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_coords(P, tm, tn);
  const int m0 = tm * PBM, n0 = tn * NBN;
  const int batch = blockIdx.y;

  const bf16_t* Ab = (const bf16_t*)P.a.p + (long)batch * P.a.bstride + (long)m0 * P.a.ld;
  const bf16_t* Bb = (const bf16_t*)P.b.p + (long)batch * P.b.bstride + (long)n0 * P.b.ld;
  long abytes = ((long)(P.M - m0) * P.a.ld - (P.a.ld - P.K)) * 2, bbytes = ((long)(P.N - n0) * P.b.ld - (P.b.ld - P.K)) * 2;
  if (abytes > 0x6fffffffL) abytes = 0x6fffffffL;
  if (bbytes > 0x6fffff00L) bbytes = 0x6fffff00L;
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)abytes, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bbytes, 0x00020000);

  // DMA: piece = 16 rows x 64 B; lane l -> row (l >> 2), physical chunk l & 3 = logical chunk ^ ((row >> 2) & 3), (row >> 2) & 3 = (l >> 4) & 3
  unsigned va[4], vb[2];
  {
    const int lc = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 4; j++) va[j] = (unsigned)((wave + 4 * j) * 16 + (lane >> 2)) * (unsigned)(P.a.ld * 2) + lc * 16;
#pragma unroll
    for (int j = 0; j < 2; j++) vb[j] = (unsigned)((wave + 4 * j) * 16 + (lane >> 2)) * (unsigned)(P.b.ld * 2) + lc * 16;
  }
  const int ns = P.K / 32;
  auto stage = [&](int st, int bo) {
    const unsigned soff = st < ns ? (unsigned)st * 64u : 0x70000000u;       // past K: beyond the descriptors' range (zeros, no traffic)
#pragma unroll
    for (int j = 0; j < 4; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(smem + bo + (wave + 4 * j) * 1024), 16, va[j], soff, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(smem + bo + 16384 + (wave + 4 * j) * 1024), 16, vb[j], soff, 0, 0);
  };
  // fragment address of row (lane & 31) of a 32-row block, k-half kk: logical chunk 2 kk + (lane >> 5)
  int L[2];
  {
    const int r = lane & 31, hi = lane >> 5, sw = (r >> 2) & 3;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) L[kk] = r * 64 + (((kk * 2 + hi) ^ sw) << 4);
  }
  const int aoff = wm * 128 * 64, boff = 16384 + wn * 64 * 64;

  bf16x8 Af[2][4];        // [register set = k-half parity][row block]
  bf16x8 Bf[2][2];        // [set][column block]
  f32x16 acc[4][2];       // [row block][column block]
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][c][r] = 0.f;

  auto read_frags = [&](auto set_c, int bo) {          // k-half `set` of the stage at byte offset bo
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int c = 0; c < 2; c++) Bf[set][c] = *(const bf16x8*)(smem + bo + boff + c * 2048 + L[set]);
#pragma unroll
    for (int i = 0; i < 4; i++) Af[set][i] = *(const bf16x8*)(smem + bo + aoff + i * 2048 + L[set]);
  };
  auto mma = [&](auto set_c) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int c = 0; c < 2; c++)
        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[set][c], Af[set][i], acc[i][c], 0, 0, 0);
  };
  auto pin = [&](auto ndma_c) {
    constexpr int ndma = decltype(ndma_c)::value;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (m >= 2 && m - 2 < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    }
  };

  // ---- prologue: stages 0 .. 2 requested, stage 0 landed, its first k-half in registers ----
  stage(0, 0); stage(1, P4_STAGE_B); stage(2, 2 * P4_STAGE_B);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(IC<0>{}, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  int bc = 0, bn = P4_STAGE_B, bnn = 2 * P4_STAGE_B;       // byte offsets of the buffers of stages s, s + 1, s + 2
  for (int st = 0; st < ns; st++) {
    // first k-half: the second half's fragments arrive meanwhile; then stage s + 1 must have landed (stage s + 2 may fly)
    read_frags(IC<1>{}, bc);
    mma(IC<0>{});
    pin(IC<0>{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // second k-half: stage s + 3 into this stage's buffer (every wave has read it), first k-half of stage s + 1 into the other set
    read_frags(IC<0>{}, bn);
    stage(st + 3, bc);
    mma(IC<1>{});
    pin(IC<6>{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int o = bc; bc = bn; bn = bnn; bnn = o;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the out-of-range requests of the last steps: nothing may land in the staging tile)
  __builtin_amdgcn_s_barrier();

  // ---- epilogue ----
  TC* Cb = (TC*)P.C + (long)batch * P.cbs;
  const TC* Rb = (const TC*)P.residual;
  if (Rb) Rb += (long)batch * P.cbs;
  if (P.dbg & 2) return;
  if constexpr (sizeof(TC) == 2) {
    if (bf16_simple(P, Rb) && P.act != DU_ACT_SWIGLU && P.store_mode != DU_STORE_QKV_ROPE) {
      bf16_t* stg = (bf16_t*)smem;
      const int hi = lane >> 5;
      float4 bv[2][4];
#pragma unroll
      for (int c = 0; c < 2; c++) load_bias4(P, n0 + wn * 64 + c * 32, hi, bv[c]);
      auto stage_all = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int c = 0; c < 2; c++)
            stage_block_bf16<ACT>(acc[i][c], bv[c], stg, N_STG_LDB, wm * 128 + i * 32 + (lane & 31), wn * 64 + c * 32, hi);
      };
      if (P.act == DU_ACT_GELU) stage_all(IC<DU_ACT_GELU>{}); else stage_all(IC<DU_ACT_NONE>{});
      __syncthreads();
      readout_bf16<NBN, false, 256>(P, stg, N_STG_LDB, 256, m0, n0, P.N, (bf16_t*)Cb, tid);
      return;
    }
  }
  // fp32 staging, two passes: pass h holds row blocks 2 h, 2 h + 1 of both wave rows = tile rows wm * 128 + h * 64 + [0, 64)
  float* stg = (float*)smem;
  bool pre = false;
  if constexpr (sizeof(TC) == 4) {
    pre = Rb && P.store_mode == DU_STORE_PLAIN && !(P.dbg & 4) && P.ldr % 4 == 0 && P.ldc % 4 == 0 && !((((uintptr_t)Rb) | ((uintptr_t)Cb)) & 15);
  }
  auto pass = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if constexpr (h > 0) __syncthreads();
    stage_block_f32(acc[2 * h][0], stg, N_STG_LDF, wm * 64 + (lane & 31), wn * 64, lane);
    stage_block_f32(acc[2 * h][1], stg, N_STG_LDF, wm * 64 + (lane & 31), wn * 64 + 32, lane);
    stage_block_f32(acc[2 * h + 1][0], stg, N_STG_LDF, wm * 64 + 32 + (lane & 31), wn * 64, lane);
    stage_block_f32(acc[2 * h + 1][1], stg, N_STG_LDF, wm * 64 + 32 + (lane & 31), wn * 64 + 32, lane);
    __syncthreads();
    // staging rows [0, 64) = tile rows h * 64 + [0, 64), staging rows [64, 128) = tile rows 128 + h * 64 + [0, 64)
    auto half = [&](auto w_c) {
      constexpr int w = decltype(w_c)::value;
      const int mrow0 = m0 + w * 128 + h * 64;
      const float* sp = stg + w * 64 * N_STG_LDF;
      if constexpr (sizeof(TC) == 4) {
        if (pre) {
          // 64 rows x 128 columns = 2048 float4 = 8 per thread: the residual requested up front (its latency is paid once per half)
          constexpr int CW = NBN / 4;
          float4 rr[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int v = tid + u * 256;
            const int m = mrow0 + v / CW, n = n0 + (v % CW) * 4;
            rr[u] = (m < P.M && n < P.N) ? *(const float4*)((const float*)Rb + (long)m * P.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int v = tid + u * 256;
            const int row = v / CW, cw = v % CW;
            const int m = mrow0 + row, n = n0 + cw * 4;
            if (m < P.M && n < P.N) {
              const float4 tt = *(const float4*)(sp + row * N_STG_LDF + cw * 4);
              float o[4] = {tt.x * P.alpha, tt.y * P.alpha, tt.z * P.alpha, tt.w * P.alpha};
              if (P.bias) {
                const float4 bb = *(const float4*)(P.bias + n);
                o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
              }
              if (P.act != DU_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = apply_act(o[e], P.act);
              }
              if (P.gamma) {
                const float4 gg = *(const float4*)(P.gamma + n);
                o[0] *= gg.x; o[1] *= gg.y; o[2] *= gg.z; o[3] *= gg.w;
              }
              if (P.row_scale) {
                const float rs = P.row_scale[m / P.rs_rows];
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] *= rs;
              }
              *(float4*)((float*)Cb + (long)m * P.ldc + n) = make_float4(o[0] + rr[u].x, o[1] + rr[u].y, o[2] + rr[u].z, o[3] + rr[u].w);
            }
          }
          return;
        }
      }
      readout_f32_any<TC, NBN, 256>(P, sp, N_STG_LDF, 64, mrow0, n0, Cb, Rb, tid);
    };
    half(IC<0>{});
    half(IC<1>{});
  };
  pass(IC<0>{});
  pass(IC<1>{});
}

// ================================================================================================================================
// PERSISTENT 256 x 128 kernel (round 5): one workgroup per CU walks a LIST of tiles; the K-tile ring never stops.
//
// Why: at K = 1024 a tile of the kernels above is 16 K-steps between a cold prologue (every workgroup of the launch asks for its first
// 96-128 KB at once) and an exposed epilogue (bias / GELU / bf16 pack / LDS-staged row stores with the matrix pipe idle): 9-17 us of a
// 36-83 us launch (profiles/r04_gemm_p8_table_v3.txt, "noepi" columns), twice per launch where a CU runs two rounds.  Two workgroups per
// CU did not hide it.  Here a workgroup keeps going:
//   * the K-tiles of tile i + 1 are requested by the last two K-steps of tile i -- the staging stream (LDS-DMA, counted vmcnt(6), three
//     48 KB buffers, the phase program of gemm_nt_p8n_kernel unchanged) does not know about tile borders; only the FIRST tile of a
//     workgroup has a prologue.  Operand rows past M / N and every request past the workgroup's last tile carry an out-of-range
//     voffset (zeros, no traffic), so the wait counts hold everywhere and there is no tail variant of the K-step;
//   * the finished accumulators (64 registers: the 256 x 128 kernel leaves 80 free) move to a second register set -- half of them in
//     the MFMA shadow of the tile's last phase, the other half in the first phase of the next tile -- and the first K-step of the next
//     tile starts its accumulators from the constant 0 (the MFMA's C operand): no zeroing pass;
//   * that second set is DRAINED in the MFMA shadows of the next tile's first four K-steps, one 32 x 16 piece per phase: bias from a
//     128-float LDS image of the tile's bias slice (LDS-DMA by waves 0 / 1 a tile ahead, two slots), erf-GELU polynomial, bf16 pack, two
//     v_permlane32_swap (lanes 0-31 then hold 8 consecutive columns of their row, lanes 32-63 the next 8 -- the strip convolution's
//     store, DESIGN 6.48, with adjacent accumulator groups paired so that a row's two lanes write 32 contiguous bytes), ONE 16-byte buffer
//     store per lane -- no staging tile, no barrier, nothing after the last MFMA
//     of a tile but the register move.  Only the workgroup's LAST tile drains in the open.
// Stores sit in the same vmcnt queue as the LDS-DMA: the counted waits are left as they are (a store among the six youngest operations
// only makes a wait stricter; a store the descriptor's range check drops retires at once, DESIGN 6.49, and is equally harmless here).
// Tiles are dealt out statically: workgroup w (XCD-contiguous index) takes tiles w, w + G, w + 2 G, ... of the banded tile order, so
// the G tiles in flight at any time are the ones a non-persistent launch would run in one round.
// Served: bf16 result, epilogue = bias (+ GELU), plain store, K >= 384 (the drain takes four K-steps of the next tile) or K = 256 (NK4:
// the tile is four K-steps and the drain of tile i runs under all of tile i + 1 -- the adapter's 43008 x 1024 x 256 products are ALL
// prologue and epilogue on the kernels above).  Results are bit-identical to gemm_nt_p8n_kernel (same K order, same epilogue arithmetic).
// ================================================================================================================================
constexpr int PP_BIAS_OFF = P8N_LDS;              // bias slots (128 floats each) behind the three K-tile buffers
constexpr int PP_LDS = P8N_LDS + 3 * 128 * 4;     // 148 992 B (three bias slots: the K = 256 form keeps an image through the next tile)
constexpr unsigned PP_OOR = 0x80000000u;          // a voffset past every descriptor's range (the range check ignores the scalar offset)

__device__ __forceinline__ void pp_swap_halves(unsigned& a, unsigned& b) {      // a's lanes 32-63 <-> b's lanes 0-31
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// (free functions: an asm operand that is a local captured by a generic lambda does not compile)
__device__ __forceinline__ void pp_lds_read1(f32x4& a, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=&v"(a) : "v"(addr)); }
__device__ __forceinline__ void pp_wait_lds(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void pp_wait_lds1(f32x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }

// RES (round 6): the bf16 RESIDUAL of the product (y = x + f(...), the adapter's output projection and ConvFFN fc2 into the query stream,
// dinov3_adapter.py:142-148) enters as TWO MORE K-STEPS of the same tile: A' = [A | R(rows, the tile's 128 columns)], W' = [W | I_128], so
// sum_k' R[m][k'] * I[n][k'] = R[m][n] lands in the accumulators through the staging ring like any other K-tile -- no residual registers
// (the kernel has none to spare), no read in the drain, the 88 MB of a 43008 x 1024 residual stream travel under the MFMAs of the
// tile before.  bf16 x 1.0 accumulated in fp32 is exact.  RES == 2: DropPath's per-sample scale (row_scale, rs_rows % 256 == 0: a
// tile lies inside one sample) multiplies the accumulators between the last K-step of A and the first of R (acc[0] in the phase that
// works on acc[1] and vice versa) and the bias in the drain: y = s * (A W^T + b) + R.  One-shot kernels: 95.6 / 111.8 us at
// 43008 x 1024 x {256, 512} (the epilogue re-reads the residual with the matrix pipe idle, profiles/r05_gemm_residual_epilogue_v1.txt).
struct PpIdent {
  unsigned short v[128 * 128];
  constexpr PpIdent() : v() { for (int i = 0; i < 128; i++) v[i * 128 + i] = 0x3F80; }      // bf16 1.0 on the diagonal
};
__device__ const PpIdent g_pp_ident{};

// PS: the pixel-shuffle store of ConvTranspose2d k2 s2 (column n = (tap, co) of input pixel m = (b, y, x) -> output pixel (b, 2y + tap / 2,
// 2x + tap % 2), channel co; ps_C % 128 == 0: a tile's 128 columns lie in one tap; ps_H, ps_W powers of two), with the residual read through
// the same mapping -- the adapter's `up` + c1 (dinov3_adapter.py:360,467), 32768 x 4096 x 1024: 428 us on the one-shot 256 x 256 kernel.
// ROPE (round 6): the ViT's qkv projection stored head-major with RoPE applied in the drain (DU_STORE_QKV_ROPE, layers/attention.py:66-85):
// a wave's 64 columns are one (q | k | v, head); the rotate-half partners d and d + 32 are the SAME accumulator register of column blocks
// c = 0 and c = 1, so a drain phase takes group g of BOTH blocks, rotates in fp32 and packs; the sin / cos of (token, d) come from a
// FACTORISED table in LDS -- the angle of dimension d is (row coordinate) / period_j for d % 32 < 16 and (column coordinate) / period_j
// above (rope_position_encoding.py:98-104, tiled twice), so 2 x (H_t + 1) x 16 (cos, sin) pairs = 8.4 KB replace the (H_t W_t) x 64 tables;
// the extra row is the identity rotation (prefix tokens, v).  The quads are fetched one phase ahead beside the bias quads.  q is scaled
// by rope_qscale.  With this the 50 MB qkv matrix and the pass that re-read it (qkv_rope_split_kernel, 21 us per block) do not exist.
constexpr int PP_ROPE_OFF = PP_BIAS_OFF + 2048;           // the factorised table behind the bias slots: [axis][33 positions][cos 16 | sin 16] fp32
constexpr int PP_ROPE_ROWS = 33;
constexpr int PP_LDS_ROPE = PP_ROPE_OFF + 2 * PP_ROPE_ROWS * 128;
__device__ __forceinline__ void pp_wait_lds4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

// QKV == 2 (DU_STORE_QKV_HEADS): the same head-major store WITHOUT the rotation -- the plain drain with another row offset, no extra
// registers, 16-byte stores; du_qkv_rope_inplace then rotates q and k where they lie (67 MB instead of the 101 MB of qkv_rope_split).
template <int ACT, bool NK4, int RES = 0, bool PS = false, int QKV = 0>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmParams P) {
  constexpr bool ROPE = QKV == 1, HEADS = QKV == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (P.tail_rows && (int)blockIdx.x >= P.main_wgs) { p8_tail<bf16_t>(P, smem); return; }
  static_assert(RES == 0 || ACT == DU_ACT_NONE, "the residual form has no activation");
  static_assert(QKV == 0 || (ACT == DU_ACT_NONE && !NK4 && RES == 0 && !PS), "the qkv stores: plain bias epilogue, K >= 384");
  constexpr bool FOUR = NK4 && RES == 0;          // the four-K-step tile form (K = 256 without a residual); with one the tile is six K-steps
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = P.main_wgs;                       // persistent workgroups of this launch
  int lin;
  {
    const int bid = blockIdx.x, q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntiles = P.tiles_m * P.tiles_n;
  const int nk = P.K / PBK + (RES ? 2 : 0);       // K-steps of a tile (RES: the residual's two behind A's)
  auto coords = [&](int tile, int& m0, int& n0) {
    int tm, tn;
    if (P.group_m > 1) {
      const int band = P.group_m * P.tiles_n;
      const int g = tile / band, l = tile - g * band;
      const int first = g * P.group_m;
      const int gsz = min(P.tiles_m - first, P.group_m);
      tn = l / gsz; tm = first + (l - tn * gsz);
    } else {
      tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n;
    }
    m0 = tm * PBM; n0 = tn * NBN;
  };

  // whole-matrix descriptors (the host checked every extent against 2^31 bytes); the tile's first row travels in the scalar offset
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)P.a.p, 0, (int)((((long)P.M - 1) * P.a.ld + P.K) * 2), 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)P.b.p, 0, (int)((((long)P.N - 1) * P.b.ld + P.K) * 2), 0x00020000);
  const long out_rows = PS ? 4L * P.M : (long)P.M;           // rows (pixels) of C and of the residual
  const int out_cols = PS ? P.ps_C : P.N;
  const auto rc = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, QKV ? (int)(3 * P.ldc * 2) : (int)(((out_rows - 1) * P.ldc + out_cols) * 2), 0x00020000);
  const int ps_lw = PS ? __builtin_ctz(P.ps_W) : 0, ps_lh = PS ? __builtin_ctz(P.ps_H) : 0;
  // PS: output pixel (before the tap's shift) of input pixel m, as a row index of C / the residual
  auto ps_pixel = [&](unsigned m) -> unsigned {
    const unsigned x = m & (unsigned)(P.ps_W - 1), y = (m >> ps_lw) & (unsigned)(P.ps_H - 1), b = m >> (ps_lw + ps_lh);
    return ((b * 2u * (unsigned)P.ps_H + 2u * y) * 2u * (unsigned)P.ps_W) + 2u * x;
  };
  const auto rbias = __builtin_amdgcn_make_buffer_rsrc(P.bias ? (void*)P.bias : P.C, 0, P.bias ? P.N * 4 : 0, 0x00020000);
  const int nrec_r = RES ? (int)(((out_rows - 1) * P.ldr + out_cols) * 2) : 0;
  const unsigned ldr2 = (unsigned)(P.ldr * 2);

  // staging: piece (round r, wave w) of a half = tile rows (r*8 + w)*8 .. +8, lane l -> row + l/8, physical 16-byte chunk l%8 holds
  // logical chunk (l%8) ^ ((row >> 1) & 7) (as gemm_nt_p8n_kernel)
  const int lc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  int srow[2];                    // this lane's tile row of round r (A half 1: + 128)
  unsigned va0[2][2], vb0[2];     // voffsets relative to the tile's first row
#pragma unroll
  for (int r = 0; r < 2; r++) {
    srow[r] = (r * 8 + wave) * 8 + (lane >> 3);
    va0[0][r] = (unsigned)srow[r] * (unsigned)(P.a.ld * 2) + lc * 16;
    va0[1][r] = (unsigned)(128 + srow[r]) * (unsigned)(P.a.ld * 2) + lc * 16;
    vb0[r] = (unsigned)srow[r] * (unsigned)(P.b.ld * 2) + lc * 16;
  }
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned va_s[2][2], vb_s[2];   // ... of the tile being STAGED, out of range for rows past M / N and past the last tile
  unsigned sa_base = 0, sb_base = 0, sr_base = 0;
  unsigned sm0 = 0;               // PS: first row of the staged tile (the residual's lane offsets are absolute output pixels)
  unsigned sn0 = 0;               // ROPE: (sm0, sn0) = first row / column of the staged tile, 0x7fffffff behind the last tile
  // the tile whose K-tiles are requested from now on; its bias slice -> LDS slot `slot` (waves 0 / 1, one 4-byte LDS-DMA each)
  auto set_stage_tile = [&](int tile, int slot) {
    if (tile < ntiles) {
      int m0, n0;
      coords(tile, m0, n0);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if constexpr (ROPE) {      // (this form has no registers for the six lane offsets: `stage` recomputes them from the tile's scalars)
          sm0 = (unsigned)m0; sn0 = (unsigned)n0;
        } else {
          va_s[0][r] = m0 + srow[r] < P.M ? va0[0][r] : PP_OOR;
          va_s[1][r] = m0 + 128 + srow[r] < P.M ? va0[1][r] : PP_OOR;
          vb_s[r] = n0 + srow[r] < P.N ? vb0[r] : PP_OOR;
        }
      }
      sa_base = (unsigned)m0 * (unsigned)(P.a.ld * 2);
      sb_base = (unsigned)n0 * (unsigned)(P.b.ld * 2);
      if constexpr (RES != 0 && !PS) sr_base = (unsigned)m0 * ldr2 + (unsigned)n0 * 2u;
      if constexpr (RES != 0 && PS) {
        const int q = n0 / P.ps_C, co0 = n0 - q * P.ps_C;           // the tile's tap: pixel shift (q / 2) rows + (q % 2) columns
        sr_base = (unsigned)((q >> 1) * 2 * P.ps_W + (q & 1)) * ldr2 + (unsigned)co0 * 2u;
        sm0 = (unsigned)m0;
      }
      if (wave < 2) {
        // inline asm: behind an LDS-DMA builtin the compiler waits vmcnt(0) in front of every read of the bias image (it cannot tell the
        // slots apart) -- once per phase of the drain.  Unseen, the image is ordered by distance: it is read a whole tile later, behind
        // >= 30 counted waits and barriers.  M0 saved / restored (the compiler's own LDS-DMA keeps its destination there).
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_base + PP_BIAS_OFF + slot * 512 + wave * 256);
        const unsigned voff = (unsigned)(n0 + wave * 64 + lane) * 4u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rbias), "s"(lds_dst) : "memory");
      }
    } else {
#pragma unroll
      for (int r = 0; r < 2; r++) { va_s[0][r] = PP_OOR; va_s[1][r] = PP_OOR; vb_s[r] = PP_OOR; }
      if constexpr (ROPE) { sm0 = 0x7fffffffu; sn0 = 0x7fffffffu; }
    }
  };
  // which: 0 = A-half0, 1 = A-half1, 2 = B of the staged tile's K-tile at byte offset kofs, into the buffer at byte offset bo.
  // EXT (RES, compile time: the K-steps that request the residual are fixed points of the tile program): K-tile e = kofs / 128 of the
  // residual -- A' rows = the tile's rows of R at columns n0 + 64 e .. + 63, W' rows = rows of the 128 x 128 identity at the same columns
  const auto rr = __builtin_amdgcn_make_buffer_rsrc(RES ? (void*)P.residual : (void*)P.a.p, 0, nrec_r, 0x00020000);
  const auto ri = __builtin_amdgcn_make_buffer_rsrc((void*)g_pp_ident.v, 0, 32768, 0x00020000);
  auto stage = [&](auto which_c, int bo, unsigned kofs, auto ext_c) {
    constexpr int which = decltype(which_c)::value;
    constexpr bool EXT = RES != 0 && decltype(ext_c)::value != 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      unsigned char* dst = smem + bo + which * HALF_B + (r * 8 + wave) * 1024;
      if constexpr (EXT) {
        // (the lane offsets are recomputed at every request -- two VALU operations -- from a value the optimiser cannot see through:
        //  hoisted out of the tile loop they are six more live registers in a kernel that has none, and it spills)
        unsigned sr = (unsigned)srow[r];
        asm volatile("" : "+v"(sr));
        if constexpr (which < 2) {
          const unsigned rowi = PS ? ps_pixel(sm0 + which * 128u + sr) : which * 128u + sr;
          const unsigned vext = va_s[which][r] == PP_OOR ? PP_OOR : rowi * ldr2 + (unsigned)lc * 16u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void*)dst, 16, vext, sr_base + kofs, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (lds_void*)dst, 16, sr * 256u + (unsigned)lc * 16u, kofs, 0, 0);
        }
      } else if constexpr (ROPE) {
        unsigned sr = (unsigned)srow[r];
        asm volatile("" : "+v"(sr));
        if constexpr (which < 2) {
          const unsigned row = which * 128u + sr;
          const unsigned vo = sm0 + row < (unsigned)P.M ? row * (unsigned)(P.a.ld * 2) + (unsigned)lc * 16u : PP_OOR;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, vo, sa_base + kofs, 0, 0);
        } else {
          const unsigned vo = sn0 + sr < (unsigned)P.N ? sr * (unsigned)(P.b.ld * 2) + (unsigned)lc * 16u : PP_OOR;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)dst, 16, vo, sb_base + kofs, 0, 0);
        }
      } else {
        if constexpr (which < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, va_s[which][r], sa_base + kofs, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)dst, 16, vb_s[r], sb_base + kofs, 0, 0);
      }
    }
  };
  int L[4];
  {
    const int x = hi ^ ((lane >> 1) & 7);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) L[kk] = (lane & 31) * 128 + ((x ^ (2 * kk)) << 4);
  }
  const int aoff = wm * 32 * 128, boff = 2 * HALF_B + wn * 64 * 128;

  bf16x8 Af[2][4];        // [set = A half][kk]
  bf16x8 Bf[2][2][4];     // [set][column block][kk]
  f32x16 acc[2][2];       // [i = A half][column block]: the tile being computed
  f32x16 prev[2][2];      // the finished tile, being stored
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) prev[i][c][r] = 0.f;
  auto readA = [&](auto set_c, int bo) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) Af[set][kk] = *(const bf16x8*)(smem + bo + set * HALF_B + aoff + L[kk]);
  };
  auto readB = [&](auto set_c, int bo) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int kk = 0; kk < 4; kk++) Bf[set][c][kk] = *(const bf16x8*)(smem + bo + boff + c * 4096 + L[kk]);
  };
  // FIRST: the tile's first K-step -- the accumulators start from the MFMA's constant C operand
  auto mma = [&](auto i_c, auto bset_c, auto first_c) {
    constexpr int i = decltype(i_c)::value, bset = decltype(bset_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        if constexpr (FIRST) {
          if (kk == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[bset][c][kk], Af[i][kk], z, 0, 0, 0);
            continue;
          }
        }
        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bf[bset][c][kk], Af[i][kk], acc[i][c], 0, 0, 0);
      }
  };
  auto pin = [&](auto nrd_c, auto ndma_c) {
    constexpr int nrd = decltype(nrd_c)::value, ndma = decltype(ndma_c)::value;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m < 4) __builtin_amdgcn_sched_group_barrier(0x100, nrd / 4, 0);
      else if (m - 4 < ndma) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // VMEM READ: the LDS-DMA, not the drain's store
    }
  };
  // ---- drain of the finished tile.  Unit u = (i, c, h2): the 32 rows x 16 columns 16 h2 .. + 15 of block (i, c), one 16-byte store per
  // lane (a row's two lanes store adjacent pieces: 32 contiguous bytes per row and instruction).  A unit is two HALVES (h = 2 u + part):
  // part 0 = accumulator group g = 2 h2 (4 values per lane: bias, activation, bf16 pair x 2, kept in two registers), part 1 = group
  // 2 h2 + 1, then the two v_permlane32_swap and the store.  The 16 halves go over FOUR K-steps, one unit per phase (dealing them over
  // eight K-steps, one half per phase, was built for the GELU drain and measured the same: 77.8 vs 77.6 us on fc1,
  // profiles/r05_gemm_p8_table_v2.txt).  NK4 (K = 256): the tile IS four K-steps, so the drain of tile i runs under ALL of tile i + 1,
  // the K-steps that request tile i + 2 included, and a bias image lives through the next tile: three slots. ----
  const int nlim = (P.dbg & 1) ? 0 : P.N;    // (du_set_option key 3 bit 0: no stores -- timing ablation)
  unsigned pbias = lds_base + PP_BIAS_OFF + (unsigned)((wn * 64 + 4 * hi) * 4);     // this lane's corner of the finished tile's bias image
  unsigned crow[2] = {PP_OOR, PP_OOR};   // byte offset of this lane's row of A half i, at its first column (wave's 64 + 8 hi), in C
  int pcol = 0;                   // that first column
  float rs_cur = 1.0f, rs_prev = 1.0f;   // RES == 2: DropPath scale of the tile being computed / drained (uniform: a tile lies inside one sample)
  unsigned trow[2] = {0u, 0u};    // ROPE: this lane's rows' table positions, (row position) | (column position) << 8; 32 = the identity rotation
  float qs_prev = 1.0f;           // ROPE: rope_qscale for a q tile, else 1
  const unsigned rope_tbl = lds_base + PP_ROPE_OFF + (unsigned)(4 * hi * 4);     // this lane's corner (+ 4 hi entries) of a table row
  auto set_prev_tile = [&](int tile) {
    int m0, n0;
    coords(tile, m0, n0);
    pcol = n0 + wn * 64 + hi * 8;
    if constexpr (QKV != 0) {
      const int hd = P.ps_C * 64, which = n0 / hd, head = ((n0 - which * hd) >> 6) + wn;     // this wave's (q | k | v, head)
      if constexpr (ROPE) qs_prev = which == 0 ? P.rope_qscale : 1.0f;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int m = m0 + i * 128 + wm * 32 + (lane & 31);
        const int b = m / P.ps_H, t = m - b * P.ps_H;
        // (ROPE stores 8 bytes per lane: + 4 hi elements; HEADS the plain drain's 16 bytes: + 8 hi, and its column offsets c * 32 + 16 h2 stay)
        crow[i] = m < P.M ? (unsigned)(((long)which * P.ldc + (((long)b * P.ps_C + head) * P.ps_W + t) * 64 + hi * (ROPE ? 4 : 8)) * 2) : PP_OOR;
        if constexpr (ROPE) {
          const int tt = t - P.rope_prefix, ty = tt / P.b.Wi, tx = tt - ty * P.b.Wi;
          trow[i] = (which < 2 && tt >= 0 && m < P.M) ? (unsigned)ty | ((unsigned)tx << 8) : (32u | (32u << 8));
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = m0 + i * 128 + wm * 32 + (lane & 31);
      if constexpr (PS) {
        const int q = n0 / P.ps_C, co0 = n0 - q * P.ps_C;
        const unsigned pix = ps_pixel((unsigned)m) + (unsigned)((q >> 1) * 2 * P.ps_W + (q & 1));
        crow[i] = m < P.M ? (unsigned)(((long)pix * P.ldc + co0 + wn * 64 + hi * 8) * 2) : PP_OOR;
      } else {
        crow[i] = m < P.M ? (unsigned)(((long)m * P.ldc + pcol) * 2) : PP_OOR;
      }
    }
  };
  // The bias quad of a half is fetched ONE PHASE AHEAD by an inline-asm ds_read the compiler does not see: behind the LDS-DMA builtins it
  // puts s_waitcnt vmcnt(0) in front of any read of the bias image (it cannot tell the image from the K-tile ring), which drains the
  // ring once per phase.  The phase-end wait (lgkmcnt(0), naming the destinations so they stay opaque until then) retires the reads.
  f32x4 bqa = {0.f, 0.f, 0.f, 0.f}, bqb = {0.f, 0.f, 0.f, 0.f};     // bias of the next phase's first / second half
  unsigned keep0 = 0u, keep1 = 0u;                                  // packed group of the unit in progress (part 0)
  f32x4 cq = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};       // ROPE: cos / sin of the next phase's group
  auto fetch = [&](auto h_c, f32x4& bq, unsigned base) {            // columns c*32 + 8 g + 4 hi + e of the image at `base`
    constexpr int h = decltype(h_c)::value;
    if constexpr (h >= 0) {
      if constexpr (ROPE) {
        // ROPE: half h = 2 ph + c: phase ph takes group g = ph & 3 of row block i = ph >> 2 for BOTH column blocks; with the c = 0 half the
        // cos / sin quads of (row, dimensions 8 g + 4 hi .. + 3): axis = g >> 1, 16-entry table rows, j = (8 g) % 16 + 4 hi
        constexpr int ph = h >> 1, c = h & 1, i = ph >> 2, g = ph & 3, axis = g >> 1;
        pp_lds_read1(bq, base + (unsigned)((c * 32 + 8 * g) * 4));
        if constexpr (c == 0) {
          const unsigned pos = axis ? (trow[i] >> 8) : (trow[i] & 0xffu);
          const unsigned ta = rope_tbl + (unsigned)(axis * PP_ROPE_ROWS * 128 + ((8 * g) & 15) * 4) + pos * 128u;
          pp_lds_read1(cq, ta);
          pp_lds_read1(sq, ta + 64u);
        }
      } else {
        constexpr int u = h >> 1, c = (u >> 1) & 1, g = 2 * (u & 1) + (h & 1);
        pp_lds_read1(bq, base + (unsigned)((c * 32 + 8 * g) * 4));
      }
    }
  };
  // ROPE: one phase = group g of both column blocks of row block i (bqa / bqb: their bias quads, cq / sq: the rotation, fp32 throughout)
  auto rope_pair = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if constexpr (h >= 0) {
      constexpr int ph = h >> 1, i = ph >> 2, g = ph & 3, h2 = g >> 1;
      const f32x16& a0 = prev[i][0];
      const f32x16& a1 = prev[i][1];
      unsigned pk[4];
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const float ca = cq[e], sa = sq[e], cb = cq[e + 1], sb = sq[e + 1];
        const float x1a = a0[4 * g + e] + bqa[e], x1b = a0[4 * g + e + 1] + bqa[e + 1];
        const float x2a = a1[4 * g + e] + bqb[e], x2b = a1[4 * g + e + 1] + bqb[e + 1];
        // layers/attention.py:16-27: x cos + rotate_half(x) sin, rotate_half = [-x2, x1]
        const float o1a = (x1a * ca - x2a * sa) * qs_prev, o1b = (x1b * cb - x2b * sb) * qs_prev;
        const float o2a = (x2a * ca + x1a * sa) * qs_prev, o2b = (x2b * cb + x1b * sb) * qs_prev;
        const bf16x2 t1 = {(bf16_t)o1a, (bf16_t)o1b}, t2 = {(bf16_t)o2a, (bf16_t)o2b};
        pk[e >> 1] = __builtin_bit_cast(unsigned, t1);
        pk[2 + (e >> 1)] = __builtin_bit_cast(unsigned, t2);
      }
      // 8 bytes per lane (dimensions 8 g + 4 hi .. + 3 of this lane's row, a row's two lanes adjacent).  The 16-byte form of the plain
      // drain (pairs kept across a phase + v_permlane32_swap) needs four more registers than this kernel has: it spills (104 B / lane)
      (void)h2;
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t v0 = {pk[0], pk[1]}, v1 = {pk[2], pk[3]};
      const unsigned off = nlim ? crow[i] + (unsigned)(8 * g * 2) : PP_OOR;
      __builtin_amdgcn_raw_buffer_store_b64(v0, rc, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(v1, rc, off == PP_OOR ? PP_OOR : off + 64u, 0, 0);
    }
  };
  auto half = [&](auto h_c, const f32x4& bq) {
    constexpr int h = decltype(h_c)::value;
    if constexpr (h >= 0) {
      constexpr int u = h >> 1, part = h & 1, i = u >> 2, c = (u >> 1) & 1, h2 = u & 1, g = 2 * h2 + part;
      // (no "is there a finished tile" branch: before the first one `prev` holds zeros and crow is out of range -- the store is dropped;
      //  a branch would end the phase's scheduling region in front of the drain and push it behind the MFMAs)
      const f32x16& a = prev[i][c];
      f32x2 o0, o1;
      if constexpr (RES == 2) {
        o0 = {__builtin_fmaf(bq[0], rs_prev, a[4 * g]), __builtin_fmaf(bq[1], rs_prev, a[4 * g + 1])};
        o1 = {__builtin_fmaf(bq[2], rs_prev, a[4 * g + 2]), __builtin_fmaf(bq[3], rs_prev, a[4 * g + 3])};
      } else {
        o0 = {a[4 * g] + bq[0], a[4 * g + 1] + bq[1]}; o1 = {a[4 * g + 2] + bq[2], a[4 * g + 3] + bq[3]};
      }
      // (packed fp32: a one-element-at-a-time form on plain v_fma_f32 -- the micro-architecture guide prices a packed operation beside MFMAs
      //  at +22 cycles -- measured SLOWER here, 87.0 vs 84.5 us on the fc1 product, profiles/r05_gemm_p8_table_v3.txt: 11 instead of 7
      //  instructions per element at two waves per SIMD)
      if constexpr (ACT == DU_ACT_GELU) { o0 = gelu_pk(o0); o1 = gelu_pk(o1); }
      const bf16x2 t0 = {(bf16_t)o0.x, (bf16_t)o0.y}, t1 = {(bf16_t)o1.x, (bf16_t)o1.y};
      if constexpr (part == 0) {
        keep0 = __builtin_bit_cast(unsigned, t0); keep1 = __builtin_bit_cast(unsigned, t1);
      } else {
        unsigned k4 = __builtin_bit_cast(unsigned, t0), k5 = __builtin_bit_cast(unsigned, t1);
        pp_swap_halves(keep0, k4);     // lanes 0-31: (own group 2 h2 | partner's group 2 h2) = columns 16 h2 .. + 7;
        pp_swap_halves(keep1, k5);     // lanes 32-63: (partner's group 2 h2 + 1 | own) = columns 16 h2 + 8 .. + 15: 32 contiguous bytes per row
        const u32x4_t v = {keep0, keep1, k4, k5};
        const unsigned off = pcol + c * 32 + 16 * h2 < nlim ? crow[i] + (unsigned)((c * 32 + 16 * h2) * 2) : PP_OOR;
        __builtin_amdgcn_raw_buffer_store_b128(v, rc, off, 0, 0);
      }
    }
  };
  auto finish6 = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (ROPE) pp_wait_lds4(bqa, bqb, cq, sq); else pp_wait_lds(bqa, bqb);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // one K-step.  p: fragment set of this K-tile's B; kofs: byte offset (inside the operand rows) of the K-tile requested now (the staged
  // tile's); HA0 / HB0 (HA1 / HB1): the halves drained in phase 0 (1), -1 = none; the bias of the NEXT phase's halves is fetched at the end
  // of a phase (phase 1 fetches NA / NB = the following K-step's phase-0 halves) from the image at fbase; CP1: prev[1] <- acc[1] in phase
  // 0 (first K-step of a tile, before its phase 1 restarts acc[1]); CP0: prev[0] <- acc[0] in phase 1 (last K-step: acc[0] is final)
  // SC (RES == 2): 1 = this is A's last K-step: acc[0] (final after phase 0) is scaled in phase 1; 2 = the residual's first K-step: acc[1]
  // is scaled in phase 0, before phase 1 adds the residual to it
  auto scale_acc = [&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][c][r] *= rs_cur;
  };
  auto kstep = [&](auto par_c, auto first_c, auto ha0, auto hb0, auto ha1, auto hb1, auto na, auto nb, auto cp1_c, auto cp0_c, int bc, int bn,
                   int bnn, unsigned kofs, unsigned fbase, unsigned fbase2, auto sc_c, auto ext_c) {
    constexpr int p = decltype(par_c)::value;
    constexpr int SC = RES == 2 ? decltype(sc_c)::value : 0;
    // phase 0
    readA(IC<1>{}, bc);
    stage(IC<0>{}, bnn, kofs, ext_c); stage(IC<2>{}, bnn, kofs, ext_c);
    if constexpr (decltype(cp1_c)::value) { prev[1][0] = acc[1][0]; prev[1][1] = acc[1][1]; }
    if constexpr (SC == 2) scale_acc(IC<1>{});
    mma(IC<0>{}, IC<p>{}, first_c);
    if constexpr (ROPE) rope_pair(ha0); else { half(ha0, bqa); half(hb0, bqb); }
    pin(IC<4>{}, IC<4>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(ha1, bqa, fbase); fetch(hb1, bqb, fbase);
    finish6();
    // phase 1
    readA(IC<0>{}, bn); readB(IC<1 - p>{}, bn);
    stage(IC<1>{}, bnn, kofs, ext_c);
    if constexpr (decltype(cp0_c)::value) { prev[0][0] = acc[0][0]; prev[0][1] = acc[0][1]; }
    if constexpr (SC == 1) scale_acc(IC<0>{});
    mma(IC<1>{}, IC<p>{}, first_c);
    if constexpr (ROPE) rope_pair(ha1); else { half(ha1, bqa); half(hb1, bqb); }
    pin(IC<12>{}, IC<2>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(na, bqa, fbase2); fetch(nb, bqb, fbase2);      // (fbase2: the image the FOLLOWING K-step's halves belong to)
    finish6();
  };
  using T_ = IC<1>; using F_ = IC<0>; using N_ = IC<-1>;
  // drain K-step j (0 .. 3): phase 0 halves (4 j, 4 j + 1), phase 1 (4 j + 2, 4 j + 3).  NK4: K-step 3 is also the tile's last (CP0, and
  // its last phase fetches the bias of the first unit of THIS tile, from the image at nbase)
  auto dstep = [&](auto j_c, auto par_c, int bc_, int bn_, int bnn_, unsigned kofs, unsigned nbase) {
    constexpr int j = decltype(j_c)::value;
    constexpr bool wrap = FOUR && j == 3;
    // (RES with K = 256: the tile is six K-steps; K-steps 2 and 3 request the residual's K-tiles 0 and 1, K-step 3 is A's last)
    constexpr bool ext = NK4 && RES != 0 && j >= 2;
    kstep(par_c, IC<(j == 0)>{}, IC<4 * j>{}, IC<4 * j + 1>{}, IC<4 * j + 2>{}, IC<4 * j + 3>{}, IC<(j < 3 ? 4 * j + 4 : (wrap ? 0 : -1))>{},
          IC<(j < 3 ? 4 * j + 5 : (wrap ? 1 : -1))>{}, IC<(j == 0)>{}, IC<wrap>{}, bc_, bn_, bnn_, ext ? (unsigned)(j - 2) * 128u : kofs, pbias,
          wrap ? nbase : pbias, IC<((ext && j == 3) ? 1 : 0)>{}, IC<(ext ? 1 : 0)>{});
  };

  // ---- this workgroup's first tile: the only prologue (K-tiles 0 and 1) ----
  int tile = lin;
  if (tile >= ntiles) return;
  if constexpr (ROPE) {
    // the factorised rotation table: row `pos` of axis 0 = dimensions 0..15 of token (pos, 0), of axis 1 = dimensions 16..31 of token
    // (0, pos) -- the tables are separable by construction (rope_position_encoding.py:98-104) --, row 32 of either = the identity
    float* tb = (float*)(smem + PP_ROPE_OFF);
    for (int v = tid; v < 2 * PP_ROPE_ROWS * 32; v += 512) {
      const int axis = v / (PP_ROPE_ROWS * 32), r = v - axis * (PP_ROPE_ROWS * 32), pos = r >> 5, e = r & 31, j = e & 15, is_sin = e >> 4;
      float val = is_sin ? 0.f : 1.f;
      const bool live = axis ? pos < P.b.Wi : pos < P.b.Hi;
      if (live) {
        const long tok = axis ? pos : (long)pos * P.b.Wi;
        val = (is_sin ? P.rope_sin : P.rope_cos)[tok * 64 + axis * 16 + j];
      }
      tb[v] = val;
    }
    __syncthreads();
  }
  set_stage_tile(tile, 0);
  stage(IC<0>{}, 0, 0u, F_{}); stage(IC<2>{}, 0, 0u, F_{}); stage(IC<1>{}, 0, 0u, F_{});
  stage(IC<0>{}, NBUF_B, 128u, F_{}); stage(IC<2>{}, NBUF_B, 128u, F_{}); stage(IC<1>{}, NBUF_B, 128u, F_{});
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // K-tile 0 landed
  __builtin_amdgcn_s_barrier();
  readA(IC<0>{}, 0);
  readB(IC<0>{}, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  int bc = 0, bn = NBUF_B, bnn = 2 * NBUF_B;
  int slot = 0;
  auto rot2 = [&]() { const int o = bc; bc = bnn; bnn = bn; bn = o; };      // advance two K-tiles: (bc, bn, bnn) <- (bnn, bc, bn)
  auto bias_at = [&](int sl) { return lds_base + PP_BIAS_OFF + (unsigned)(sl * 512 + (wn * 64 + 4 * hi) * 4); };
  for (;;) {
    const int next = tile + G;
    if constexpr (RES == 2) {
      int m0c, n0c;
      coords(tile, m0c, n0c);
      rs_cur = P.row_scale[__builtin_amdgcn_readfirstlane(m0c / P.rs_rows)];
    }
    if constexpr (FOUR) {
      // four K-steps = the tile; the previous tile drains under all of them.  K-steps 0, 1 request this tile's K-tiles 2, 3; then the
      // staging moves on: K-steps 2, 3 request K-tiles 0, 1 of the NEXT tile (out of range behind the last one)
      const int nslot = slot == 2 ? 0 : slot + 1;
      dstep(IC<0>{}, IC<0>{}, bc, bn, bnn, 2u * 128u, 0u);
      dstep(IC<1>{}, IC<1>{}, bn, bnn, bc, 3u * 128u, 0u);
      rot2();
      set_stage_tile(next, nslot);
      dstep(IC<2>{}, IC<0>{}, bc, bn, bnn, 0u, 0u);
      dstep(IC<3>{}, IC<1>{}, bn, bnn, bc, 128u, bias_at(slot));
      rot2();
      pbias = bias_at(slot);
      set_prev_tile(tile);
      slot = nslot;
    } else {
      // K-steps 0 .. 3: the previous tile drains (the bias of its first unit was fetched by the last phase of that tile)
      dstep(IC<0>{}, IC<0>{}, bc, bn, bnn, 2u * 128u, 0u);
      dstep(IC<1>{}, IC<1>{}, bn, bnn, bc, 3u * 128u, 0u);
      rot2();
      dstep(IC<2>{}, IC<0>{}, bc, bn, bnn, 4u * 128u, 0u);
      dstep(IC<3>{}, IC<1>{}, bn, bnn, bc, 5u * 128u, 0u);
      rot2();
      int t = 4;
      // (RES: the requests (t + 2) * 128 >= kext of the last pair before the final one are the residual's two K-tiles; that pair is peeled
      //  because its second K-step is A's last -- the DropPath scale hook)
      for (; t + (RES != 0 && !NK4 ? 4 : 2) < nk; t += 2) {
        kstep(IC<0>{}, F_{}, N_{}, N_{}, N_{}, N_{}, N_{}, N_{}, F_{}, F_{}, bc, bn, bnn, (unsigned)(t + 2) * 128u, 0u, 0u, IC<0>{}, IC<0>{});
        kstep(IC<1>{}, F_{}, N_{}, N_{}, N_{}, N_{}, N_{}, N_{}, F_{}, F_{}, bn, bnn, bc, (unsigned)(t + 3) * 128u, 0u, 0u, IC<0>{}, IC<0>{});
        rot2();
      }
      if constexpr (RES != 0 && !NK4) {      // A's last two K-steps request the residual's K-tiles 0 and 1
        kstep(IC<0>{}, F_{}, N_{}, N_{}, N_{}, N_{}, N_{}, N_{}, F_{}, F_{}, bc, bn, bnn, 0u, 0u, 0u, IC<0>{}, IC<1>{});
        kstep(IC<1>{}, F_{}, N_{}, N_{}, N_{}, N_{}, N_{}, N_{}, F_{}, F_{}, bn, bnn, bc, 128u, 0u, 0u, IC<1>{}, IC<1>{});
        rot2();
      }
      // the last two K-steps request K-tiles 0 and 1 of the NEXT tile (out of range behind the last one); the very last phase fetches the
      // bias of the first unit of THIS tile (its image: slot `slot`)
      set_stage_tile(next, slot ^ 1);
      pbias = bias_at(slot);
      // (ROPE: the last phase below fetches the first rotation of THIS tile, which needs its rows' table positions -- the drain of the
      //  previous tile ended with K-step 3, so its store offsets may be replaced here already)
      if constexpr (ROPE) set_prev_tile(tile);
      kstep(IC<0>{}, F_{}, N_{}, N_{}, N_{}, N_{}, N_{}, N_{}, F_{}, F_{}, bc, bn, bnn, 0u, 0u, 0u, IC<2>{}, IC<0>{});
      kstep(IC<1>{}, F_{}, N_{}, N_{}, N_{}, N_{}, IC<0>{}, IC<1>{}, F_{}, T_{}, bn, bnn, bc, 128u, 0u, pbias, IC<0>{}, IC<0>{});
      rot2();
      if constexpr (!ROPE) set_prev_tile(tile);
      rs_prev = rs_cur;
      slot ^= 1;
    }
    tile = next;
    if (tile >= ntiles) break;
  }
  // ---- the last tile drains in the open ----
  prev[1][0] = acc[1][0]; prev[1][1] = acc[1][1];
  if (P.dbg & 2) return;
  auto last = [&](auto h_c) {          // half h (its bias is in bqa when h is even, else in bqb), then behind every unit the fetch of the next one's pair
    constexpr int h = decltype(h_c)::value;
    pp_wait_lds(bqa, bqb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (h & 1) half(h_c, bqb); else half(h_c, bqa);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (h & 1) { fetch(IC<(h + 1 < 16 ? h + 1 : -1)>{}, bqa, pbias); fetch(IC<(h + 2 < 16 ? h + 2 : -1)>{}, bqb, pbias); }
  };
  auto last_pair = [&](auto ph_c) {    // ROPE: phase ph of the drain in the open, then the fetch of the next phase's quads
    constexpr int ph = decltype(ph_c)::value;
    pp_wait_lds4(bqa, bqb, cq, sq);
    __builtin_amdgcn_sched_barrier(0);
    rope_pair(IC<2 * ph>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(IC<(ph + 1 < 8 ? 2 * ph + 2 : -1)>{}, bqa, pbias); fetch(IC<(ph + 1 < 8 ? 2 * ph + 3 : -1)>{}, bqb, pbias);
  };
  if constexpr (ROPE) {
    last_pair(IC<0>{}); last_pair(IC<1>{}); last_pair(IC<2>{}); last_pair(IC<3>{});
    last_pair(IC<4>{}); last_pair(IC<5>{}); last_pair(IC<6>{}); last_pair(IC<7>{});
  } else {
    last(IC<0>{}); last(IC<1>{}); last(IC<2>{}); last(IC<3>{}); last(IC<4>{}); last(IC<5>{}); last(IC<6>{}); last(IC<7>{});
    last(IC<8>{}); last(IC<9>{}); last(IC<10>{}); last(IC<11>{}); last(IC<12>{}); last(IC<13>{}); last(IC<14>{}); last(IC<15>{});
  }
  if (P.tail_rows && (int)gridDim.x == G) p8_tail_inline<bf16_t>(P, smem, lin, G);     // the ragged rows: units lin, lin + G, ... (round 6)
}

int g_p8_mode = -1;      // -1: heuristic, 0: off, 1: 256 x 256 wherever legal, 2: 256 x 128 wherever legal, 3: the 4-wave 256 x 128 kernel wherever legal
int g_p8_sched = 1;
int g_p8_corun = 1;      // du_set_option key 9: independent products the caller keeps in flight on different streams (the frozen ViT run as
                         // two half-batch chains): the tile choice then counts rounds on 256 / corun CUs -- a product that fills half the chip
                         // alone is a full round beside its twin
int g_p8_persist = 1;    // du_set_option key 10: 0 = never, 1 = the persistent 256 x 128 kernel where its epilogue / shape rules hold and a CU gets
                         // >= 2 tiles (default), 2 = wherever legal
int g_p8_tail_inline = 0; // du_set_option key 15: the ragged rows behind the last full tile row run as extra workgroups behind the tiles (0, round 3,
                         // default) or inside the tile workgroups (1, round 6: 1-1.4 us faster per ViT product alone, but x0.998 in the replayed
                         // step against the extra workgroups, profiles/r06_ab_tail_inline_v2.txt -- opt-in)
int g_p8_res = 1;        // du_set_option key 14 (A-B aid): 0 = products with a bf16 residual stay on the one-shot kernels (round 5)
int g_p8_pp_full = 0;    // du_set_option key 11 (A-B aid): 1 = the persistent kernel always launches min(tiles, 256) workgroups, co-running or not
int g_p8_group = 4;
int g_p8_debug = 0;      // bit 0: skip the bf16 global stores, bit 1: skip the whole epilogue (timing ablations only)

constexpr long KS_STATE_BYTES = 131072;       // [0, 64 KB): 4 words per tile (<= 4095 tiles) + the error word at int index 16380; [64 KB, 128 KB): tail tickets
int g_p8_tail_slices = 1;      // du_set_option key 17: 1 = ragged-row units of a long contraction (K >= 2048) are cut into K slices that meet through ks_ws
// slices per 32-column block of the ragged-row units (1 = whole contraction per unit): <= 8, >= 512 contraction elements each, at most one
// round of 256 workgroups; needs the persistent scratch
static int tail_slices_for(const du_gemm_args& a, bool have_ws) {
  if (!g_p8_tail_slices || a.K < (g_p8_tail_slices >= 2 ? 1024 : 2048) || g_p8_tail_inline) return 1;      // (key 17 = 2: A-B aid, K = 1024 too)
  const int units = (a.N + SK_BN - 1) / SK_BN;
  int s = a.K / 512;
  if (s > 8) s = 8;
  while (s > 1 && (a.K % (s * SK_CHUNK) || units * s > 256)) s--;
  if (s < 2 || units > 16384) return 1;
  if (have_ws && (!a.ks_ws || a.ks_ws_bytes < KS_STATE_BYTES + (long)units * s * 64 * SK_BN * 4)) return 1;
  return s;
}

template <typename TC, int SCHED, bool NARROW, bool GA = false>
int launch_p8(const du_gemm_args& a, hipStream_t st, int tail_rows = 0) {
  static_assert(!(NARROW && GA), "the gather form exists for the 256 x 256 kernel only");
  constexpr int LDS_BYTES = NARROW ? P8N_LDS : P8_LDS;
  constexpr int TBN = NARROW ? NBN : PBN;
  GemmParams P = make_params(a, GA ? DU_IM2COL_ROW : DU_PLAIN_ROW, DU_PLAIN_ROW, PBM, TBN, PBK);
  P.tiles_m = (a.M + PBM - 1) / PBM;
  P.group_m = g_p8_group;
  P.dbg = g_p8_debug;
  dim3 grid(P.tiles_m * P.tiles_n, a.batch < 1 ? 1 : a.batch);
  if (tail_rows > 0 && !GA) {       // the ragged rows behind M ride along: inside the tile workgroups (round 6), or one extra workgroup per 32 output columns
    P.tail_rows = tail_rows; P.main_wgs = P.tiles_m * P.tiles_n;
    // (inline only when the tiles fill the chip: with idle CUs -- 128 tiles of 256 x 256 -- the extra workgroups run beside the tiles for free)
    if (!(g_p8_tail_inline && grid.y == 1 && (long)grid.x * g_p8_corun >= 256)) {
      P.tail_slices = tail_slices_for(a, true); P.ks_ws = a.ks_ws;
      grid.x += (a.N + SK_BN - 1) / SK_BN * P.tail_slices;
    }
  }
  void (*kfn)(GemmParams);
  if constexpr (NARROW) kfn = gemm_nt_p8n_kernel<TC, SCHED>; else kfn = gemm_nt_p8_kernel<TC, SCHED, false, GA>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(512), LDS_BYTES, st, P);
  return du_check_launch();
}

// persistent 256 x 128 kernel: G = min(tiles, CUs this product can count on) workgroups, each walks tiles w, w + G, ...
int launch_pp(const du_gemm_args& a, hipStream_t st, int tail_rows = 0) {
  GemmParams P = make_params(a, DU_PLAIN_ROW, DU_PLAIN_ROW, PBM, NBN, PBK);
  P.tiles_m = (a.M + PBM - 1) / PBM;
  P.group_m = g_p8_group;
  P.dbg = g_p8_debug;
  const long ntiles = (long)P.tiles_m * P.tiles_n;
  const long cus = g_p8_pp_full ? 256 : 256 / g_p8_corun;
  P.main_wgs = (int)(ntiles < cus ? ntiles : cus);
  dim3 grid(P.main_wgs, 1);
  if (tail_rows > 0) {
    P.tail_rows = tail_rows;
    if (!g_p8_tail_inline) grid.x += (a.N + SK_BN - 1) / SK_BN;
  }
  const bool nk4 = a.K == 256;
  const int res = a.residual ? (a.row_scale ? 2 : 1) : 0;       // the residual as two more K-steps (+ DropPath's per-sample scale)
  const bool ps = a.store_mode == DU_STORE_PIXEL_SHUFFLE2;     // (pp_legal: with a residual, K >= 384)
  const bool rope = a.store_mode == DU_STORE_QKV_ROPE;         // (pp_legal: token grid <= 32 x 32, K >= 384)
  const bool heads = a.store_mode == DU_STORE_QKV_HEADS;
  if (rope) { P.b.Hi = a.geom.Hi; P.b.Wi = a.geom.Wi; }
  void (*kfn)(GemmParams);
  if (rope) kfn = gemm_nt_pp_kernel<DU_ACT_NONE, false, 0, false, 1>;
  else if (heads) kfn = gemm_nt_pp_kernel<DU_ACT_NONE, false, 0, false, 2>;
  else if (ps) kfn = gemm_nt_pp_kernel<DU_ACT_NONE, false, 1, true>;
  else if (res == 2) kfn = nk4 ? gemm_nt_pp_kernel<DU_ACT_NONE, true, 2> : gemm_nt_pp_kernel<DU_ACT_NONE, false, 2>;
  else if (res == 1) kfn = nk4 ? gemm_nt_pp_kernel<DU_ACT_NONE, true, 1> : gemm_nt_pp_kernel<DU_ACT_NONE, false, 1>;
  else if (nk4) kfn = a.act == DU_ACT_GELU ? gemm_nt_pp_kernel<DU_ACT_GELU, true> : gemm_nt_pp_kernel<DU_ACT_NONE, true>;
  else kfn = a.act == DU_ACT_GELU ? gemm_nt_pp_kernel<DU_ACT_GELU, false> : gemm_nt_pp_kernel<DU_ACT_NONE, false>;
  static bool attr_set[11] = {false, false, false, false, false, false, false, false, false, false, false};
  const int lds_bytes = rope ? PP_LDS_ROPE : PP_LDS;
  const int ai = heads ? 10 : rope ? 9 : ps ? 8 : (res ? 2 + 2 * res + (nk4 ? 1 : 0) : (a.act == DU_ACT_GELU ? 1 : 0) + (nk4 ? 2 : 0));
  if (!attr_set[ai]) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set[ai] = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(512), lds_bytes, st, P);
  return du_check_launch();
}

template <typename TC>
int launch_p4(const du_gemm_args& a, hipStream_t st, int tail_rows = 0) {
  GemmParams P = make_params(a, DU_PLAIN_ROW, DU_PLAIN_ROW, PBM, NBN, 32);
  P.tiles_m = (a.M + PBM - 1) / PBM;
  P.group_m = g_p8_group;
  P.dbg = g_p8_debug;
  dim3 grid(P.tiles_m * P.tiles_n, a.batch < 1 ? 1 : a.batch);
  if (tail_rows > 0) {
    P.tail_rows = tail_rows; P.main_wgs = P.tiles_m * P.tiles_n;
    grid.x += (a.N + SK_BN - 1) / SK_BN;
  }
  void (*kfn)(GemmParams) = gemm_nt_p4_kernel<TC>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), P4_LDS, st, P);
  return du_check_launch();
}

// weight-gradient form: C[m][n] += alpha * sum_k A[k][m] * B[k][n] (both operands contraction-major, fp32 atomics into a zeroed C)
int g_p8_ks = 0;         // du_set_option key 16: 1 = fp32-result products with <= 128 tiles of 256 x 256 run as K-split pairs (gemm_nt_p8ks_kernel)
inline long ks_ws_need(long tiles) { return KS_STATE_BYTES + tiles * 2 * PBM * PBN * 4; }

int launch_p8ks(const du_gemm_args& a, hipStream_t st, int tail_rows) {
  GemmParams P = make_params(a, DU_PLAIN_ROW, DU_PLAIN_ROW, PBM, PBN, PBK);
  P.tiles_m = (a.M + PBM - 1) / PBM;
  P.group_m = g_p8_group;
  P.dbg = g_p8_debug;
  P.ks_ws = a.ks_ws;
  P.main_wgs = 2 * P.tiles_m * P.tiles_n;
  dim3 grid(P.main_wgs, 1);
  if (tail_rows > 0) { P.tail_rows = tail_rows; grid.x += (a.N + SK_BN - 1) / SK_BN; }
  void (*kfn)(GemmParams) = g_p8_sched ? gemm_nt_p8ks_kernel<1> : gemm_nt_p8ks_kernel<0>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[g_p8_sched ? 1 : 0]) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set[g_p8_sched ? 1 : 0] = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(512), P8_LDS, st, P);
  return du_check_launch();
}

int g_p8_tn = 1;         // du_set_option key 5: 0 = keep these products on the 128 x 128 register-staged kernel, 1 = where it pays, 2 = wherever legal

template <int SCHED, bool GA>
int launch_p8_tn(const du_gemm_args& a, int splits, hipStream_t st) {
  GemmParams P = make_params(a, DU_PLAIN_COL, GA ? DU_IM2COL_COL : DU_PLAIN_COL, PBM, PBN, PBK);
  P.tiles_m = (a.M + PBM - 1) / PBM;
  P.split_k = splits;
  P.k_per_split = 0;       // unused: the kernel deals out K-tile pairs itself
  P.dbg = g_p8_debug;
  dim3 grid(P.tiles_m * P.tiles_n * splits, 1);
  void (*kfn)(GemmParams) = GA ? gemm_nt_p8_kernel<float, SCHED, true, true> : gemm_nt_p8_kernel<float, SCHED, true, false>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(512), P8_LDS, st, P);
  return du_check_launch();
}

}  // namespace

// debugging / A-B knobs (within-process variant switching for tools/gemm_p8_bench.py); not part of the hot-path contract
extern int g_attn_w;      // attention.hip
extern int g_attn_impl, g_attn_thresh_log2, g_attn_var;
extern int g_rk_mode;     // gemm_rk.hip
extern int g_wgrad_rows;  // conv_halo.hip
extern int g_ln_rows2;    // norm.hip

extern "C" int du_set_option(int key, int value) {
  switch (key) {
    case 4: g_attn_w = value; return DU_OK;
    case 6: g_attn_impl = value; return DU_OK;
    case 7: g_attn_thresh_log2 = value; return DU_OK;
    case 8: g_attn_var = value; return DU_OK;
    case 0: g_p8_mode = value; return DU_OK;
    case 1: g_p8_sched = value; return DU_OK;
    case 2: g_p8_group = value; return DU_OK;
    case 3: g_p8_debug = value; return DU_OK;
    case 5: g_p8_tn = value; return DU_OK;
    case 9: g_p8_corun = value < 1 ? 1 : (value > 8 ? 8 : value); return DU_OK;
    case 10: g_p8_persist = value; return DU_OK;
    case 11: g_p8_pp_full = value; return DU_OK;
    case 12: g_rk_mode = value; return DU_OK;
    case 13: g_wgrad_rows = value; return DU_OK;
    case 14: g_p8_res = value; return DU_OK;
    case 15: g_p8_tail_inline = value; return DU_OK;
    case 16: g_p8_ks = value; return DU_OK;
    case 17: g_p8_tail_slices = value; return DU_OK;
    case 18: g_ln_rows2 = value; return DU_OK;
    default: return DU_ERR_BAD_ARG;
  }
}

// the im2col operand of a ConvTranspose2d k2 s2 backward product (a 2 x 2 stride-2 patch gather over dy) in the form the gather
// kernels address in place: one source tensor, power-of-two channel count (a K- or N-tile lies inside one tap), < 2 GB
static bool convt_gather_geom(const du_gemm_args& a, const void* src, long ld, long pixels_in) {
  const du_conv_geom& g = a.geom;
  if (g.KH != 2 || g.KW != 2 || g.stride != 2 || g.pad != 0 || g.transposed || g.p2) return false;
  if (g.Hi != 2 * g.Ho || g.Wi != 2 * g.Wo || g.Ho <= 0 || g.Wo <= 0) return false;
  if (g.C <= 0 || (g.C & (g.C - 1)) || g.C % 64 || ld % 8 || (((uintptr_t)src) & 15)) return false;
  if (pixels_in % ((long)g.Ho * g.Wo)) return false;
  return 4 * pixels_in * ld * 2 <= 0x7fffffffL;
}
// data gradient of ConvTranspose2d k2 s2 as an NT product with the A rows gathered from dy
static bool p8_gather_legal(const du_gemm_args& a) {
  if (a.a_mode != DU_IM2COL_ROW || a.b_mode != DU_PLAIN_ROW || a.dtype != DU_BF16 || a.out_dtype != DU_BF16) return false;
  if (a.split_k > 1 || a.batch > 1 || a.store_mode != DU_STORE_PLAIN || a.act == DU_ACT_SWIGLU) return false;
  if (!convt_gather_geom(a, a.A, a.lda, a.M) || a.K != 4 * a.geom.C) return false;
  if (a.K % 128 || a.M < 256 || a.N < 128 || a.N % 4 || a.ldb % 8 || (((uintptr_t)a.B) & 15)) return false;
  return (long)a.ldb * 2 * 256 <= 0x7fffffffL;
}

// true when the multi-phase kernels can run this product at all
static bool p8_legal(const du_gemm_args& a) {
  if (a.a_mode != DU_PLAIN_ROW || a.b_mode != DU_PLAIN_ROW || a.dtype != DU_BF16) return false;
  if (a.split_k > 1) return false;
  if (a.store_mode == DU_STORE_MSDA_PREP) {       // offsets | weights product with the sampling-location / softmax epilogue: one tile column
    if (a.out_dtype != DU_F32 || a.residual || a.gamma || a.row_scale || a.alpha != 1.0f || a.act != DU_ACT_NONE || a.batch > 1) return false;
    if (a.N % 12 || a.N > 256 || a.ps_H <= 0 || a.ps_W <= 0 || a.ps_C <= 0 || !a.rope_sin || !a.C2) return false;
    if ((((uintptr_t)a.C) | ((uintptr_t)a.C2)) & 15) return false;
  } else
  if (a.store_mode == DU_STORE_QKV_ROPE || a.store_mode == DU_STORE_QKV_HEADS) {       // bias-only bf16 epilogue, d_head 64
    if (a.out_dtype != DU_BF16 || a.residual || a.gamma || a.row_scale || a.alpha != 1.0f || a.act != DU_ACT_NONE || a.batch > 1) return false;
    if (a.ps_H <= 0 || a.ps_W < a.ps_H || a.ps_C <= 0 || a.N != 3 * a.ps_C * 64 || a.ldc % 8 || (((uintptr_t)a.C) & 15)) return false;
    if (a.store_mode == DU_STORE_QKV_ROPE && (!a.rope_sin || !a.rope_cos || ((((uintptr_t)a.rope_sin) | ((uintptr_t)a.rope_cos)) & 15))) return false;
  } else
  if (a.store_mode != DU_STORE_PLAIN && (a.store_mode != DU_STORE_PIXEL_SHUFFLE2 || a.ps_C % 4 || a.act == DU_ACT_SWIGLU)) return false;
  if (a.K % 128 || a.K < 256 || a.M < 256 || a.N < 128 || a.N % 4) return false;
  if (a.lda % 8 || a.ldb % 8 || (((uintptr_t)a.A) & 15) || (((uintptr_t)a.B) & 15)) return false;
  if (a.a_batch_stride % 8 || a.b_batch_stride % 8) return false;
  if ((long)a.lda * 2 * 256 > 0x7fffffffL || (long)a.ldb * 2 * 256 > 0x7fffffffL) return false;
  if (a.act == DU_ACT_SWIGLU &&
      (a.out_dtype != DU_BF16 || a.residual || a.gamma || a.row_scale || a.alpha != 1.0f || a.N % 16 || a.ldc % 4)) return false;
  return true;
}

// the persistent 256 x 128 kernel: bf16 result with a bias (+ GELU) epilogue, plain store, K >= 512, every extent below 2^31 bytes
static bool pp_legal(const du_gemm_args& a) {
  if (!p8_legal(a) || a.out_dtype != DU_BF16 || a.batch > 1) return false;
  if (a.store_mode == DU_STORE_QKV_HEADS) {     // the head-major store alone (p8_legal: bias-only bf16 epilogue, the planes)
    if (a.K < 384 || a.N % 128 || a.M % 256 || 3L * a.ldc * 2 >= 0x7fffffffL) return false;
    return ((long)a.M * a.lda + a.K) * 2 < 0x7fffffffL && ((long)a.N * a.ldb + a.K) * 2 < 0x7fffffffL;
  }
  if (a.store_mode == DU_STORE_QKV_ROPE) {      // (p8_legal checked the bias-only bf16 epilogue, the planes and the tables)
    // RoPE in the drain: the factorised table holds a token grid of <= 32 x 32 (du_gemm_args.geom.Hi x Wi, Hi * Wi = tokens behind the prefix)
    const du_conv_geom& g = a.geom;
    if (g.Hi <= 0 || g.Wi <= 0 || g.Hi > 32 || g.Wi > 32 || g.Hi * g.Wi != a.ps_H - a.rope_prefix || a.rope_prefix < 0) return false;
    if (a.K < 384 || a.N % 128 || a.M % 256 || 3L * a.ldc * 2 >= 0x7fffffffL) return false;
    return ((long)a.M * a.lda + a.K) * 2 < 0x7fffffffL && ((long)a.N * a.ldb + a.K) * 2 < 0x7fffffffL;
  }
  const bool ps = a.store_mode == DU_STORE_PIXEL_SHUFFLE2;
  if (ps) {     // ConvTranspose2d k2 s2 forward + residual: a 128-column tile inside one tap, power-of-two pixel grid
    if (!a.residual || a.row_scale || a.K < 384 || a.ps_C % 128 || a.N != 4 * a.ps_C || a.ps_H <= 0 || a.ps_W <= 0) return false;
    if ((a.ps_H & (a.ps_H - 1)) || (a.ps_W & (a.ps_W - 1)) || a.M % (a.ps_H * a.ps_W)) return false;
    if ((4L * a.M * a.ldc + a.N) * 2 >= 0x7fffffffL || (4L * a.M * a.ldr + a.N) * 2 >= 0x7fffffffL) return false;
  } else if (a.store_mode != DU_STORE_PLAIN) return false;
  if (a.gamma || a.alpha != 1.0f || (a.act != DU_ACT_NONE && a.act != DU_ACT_GELU)) return false;
  if ((a.K != 256 && a.K < 384) || a.N % 8 || a.ldc % 8 || (((uintptr_t)a.C) & 15)) return false;      // (K = 256: the four-K-step form)
  const long lim = 0x7fffffffL;
  if (a.residual) {
    // the residual as two more K-steps of the tile (bf16, the result's dtype): whole 128-column tiles, 16-byte rows; DropPath's scale only
    // where a 256-row tile lies inside one sample
    if (a.act != DU_ACT_NONE || a.N % 128 || a.ldr % 8 || (((uintptr_t)a.residual) & 15)) return false;
    if (a.row_scale && (a.rs_rows < 256 || a.rs_rows % 256)) return false;
    if (((long)a.M * a.ldr + a.N) * 2 >= lim) return false;
  } else if (a.row_scale) return false;
  return ((long)a.M * a.lda + a.K) * 2 < lim && ((long)a.N * a.ldb + a.K) * 2 < lim && ((long)a.M * a.ldc + a.N) * 2 < lim;
}

// the K-split pair form of the 256 x 256 kernel: fp32 result, plain store, whole pairs of K-tile pairs per half, tiles a multiple of 8 (a pair =
// workgroups 16 g + x and 16 g + 8 + x), both halves of every tile resident at once on an otherwise idle chip (<= 128 tiles)
static bool ks_shape_ok(const du_gemm_args& a) {
  if (!p8_legal(a) || a.out_dtype != DU_F32 || a.store_mode != DU_STORE_PLAIN || a.batch > 1 || a.act == DU_ACT_SWIGLU) return false;
  if (a.K % 256 || a.K < 1024 || g_p8_corun != 1) return false;
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  return tiles % 8 == 0 && tiles >= 64 && tiles <= 128;
}
static bool ks_legal(const du_gemm_args& a) {
  if (!ks_shape_ok(a) || !a.ks_ws || (((uintptr_t)a.ks_ws) & 15)) return false;
  return a.ks_ws_bytes >= ks_ws_need((long)((a.M + 255) / 256) * ((a.N + 255) / 256));
}

// 0: not served by gemm_p8.hip, 1: 256 x 256 tiles, 2: 256 x 128 tiles (8 waves, one workgroup per CU), 3: 256 x 128 tiles on 4 waves, two workgroups per CU,
// 4: 256 x 128 tiles, persistent workgroups, 5: 256 x 256 tiles as K-split pairs
int du_gemm_p8_choice(const du_gemm_args& a) {
  if (g_p8_mode != 0 && g_p8_tn && p8_gather_legal(a))      // ConvT data gradient: 256 x 256 tiles once they fill most of the CUs
    return (g_p8_mode > 0 || (long)((a.M + 255) / 256) * ((a.N + 255) / 256) >= 192) ? 1 : 0;
  if (g_p8_mode == 0 || !p8_legal(a)) return 0;
  if (a.store_mode == DU_STORE_MSDA_PREP) return 1;                                            // the 256 x 256 kernel's fp32 staging epilogue or nothing
  if (a.store_mode == DU_STORE_QKV_HEADS) return (g_p8_mode != 0 && pp_legal(a)) ? 4 : 0;      // the persistent kernel or nothing
  if (a.store_mode == DU_STORE_QKV_ROPE) {     // round 6: on the persistent kernel (RoPE in the drain) where a CU gets >= 2 tiles, else 256 x 128
    const long t = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    return (g_p8_persist && g_p8_mode != 2 && pp_legal(a) && (g_p8_persist > 1 || g_p8_mode == 4 || t >= 2 * (256 / g_p8_corun))) ? 4 : 2;
  }
  if (g_p8_ks && g_p8_mode < 0 && ks_legal(a)) return 5;
  if (g_p8_mode == 5) return ks_legal(a) ? 5 : 2;
  if (g_p8_mode == 3) return a.act == DU_ACT_SWIGLU ? 2 : 3;
  if (g_p8_mode == 4) return pp_legal(a) ? 4 : 2;
  if (g_p8_mode > 0) return g_p8_mode == 2 ? 2 : 1;
  // rounds of workgroups on the 256 CUs (one 8-wave workgroup per CU) x cost per workgroup (a 256 x 128 tile costs ~0.56 of a
  // 256 x 256 one: half the MFMAs at a lower operand reuse); measured crossovers: tools/gemm_p8_bench.py
  const long batch = a.batch < 1 ? 1 : a.batch;
  const long tm = (a.M + 255) / 256;
  const long t256 = tm * ((a.N + 255) / 256) * batch, t128 = tm * ((a.N + 127) / 128) * batch;
  const long cus = 256 / g_p8_corun;           // CUs this product can count on (co-running products share the chip)
  if (t128 * g_p8_corun < 192) return 0;
  // (round 4: at K = 1024 the narrow tile's fixed costs -- prologue, exposed epilogue -- weigh more: 3 rounds of 256 x 128 tiles cost 62.8-67.6 us
  //  against 62.8-63.5 us for 2 rounds of 256 x 256 on the 8192 / 8232 x 3072 qkv product, ratio 0.66-0.72 per round, not 0.56)
  const double c256 = (double)((t256 + cus - 1) / cus) * 1.0, c128 = (double)((t128 + cus - 1) / cus) * (a.K >= 1024 ? 0.68 : 0.56);
  // (>= 128 tiles since round 3: the tall products of the adapter with ONE 256-wide tile column -- 43008 x {192, 256} x 1024 -- read their A
  //  operand once with the 256 x 256 tile and twice with two half-empty 128-wide columns: 31.8 / 32.6 us against 34.7 / 37.3 us)
  // (round 5) the persistent kernel walks ceil(t128 / CUs) narrow tiles per workgroup and pays prologue / epilogue once: a tile costs it
  // 0.53 of what a ROUND of 256 x 256 tiles costs the kernels above (0.49 against their exposed GELU epilogue, 0.45 at K <= 512 where the
  // fixed costs weigh more) -- qkv 3 tiles against 2 rounds (56.9 vs 62.7 us), fc1 4 against 2 (77.8 vs 81.3), the adapter's 43008 x 1024
  // x 512 six against three (56.3 vs 66.8); FAPM's 131072 x 512 x 1024 (8 tiles against 4 full rounds: 161 vs 141 us) and the big
  // square / 7B products stay on the wide tile (profiles/r05_gemm_p8_table_v2.txt)
  if (g_p8_persist && pp_legal(a)) {
    // (round 6) a bf16 residual: the one-shot kernels re-read it in an exposed epilogue (95.6 / 111.8 us at 43008 x 1024 x {256, 512} against
    // 36.5 / 56.3 without one); the persistent kernel takes it in as two more K-steps
    if (a.residual) {
      if (g_p8_res && (g_p8_persist > 1 || t128 >= 2 * cus)) return 4;
    } else {
    const double r = a.K <= 256 ? 0.40 : (a.K <= 512 ? 0.45 : (a.act == DU_ACT_GELU ? 0.49 : 0.53));
    const double cpp = (double)((t128 + cus - 1) / cus) * r;
    const double best = (t256 * g_p8_corun >= 128 && c256 <= c128) ? c256 : c128;
    if (g_p8_persist > 1 || (t128 >= 2 * cus && cpp < best)) return 4;
    }
  }
  if (t256 * g_p8_corun >= 128 && c256 <= c128) return 1;
  return 2;
}
bool du_gemm_p8_wants(const du_gemm_args& a) { return du_gemm_p8_choice(a) != 0; }
// bytes of du_gemm_args.ks_ws the K-sliced ragged-row units want for `whole` (0: one unit per column block)
long du_gemm_p8_tail_bytes(const du_gemm_args& whole) {
  const int s = tail_slices_for(whole, false);
  return s > 1 ? KS_STATE_BYTES + (long)((whole.N + SK_BN - 1) / SK_BN) * s * 64 * SK_BN * 4 : 0;
}
// bytes of du_gemm_args.ks_ws the pair kernel wants for this product (its full tile rows): 0 = it would not run there
long du_gemm_p8_ks_bytes(const du_gemm_args& a) {
  if (!(g_p8_ks && g_p8_mode < 0) && g_p8_mode != 5) return 0;
  return ks_shape_ok(a) ? ks_ws_need((long)((a.M + 255) / 256) * ((a.N + 255) / 256)) : 0;
}

// returns DU_ERR_UNSUPPORTED when these kernels cannot serve the product; the caller then uses gemm_glds.hip
// tail_rows > 0: rows [a.M, a.M + tail_rows) of the same operands / result are computed in the same launch (p8_tail); the caller has checked
// du_gemm_p8_tail_ok
int du_gemm_nt_p8(const du_gemm_args& a, hipStream_t st, int tail_rows) {
  int c = du_gemm_p8_choice(a);
  if (c == 0 && a.act == DU_ACT_SWIGLU && p8_legal(a)) c = 2;     // the gate epilogue exists only here: take the narrow tile when the
                                                                  // heuristic would have preferred another kernel family
  if (c == 0) return DU_ERR_UNSUPPORTED;
  if (a.a_mode == DU_IM2COL_ROW) return g_p8_sched ? launch_p8<bf16_t, 1, false, true>(a, st) : launch_p8<bf16_t, 0, false, true>(a, st);
  const bool bf = a.out_dtype == DU_BF16;
  const int tr = tail_rows;
  if (c == 5) return launch_p8ks(a, st, tr);
  if (c == 4) return launch_pp(a, st, tr);
  if (c == 3) return bf ? launch_p4<bf16_t>(a, st, tr) : launch_p4<float>(a, st, tr);
  if (c == 1) {
    if (bf) return g_p8_sched ? launch_p8<bf16_t, 1, false>(a, st, tr) : launch_p8<bf16_t, 0, false>(a, st, tr);
    return g_p8_sched ? launch_p8<float, 1, false>(a, st, tr) : launch_p8<float, 0, false>(a, st, tr);
  }
  if (bf) return g_p8_sched ? launch_p8<bf16_t, 1, true>(a, st, tr) : launch_p8<bf16_t, 0, true>(a, st, tr);
  return g_p8_sched ? launch_p8<float, 1, true>(a, st, tr) : launch_p8<float, 0, true>(a, st, tr);
}

// the ragged tail of `whole` (rows M - r .. M) may ride in the head's launch: the single-launch skinny form serves it (K <= 2048: longer
// rows put every load of the tail on the same few memory channels, gemm_skinny.hip), plain store, no gate epilogue
bool du_gemm_p8_tail_ok(const du_gemm_args& whole, int r) {
  static const bool off = DU_GETENV("DU_P8_NO_TAIL") != nullptr;      // debugging / A-B aid
  // (round 5: 4096 -- fc2's 40 tail rows inside its launch instead of the partial + finish kernel pair: +0.6 % step rate on one box, three
  //  interleaved rounds, profiles/r05_ab_tail_fuse_v1.txt; alone the fused K = 4096 tail is slower than the pair, 21 vs 14 us, but here it
  //  runs beside the stragglers of the tile grid and two ~5 us launch slots per block go away)
  static const int kmax = DU_GETENV("DU_SKINNY_FUSE_KMAX") ? atoi(DU_GETENV("DU_SKINNY_FUSE_KMAX")) : 4096;
  if (off || r < 1 || r > 64 || whole.K > kmax || whole.K % SK_CHUNK || whole.N % 4 || whole.batch > 1) return false;
  if ((whole.store_mode != DU_STORE_PLAIN && whole.store_mode != DU_STORE_QKV_HEADS) || whole.act == DU_ACT_SWIGLU ||
      whole.a_mode != DU_PLAIN_ROW || whole.b_mode != DU_PLAIN_ROW) return false;
  return 8 * 64 * (SK_BN + 1) * 4 <= P8_LDS;
}

// Weight gradients on the multi-phase kernel: 0 = not served, else the number of K splits it would run with.  Only products the caller
// already runs split-K (fp32 result zeroed beforehand, no epilogue terms) are taken; the split count is re-chosen for ONE 8-wave
// workgroup per CU: tiles x splits <= 256 workgroups, at least two K-tile pairs (256 contraction rows) per split.
int du_gemm_tn_p8_splits(const du_gemm_args& a) {
  if (!g_p8_tn || g_p8_mode == 0) return 0;
  if (a.dtype != DU_BF16 || a.out_dtype != DU_F32 || a.a_mode != DU_PLAIN_COL) return 0;
  if (a.b_mode != DU_PLAIN_COL && a.b_mode != DU_IM2COL_COL) return 0;
  if (a.split_k <= 1 || a.batch > 1 || a.store_mode != DU_STORE_PLAIN) return 0;
  if (a.bias || a.act || a.gamma || a.row_scale || a.residual) return 0;
  if (a.K % 128 || a.K < 512 || a.M <= 128 || a.N <= 128) return 0;
  if (a.lda % 8 || a.ldb % 8 || (((uintptr_t)a.A) & 15) || (((uintptr_t)a.B) & 15)) return 0;
  if (a.b_mode == DU_IM2COL_COL) {     // ConvT weight gradient: B gathered from dy; an N-tile must lie inside one tap
    if (!convt_gather_geom(a, a.B, a.ldb, a.K) || a.N != 4 * a.geom.C || a.geom.C % 256) return 0;
    if (a.geom.Wo & (a.geom.Wo - 1)) return 0;
  }
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const int npairs = a.K / 128;
  long s = tiles >= 256 ? 1 : 256 / tiles;
  if (s > npairs / 2) s = npairs / 2;
  if (s < 1) s = 1;
  // every workgroup ends with 256 x 256 fp32 atomics whatever its K range (measured ~35-60 us for a 256-workgroup launch, more the
  // more splits share a tile): below ~16 K-tile pairs per split the 128 x 128 kernel's smaller partial tiles win
  // (tools/gemm_tn_bench.py: 43008-row adapter linears 88 vs 94 us; 131072 x 512 x 1024 237 -> 162 us, ConvT 1024 -> 1024 660 -> 249 us)
  if (g_p8_tn < 2 && npairs / s < 16) return 0;
  // a split may not run more than 2^31 bytes past its base (buffer offsets are 32-bit): (npairs / s + 1) pairs of 128 rows
  const long ldmax = a.b_mode == DU_PLAIN_COL && a.ldb > a.lda ? a.ldb : a.lda;
  if ((long)(npairs / s + 1) * 128 * ldmax * 2 > 0x7fffffffL) return 0;
  return (int)s;
}

// ---- grouped weight gradients (see gemm_tn_group_kernel) ----
static bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
static bool tn_group_legal(const du_tn_job& j) {
  if (!j.A || !j.B || !j.C || j.M <= 0 || j.N <= 0) return false;
  if (j.K % 128 || j.K < 512) return false;                       // whole K-tile pairs, at least two pairs per split
  if (j.lda % 8 || j.ldb % 8 || j.lda < j.M) return false;
  if ((((uintptr_t)j.A) | ((uintptr_t)j.B)) & 15) return false;
  if (j.lda > 0x7fffffffL / 4 || j.ldb > 0x7fffffffL / 4 || j.ldc > 0x7fffffffL) return false;
  if (j.taps > 1 && (j.inner <= 0 || j.N != j.taps * j.inner)) return false;
  if (j.taps <= 1 && j.ldc < j.N) return false;
  if (((long)j.K + 128) * j.lda * 2 > 0x7fffffffL) return false;   // an unsplit product addresses its operand through one 32-bit offset
  if (j.taps > 1 && (j.c_off < 0 || j.inner_total < j.c_off + j.inner)) return false;
  if (j.gather == 0) {
    if (j.ldb < j.N || j.b_colsum) return false;
    return ((long)j.K + 128) * j.ldb * 2 <= 0x7fffffffL;
  }
  if (j.gather != 2 && j.gather != 3) return false;
  if (j.Cb <= 0 || j.Cb % 8 || j.ldb < j.Cb || !pow2(j.Ws) || j.Hs <= 0 || j.K % (j.Hs * j.Ws)) return false;
  if (j.gather == 2) {
    if (j.N != 4 * j.Cb) return false;
    return (long)j.K * 4 * j.ldb * 2 <= 0x7fffff00L;
  }
  if (j.N != 9 * j.Cb || !pow2(j.Hs) || j.b_colsum) return false;
  if (j.Ws % 64) return false;          // the 3 x 3 gather assumes a K-tile (64 consecutive pixels) inside ONE image row
  return (long)j.K * j.ldb * 2 <= 0x7fffff00L;
}
extern "C" int du_gemm_tn_group_legal(const du_tn_job* job) { return (job && g_p8_mode != 0 && tn_group_legal(*job)) ? 1 : 0; }

template <int GG>
static int tn_group_launch(const du_tn_job* const* jobs, int njobs, hipStream_t st) {
  static const int target_units = DU_GETENV("DU_TN_GROUP_UNITS") ? atoi(DU_GETENV("DU_TN_GROUP_UNITS")) : 256;   // one 8-wave workgroup per CU
  static const int min_pairs = DU_GETENV("DU_TN_GROUP_MINPAIRS") ? atoi(DU_GETENV("DU_TN_GROUP_MINPAIRS")) : 8;  // >= 1024 contraction rows per split
  void (*kfn)(TnGroupArgs) = g_p8_sched ? gemm_tn_group_kernel<1, GG> : gemm_tn_group_kernel<0, GG>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[g_p8_sched ? 1 : 0]) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set[g_p8_sched ? 1 : 0] = true;
  }
  static const int max_jobs = DU_GETENV("DU_TN_GROUP_MAXJOBS") ? std::max(1, std::min(TN_GROUP_MAX, atoi(DU_GETENV("DU_TN_GROUP_MAXJOBS")))) : TN_GROUP_MAX;
  int i0 = 0;
  while (i0 < njobs) {
    // one launch: consecutive jobs while their tiles fit one round of workgroups
    int tiles[TN_GROUP_MAX], pairs[TN_GROUP_MAX], splits[TN_GROUP_MAX];
    int n = 0, units = 0;
    while (i0 + n < njobs && n < max_jobs) {
      const du_tn_job& j = *jobs[i0 + n];
      const int t = ((j.M + PBM - 1) / PBM) * ((j.N + PBN - 1) / PBN);
      if (n > 0 && units + t > target_units) break;
      tiles[n] = t; pairs[n] = j.K / 128; splits[n] = 1;
      units += t; n++;
    }
    // K splits in proportion to the contraction length: repeatedly split the job with the longest units while the round has room
    for (;;) {
      int best = -1; double len = 0.0;
      for (int k = 0; k < n; k++) {
        if (units + tiles[k] > target_units || pairs[k] / (splits[k] + 1) < min_pairs) continue;
        const double l = (double)pairs[k] / splits[k];
        if (l > len) { len = l; best = k; }
      }
      if (best < 0) break;
      splits[best]++; units += tiles[best];
    }
    TnGroupArgs G{};
    G.njobs = n; G.dbg = g_p8_debug;
    int u = 0;
    for (int k = 0; k < n; k++) {
      const du_tn_job& j = *jobs[i0 + k];
      TnJob& d = G.jobs[k];
      d.A = j.A; d.B = j.B; d.C = j.C; d.a_colsum = j.a_colsum; d.alpha = j.alpha; d.accumulate = j.accumulate;
      d.b_colsum = j.b_colsum;
      d.lda = (int)j.lda; d.ldb = (int)j.ldb; d.ldc = (int)j.ldc; d.M = j.M; d.N = j.N; d.K = j.K;
      d.gather = j.gather; d.Hs = j.Hs; d.Ws = j.Ws; d.Cb = j.Cb;
      d.taps = j.taps; d.inner = j.inner; d.inner_total = j.taps > 1 ? j.inner_total : 0; d.c_off = j.taps > 1 ? j.c_off : 0;
      d.splits = splits[k]; d.unit0 = u;
      u += tiles[k] * splits[k];
    }
    G.nunits = u;
    hipLaunchKernelGGL(kfn, dim3(u), dim3(512), P8_LDS, st, G);
    const int rc = du_check_launch();
    if (rc != DU_OK) return rc;
    i0 += n;
  }
  return DU_OK;
}

extern "C" int du_gemm_tn_group(const du_tn_job* jobs, int njobs, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (njobs < 0 || (njobs > 0 && !jobs)) return DU_ERR_BAD_ARG;
  for (int i = 0; i < njobs; i++)
    if (!tn_group_legal(jobs[i])) return DU_ERR_UNSUPPORTED;
  // plain products and in-place gathers run different instantiations of the tile program: one launch sequence each, queue order kept
  std::vector<const du_tn_job*> plain, convt, conv3;
  for (int i = 0; i < njobs; i++) (jobs[i].gather == 0 ? plain : jobs[i].gather == 2 ? convt : conv3).push_back(&jobs[i]);
  int rc = DU_OK;
  if (!plain.empty()) rc = tn_group_launch<0>(plain.data(), (int)plain.size(), st);
  if (rc == DU_OK && !convt.empty()) rc = tn_group_launch<2>(convt.data(), (int)convt.size(), st);
  if (rc == DU_OK && !conv3.empty()) rc = tn_group_launch<3>(conv3.data(), (int)conv3.size(), st);
  return rc;
}

int du_gemm_tn_p8(const du_gemm_args& a, hipStream_t st) {
  const int s = du_gemm_tn_p8_splits(a);
  if (s == 0) return DU_ERR_UNSUPPORTED;
  if (a.b_mode == DU_IM2COL_COL) return g_p8_sched ? launch_p8_tn<1, true>(a, s, st) : launch_p8_tn<0, true>(a, s, st);
  return g_p8_sched ? launch_p8_tn<1, false>(a, s, st) : launch_p8_tn<0, false>(a, s, st);
}
