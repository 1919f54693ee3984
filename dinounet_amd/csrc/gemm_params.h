// Shared by gemm.hip (generic fp32/bf16 kernel) and gemm_bf16.hip (bf16 throughput kernel): kernel parameter block,
// im2col addressing and the host-side packing of du_gemm_args.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace {

struct Operand {
  const void* p; long ld; long bstride;
  const void* p2; long ld2; int C1;
  int Hi, Wi, C, KH, KW, stride, pad, Ho, Wo, transposed;
  int logC;  // log2(C) if C is a power of two else -1
};

struct GemmParams {
  Operand a, b;
  void* C; long ldc; long cbs;
  int M, N, K;
  int split_k, k_per_split;
  float alpha; const float* bias; int act; const float* gamma; const float* row_scale; int rs_rows;
  const void* residual; long ldr;
  int store_mode, ps_H, ps_W, ps_C;
  int tiles_n;
  int tiles_m, group_m;   // grouped tile order (gemm_glds.hip): bands of group_m tile rows are walked column by column; 0 = row-major
  int dbg;                // measurement aids of gemm_p8.hip (du_set_option key 3); 0 in production
  float* a_colsum;        // weight-gradient kernels: a_colsum[m] += sum_k A(m, k) (bias gradient), fp32 atomics; NULL = off
  float* b_colsum;        // ConvT weight gradient: b_colsum[n % b.C] += sum_k B(n, k) (bias gradient over the four taps); NULL = off
  const float* rope_sin; const float* rope_cos; int rope_prefix; float rope_qscale;     // DU_STORE_QKV_ROPE
  int k_scale;            // weight-gradient form (A contraction-major): row_scale[k / rs_rows] scales the CONTRACTION rows of A, not output rows
  int tail_rows;          // gemm_p8.hip NT kernels: rows [M, M + tail_rows) (<= 64) are computed by the workgroups behind the first main_wgs
  int main_wgs;           //   ones (skinny_fused_body, gemm_skinny_body.h); 0 = no tail in this launch
  float* C2;              // DU_STORE_MSDA_PREP: the attention weights (C = the sampling locations)
  void* ks_ws;            // du_gemm_args.ks_ws: [0, 64 KB) pair state of gemm_nt_p8ks_kernel, [64 KB, 128 KB) tail-unit tickets, then slabs
  int tail_slices;        // > 1: the tail units are (32 columns, K slice) pairs that meet through ks_ws (gemm_skinny_body.h)
};

// element offset of C / residual element (m, n) for row stride ld: plain rows, or the pixel-shuffle store of ConvTranspose2d k2 s2
// (column n = (dy*2+dx)*Cout + co of input pixel m = (b, y, x) -> output pixel (b, 2y+dy, 2x+dx), channel co)
__device__ __forceinline__ long out_offset(const GemmParams& P, int m, int n, long ld) {
  if (P.store_mode != DU_STORE_PIXEL_SHUFFLE2) return (long)m * ld + n;
  const int q = n / P.ps_C, co = n - q * P.ps_C;
  const int x = m % P.ps_W, t2 = m / P.ps_W, y = t2 % P.ps_H, b = t2 / P.ps_H;
  const long opix = ((long)b * 2 * P.ps_H + 2 * y + (q >> 1)) * (2 * P.ps_W) + 2 * x + (q & 1);
  return opix * ld + co;
}

// DU_STORE_QKV_ROPE: (b, t) x (which, h, d) -> [which][b][h][t][d]; ld = elements between the q / k / v planes.  Kept out of out_offset():
// inlined into every epilogue it pushed the 256 x 256 kernel's fp32-staged readout into scratch (+7 ms per step).
__device__ __forceinline__ long qkv_heads_offset(const GemmParams& P, int m, int n, long ld) {
  const int b = m / P.ps_H, t = m - b * P.ps_H;
  const int hd = P.ps_C * 64;
  const int which = n / hd, rem = n - which * hd;
  return (long)which * ld + (((long)b * P.ps_C + (rem >> 6)) * P.ps_W + t) * 64 + (rem & 63);
}

__device__ __forceinline__ int div_small(int x, int d) {
  if (d == 3) return (x * 11) >> 5;  // exact for x < 32
  if (d == 2) return x >> 1;
  if (d == 1) return x;
  return x / d;
}

// address of element (pixel (b,yo,xo), column c=(tap,ci)) of the im2col matrix; returns nullptr if padding
template <typename T>
__device__ __forceinline__ const T* im2col_ptr(const Operand& op, int b, int yo, int xo, int c) {
  int tap = op.logC >= 0 ? (c >> op.logC) : (c / op.C);
  int ci = c - tap * op.C;
  int dy = div_small(tap, op.KW);
  int dx = tap - dy * op.KW;
  int yi, xi;
  bool ok;
  if (!op.transposed) {
    yi = yo * op.stride - op.pad + dy;
    xi = xo * op.stride - op.pad + dx;
    ok = (yi >= 0) & (yi < op.Hi) & (xi >= 0) & (xi < op.Wi);
  } else {
    int ty = yo + op.pad - dy, tx = xo + op.pad - dx;
    if (op.stride == 1) { yi = ty; xi = tx; ok = true; }
    else if (op.stride == 2) { yi = ty >> 1; xi = tx >> 1; ok = ((ty & 1) == 0) & ((tx & 1) == 0); }
    else { yi = ty / op.stride; xi = tx / op.stride; ok = (yi * op.stride == ty) & (xi * op.stride == tx); }
    ok = ok & (ty >= 0) & (tx >= 0) & (yi < op.Hi) & (xi < op.Wi);
  }
  if (!ok) return nullptr;
  long sp = ((long)b * op.Hi + yi) * op.Wi + xi;
  if (ci < op.C1) return (const T*)op.p + sp * op.ld + ci;
  return (const T*)op.p2 + sp * op.ld2 + (ci - op.C1);
}


inline int ilog2_exact(int x) {
  if (x <= 0 || (x & (x - 1))) return -1;
  int l = 0; while ((1 << l) < x) l++;
  return l;
}

inline Operand make_operand(const void* p, long ld, long bs, int mode, const du_conv_geom& g) {
  Operand o{};
  o.p = p; o.ld = ld; o.bstride = bs;
  if (mode == DU_IM2COL_ROW || mode == DU_IM2COL_COL) {
    o.p2 = g.p2; o.ld2 = g.ld2; o.C1 = g.p2 ? g.C1 : g.C;
    o.Hi = g.Hi; o.Wi = g.Wi; o.C = g.C; o.KH = g.KH; o.KW = g.KW; o.stride = g.stride; o.pad = g.pad;
    o.Ho = g.Ho; o.Wo = g.Wo; o.transposed = g.transposed; o.logC = ilog2_exact(g.C);
  }
  return o;
}


// BM/BN/BK: tile shape of the kernel about to be launched (split-K chunks are rounded to whole K tiles)
inline GemmParams make_params(const du_gemm_args& a, int amode, int bmode, int BM, int BN, int BKt) {
  GemmParams P{};
  P.a = make_operand(a.A, a.lda, a.a_batch_stride, amode, a.geom);
  P.b = make_operand(a.B, a.ldb, a.b_batch_stride, bmode, a.geom);
  P.C = a.C; P.ldc = a.ldc; P.cbs = a.c_batch_stride;
  P.M = a.M; P.N = a.N; P.K = a.K;
  P.split_k = a.split_k < 1 ? 1 : a.split_k;
  int kps = (a.K + P.split_k - 1) / P.split_k;
  kps = ((kps + BKt - 1) / BKt) * BKt;
  P.k_per_split = kps;
  P.split_k = (a.K + kps - 1) / kps;  // drop empty splits
  P.alpha = a.alpha; P.bias = a.bias; P.act = a.act; P.gamma = a.gamma; P.row_scale = a.row_scale;
  P.rs_rows = a.rs_rows > 0 ? a.rs_rows : 1; P.residual = a.residual; P.ldr = a.ldr;
  P.store_mode = a.store_mode; P.ps_H = a.ps_H; P.ps_W = a.ps_W; P.ps_C = a.ps_C;
  P.a_colsum = a.a_colsum; P.b_colsum = a.b_colsum;
  P.rope_sin = a.rope_sin; P.rope_cos = a.rope_cos; P.rope_prefix = a.rope_prefix; P.rope_qscale = a.rope_qscale;
  P.C2 = a.C2;
  P.k_scale = (amode == DU_PLAIN_COL && a.row_scale) ? 1 : 0;
  P.tiles_n = (a.N + BN - 1) / BN;
  (void)BM;
  return P;
}

}  // namespace
