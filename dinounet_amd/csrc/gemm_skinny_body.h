// The ragged-tail product of a tall NT GEMM (<= 64 rows x N x K, see gemm_skinny.hip) as a device function: one workgroup = 32 output columns
// x the WHOLE contraction, NW waves each taking every NW-th 64-wide K chunk, fragments straight from global memory, wave tiles parked in
// LDS slots [NW][64][33], then the du_gemm epilogue.  Used by gemm_skinny_fused_kernel (its own launch) and, for the ViT's 40-row tails, by
// EXTRA workgroups appended to the grid of the multi-phase tile kernels (gemm_p8.hip): the tail then costs no launch of its own.
#pragma once
#include "common.h"

namespace {

struct SkinnyEpi {
  void* C; long ldc; const void* residual; long ldr;
  const float* bias; const float* gamma; const float* row_scale;
  float alpha; int act, rs_rows, out_bf16;
  int row0;        // first output row's index for row_scale (the tail sits behind the head's rows)
  int qkv_H;       // > 0: DU_STORE_QKV_HEADS -- C is the base of the three head-major planes (ldc elements apart), row row0 + m = (b, token) of
  int qkv_N, qkv_Npad;   //   qkv_N tokens per sample, column n = (which, head, d): element [which][b][head][token][d] (bf16, no residual)
};

constexpr int SK_BN = 32;          // output columns per workgroup
constexpr int SK_CHUNK = 64;       // contraction elements per wave step (4 MFMAs)

template <int NW>
__device__ __forceinline__ void skinny_fused_body(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, int M, int N,
                                                  int K, const SkinnyEpi& P, float* sk_red, int blk) {
  constexpr int SLOT = 64 * (SK_BN + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blk * SK_BN;
  const int nrb = (M + 31) >> 5;
  const int kh = (lane >> 5) * 32;
  const int n = n0 + (lane & 31);
  const bf16_t* bp = B + (long)min(n, N - 1) * ldb + kh;
  const bf16_t* ap0 = A + (long)min(lane & 31, M - 1) * lda + kh;
  const bf16_t* ap1 = A + (long)min(32 + (lane & 31), M - 1) * lda + kh;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // workgroup j starts its walk over the K chunks at a different chunk (and wraps): with long rows (K = 4096: 8 KB pitch) every load of
  // the launch otherwise hits the same few memory channels at the same time (32 rows at ONE column offset per load, all workgroups in step)
  const int nchunk = K / SK_CHUNK;
  const int rot = (int)((blk * 5u) % (unsigned)nchunk);
#pragma unroll 4
  for (int ci = wave; ci < nchunk; ci += NW) {
    int cc = ci + rot; if (cc >= nchunk) cc -= nchunk;
    const int k = cc * SK_CHUNK;
    bf16x8 fb[4], fa0[4], fa1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) fb[j] = *(const bf16x8*)(bp + k + j * 8);
#pragma unroll
    for (int j = 0; j < 4; j++) fa0[j] = *(const bf16x8*)(ap0 + k + j * 8);
    if (nrb > 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) fa1[j] = *(const bf16x8*)(ap1 + k + j * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[j], fb[j], acc0, 0, 0, 0);
    if (nrb > 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[j], fb[j], acc1, 0, 0, 0);
    }
  }
  {   // D layout: column lane & 31, row (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* slot = sk_red + wave * SLOT;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      slot[row * (SK_BN + 1) + (lane & 31)] = acc0[r];
      if (nrb > 1) slot[(32 + row) * (SK_BN + 1) + (lane & 31)] = acc1[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < M * (SK_BN / 4); i += NW * 64) {
    const int m = i / (SK_BN / 4), c = (i % (SK_BN / 4)) * 4;
    const int nn = n0 + c;
    if (nn >= N) continue;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float* q = sk_red + w * SLOT + m * (SK_BN + 1) + c;
      o[0] += q[0]; o[1] += q[1]; o[2] += q[2]; o[3] += q[3];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] *= P.alpha;
    if (P.bias) {
      const float4 bb = *(const float4*)(P.bias + nn);
      o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
    }
    if (P.act != DU_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = apply_act(o[e], P.act);
    }
    if (P.gamma) {
      const float4 gg = *(const float4*)(P.gamma + nn);
      o[0] *= gg.x; o[1] *= gg.y; o[2] *= gg.z; o[3] *= gg.w;
    }
    if (P.row_scale) {
      const float rs = P.row_scale[(P.row0 + m) / P.rs_rows];
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] *= rs;
    }
    if (P.qkv_H > 0) {
      const int gm = P.row0 + m, b = gm / P.qkv_N, tk = gm - b * P.qkv_N;
      const int hd = P.qkv_H * 64, which = nn / hd, rem = nn - which * hd;
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (bf16_t)o[e];
      *(uint2*)((bf16_t*)P.C + (long)which * P.ldc + (((long)b * P.qkv_H + (rem >> 6)) * P.qkv_Npad + tk) * 64 + (rem & 63)) = __builtin_bit_cast(uint2, t);
    } else if (P.out_bf16) {
      if (P.residual) {
        const bf16_t* rp = (const bf16_t*)P.residual + (long)m * P.ldr + nn;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] += (float)rp[e];
      }
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (bf16_t)o[e];
      *(uint2*)((bf16_t*)P.C + (long)m * P.ldc + nn) = __builtin_bit_cast(uint2, t);
    } else {
      if (P.residual) {
        const float4 rr = *(const float4*)((const float*)P.residual + (long)m * P.ldr + nn);
        o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
      }
      *(float4*)((float*)P.C + (long)m * P.ldc + nn) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

}  // namespace
