// Multi-scale deformable attention (bilinear gather / scatter) for gfx950 -- replaces the reference's CUDA
// extension `MultiScaleDeformableAttention` (ops/src/cuda/ms_deform_im2col_cuda.cuh).
//
// Semantics restated from the reference kernels (cuh:242-304 forward, cuh:92-164 backward partials):
//   pixel coordinate = loc * size - 0.5, sample contributes iff  -1 < h < H and -1 < w < W,
//   each of the 4 corners contributes iff it lies inside the level (zero padding),
//   grad_attn = top_grad . val,  grad_loc = (W * g_w, H * g_h) * top_grad * attn,  grad_value += corner_w * top_grad * attn.
//
// Design (wave64): a group of LPP lanes (power of two, <= 64) owns one (batch, query, head); each lane carries
// CPT consecutive channels (16-byte value loads for D % 4 == 0).  The reference launches D-thread blocks (12/24/32
// threads for the s/b/l models -- half a wavefront or less) and reduces through shared memory with a serial loop;
// here the D-reduction of grad_loc / grad_attn is an in-wave xor-shuffle tree and several (q, head) pairs share a wave.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <typename T, int CPT> struct ChanVec;
template <> struct ChanVec<float, 4> { typedef float4 Raw; };
template <> struct ChanVec<float, 1> { typedef float Raw; };
template <> struct ChanVec<bf16_t, 4> { typedef uint2 Raw; };
template <> struct ChanVec<bf16_t, 1> { typedef bf16_t Raw; };

template <typename T, int CPT>
__device__ __forceinline__ void load_chan(const T* p, float* out) {
  if constexpr (CPT == 1) {
    out[0] = to_f32(*p);
  } else if constexpr (sizeof(T) == 4) {
    float4 v = *(const float4*)p;
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
    uint2 u = *(const uint2*)p;
    bf16x4 v = __builtin_bit_cast(bf16x4, u);
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = (float)v[j];
  }
}
template <typename T, int CPT>
__device__ __forceinline__ void store_chan(T* p, const float* in) {
  if constexpr (CPT == 1) {
    *p = from_f32<T>(in[0]);
  } else if constexpr (sizeof(T) == 4) {
    *(float4*)p = make_float4(in[0], in[1], in[2], in[3]);
  } else {
    bf16x4 v;
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (bf16_t)in[j];
    *(uint2*)p = __builtin_bit_cast(uint2, v);
  }
}

struct Corner { int off[4]; float wt[4]; bool ok[4]; float lh, lw; };

// bilinear corner setup for pixel coords (h, w) inside a level of size H x W; offsets in units of pixels
__device__ __forceinline__ Corner corners(float h, float w, int H, int W) {
  Corner c;
  const int h0 = (int)floorf(h), w0 = (int)floorf(w);
  const float lh = h - h0, lw = w - w0, hh = 1.f - lh, hw = 1.f - lw;
  c.lh = lh; c.lw = lw;
  c.off[0] = h0 * W + w0;       c.wt[0] = hh * hw; c.ok[0] = (h0 >= 0) & (w0 >= 0);
  c.off[1] = h0 * W + w0 + 1;   c.wt[1] = hh * lw; c.ok[1] = (h0 >= 0) & (w0 + 1 <= W - 1);
  c.off[2] = (h0 + 1) * W + w0; c.wt[2] = lh * hw; c.ok[2] = (h0 + 1 <= H - 1) & (w0 >= 0);
  c.off[3] = (h0 + 1) * W + w0 + 1; c.wt[3] = lh * lw; c.ok[3] = (h0 + 1 <= H - 1) & (w0 + 1 <= W - 1);
  return c;
}

// ------------------------------------------------------------------------------------------------------
// forward: one thread per (b, q, m, channel chunk)
// ------------------------------------------------------------------------------------------------------
template <typename T, int CPT>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, T* __restrict__ out, int N, int S, int M,
                                                       int D, int L, int Lq, int P, long total) {
  const int chunks = D / CPT;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ch = (int)(idx % chunks);
    const long pair = idx / chunks;          // (b*Lq + q)*M + m
    const int m = (int)(pair % M);
    const long bq = pair / M;
    const int b = (int)(bq / Lq);
    const int c0 = ch * CPT;
    const float* lp = loc + pair * (long)L * P * 2;
    const float* ap = attn + pair * (long)L * P;
    const long qstride = (long)M * D;
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; j++) acc[j] = 0.f;
    for (int l = 0; l < L; l++) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* vbase = value + ((long)b * S + lsi[l]) * qstride + (long)m * D + c0;
      // every corner is gathered unconditionally from a clamped address with a zeroed weight when it is outside the level: no
      // branch separates the loads, so all corners of all points are in flight together (the kernel is gather-latency bound)
      auto point = [&](int p) {
        const float lw_ = lp[(l * P + p) * 2], lh_ = lp[(l * P + p) * 2 + 1];
        const float a = ap[l * P + p];
        float h = lh_ * H - 0.5f, w = lw_ * W - 0.5f;
        const bool in = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
        h = in ? h : 0.f; w = in ? w : 0.f;
        Corner c = corners(h, w, H, W);
        float v[4][CPT];
#pragma unroll
        for (int k = 0; k < 4; k++) load_chan<T, CPT>(vbase + (long)((in && c.ok[k]) ? c.off[k] : 0) * qstride, v[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float f = (in && c.ok[k]) ? c.wt[k] * a : 0.f;
#pragma unroll
          for (int j = 0; j < CPT; j++) acc[j] += f * v[k][j];
        }
      };
      if (P == 4) {
#pragma unroll
        for (int p = 0; p < 4; p++) point(p);
      } else {
        for (int p = 0; p < P; p++) point(p);
      }
    }
    store_chan<T, CPT>(out + pair * (long)D + c0, acc);
  }
}

// ------------------------------------------------------------------------------------------------------
// backward: LPP lanes per (b, q, m); every lane loops over its channel chunks; per-sample partials of
// grad_attn / grad_loc are reduced across the LPP lanes with xor shuffles; grad_value uses fp32 atomics
// (contiguous per (pixel, head): D floats = one or two cache lines).
// ------------------------------------------------------------------------------------------------------
template <typename T, int CPT, int MAXLP, bool GV>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, const T* __restrict__ gout,
                                                       float* __restrict__ gvalue, float* __restrict__ gloc,
                                                       float* __restrict__ gattn, int N, int S, int M, int D, int L, int Lq,
                                                       int P, int LPP, long npairs) {
  const int gpw = 256 / LPP;                 // pairs per block
  const int sub = threadIdx.x % LPP;         // lane within the pair group
  const int chunks = (D + CPT - 1) / CPT;
  const long qstride = (long)M * D;
  const int LP = L * P;
  // the LPP lanes of a group share `pair`, so they leave the loop together and the xor shuffles below only
  // ever exchange data between lanes that are all active
  for (long pair = (long)blockIdx.x * gpw + threadIdx.x / LPP; pair < npairs; pair += (long)gridDim.x * gpw) {
    const bool live = true;
    const long pr = pair;
    const int m = (int)(pr % M);
    const long bq = pr / M;
    const int b = (int)(bq / Lq);
    const float* lp = loc + pr * (long)LP * 2;
    const float* ap = attn + pr * (long)LP;
    float ga[MAXLP], gx[MAXLP], gy[MAXLP];
#pragma unroll
    for (int s = 0; s < MAXLP; s++) { ga[s] = 0.f; gx[s] = 0.f; gy[s] = 0.f; }
    for (int ch = sub; ch < chunks; ch += LPP) {
      const int c0 = ch * CPT;
      float tg[CPT];
      load_chan<T, CPT>(gout + pr * (long)D + c0, tg);
#pragma unroll
      for (int s = 0; s < MAXLP; s++) {
        if (s < LP) {
          const int l = s / P;
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const long base = ((long)b * S + lsi[l]) * qstride + (long)m * D + c0;
          const float lw_ = lp[s * 2], lh_ = lp[s * 2 + 1];
          const float a = ap[s];
          const float h = lh_ * H - 0.5f, w = lw_ * W - 0.5f;
          const bool in = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
          if (GV ? in : true) {
            // without the grad_value atomics (GV = false) nothing below needs a branch: corners outside the level are gathered from a
            // clamped address with zero coefficients, so the loads of all corners and samples overlap
            Corner c = corners(in ? h : 0.f, in ? w : 0.f, H, W);
            const float hh = 1.f - c.lh, hw = 1.f - c.lw;
            // d(val)/dh and d(val)/dw corner coefficients (cuh:122-158)
            const float dh[4] = {-hw, -c.lw, hw, c.lw};
            const float dw[4] = {-hh, hh, -c.lh, c.lh};
            float val[CPT], ghw[CPT], gww[CPT];
#pragma unroll
            for (int j = 0; j < CPT; j++) { val[j] = 0.f; ghw[j] = 0.f; gww[j] = 0.f; }
            if constexpr (GV) {
#pragma unroll
              for (int k = 0; k < 4; k++) {
                if (c.ok[k]) {
                  float v[CPT];
                  load_chan<T, CPT>(value + base + (long)c.off[k] * qstride, v);
#pragma unroll
                  for (int j = 0; j < CPT; j++) {
                    val[j] += c.wt[k] * v[j];
                    ghw[j] += dh[k] * v[j];
                    gww[j] += dw[k] * v[j];
                    atomic_add_f32(gvalue + base + (long)c.off[k] * qstride + j, c.wt[k] * tg[j] * a);
                  }
                }
              }
            } else {
              float v[4][CPT];
#pragma unroll
              for (int k = 0; k < 4; k++) load_chan<T, CPT>(value + base + (long)((in && c.ok[k]) ? c.off[k] : 0) * qstride, v[k]);
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const bool okk = in && c.ok[k];
                const float wk = okk ? c.wt[k] : 0.f, dhk = okk ? dh[k] : 0.f, dwk = okk ? dw[k] : 0.f;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                  val[j] += wk * v[k][j];
                  ghw[j] += dhk * v[k][j];
                  gww[j] += dwk * v[k][j];
                }
              }
            }
#pragma unroll
            for (int j = 0; j < CPT; j++) {
              ga[s] += tg[j] * val[j];
              gx[s] += (float)W * gww[j] * tg[j] * a;
              gy[s] += (float)H * ghw[j] * tg[j] * a;
            }
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MAXLP; s++) {
      if (s < LP) {
        float a = ga[s], x = gx[s], y = gy[s];
        for (int o = LPP >> 1; o > 0; o >>= 1) {
          a += __shfl_xor(a, o, 64); x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64);
        }
        if (live && sub == 0) {
          gattn[pr * (long)LP + s] = a;
          gloc[(pr * (long)LP + s) * 2] = x;
          gloc[(pr * (long)LP + s) * 2 + 1] = y;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------
// backward, LDS-resident grad_value: one workgroup owns one (batch, head) value plane (S x D fp32 <= 144 KB of the
// 160 KB LDS) and a chunk of the queries; the 4-corner scatter goes to LDS atomics (ds_add_f32) and is flushed once
// with one global atomic per plane element.  Cuts global fp32 atomics from Lq*P*4*D to S*D per (b, head, chunk).
// ------------------------------------------------------------------------------------------------------
template <typename T, int CPT, int MAXLP>
__global__ __launch_bounds__(1024) void msda_bwd_lds_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                            const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                            const float* __restrict__ attn, const T* __restrict__ gout,
                                                            float* __restrict__ gvalue, float* __restrict__ gloc,
                                                            float* __restrict__ gattn, int N, int S, int M, int D, int L, int Lq,
                                                            int P, int LPP, int q_per_chunk, float* __restrict__ part, int PLD) {
  // gacc[S][PLD]: one pixel row per value pixel, padded (PLD = D + 4 when it fits) so that the rows of different pixels start
  // on different banks; within a row channel c sits at position (c % CPT) * (D / CPT) + c / CPT, so the lanes of a query group
  // (lane `sub` carries channels CPT*sub .. CPT*sub + CPT-1) hit CONSECUTIVE banks in each of the CPT atomic instructions
  // instead of every CPT-th one (4-way+ bank conflicts on ds_add_f32 were 80 % of this kernel's time).
  extern __shared__ __attribute__((aligned(16))) float gacc[];
  const int tid = threadIdx.x;
  const int b = blockIdx.y / M, m = blockIdx.y % M;
  const int q0 = blockIdx.x * q_per_chunk;
  const int q1 = min(Lq, q0 + q_per_chunk);
  for (int i = tid; i < S * PLD; i += 1024) gacc[i] = 0.f;
  __syncthreads();
  const int DQ = D / CPT;                    // channels-per-position stride of the permuted row layout
  const int gpw = 1024 / LPP;
  const int sub = tid % LPP;
  const int chunks = (D + CPT - 1) / CPT;
  const long qstride = (long)M * D;
  const int LP = L * P;
  for (int q = q0 + tid / LPP; q < q1; q += gpw) {
    const long pr = ((long)b * Lq + q) * M + m;
    const float* lp = loc + pr * (long)LP * 2;
    const float* ap = attn + pr * (long)LP;
    float ga[MAXLP], gx[MAXLP], gy[MAXLP];
#pragma unroll
    for (int s = 0; s < MAXLP; s++) { ga[s] = 0.f; gx[s] = 0.f; gy[s] = 0.f; }
    for (int ch = sub; ch < chunks; ch += LPP) {
      const int c0 = ch * CPT;
      float tg[CPT];
      load_chan<T, CPT>(gout + pr * (long)D + c0, tg);
#pragma unroll
      for (int s = 0; s < MAXLP; s++) {
        if (s < LP) {
          const int l = s / P;
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const int ls = (int)lsi[l];
          const long base = ((long)b * S + ls) * qstride + (long)m * D + c0;
          const float lw_ = lp[s * 2], lh_ = lp[s * 2 + 1];
          const float a = ap[s];
          const float h = lh_ * H - 0.5f, w = lw_ * W - 0.5f;
          if (h > -1.f && w > -1.f && h < (float)H && w < (float)W) {
            Corner c = corners(h, w, H, W);
            const float hh = 1.f - c.lh, hw = 1.f - c.lw;
            const float dh[4] = {-hw, -c.lw, hw, c.lw};
            const float dw[4] = {-hh, hh, -c.lh, c.lh};
            float val[CPT], ghw[CPT], gww[CPT];
#pragma unroll
            for (int j = 0; j < CPT; j++) { val[j] = 0.f; ghw[j] = 0.f; gww[j] = 0.f; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
              if (c.ok[k]) {
                float v[CPT];
                load_chan<T, CPT>(value + base + (long)c.off[k] * qstride, v);
                float* ldst = gacc + (ls + c.off[k]) * PLD + ch;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                  val[j] += c.wt[k] * v[j];
                  ghw[j] += dh[k] * v[j];
                  gww[j] += dw[k] * v[j];
                  atomicAdd(ldst + j * DQ, c.wt[k] * tg[j] * a);
                }
              }
            }
#pragma unroll
            for (int j = 0; j < CPT; j++) {
              ga[s] += tg[j] * val[j];
              gx[s] += (float)W * gww[j] * tg[j] * a;
              gy[s] += (float)H * ghw[j] * tg[j] * a;
            }
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MAXLP; s++) {
      if (s < LP) {
        float a = ga[s], x = gx[s], y = gy[s];
        for (int o = LPP >> 1; o > 0; o >>= 1) {
          a += __shfl_xor(a, o, 64); x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64);
        }
        if (sub == 0) {
          gattn[pr * (long)LP + s] = a;
          gloc[(pr * (long)LP + s) * 2] = x;
          gloc[(pr * (long)LP + s) * 2 + 1] = y;
        }
      }
    }
  }
  __syncthreads();
  if (part) {
    // two-stage: this query chunk's plane goes to part[chunk][b][s][m][:] with plain stores; msda_gv_finalize_kernel adds the chunks
    float* dst = part + (long)blockIdx.x * N * S * qstride;
    for (int i = tid; i < S * D; i += 1024) {
      const int sidx = i / D, c = i - sidx * D;
      dst[((long)b * S + sidx) * qstride + (long)m * D + c] = gacc[sidx * PLD + (c % CPT) * DQ + c / CPT];
    }
    return;
  }
  for (int i = tid; i < S * D; i += 1024) {
    const int sidx = i / D, c = i - sidx * D;
    const float v = gacc[sidx * PLD + (c % CPT) * DQ + c / CPT];
    if (v != 0.f) atomic_add_f32(gvalue + ((long)b * S + sidx) * qstride + (long)m * D + c, v);
  }
}

// grad_value = sum over the query chunks of the partial planes
__global__ __launch_bounds__(256) void msda_gv_finalize_kernel(const float* __restrict__ part, float* __restrict__ gvalue, int nchunk,
                                                               long n4, int out_bf16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = ((const float4*)part)[i];
    for (int c = 1; c < nchunk; c++) {
      float4 t = ((const float4*)part)[i + (long)c * n4];
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    if (out_bf16) {
      bf16x4 o; o[0] = (bf16_t)a.x; o[1] = (bf16_t)a.y; o[2] = (bf16_t)a.z; o[3] = (bf16_t)a.w;
      ((uint2*)gvalue)[i] = __builtin_bit_cast(uint2, o);
    } else {
      ((float4*)gvalue)[i] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// grad_value as a matrix product on the MFMA pipe (bf16 throughput mode, single level).
//
//   grad_value[pix][c] = sum over (query q, point p) of  W[pix][(q,p)] * grad_out[q][c]
//   W[pix][(q,p)] = attn(q,p) * (bilinear weight of sample (q,p) at corner pixel pix), 0 elsewhere (<= 4 non-zeros per column)
//
// The scatter of the reference (atomicAdd per corner and channel, ms_deform_im2col_cuda.cuh:130-157) becomes: write the <= 4
// non-zeros of each (q,p) column into an LDS image of W (no two threads ever write the same element: distinct columns, or
// distinct corner pixels of one column), multiply with v_mfma_f32_32x32x16_bf16, accumulate the plane tile in registers.  Measured
// motive: ds_add_f32 runs lane-serially on gfx950 (the LDS-atomic kernel spent 1.75 ms per call, independent of bank layout).
//
// Workgroup = 1024 threads = 16 waves; it owns PT = 512 consecutive value pixels of one (batch, head) plane: wave w accumulates
// pixels [32w, 32w+32) x 32 channels (one f32x16).  K step = 128 columns = 32 queries x 4 points:
//   clear own previous non-zero, install the new column entries + G^T tile, OR the touched 32-pixel blocks into a step mask
//   -> barrier -> waves whose pixel block was touched: 8 MFMAs (the others skip) -> barrier (skipped when nothing was touched).
// The clear and the install of one column are done by the 4 corner lanes of that column -- adjacent lanes of ONE wave -- so program
// order keeps them apart without a barrier.  Consecutive queries sample neighbouring pixels (reference point + learned offset), so
// a step usually touches 2-3 of the 16 pixel blocks: the skip removes most of the dense-MFMA redundancy, and is only a speed-up
// (any offset pattern stays correct).  A 64-column double-buffered variant (one barrier per step) measured slower: the per-step
// scatter + barrier cost, not the barrier count, paces the loop.
// ------------------------------------------------------------------------------------------------------
// Round 3: most K steps touch nothing.  A workgroup owns 512 consecutive plane pixels (8 rows of the 64 x 64 level) but walked ALL
// Lq / 32 query blocks -- scatter arithmetic + a 16-wave barrier each -- although a query samples around its own reference point: ~33 of
// the 168 blocks of the dinounet_l adapter reach a given 8-row tile.  msda_gv_rows_kernel records, per (batch, head, 32-query block), the
// first and last plane row any of its 128 samples can touch (exact bilinear support, empty = 0xffff / 0); the product kernel builds the
// ordered list of the blocks whose rows meet its tile and walks only those.  Skipping is decided on exact bounds, so results are
// unchanged for any offsets (a model with far-reaching offsets simply skips less).
__global__ __launch_bounds__(256) void msda_gv_rows_kernel(const int64_t* __restrict__ shapes, const float* __restrict__ loc,
                                                           unsigned* __restrict__ rows, int N, int M, int Lq, int nblk) {
  // workgroup = one (batch, 32-query block), all heads: a query's locations of all heads and points are one contiguous M * 32-byte run,
  // so thread (query, head) reads its 4 points as two 16-byte loads next to its neighbours'; per-head min / max through LDS atomics
  __shared__ int lo_s[64], hi_s[64];
  const int Hs = (int)shapes[0], Ws = (int)shapes[1];
  const int b = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  for (int i = threadIdx.x; i < M; i += 256) { lo_s[i] = 0xffff; hi_s[i] = -1; }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * M; i += 256) {
    const int m = i % M, q = blk * 32 + i / M;
    if (q >= Lq) break;
    const float4* lp = (const float4*)(loc + (((long)b * Lq + q) * M + m) * 8);
    const float4 l0 = lp[0], l1 = lp[1];
    const float xs[4] = {l0.x, l0.z, l1.x, l1.z}, ys[4] = {l0.y, l0.w, l1.y, l1.w};
    int lo = 0xffff, hi = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float h = ys[k] * Hs - 0.5f, w = xs[k] * Ws - 0.5f;
      if (h > -1.f && w > -1.f && h < (float)Hs && w < (float)Ws) {
        const int h0 = (int)floorf(h);
        lo = min(lo, h0 < 0 ? 0 : h0); hi = max(hi, h0 + 1 > Hs - 1 ? Hs - 1 : h0 + 1);
      }
    }
    if (hi >= 0) { atomicMin(&lo_s[m], lo); atomicMax(&hi_s[m], hi); }
  }
  __syncthreads();
  for (int m = threadIdx.x; m < M; m += 256)
    rows[((long)b * M + m) * nblk + blk] = hi_s[m] < 0 ? 0xffffu : ((unsigned)lo_s[m] | ((unsigned)hi_s[m] << 16));
}

constexpr int GV_PT = 512;        // plane pixels per 1024-thread workgroup (32 per wave); the 512-thread form owns 256
template <int N_> struct GvSlot { static constexpr int value = N_; };
constexpr int GV_MAXBLK = 1024;   // query blocks per chunk the skip list holds (more: every block is walked)
constexpr int GV_KQ = 32;         // queries per K step
constexpr int GV_WLD = 128 + 8;   // W row pitch (bf16): 272 B, conflict-free ds_read_b128 fragments
constexpr int GV_GLD = 128 + 8;

// NT = 1024: 16 waves, 512 plane pixels, one workgroup per CU (139 KB of LDS).  NT = 512 (round 4): 8 waves, 256 pixels, 78 KB -- TWO
// workgroups per CU, so one's barrier-paced steps run under the other's, and a 256-pixel tile is reached by fewer query blocks.
template <int NT>
__global__ __launch_bounds__(NT) void msda_gv_mfma_kernel(const bf16_t* __restrict__ gout, const int64_t* __restrict__ shapes,
                                                            const float* __restrict__ loc, const float* __restrict__ attn,
                                                            float* __restrict__ gvalue, int N, int S, int M, int D, int Lq,
                                                            int q_per_chunk, int tiles, long chunk_stride, int out_bf16,
                                                            const unsigned* __restrict__ rows, int nblk_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ unsigned step_mask[3];
  __shared__ unsigned short need[GV_MAXBLK];
  __shared__ int wave_cnt[16];
  constexpr int PT = NT / 2;                            // plane pixels of this workgroup: 32 per wave
  constexpr int GQ = 1024 / NT;                         // grad_out elements per thread and step (32 queries x 32 channels)
  bf16_t* Wl = (bf16_t*)smem_raw;                       // [GV_PT][GV_WLD]
  bf16_t* Gt = Wl + PT * GV_WLD;                     // [32 channels][GV_GLD]   (G^T: k contiguous)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hs = (int)shapes[0], Ws = (int)shapes[1];   // single level
  // XCD-aware order (workgroup y lands on XCD y % 8 for the usual one-chunk grid): the loc / attn / grad_out rows of a query hold
  // all M heads back to back, so neighbouring heads must share an XCD for the 128-byte lines to be reused in its L2.  XCD x serves
  // the (head, pixel tile) pairs x*(M*tiles/8) .. of every batch sample instead of heads x, x+4, x+8, ... (4 different lines).
  int bm = blockIdx.y / tiles, ptile = blockIdx.y % tiles;
  const int mt = M * tiles;
  if ((mt & 7) == 0 && gridDim.x == 1) {
    const int per = mt >> 3, xcd = blockIdx.y & 7, idx = blockIdx.y >> 3;      // idx in [0, N * per)
    const int bb = idx / per, r = idx - bb * per, hp = xcd * per + r;          // hp: (head, tile) pair of this sample
    bm = bb * M + hp / tiles; ptile = hp % tiles;
  }
  const int b = bm / M, m = bm % M;
  const int pix0 = ptile * PT;
  const int q0 = blockIdx.x * q_per_chunk;
  const int q1 = min(Lq, q0 + q_per_chunk);
  for (int i = tid; i < (PT * GV_WLD + 32 * GV_GLD) / 8; i += NT) ((uint4*)Wl)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 3) step_mask[tid] = 0u;
  // ---- ordered list of the query blocks of this chunk whose rows meet this tile (thread t = block t of the chunk) ----
  const int nblk = (q1 - q0 + GV_KQ - 1) / GV_KQ;
  const bool listed = rows != nullptr && nblk <= (GV_MAXBLK < NT ? GV_MAXBLK : NT);
  int nneed = nblk;
  if (listed) {
    bool want = false;
    if (tid < nblk) {
      const unsigned r = rows[((long)b * M + m) * nblk_all + q0 / GV_KQ + tid];
      const int lo = (int)(r & 0xffffu), hi = (int)(r >> 16);
      // pixels the block can touch: [lo * Ws, hi * Ws + Ws - 1]; this tile: [pix0, pix0 + PT - 1]
      want = lo != 0xffff && lo * Ws <= pix0 + PT - 1 && hi * Ws + Ws - 1 >= pix0;
    }
    const unsigned long long bal = __ballot(want);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < NT / 64; w2++) { const int c2 = wave_cnt[w2]; if (w2 < wave) base += c2; total += c2; }
    if (want) need[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)tid;
    nneed = total;
  }
  __syncthreads();
  auto block_q = [&](int i) { return q0 + (listed ? (int)need[i] : i) * GV_KQ; };

  f32x16 acc, acc2;          // even / odd k-steps: two chains of 4 dependent MFMAs instead of one of 8 (only the 2-3 touched waves work in
                             // a step, so the chain latency is the step's MFMA phase)
#pragma unroll
  for (int r = 0; r < 16; r++) { acc[r] = 0.f; acc2[r] = 0.f; }
  // scatter role (threads 0..511): column col = (query ql, point p), one bilinear corner each
  const int col = tid >> 2, corner = tid & 3;
  const int ql_s = col >> 2, p_s = col & 3;
  // G role (all threads): queries ql_g + (NT / 32) j, j < GQ, channel c_g
  const int ql_g = tid >> 5, c_g = tid & 31;
  int my_off = -1;
  // the per-step operands (sampling location, attention weight, grad_out element) are fetched THREE steps ahead into a ring of three
  // register sets (round 3: a step is ~1 us, about one L2 / HBM latency -- with one step of prefetch every step began by waiting for its
  // own operands).  The step body exists three times, one per set, so set indices are static and the compiler's vmcnt stays counted.
  float lx_r[3] = {0.f, 0.f, 0.f}, ly_r[3] = {0.f, 0.f, 0.f}, a_r[3] = {0.f, 0.f, 0.f};
  bf16_t g_r[3][GQ];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < GQ; j++) g_r[i][j] = (bf16_t)0.f;
  // every thread issues the same three loads in every fetch (clamped addresses; validity is applied when the values are used): with
  // loads under divergent or wave-dependent conditions the number of younger loads differs by path and the compiler falls back to
  // s_waitcnt vmcnt(0) in front of every use -- which drains the ring
  auto fetch = [&](auto slot, int qs) {
    constexpr int SL = decltype(slot)::value;
    const int q = min(qs + ql_s, Lq - 1);
    const long pr = ((long)b * Lq + q) * M + m;
    const float2 l2 = *(const float2*)(loc + (pr * 4 + p_s) * 2);
    lx_r[SL] = l2.x; ly_r[SL] = l2.y;
    a_r[SL] = attn[pr * 4 + p_s];
#pragma unroll
    for (int j = 0; j < GQ; j++) {
      const int qg = min(qs + ql_g + (NT / 32) * j, Lq - 1);
      g_r[SL][j] = gout[(((long)b * Lq + qg) * M + m) * D + min(c_g, D - 1)];
    }
  };
  auto block_q_or_end = [&](int i) { return i < nneed ? block_q(i) : q1; };
  fetch(GvSlot<0>{}, block_q_or_end(0));
  fetch(GvSlot<1>{}, block_q_or_end(1));
  fetch(GvSlot<2>{}, block_q_or_end(2));
  // one K step on register set / step mask SL (= step % 3: three masks rotate so that a mask is cleared a full step before its next writers)
  auto do_step = [&](auto slot, int step) {
    constexpr int par = decltype(slot)::value;
    const int qs = block_q(step);
    const bool ok_s = qs + ql_s < q1;
    const float lx = ok_s ? lx_r[par] : -4.f, ly = ok_s ? ly_r[par] : -4.f, a = ok_s ? a_r[par] : 0.f;
    bf16_t gval[GQ];
#pragma unroll
    for (int j = 0; j < GQ; j++) gval[j] = (qs + ql_g + (NT / 32) * j < q1 && c_g < D) ? g_r[par][j] : (bf16_t)0.f;
    fetch(slot, block_q_or_end(step + 3));
    if (tid < 512) {                                   // waves 0..7 (wave-uniform)
      float wv = 0.f; int woff = -1; unsigned blk = 0u;
      const float h = ly * Hs - 0.5f, w = lx * Ws - 0.5f;
      if (h > -1.f && w > -1.f && h < (float)Hs && w < (float)Ws) {
        const int h0 = (int)floorf(h), w0 = (int)floorf(w);
        const float lh = h - h0, lw = w - w0;
        const int hy = h0 + (corner >> 1), wx = w0 + (corner & 1);
        if (hy >= 0 && hy < Hs && wx >= 0 && wx < Ws) {
          const int pl = hy * Ws + wx - pix0;
          if (pl >= 0 && pl < PT) {
            wv = ((corner >> 1) ? lh : 1.f - lh) * ((corner & 1) ? lw : 1.f - lw) * a;
            woff = pl * GV_WLD + col;
            blk = 1u << (pl >> 5);
          }
        }
      }
      if (my_off >= 0) Wl[my_off] = (bf16_t)0.f;
      if (woff >= 0) Wl[woff] = (bf16_t)wv;
      my_off = woff;
      // OR over the wave: DPP inside the 16-lane rows (4 full-rate VALU ops), scalar reads across the 4 rows -- the xor-shuffle tree
      // it replaces was six dependent ds_bpermute round trips on the critical path of every one of the 168 barrier-paced steps
      blk |= (unsigned)__builtin_amdgcn_mov_dpp((int)blk, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
      blk |= (unsigned)__builtin_amdgcn_mov_dpp((int)blk, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
      blk |= (unsigned)__builtin_amdgcn_mov_dpp((int)blk, 0x141, 0xf, 0xf, true);     // row_half_mirror: lane i <-> 7 - i of its 8
      blk |= (unsigned)__builtin_amdgcn_mov_dpp((int)blk, 0x140, 0xf, 0xf, true);     // row_mirror: lane i <-> 15 - i of its 16
      blk = (unsigned)(__builtin_amdgcn_readlane((int)blk, 0) | __builtin_amdgcn_readlane((int)blk, 16) |
                       __builtin_amdgcn_readlane((int)blk, 32) | __builtin_amdgcn_readlane((int)blk, 48));
      if (lane == 0 && blk) atomicOr(&step_mask[par], blk);
    }
#pragma unroll
    for (int j = 0; j < GQ; j++) {
      bf16x4 g4; g4[0] = gval[j]; g4[1] = gval[j]; g4[2] = gval[j]; g4[3] = gval[j];
      *(bf16x4*)(Gt + c_g * GV_GLD + (ql_g + (NT / 32) * j) * 4) = g4;
    }
    __syncthreads();
    const unsigned touched = step_mask[par];           // block-uniform
    // the mask of step s+2: its last readers (step s-1) are behind the barrier above, its next writers behind the next step's
    if (tid == 0) step_mask[par == 0 ? 2 : par - 1] = 0u;
    if (touched) {
      if ((touched >> wave) & 1u) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          bf16x8 fa = *(const bf16x8*)(Wl + (wave * 32 + (lane & 31)) * GV_WLD + kk * 16 + (lane >> 5) * 8);
          bf16x8 fb = *(const bf16x8*)(Gt + (lane & 31) * GV_GLD + kk * 16 + (lane >> 5) * 8);
          if (kk & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
      }
      __syncthreads();                                 // W / G^T are rewritten by the next step (nobody read them if nothing was touched)
    }
  };
  // whole triples unconditionally (a conditional copy inside the loop makes the number of loads younger than a set path-dependent and
  // the compiler's waits collapse towards vmcnt(0)); the last one or two steps peeled
  int step = 0;
  for (; step + 3 <= nneed; step += 3) {
    do_step(GvSlot<0>{}, step);
    do_step(GvSlot<1>{}, step + 1);
    do_step(GvSlot<2>{}, step + 2);
  }
  if (step < nneed) {
    do_step(GvSlot<0>{}, step);
    if (step + 1 < nneed) do_step(GvSlot<1>{}, step + 1);
  }
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] += acc2[r];
  // lane: channel (lane & 31), pixels (r&3) + 8*(r>>2) + 4*(lane>>5) of the wave's 32-pixel block
  const int c = lane & 31;
  if (c < D) {
    float* dst = gvalue + (long)blockIdx.x * chunk_stride;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int pix = pix0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const long e = ((long)b * S + pix) * ((long)M * D) + (long)m * D + c;
      if (pix < S) {
        if (out_bf16) ((bf16_t*)gvalue)[e] = (bf16_t)acc[r];       // (one query chunk: the result itself, in the activation dtype)
        else dst[e] = acc[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// bf16 fast path (round 3): D % 8 == 0 with D / 8 in {4, 8, 16} (dinounet_l: 32 channels per head; the 7B adapter: 128), 4 points.
//
// The kernels above were VALU-bound (SQ counters: VALU issue 0.92 / 0.89 of the launch): every one of the D / 4 lanes of a (query, head)
// pair recomputed the bilinear setup of all 4 points (floor, 4 weights, 4 range tests, 4 corner offsets through 64-bit multiplies) and
// gathered 8 bytes per corner.  Here
//   * a pair is served by LPP = D / 8 lanes (16-byte gathers: half the gather instructions),
//   * lane (l & 3) of every quad sets up ONE of the 4 points and the others fetch its 4 corner offsets / coefficients with
//     v_mov_b32_dpp quad_perm broadcasts (one full-rate VALU op each; the LPP >= 4 lanes of a pair are whole quads),
//   * offsets are 32-bit BYTE offsets into one buffer descriptor over `value` (the product pixel * (M * D * 2) is a 24-bit multiply done
//     once in the setup lane), so a gather costs one v_add + one buffer_load_dwordx4,
//   * corners outside the level are gathered from offset 0 with zero coefficients (no branches: all 16 gathers of a pair in flight).
// The backward variant adds the three per-point reductions (grad_attn, grad_loc x / y) over the channels: 8 FMAs per corner into one
// dot product with this lane's grad_out slice, three scalar FMAs with the corner's coefficients, quad butterflies (DPP) at the end.
// ------------------------------------------------------------------------------------------------------
template <int P_> __device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, P_ | (P_ << 2) | (P_ << 4) | (P_ << 6), 0xf, 0xf, true);
}
template <int P_> __device__ __forceinline__ float quad_bcastf(float v) { return __builtin_bit_cast(float, quad_bcast<P_>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)); }   // quad_perm [2,3,0,1]

struct PointSetup { int off[4]; float cw[4]; float lh, lw; bool in; bool ok[4]; };
// this lane's point: pixel coordinates -> 4 corner byte offsets (pixel * pix_bytes, 0 when the corner is outside) and bilinear weights (0 outside)
__device__ __forceinline__ PointSetup point_setup(float lx, float ly, int H, int W, unsigned pix_bytes) {
  PointSetup s;
  float h = ly * H - 0.5f, w = lx * W - 0.5f;
  s.in = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
  h = s.in ? h : 0.f; w = s.in ? w : 0.f;
  const float hf = floorf(h), wf = floorf(w);
  const int h0 = (int)hf, w0 = (int)wf;
  s.lh = h - hf; s.lw = w - wf;
  const float hh = 1.f - s.lh, hw = 1.f - s.lw;
  const bool okh0 = s.in && h0 >= 0, okh1 = s.in && h0 + 1 <= H - 1, okw0 = w0 >= 0, okw1 = w0 + 1 <= W - 1;
  const bool ok[4] = {okh0 && okw0, okh0 && okw1, okh1 && okw0, okh1 && okw1};
#pragma unroll
  for (int k = 0; k < 4; k++) s.ok[k] = ok[k];
  const int pix[4] = {h0 * W + w0, h0 * W + w0 + 1, (h0 + 1) * W + w0, (h0 + 1) * W + w0 + 1};
  const float wt[4] = {hh * hw, hh * s.lw, s.lh * hw, s.lh * s.lw};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    s.off[k] = ok[k] ? (int)__umul24((unsigned)pix[k], pix_bytes) : 0;
    s.cw[k] = ok[k] ? wt[k] : 0.f;
  }
  return s;
}
__device__ __forceinline__ void bf16x8_to_f32(const uint4& u, float* f) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
  }
}

template <int LPP>
__global__ __launch_bounds__(256) void msda_fwd_q8_kernel(const bf16_t* __restrict__ value, const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                          const float* __restrict__ attn, bf16_t* __restrict__ out, int N, int S, int M,
                                                          int D, int L, int Lq, long npairs) {
  constexpr int GPW = 256 / LPP;
  const int tid = threadIdx.x;
  const int sub = tid % LPP, pt = tid & 3;
  const unsigned pix_bytes = (unsigned)(M * D * 2);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)value, 0, (int)((long)N * S * M * D * 2), 0x00020000);
  const long rounds = (npairs + (long)gridDim.x * GPW - 1) / ((long)gridDim.x * GPW);
  for (long it = 0; it < rounds; it++) {
    const long pair = (it * gridDim.x + blockIdx.x) * GPW + tid / LPP;
    const bool live = pair < npairs;
    const long pr = live ? pair : npairs - 1;             // (whole quads stay active for the DPP exchanges)
    const int m = (int)(pr % M);
    const int b = (int)(pr / ((long)M * Lq));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    for (int l = 0; l < L; l++) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float2 lxy = *(const float2*)(loc + ((pr * L + l) * 4 + pt) * 2);
      const float a = attn[(pr * L + l) * 4 + pt];
      PointSetup ps = point_setup(lxy.x, lxy.y, H, W, pix_bytes);
      float ca[4];
#pragma unroll
      for (int k = 0; k < 4; k++) ca[k] = ps.cw[k] * a;
      const unsigned base = (unsigned)(((b * S + (int)lsi[l]) * M + m) * D + sub * 8) * 2u;
      // two batches of 8 gathers (two points each): 32 registers of loads in flight per wave, 4-5 waves per SIMD
      auto two_points = [&](auto p_c) {
        constexpr int p = decltype(p_c)::value;
        uint4 v[8];
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          f[k] = quad_bcastf<p>(ca[k]);
          v[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (unsigned)quad_bcast<p>(ps.off[k]), 0, 0));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          f[4 + k] = quad_bcastf<p + 1>(ca[k]);
          v[4 + k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + (unsigned)quad_bcast<p + 1>(ps.off[k]), 0, 0));
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
          float x[8];
          bf16x8_to_f32(v[c], x);
#pragma unroll
          for (int j = 0; j < 8; j++) acc[j] = fmaf(f[c], x[j], acc[j]);
        }
      };
      two_points(std::integral_constant<int, 0>{});
      two_points(std::integral_constant<int, 2>{});
    }
    if (live) {
      bf16x8 o8;
#pragma unroll
      for (int j = 0; j < 8; j++) o8[j] = (bf16_t)acc[j];
      *(uint4*)(out + pr * (long)D + sub * 8) = __builtin_bit_cast(uint4, o8);
    }
  }
}

// gather-only backward (grad_sampling_loc, grad_attn_weight): same lane assignment; lane (l & 3) of the pair's FIRST quad writes point (l & 3)
template <int LPP>
__global__ __launch_bounds__(256) void msda_bwd_q8_kernel(const bf16_t* __restrict__ value, const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                          const float* __restrict__ attn, const bf16_t* __restrict__ gout,
                                                          float* __restrict__ gloc, float* __restrict__ gattn, int N, int S, int M, int D,
                                                          int L, int Lq, long npairs) {
  constexpr int GPW = 256 / LPP;
  const int tid = threadIdx.x;
  const int sub = tid % LPP, pt = tid & 3;
  const unsigned pix_bytes = (unsigned)(M * D * 2);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)value, 0, (int)((long)N * S * M * D * 2), 0x00020000);
  const long rounds = (npairs + (long)gridDim.x * GPW - 1) / ((long)gridDim.x * GPW);
  for (long it = 0; it < rounds; it++) {
    const long pair = (it * gridDim.x + blockIdx.x) * GPW + tid / LPP;
    const bool live = pair < npairs;
    const long pr = live ? pair : npairs - 1;
    const int m = (int)(pr % M);
    const int b = (int)(pr / ((long)M * Lq));
    float tg[8];
    bf16x8_to_f32(*(const uint4*)(gout + pr * (long)D + sub * 8), tg);
    for (int l = 0; l < L; l++) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float2 lxy = *(const float2*)(loc + ((pr * L + l) * 4 + pt) * 2);
      const float a = attn[(pr * L + l) * 4 + pt];
      PointSetup ps = point_setup(lxy.x, lxy.y, H, W, pix_bytes);
      // d(val)/dh and d(val)/dw corner coefficients (cuh:122-158), scaled by H * a / W * a like the reference's grad_loc; zero outside
      const float hh = 1.f - ps.lh, hw = 1.f - ps.lw;
      const float dh0[4] = {-hw, -ps.lw, hw, ps.lw}, dw0[4] = {-hh, hh, -ps.lh, ps.lh};
      float dh[4], dw[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        dh[k] = ps.ok[k] ? dh0[k] * ((float)H * a) : 0.f;
        dw[k] = ps.ok[k] ? dw0[k] * ((float)W * a) : 0.f;
      }
      const unsigned base = (unsigned)(((b * S + (int)lsi[l]) * M + m) * D + sub * 8) * 2u;
      float ga[4], gx[4], gy[4];
      auto point = [&](auto p_c) {
        constexpr int p = decltype(p_c)::value;
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned o = base + (unsigned)quad_bcast<p>(ps.off[k]);
          v[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0));
        }
        float sa = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float x[8];
          bf16x8_to_f32(v[k], x);
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < 8; j++) t = fmaf(tg[j], x[j], t);
          sa = fmaf(quad_bcastf<p>(ps.cw[k]), t, sa);
          sx = fmaf(quad_bcastf<p>(dw[k]), t, sx);
          sy = fmaf(quad_bcastf<p>(dh[k]), t, sy);
        }
        ga[p] = sa; gx[p] = sx; gy[p] = sy;
      };
      point(std::integral_constant<int, 0>{}); point(std::integral_constant<int, 1>{});
      point(std::integral_constant<int, 2>{}); point(std::integral_constant<int, 3>{});
      // sum over the LPP lanes of the pair: inside the quad by DPP, across the pair's quads by shuffles
#pragma unroll
      for (int p = 0; p < 4; p++) {
        ga[p] += quad_xor1(ga[p]); gx[p] += quad_xor1(gx[p]); gy[p] += quad_xor1(gy[p]);
        ga[p] += quad_xor2(ga[p]); gx[p] += quad_xor2(gx[p]); gy[p] += quad_xor2(gy[p]);
#pragma unroll
        for (int o = 4; o < LPP; o <<= 1) {
          ga[p] += __shfl_xor(ga[p], o, 64); gx[p] += __shfl_xor(gx[p], o, 64); gy[p] += __shfl_xor(gy[p], o, 64);
        }
      }
      const float oa = pt == 0 ? ga[0] : pt == 1 ? ga[1] : pt == 2 ? ga[2] : ga[3];
      const float ox = pt == 0 ? gx[0] : pt == 1 ? gx[1] : pt == 2 ? gx[2] : gx[3];
      const float oy = pt == 0 ? gy[0] : pt == 1 ? gy[1] : pt == 2 ? gy[2] : gy[3];
      if (live && sub < 4) {
        const long e = (pr * L + l) * 4 + pt;
        gattn[e] = oa;
        *(float2*)(gloc + e * 2) = make_float2(ox, oy);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LDS-resident value plane (round 4): single level, 4 points, 32 channels per head, S * 128 B <= 128 KB (the adapter's Extractor: the
// ViT's 32 x 32 token map, 16 heads).  The q8 kernels above gather their 16 corners per (query, head) from L2 -- 64-byte segments, 16 of
// them per wave instruction through the texture-address path -- and sit at ~2x their VALU time (51 / 78 us, profiles/r03 steady trace).
// Here a workgroup owns the value plane of TWO adjacent heads of one image ([pixel][2 x 32 channels] = 128 B per pixel, as it lies in
// `value`) in LDS and walks a chunk of the queries: the gathers are ds_read_b128 (256 B / clock / CU, ~100-cycle latency).  8 lanes serve
// one query (2 heads x 4 channel slices), so the output / grad_out rows are full 128-byte lines and loc / attn are read in 64 / 32-byte
// pieces; the workgroups of the 8 head pairs of the same (image, query chunk) are dispatched 8 apart = onto the same XCD, whose L2
// then serves the other halves of those lines.
// Backward (grad_sampling_loc / grad_attn_weight): the channel dot products of a corner with this lane's grad_out slice are four
// v_dot2_f32_bf16 on the packed pairs as they come from LDS -- no unpacking.
// ------------------------------------------------------------------------------------------------------
constexpr int MP_THREADS = 1024;
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct PlaneJob { int b, hp, q0, q1; };
__device__ __forceinline__ PlaneJob plane_job(int N, int M, int Lq, int nchunk) {
  // blockIdx -> (slot = image x chunk, head pair): head pairs of one slot sit 8 workgroups apart (same XCD)
  const int HP = M / 2;
  const int bid = blockIdx.x;
  const int lane8 = bid & 7, rest = bid >> 3;
  const int hp = rest % HP, grp = rest / HP;
  const int slot = grp * 8 + lane8;
  PlaneJob j;
  j.hp = hp;
  j.b = slot / nchunk;
  const int ch = slot - j.b * nchunk;
  const int qpc = (Lq + nchunk - 1) / nchunk;
  j.q0 = ch * qpc; j.q1 = j.q0 + qpc < Lq ? j.q0 + qpc : Lq;
  if (j.b >= N) { j.q0 = 0; j.q1 = 0; j.b = 0; }
  return j;
}
__device__ __forceinline__ void plane_load(unsigned char* plane, const bf16_t* value, int b, int hp, int S, int M) {
  // [pixel][128 B] <- value[b][pixel][2 hp .. 2 hp + 1][0 .. 31]
  const bf16_t* src = value + ((long)b * S * M + 2 * hp) * 32;
  for (int v = threadIdx.x; v < S * 8; v += MP_THREADS)
    *(uint4*)(plane + v * 16) = *(const uint4*)(src + (long)(v >> 3) * M * 32 + (v & 7) * 8);
  __syncthreads();
}

__global__ __launch_bounds__(MP_THREADS) void msda_fwd_plane_kernel(const bf16_t* __restrict__ value, const int64_t* __restrict__ shapes,
                                                                   const float* __restrict__ loc, const float* __restrict__ attn,
                                                                   bf16_t* __restrict__ out, int N, int S, int M, int Lq, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mp_plane[];
  const PlaneJob J = plane_job(N, M, Lq, nchunk);
  plane_load(mp_plane, value, J.b, J.hp, S, M);
  const int H = (int)shapes[0], W = (int)shapes[1];
  const int tid = threadIdx.x;
  const int pt = tid & 3, hsel = (tid >> 2) & 1;
  const int m = 2 * J.hp + hsel;
  const unsigned lbase = (unsigned)((tid & 7) * 16);                  // this lane's 16 bytes inside a pixel's 128
  const int qend = J.q0 + ((J.q1 - J.q0 + 7) & ~7);                    // whole groups of 8 queries = whole waves in the loop (DPP exchanges)
  for (int q = J.q0 + (tid >> 3); q < qend; q += MP_THREADS / 8) {
    const bool live = q < J.q1;
    const long pr = ((long)J.b * Lq + (live ? q : J.q1 - 1)) * M + m;
    const float2 lxy = *(const float2*)(loc + (pr * 4 + pt) * 2);
    const float a = attn[pr * 4 + pt];
    PointSetup ps = point_setup(lxy.x, lxy.y, H, W, 128u);
    float ca[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ca[k] = ps.cw[k] * a;
    f32x2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x2{0.f, 0.f};
    auto point = [&](auto p_c) {
      constexpr int p = decltype(p_c)::value;
      uint4 v[4];
      float f[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        f[k] = quad_bcastf<p>(ca[k]);
        v[k] = *(const uint4*)(mp_plane + lbase + (unsigned)quad_bcast<p>(ps.off[k]));
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        const f32x2 ff = {f[k], f[k]};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const f32x2 x = {__builtin_bit_cast(float, w[j] << 16), __builtin_bit_cast(float, w[j] & 0xffff0000u)};
          acc[j] = __builtin_elementwise_fma(ff, x, acc[j]);
        }
      }
    };
    point(std::integral_constant<int, 0>{}); point(std::integral_constant<int, 1>{});
    point(std::integral_constant<int, 2>{}); point(std::integral_constant<int, 3>{});
    if (live) {
      bf16x8 o8;
#pragma unroll
      for (int j = 0; j < 4; j++) { o8[2 * j] = (bf16_t)acc[j][0]; o8[2 * j + 1] = (bf16_t)acc[j][1]; }
      *(uint4*)(out + pr * 32 + (tid & 3) * 8) = __builtin_bit_cast(uint4, o8);
    }
  }
}

__global__ __launch_bounds__(MP_THREADS) void msda_bwd_plane_kernel(const bf16_t* __restrict__ value, const int64_t* __restrict__ shapes,
                                                                   const float* __restrict__ loc, const float* __restrict__ attn,
                                                                   const bf16_t* __restrict__ gout, float* __restrict__ gloc,
                                                                   float* __restrict__ gattn, int N, int S, int M, int Lq, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mp_plane[];
  const PlaneJob J = plane_job(N, M, Lq, nchunk);
  plane_load(mp_plane, value, J.b, J.hp, S, M);
  const int H = (int)shapes[0], W = (int)shapes[1];
  const int tid = threadIdx.x;
  const int pt = tid & 3, hsel = (tid >> 2) & 1;
  const int m = 2 * J.hp + hsel;
  const unsigned lbase = (unsigned)((tid & 7) * 16);
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const int qend = J.q0 + ((J.q1 - J.q0 + 7) & ~7);                    // whole groups of 8 queries = whole waves in the loop
  for (int q = J.q0 + (tid >> 3); q < qend; q += MP_THREADS / 8) {
    const bool live = q < J.q1;
    const long pr = ((long)J.b * Lq + (live ? q : J.q1 - 1)) * M + m;
    const uint4 tgv = *(const uint4*)(gout + pr * 32 + (tid & 3) * 8);
    const unsigned tgw[4] = {tgv.x, tgv.y, tgv.z, tgv.w};
    const float2 lxy = *(const float2*)(loc + (pr * 4 + pt) * 2);
    const float a = attn[pr * 4 + pt];
    PointSetup ps = point_setup(lxy.x, lxy.y, H, W, 128u);
    // d(val)/dh and d(val)/dw corner coefficients (cuh:122-158), scaled by H * a / W * a like the reference's grad_loc; zero outside
    const float hh = 1.f - ps.lh, hw = 1.f - ps.lw;
    const float dh0[4] = {-hw, -ps.lw, hw, ps.lw}, dw0[4] = {-hh, hh, -ps.lh, ps.lh};
    float dh[4], dw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      dh[k] = ps.ok[k] ? dh0[k] * ((float)H * a) : 0.f;
      dw[k] = ps.ok[k] ? dw0[k] * ((float)W * a) : 0.f;
    }
    float ga[4], gx[4], gy[4];
    auto point = [&](auto p_c) {
      constexpr int p = decltype(p_c)::value;
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = *(const uint4*)(mp_plane + lbase + (unsigned)quad_bcast<p>(ps.off[k]));
      float sa = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++)
          t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[j]), __builtin_bit_cast(bf16x2_t, tgw[j]), t, false);
        sa = fmaf(quad_bcastf<p>(ps.cw[k]), t, sa);
        sx = fmaf(quad_bcastf<p>(dw[k]), t, sx);
        sy = fmaf(quad_bcastf<p>(dh[k]), t, sy);
      }
      ga[p] = sa; gx[p] = sx; gy[p] = sy;
    };
    point(std::integral_constant<int, 0>{}); point(std::integral_constant<int, 1>{});
    point(std::integral_constant<int, 2>{}); point(std::integral_constant<int, 3>{});
    // sum over the 4 lanes of the (query, head) pair: inside the quad by DPP
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ga[p] += quad_xor1(ga[p]); gx[p] += quad_xor1(gx[p]); gy[p] += quad_xor1(gy[p]);
      ga[p] += quad_xor2(ga[p]); gx[p] += quad_xor2(gx[p]); gy[p] += quad_xor2(gy[p]);
    }
    const float oa = pt == 0 ? ga[0] : pt == 1 ? ga[1] : pt == 2 ? ga[2] : ga[3];
    const float ox = pt == 0 ? gx[0] : pt == 1 ? gx[1] : pt == 2 ? gx[2] : gx[3];
    const float oy = pt == 0 ? gy[0] : pt == 1 ? gy[1] : pt == 2 ? gy[2] : gy[3];
    if (live) {
      const long e = pr * 4 + pt;
      gattn[e] = oa;
      *(float2*)(gloc + e * 2) = make_float2(ox, oy);
    }
  }
}

// the plane kernels serve: bf16, one level, 4 points, 32 channels per head, an even number of heads, the two-head plane within 128 KB
static bool plane_serves(int N, int S, int M, int D, int L, int P, int Lq) {
  static const bool off = DU_GETENV("DU_MSDA_NO_PLANE") != nullptr;      // debugging / A-B aid
  return !off && L == 1 && P == 4 && D == 32 && M % 2 == 0 && (long)S * 128 <= 128 * 1024 && Lq >= 64;
}
// query chunks per (image, head pair): one workgroup per CU with room to spare, slots (image x chunk) in multiples of 8
static int plane_chunks(int N, int M, int Lq) {
  int nchunk = (int)((256 + (long)N * (M / 2) - 1) / ((long)N * (M / 2)));
  while ((N * nchunk) % 8) nchunk++;
  const int maxc = Lq / 64 > 0 ? Lq / 64 : 1;
  return nchunk > maxc ? maxc : nchunk;
}

int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }
// threads per workgroup of the MFMA grad_value kernel: 512 (256-pixel tiles, two workgroups per CU) unless DU_MSDA_GV_THREADS=1024
static int gv_threads() {
  static const int nt = (DU_GETENV("DU_MSDA_GV_THREADS") && atoi(DU_GETENV("DU_MSDA_GV_THREADS")) == 1024) ? 1024 : 512;
  return nt;
}

// the round-3 fast kernels serve: bf16, 4 points, D / 8 in {4, 8, 16}, value < 2 GB (32-bit byte offsets), < 2^24 pixels and row bytes
static bool q8_serves(int N, int S, int M, int D, int P) {
  static const bool off = DU_GETENV("DU_MSDA_NO_Q8") != nullptr;      // debugging / A-B aid
  if (off || P != 4 || D % 8) return false;
  const int lpp = D / 8;
  if (lpp != 4 && lpp != 8 && lpp != 16) return false;
  return (long)N * S * M * D * 2 < 0x7fffffffL && (long)N * S < (1L << 24) && (long)M * D * 2 < (1L << 24);
}

template <typename T>
int fwd_dispatch(const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn, void* out,
                 int N, int S, int M, int D, int L, int Lq, int P, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (plane_serves(N, S, M, D, L, P, Lq)) {
      const int nchunk = plane_chunks(N, M, Lq);
      const int slots = ((N * nchunk + 7) / 8) * 8;
      const int lds = S * 128;
      if (hipFuncSetAttribute((const void*)msda_fwd_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DU_ERR_LAUNCH;
      hipLaunchKernelGGL(msda_fwd_plane_kernel, dim3((unsigned)(slots * (M / 2))), dim3(MP_THREADS), lds, st, (const bf16_t*)value, shapes, loc, attn,
                         (bf16_t*)out, N, S, M, Lq, nchunk);
      return du_check_launch();
    }
    if (q8_serves(N, S, M, D, P)) {
      const int lpp = D / 8;
      const long npairs = (long)N * Lq * M;
      const long gpw = 256 / lpp;
      long blocks = (npairs + gpw - 1) / gpw;
      if (blocks > 256 * 32) blocks = 256 * 32;
#define MSDA_FWD_Q8(LPP_) hipLaunchKernelGGL((msda_fwd_q8_kernel<LPP_>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)value, shapes, lsi, loc, attn, (bf16_t*)out, N, S, M, D, L, Lq, npairs)
      if (lpp == 4) MSDA_FWD_Q8(4); else if (lpp == 8) MSDA_FWD_Q8(8); else MSDA_FWD_Q8(16);
#undef MSDA_FWD_Q8
      return du_check_launch();
    }
  }
  if (D % 4 == 0) {
    long total = (long)N * Lq * M * (D / 4);
    int grid = (int)((total + 255) / 256 > 65535 * 4 ? 65535 * 4 : (total + 255) / 256);
    hipLaunchKernelGGL((msda_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)value, shapes, lsi, loc, attn, (T*)out, N, S,
                       M, D, L, Lq, P, total);
  } else {
    long total = (long)N * Lq * M * D;
    int grid = (int)((total + 255) / 256 > 65535 * 4 ? 65535 * 4 : (total + 255) / 256);
    hipLaunchKernelGGL((msda_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)value, shapes, lsi, loc, attn, (T*)out, N, S,
                       M, D, L, Lq, P, total);
  }
  return du_check_launch();
}

// query chunking of the LDS-resident backward: enough workgroups to fill 256 CUs; each chunk costs one S*D flush
static void lds_chunking(int N, int M, int Lq, int LPP, int* nchunk_out, int* qpc_out) {
  int nchunk = (int)((256 + (long)N * M - 1) / ((long)N * M));   // one 1024-thread workgroup (~130-150 KB LDS) per CU
  if (nchunk < 1) nchunk = 1;
  if (nchunk > 16) nchunk = 16;
  int qpc = (Lq + nchunk - 1) / nchunk;
  const int gq = 1024 / LPP;
  qpc = ((qpc + gq - 1) / gq) * gq;
  *nchunk_out = (Lq + qpc - 1) / qpc;
  *qpc_out = qpc;
}

template <typename T, int CPT>
int bwd_launch(const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn, const void* gout,
               float* gv, float* gl, float* ga, int N, int S, int M, int D, int L, int Lq, int P, float* ws, long ws_elems,
               hipStream_t st, int gv_bf16 = 0) {
  const int chunks = (D + CPT - 1) / CPT;
  int LPP = next_pow2(chunks);
  if (LPP > 64) LPP = 64;
  const long npairs = (long)N * Lq * M;
  const int gpw = 256 / LPP;
  long blocks = (npairs + gpw - 1) / gpw;
  if (blocks > 65535 * 8) blocks = 65535 * 8;
  const int LP = L * P;
  const size_t plane = (size_t)S * D * sizeof(float);
  static const bool no_mfma = DU_GETENV("DU_MSDA_NO_MFMA") != nullptr;  // debugging / A-B aid
  if constexpr (sizeof(T) == 2) {
    if (!no_mfma && L == 1 && P == 4 && D <= 32 && LP <= 4) {
      // (1) gather-only pass: grad_sampling_loc / grad_attn_weight (no atomics)
      if (plane_serves(N, S, M, D, L, P, Lq)) {
        const int nchunk = plane_chunks(N, M, Lq);
        const int slots = ((N * nchunk + 7) / 8) * 8;
        const int lds = S * 128;
        if (hipFuncSetAttribute((const void*)msda_bwd_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DU_ERR_LAUNCH;
        hipLaunchKernelGGL(msda_bwd_plane_kernel, dim3((unsigned)(slots * (M / 2))), dim3(MP_THREADS), lds, st, (const bf16_t*)value, shapes, loc, attn,
                           (const bf16_t*)gout, gl, ga, N, S, M, Lq, nchunk);
      } else if (q8_serves(N, S, M, D, P)) {
        const long gpw8 = 256 / (D / 8);
        long b8 = (npairs + gpw8 - 1) / gpw8;
        if (b8 > 256 * 32) b8 = 256 * 32;
#define MSDA_BWD_Q8(LPP_) hipLaunchKernelGGL((msda_bwd_q8_kernel<LPP_>), dim3((unsigned)b8), dim3(256), 0, st, (const bf16_t*)value, shapes, lsi, loc, attn, (const bf16_t*)gout, gl, ga, N, S, M, D, L, Lq, npairs)
        if (D == 32) MSDA_BWD_Q8(4); else if (D == 64) MSDA_BWD_Q8(8); else MSDA_BWD_Q8(16);
#undef MSDA_BWD_Q8
      } else
      hipLaunchKernelGGL((msda_bwd_kernel<T, CPT, 4, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)value, shapes, lsi, loc, attn,
                         (const T*)gout, gv, gl, ga, N, S, M, D, L, Lq, P, LPP, npairs);
      // (2) grad_value on the MFMA pipe
      {
        const int NT = gv_threads();
        const int PT = NT / 2;
        const int tiles = (S + PT - 1) / PT;
        const int want_wgs = NT == 512 ? 512 : 256;                // two resident workgroups per CU in the 512-thread form
        int nchunk = (int)((want_wgs + (long)N * M * tiles - 1) / ((long)N * M * tiles));
        if (nchunk < 1) nchunk = 1;
        if (nchunk > 16) nchunk = 16;
        int qpc = (Lq + nchunk - 1) / nchunk;
        qpc = ((qpc + GV_KQ - 1) / GV_KQ) * GV_KQ;
        nchunk = (Lq + qpc - 1) / qpc;
        const long plane_all = (long)N * S * M * D;
        float* dstp = gv; long cstride = 0;
        if (nchunk > 1) {
          if (ws && ws_elems >= (long)nchunk * plane_all && plane_all % 4 == 0) { dstp = ws; cstride = plane_all; }
          else { nchunk = 1; qpc = ((Lq + GV_KQ - 1) / GV_KQ) * GV_KQ; }
        }
        const int lds_bytes = (PT * GV_WLD + 32 * GV_GLD) * 2;
        if (hipFuncSetAttribute(NT == 512 ? (const void*)msda_gv_mfma_kernel<512> : (const void*)msda_gv_mfma_kernel<1024>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return DU_ERR_LAUNCH;
        // rows table of the query blocks (behind the chunk partials in the workspace); without room for it every block is walked
        const int nblk_all = (Lq + GV_KQ - 1) / GV_KQ;
        const long tab_off = nchunk > 1 ? (long)nchunk * plane_all : 0, tab_n = (long)N * M * nblk_all;
        static const bool no_skip = DU_GETENV("DU_MSDA_GV_NO_SKIP") != nullptr;      // debugging / A-B aid
        unsigned* rows = (!no_skip && M <= 64 && ws && ws_elems >= tab_off + tab_n) ? (unsigned*)(ws + tab_off) : nullptr;
        if (rows)
          hipLaunchKernelGGL(msda_gv_rows_kernel, dim3((unsigned)(N * nblk_all)), dim3(256), 0, st, shapes, loc, rows, N, M, Lq, nblk_all);
        if (NT == 512)
          hipLaunchKernelGGL(msda_gv_mfma_kernel<512>, dim3(nchunk, N * M * tiles), dim3(512), lds_bytes, st, (const bf16_t*)gout, shapes, loc, attn,
                             dstp, N, S, M, D, Lq, qpc, tiles, cstride, (gv_bf16 && nchunk == 1) ? 1 : 0, (const unsigned*)rows, nblk_all);
        else
          hipLaunchKernelGGL(msda_gv_mfma_kernel<1024>, dim3(nchunk, N * M * tiles), dim3(1024), lds_bytes, st, (const bf16_t*)gout, shapes, loc, attn,
                             dstp, N, S, M, D, Lq, qpc, tiles, cstride, (gv_bf16 && nchunk == 1) ? 1 : 0, (const unsigned*)rows, nblk_all);
        if (nchunk > 1) {
          const long n4 = plane_all / 4;
          long g = (n4 + 255) / 256; if (g > 4096) g = 4096;
          hipLaunchKernelGGL(msda_gv_finalize_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)ws, gv, nchunk, n4, gv_bf16);
        }
        return du_check_launch();
      }
    }
  }
  if (gv_bf16) return DU_ERR_UNSUPPORTED;                            // only the MFMA grad_value path above writes the activation dtype
  static const bool no_lds = DU_GETENV("DU_MSDA_NO_LDS") != nullptr;   // debugging aid: force the global-atomics kernel
  if (!no_lds && plane <= 144 * 1024 && LP <= 8 && LPP <= 64) {
    int nchunk, qpc;
    lds_chunking(N, M, Lq, LPP, &nchunk, &qpc);
    const long plane_all = (long)N * S * M * D;
    const int PLD = ((size_t)S * (D + 4) * sizeof(float) <= 156 * 1024 && D % CPT == 0) ? D + 4 : D;
    const size_t lds_bytes = (size_t)S * PLD * sizeof(float);
    float* part = (ws && ws_elems >= (long)nchunk * plane_all && plane_all % 4 == 0 && nchunk > 1) ? ws : nullptr;
    if (!part && hipMemsetAsync(gv, 0, (size_t)plane_all * sizeof(float), st) != hipSuccess) return DU_ERR_LAUNCH;   // atomic flush target
    dim3 grid(nchunk, N * M);
#define MSDA_BWD_LDS(MAXLP) do { \
      auto kfn = msda_bwd_lds_kernel<T, CPT, MAXLP>; \
      if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return DU_ERR_LAUNCH; \
      hipLaunchKernelGGL(kfn, grid, dim3(1024), lds_bytes, st, (const T*)value, shapes, lsi, loc, attn, (const T*)gout, gv, gl, ga, N, S, M, D, L, Lq, P, LPP, qpc, part, PLD); \
    } while (0)
    if (LP <= 4) MSDA_BWD_LDS(4); else MSDA_BWD_LDS(8);
#undef MSDA_BWD_LDS
    if (part) {
      const long n4 = plane_all / 4;
      long g = (n4 + 255) / 256; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(msda_gv_finalize_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)part, gv, nchunk, n4, 0);
    }
    return du_check_launch();
  }
  if (hipMemsetAsync(gv, 0, (size_t)N * S * M * D * sizeof(float), st) != hipSuccess) return DU_ERR_LAUNCH;             // atomic scatter target
#define MSDA_BWD(MAXLP) hipLaunchKernelGGL((msda_bwd_kernel<T, CPT, MAXLP, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)value, shapes, lsi, loc, attn, (const T*)gout, gv, gl, ga, N, S, M, D, L, Lq, P, LPP, npairs)
  if (LP <= 4) MSDA_BWD(4); else if (LP <= 8) MSDA_BWD(8); else if (LP <= 16) MSDA_BWD(16); else return DU_ERR_UNSUPPORTED;
#undef MSDA_BWD
  return du_check_launch();
}

// ---- fp64 (the reference extension dispatches AT_DISPATCH_FLOATING_TYPES: fp32 AND fp64, ms_deform_attn_cuda.cu:69,139; its own
// acceptance script ops/test.py feeds .double() tensors through MSDeformAttnFunction and torch.autograd.gradcheck).  Not a hot path: one
// workgroup per (batch, query, head), threads over the channels; everything (value, locations, weights, gradients) in double.
// Sampling rule of ms_deform_im2col_cuda.cuh:242-304 (forward) and :92-164 (backward): pixel = loc * size - 0.5, a sample counts when
// -1 < pixel < size, corners outside the level contribute zero.
struct Corner64 { int h0, w0; double lh, lw; bool ok; };
__device__ __forceinline__ Corner64 corner64(double ly, double lx, int H, int W) {
  Corner64 c;
  const double h = ly * H - 0.5, w = lx * W - 0.5;
  c.ok = h > -1.0 && w > -1.0 && h < (double)H && w < (double)W;
  const double fh = floor(h), fw = floor(w);
  c.h0 = (int)fh; c.w0 = (int)fw; c.lh = h - fh; c.lw = w - fw;
  return c;
}
__global__ __launch_bounds__(256) void msda_fwd_f64_kernel(const double* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lsi, const double* __restrict__ loc,
                                                           const double* __restrict__ attn, double* __restrict__ out, int N, int S, int M,
                                                           int D, int L, int Lq, int P) {
  const long nqm = blockIdx.x;                  // (n, q, m)
  const int m = (int)(nqm % M);
  const long nq = nqm / M;
  const int n = (int)(nq / Lq);
  const double* lp = loc + nqm * L * P * 2;
  const double* ap = attn + nqm * L * P;
  for (int d = threadIdx.x; d < D; d += 256) {
    double acc = 0.0;
    for (int l = 0; l < L; l++) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const double* vb = value + ((long)n * S + lsi[l]) * M * D + (long)m * D + d;
      for (int p = 0; p < P; p++) {
        const Corner64 c = corner64(lp[(l * P + p) * 2 + 1], lp[(l * P + p) * 2], H, W);
        if (!c.ok) continue;
        const double hh = 1.0 - c.lh, hw = 1.0 - c.lw;
        double v = 0.0;
        if (c.h0 >= 0 && c.w0 >= 0) v += hh * hw * vb[((long)c.h0 * W + c.w0) * M * D];
        if (c.h0 >= 0 && c.w0 + 1 <= W - 1) v += hh * c.lw * vb[((long)c.h0 * W + c.w0 + 1) * M * D];
        if (c.h0 + 1 <= H - 1 && c.w0 >= 0) v += c.lh * hw * vb[((long)(c.h0 + 1) * W + c.w0) * M * D];
        if (c.h0 + 1 <= H - 1 && c.w0 + 1 <= W - 1) v += c.lh * c.lw * vb[((long)(c.h0 + 1) * W + c.w0 + 1) * M * D];
        acc += ap[l * P + p] * v;
      }
    }
    out[nq * M * D + (long)m * D + d] = acc;
  }
}
// grad_value by fp64 atomics (zeroed by the launcher); grad_loc / grad_attn: per-thread partial sums over its channels, block reduction
__global__ __launch_bounds__(256) void msda_bwd_f64_kernel(const double* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lsi, const double* __restrict__ loc,
                                                           const double* __restrict__ attn, const double* __restrict__ gout,
                                                           double* __restrict__ gv, double* __restrict__ gl, double* __restrict__ ga,
                                                           int N, int S, int M, int D, int L, int Lq, int P) {
  __shared__ double red[3][256];
  const long nqm = blockIdx.x;
  const int m = (int)(nqm % M);
  const long nq = nqm / M;
  const int n = (int)(nq / Lq);
  const double* lp = loc + nqm * L * P * 2;
  const double* ap = attn + nqm * L * P;
  for (int l = 0; l < L; l++) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long base = ((long)n * S + lsi[l]) * M * D + (long)m * D;
    for (int p = 0; p < P; p++) {
      const Corner64 c = corner64(lp[(l * P + p) * 2 + 1], lp[(l * P + p) * 2], H, W);
      const double a = ap[l * P + p];
      double s_a = 0.0, s_h = 0.0, s_w = 0.0;
      if (c.ok) {
        const double hh = 1.0 - c.lh, hw = 1.0 - c.lw;
        const bool k1 = c.h0 >= 0 && c.w0 >= 0, k2 = c.h0 >= 0 && c.w0 + 1 <= W - 1, k3 = c.h0 + 1 <= H - 1 && c.w0 >= 0,
                   k4 = c.h0 + 1 <= H - 1 && c.w0 + 1 <= W - 1;
        const long o1 = base + ((long)c.h0 * W + c.w0) * M * D, o2 = o1 + (long)M * D, o3 = o1 + (long)W * M * D, o4 = o3 + (long)M * D;
        for (int d = threadIdx.x; d < D; d += 256) {
          const double g = gout[nq * M * D + (long)m * D + d];
          const double v1 = k1 ? value[o1 + d] : 0.0, v2 = k2 ? value[o2 + d] : 0.0, v3 = k3 ? value[o3 + d] : 0.0, v4 = k4 ? value[o4 + d] : 0.0;
          if (k1) atomicAdd(gv + o1 + d, hh * hw * a * g);
          if (k2) atomicAdd(gv + o2 + d, hh * c.lw * a * g);
          if (k3) atomicAdd(gv + o3 + d, c.lh * hw * a * g);
          if (k4) atomicAdd(gv + o4 + d, c.lh * c.lw * a * g);
          s_a += g * (hh * hw * v1 + hh * c.lw * v2 + c.lh * hw * v3 + c.lh * c.lw * v4);
          s_h += g * a * (-hw * v1 - c.lw * v2 + hw * v3 + c.lw * v4);
          s_w += g * a * (-hh * v1 + hh * v2 - c.lh * v3 + c.lh * v4);
        }
      }
      red[0][threadIdx.x] = s_a; red[1][threadIdx.x] = s_h; red[2][threadIdx.x] = s_w;
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
          red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; red[2][threadIdx.x] += red[2][threadIdx.x + st];
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        ga[nqm * L * P + l * P + p] = red[0][0];
        gl[(nqm * L * P + l * P + p) * 2] = red[2][0] * W;          // d / d loc_x
        gl[(nqm * L * P + l * P + p) * 2 + 1] = red[1][0] * H;      // d / d loc_y
      }
      __syncthreads();
    }
  }
}

}  // namespace

extern "C" int du_msda_forward_f64(const double* value, const int64_t* shapes, const int64_t* lsi, const double* loc, const double* attn,
                                   double* out, int N, int S, int M, int D, int L, int Lq, int P, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !out || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0)
    return DU_ERR_BAD_ARG;
  if ((long)N * Lq * M > 0x7fffffffL) return DU_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(msda_fwd_f64_kernel, dim3((unsigned)((long)N * Lq * M)), dim3(256), 0, st, value, shapes, lsi, loc, attn, out, N, S, M, D, L, Lq, P);
  return du_check_launch();
}
extern "C" int du_msda_backward_f64(const double* value, const int64_t* shapes, const int64_t* lsi, const double* loc, const double* attn,
                                    const double* gout, double* gv, double* gl, double* ga, int N, int S, int M, int D, int L, int Lq, int P,
                                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !gout || !gv || !gl || !ga || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 ||
      P <= 0)
    return DU_ERR_BAD_ARG;
  if ((long)N * Lq * M > 0x7fffffffL) return DU_ERR_UNSUPPORTED;
  if (hipMemsetAsync(gv, 0, (size_t)N * S * M * D * sizeof(double), st) != hipSuccess) return DU_ERR_LAUNCH;
  hipLaunchKernelGGL(msda_bwd_f64_kernel, dim3((unsigned)((long)N * Lq * M)), dim3(256), 0, st, value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D,
                     L, Lq, P);
  return du_check_launch();
}

extern "C" int du_msda_forward(int dtype, const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                               const float* attn, void* out, int N, int S, int M, int D, int L, int Lq, int P, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !out || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0)
    return DU_ERR_BAD_ARG;
  if (dtype == DU_F32) return fwd_dispatch<float>(value, shapes, lsi, loc, attn, out, N, S, M, D, L, Lq, P, st);
  if (dtype == DU_BF16) return fwd_dispatch<bf16_t>(value, shapes, lsi, loc, attn, out, N, S, M, D, L, Lq, P, st);
  return DU_ERR_BAD_ARG;
}

extern "C" int64_t du_msda_bwd_ws_elems(int N, int S, int M, int D, int L, int Lq, int P) {
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
  const int cpt = D % 4 == 0 ? 4 : 1;
  int LPP = next_pow2((D + cpt - 1) / cpt);
  if (LPP > 64) LPP = 64;
  // rows table of msda_gv_rows_kernel (MFMA grad_value path: bf16, single level, 4 points, D <= 32), behind the chunk partials
  const int64_t tab = (L == 1 && P == 4 && D <= 32) ? (int64_t)N * M * ((Lq + GV_KQ - 1) / GV_KQ) : 0;
  if (L * P > 8) return tab;
  int nchunk = 1, qpc;
  if ((size_t)S * D * sizeof(float) <= 144 * 1024) lds_chunking(N, M, Lq, LPP, &nchunk, &qpc);
  // MFMA grad_value path (bf16, single level, 4 points): query chunks so that ~256 workgroups exist
  const int PT = gv_threads() / 2;
  const int tiles = (S + PT - 1) / PT;
  int nc2 = (int)(((gv_threads() == 512 ? 512 : 256) + (long)N * M * tiles - 1) / ((long)N * M * tiles));
  if (nc2 > 16) nc2 = 16;
  if (nc2 > nchunk) nchunk = nc2;
  return (nchunk > 1 ? (int64_t)nchunk * N * S * M * D : 0) + tab;
}

extern "C" int du_msda_backward(int dtype, const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                                const float* attn, const void* gout, float* gv, float* gl, float* ga, int N, int S, int M, int D,
                                int L, int Lq, int P, float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !gout || !gv || !gl || !ga || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 ||
      Lq <= 0 || P <= 0)
    return DU_ERR_BAD_ARG;
  if (dtype == DU_F32) {
    if (D % 4 == 0) return bwd_launch<float, 4>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, ws, ws_elems, st);
    return bwd_launch<float, 1>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, ws, ws_elems, st);
  }
  if (dtype == DU_BF16) {
    if (D % 4 == 0) return bwd_launch<bf16_t, 4>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, ws, ws_elems, st);
    return bwd_launch<bf16_t, 1>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, ws, ws_elems, st);
  }
  return DU_ERR_BAD_ARG;
}

// The same with grad_value written in bf16 (the dtype of `value`: what the value projection's backward consumes) instead of fp32: saves
// the cast pass.  Served on the MFMA grad_value path only (bf16, one level, 4 points, D <= 32); DU_ERR_UNSUPPORTED otherwise -- the caller
// then uses du_msda_backward and casts.
extern "C" int du_msda_backward_bf16gv(const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                                       const void* gout, void* gv_bf16, float* gl, float* ga, int N, int S, int M, int D, int L, int Lq,
                                       int P, float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !gout || !gv_bf16 || !gl || !ga || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 ||
      Lq <= 0 || P <= 0)
    return DU_ERR_BAD_ARG;
  static const bool no_mfma = DU_GETENV("DU_MSDA_NO_MFMA") != nullptr;
  if (no_mfma || !(L == 1 && P == 4 && D <= 32 && D % 4 == 0)) return DU_ERR_UNSUPPORTED;
  return bwd_launch<bf16_t, 4>(value, shapes, lsi, loc, attn, gout, (float*)gv_bf16, gl, ga, N, S, M, D, L, Lq, P, ws, ws_elems, st, 1);
}
