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
      for (int p = 0; p < P; p++) {
        const float lw_ = lp[(l * P + p) * 2], lh_ = lp[(l * P + p) * 2 + 1];
        const float a = ap[l * P + p];
        const float h = lh_ * H - 0.5f, w = lw_ * W - 0.5f;
        if (h > -1.f && w > -1.f && h < (float)H && w < (float)W) {
          Corner c = corners(h, w, H, W);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (c.ok[k]) {
              float v[CPT];
              load_chan<T, CPT>(vbase + (long)c.off[k] * qstride, v);
              const float f = c.wt[k] * a;
#pragma unroll
              for (int j = 0; j < CPT; j++) acc[j] += f * v[j];
            }
          }
        }
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
template <typename T, int CPT, int MAXLP>
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
          if (h > -1.f && w > -1.f && h < (float)H && w < (float)W) {
            Corner c = corners(h, w, H, W);
            const float hh = 1.f - c.lh, hw = 1.f - c.lw;
            // d(val)/dh and d(val)/dw corner coefficients (cuh:122-158)
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
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                  val[j] += c.wt[k] * v[j];
                  ghw[j] += dh[k] * v[j];
                  gww[j] += dw[k] * v[j];
                  if (live) atomic_add_f32(gvalue + base + (long)c.off[k] * qstride + j, c.wt[k] * tg[j] * a);
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
                                                            int P, int LPP, int q_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float gacc[];   // [S][D]
  const int tid = threadIdx.x;
  const int b = blockIdx.y / M, m = blockIdx.y % M;
  const int q0 = blockIdx.x * q_per_chunk;
  const int q1 = min(Lq, q0 + q_per_chunk);
  for (int i = tid; i < S * D; i += 1024) gacc[i] = 0.f;
  __syncthreads();
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
                float* ldst = gacc + (ls + c.off[k]) * D + c0;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                  val[j] += c.wt[k] * v[j];
                  ghw[j] += dh[k] * v[j];
                  gww[j] += dw[k] * v[j];
                  atomicAdd(ldst + j, c.wt[k] * tg[j] * a);
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
  for (int i = tid; i < S * D; i += 1024) {
    const float v = gacc[i];
    if (v != 0.f) {
      const int sidx = i / D, c = i - sidx * D;
      atomic_add_f32(gvalue + ((long)b * S + sidx) * qstride + (long)m * D + c, v);
    }
  }
}

int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <typename T>
int fwd_dispatch(const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn, void* out,
                 int N, int S, int M, int D, int L, int Lq, int P, hipStream_t st) {
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

template <typename T, int CPT>
int bwd_launch(const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn, const void* gout,
               float* gv, float* gl, float* ga, int N, int S, int M, int D, int L, int Lq, int P, hipStream_t st) {
  const int chunks = (D + CPT - 1) / CPT;
  int LPP = next_pow2(chunks);
  if (LPP > 64) LPP = 64;
  const long npairs = (long)N * Lq * M;
  const int gpw = 256 / LPP;
  long blocks = (npairs + gpw - 1) / gpw;
  if (blocks > 65535 * 8) blocks = 65535 * 8;
  const int LP = L * P;
  const size_t plane = (size_t)S * D * sizeof(float);
  static const bool no_lds = getenv("DU_MSDA_NO_LDS") != nullptr;   // debugging aid: force the global-atomics kernel
  if (!no_lds && plane <= 144 * 1024 && LP <= 8 && LPP <= 64) {
    // enough workgroups to fill 256 CUs; each chunk costs one S*D flush
    int nchunk = (int)((512 + (long)N * M - 1) / ((long)N * M));
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 16) nchunk = 16;
    int qpc = (Lq + nchunk - 1) / nchunk;
    const int gq = 1024 / LPP;
    qpc = ((qpc + gq - 1) / gq) * gq;
    nchunk = (Lq + qpc - 1) / qpc;
    dim3 grid(nchunk, N * M);
#define MSDA_BWD_LDS(MAXLP) do { \
      auto kfn = msda_bwd_lds_kernel<T, CPT, MAXLP>; \
      if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plane) != hipSuccess) return DU_ERR_LAUNCH; \
      hipLaunchKernelGGL(kfn, grid, dim3(1024), plane, st, (const T*)value, shapes, lsi, loc, attn, (const T*)gout, gv, gl, ga, N, S, M, D, L, Lq, P, LPP, qpc); \
    } while (0)
    if (LP <= 4) MSDA_BWD_LDS(4); else MSDA_BWD_LDS(8);
#undef MSDA_BWD_LDS
    return du_check_launch();
  }
#define MSDA_BWD(MAXLP) hipLaunchKernelGGL((msda_bwd_kernel<T, CPT, MAXLP>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)value, shapes, lsi, loc, attn, (const T*)gout, gv, gl, ga, N, S, M, D, L, Lq, P, LPP, npairs)
  if (LP <= 4) MSDA_BWD(4); else if (LP <= 8) MSDA_BWD(8); else if (LP <= 16) MSDA_BWD(16); else return DU_ERR_UNSUPPORTED;
#undef MSDA_BWD
  return du_check_launch();
}

}  // namespace

extern "C" int du_msda_forward(int dtype, const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                               const float* attn, void* out, int N, int S, int M, int D, int L, int Lq, int P, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !out || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0)
    return DU_ERR_BAD_ARG;
  if (dtype == DU_F32) return fwd_dispatch<float>(value, shapes, lsi, loc, attn, out, N, S, M, D, L, Lq, P, st);
  if (dtype == DU_BF16) return fwd_dispatch<bf16_t>(value, shapes, lsi, loc, attn, out, N, S, M, D, L, Lq, P, st);
  return DU_ERR_BAD_ARG;
}

extern "C" int du_msda_backward(int dtype, const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                                const float* attn, const void* gout, float* gv, float* gl, float* ga, int N, int S, int M, int D,
                                int L, int Lq, int P, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!value || !shapes || !lsi || !loc || !attn || !gout || !gv || !gl || !ga || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 ||
      Lq <= 0 || P <= 0)
    return DU_ERR_BAD_ARG;
  if (dtype == DU_F32) {
    if (D % 4 == 0) return bwd_launch<float, 4>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, st);
    return bwd_launch<float, 1>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, st);
  }
  if (dtype == DU_BF16) {
    if (D % 4 == 0) return bwd_launch<bf16_t, 4>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, st);
    return bwd_launch<bf16_t, 1>(value, shapes, lsi, loc, attn, gout, gv, gl, ga, N, S, M, D, L, Lq, P, st);
  }
  return DU_ERR_BAD_ARG;
}
