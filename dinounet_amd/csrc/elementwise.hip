// HBM-bound NHWC helpers for gfx950: depthwise 3x3, max-pool 3x3/s2, bilinear upsample+add, layout conversion,
// MSDeformAttn sampling-location / softmax preparation.  All use 16-byte channel vectors per thread.
#include <stdlib.h>
#include "common.h"

namespace {

__host__ int grid_1d(long total) {
  long g = (total + 255) / 256;
  if (g < 1) g = 1;
  if (g > 65535L * 8) g = 65535L * 8;
  return (int)g;
}

#define GRID_STRIDE(i, total) for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (total); i += (long)gridDim.x * 256)

// ---------------- depthwise 3x3, stride 1, pad 1 (dinov3_adapter.py:94-109, dinounet_training.py:235) ----------------
// Thread = (pixel lane, 8/4-channel vector); a thread keeps the 9 x VEC filter taps of its channels in registers and walks
// pixels (grid-stride over pixel lanes), so the inner loop is 9 vector loads + 9*VEC FMAs + 1-2 vector stores per pixel.
// FLIP = true evaluates the data gradient: correlation of dy with the spatially flipped filter.
// PYR = true: the token pyramid of ConvFFN's DWConv (dinov3_adapter.py:99-109) in ONE launch: every image is N = 21 n tokens
// (n = H*W/4) holding a (2H x 2W), an (H x W) and an (H/2 x W/2) grid back to back; a pixel index decodes to (image, grid, y, x).
struct PyrPix { int b, s0, Hs, Ws, yo, xo; };
__device__ __forceinline__ PyrPix pyr_decode(long p, int H, int W) {
  const int n = (H * W) >> 2, N = 21 * n;
  PyrPix q;
  q.b = (int)(p / N);
  int t = (int)(p - (long)q.b * N);
  if (t < 16 * n) { q.s0 = 0; q.Hs = 2 * H; q.Ws = 2 * W; }
  else if (t < 20 * n) { q.s0 = 16 * n; q.Hs = H; q.Ws = W; t -= 16 * n; }
  else { q.s0 = 20 * n; q.Hs = H >> 1; q.Ws = W >> 1; t -= 20 * n; }
  q.yo = t / q.Ws; q.xo = t - q.yo * q.Ws;
  return q;
}

template <typename T, bool FLIP, bool PYR>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, long ldx, long xbs, const float* __restrict__ w,
                                                     const float* __restrict__ bias, T* __restrict__ y, long ldy, long ybs,
                                                     T* __restrict__ z, int B, int H, int W, int C, int act, int lanes_per_block) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V;
  const int cvb = min(cvn, 256);
  const int np = 256 / cvb;
  const int tp = threadIdx.x / cvb, tcv = threadIdx.x % cvb;
  if (tp >= np) return;
  const long npix = PYR ? (long)B * 21 * ((H * W) >> 2) : (long)B * H * W;
  for (int cv = tcv; cv < cvn; cv += cvb) {
    const int c0 = cv * V;
    float wt[9][V], bs[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
      bs[j] = bias ? bias[c0 + j] : 0.f;
#pragma unroll
      for (int t = 0; t < 9; t++) wt[t][j] = w[(c0 + j) * 9 + (FLIP ? 8 - t : t)];
    }
    for (long p = (long)blockIdx.x * np + tp; p < npix; p += (long)gridDim.x * np) {
      int xo, yo, b, Hs = H, Ws = W, s0 = 0;
      if constexpr (PYR) {
        const PyrPix q = pyr_decode(p, H, W);
        xo = q.xo; yo = q.yo; b = q.b; Hs = q.Hs; Ws = q.Ws; s0 = q.s0;
      } else {
        xo = (int)(p % W); long t2 = p / W;
        yo = (int)(t2 % H); b = (int)(t2 / H);
      }
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; j++) acc[j] = bs[j];
#pragma unroll
      for (int dy = 0; dy < 3; dy++) {
        const int yi = yo + dy - 1;
        if (yi < 0 || yi >= Hs) continue;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
          const int xi = xo + dx - 1;
          if (xi < 0 || xi >= Ws) continue;
          Vec16<T> v = as_vec<T>(*(const uint4*)(x + (long)b * xbs + (s0 + (long)yi * Ws + xi) * ldx + c0));
#pragma unroll
          for (int j = 0; j < V; j++) acc[j] += to_f32(v.v[j]) * wt[dy * 3 + dx][j];
        }
      }
      const long off = (long)b * ybs + (s0 + (long)yo * Ws + xo) * ldy + c0;
      Vec16<T> o;
      if (z) {
#pragma unroll
        for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(acc[j]);
        *(uint4*)(z + off) = as_u4(o);
      }
#pragma unroll
      for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(apply_act(acc[j], act));
      *(uint4*)(y + off) = as_u4(o);
    }
  }
}

// Row form of the kernel above (round 3): a thread computes XT = 4 CONSECUTIVE outputs of one image row for its channel vector and slides
// the 3-column window over the XT + 2 input columns it loads per filter row: 18 sixteen-byte loads per 4 outputs instead of 36, one pixel
// decode (integer divisions) per 4 outputs instead of 4.  The per-pixel form ran at 1.1 TB/s on the (8, 5376, 256) token pyramid -- load /
// VALU issue bound, not HBM.  Every grid row width must be a multiple of XT (the caller checks).
template <typename T, bool FLIP, bool PYR>
__global__ __launch_bounds__(256) void dwconv_row4_kernel(const T* __restrict__ x, long ldx, long xbs, const float* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ y, long ldy, long ybs,
                                                          T* __restrict__ z, int B, int H, int W, int C, int act) {
  constexpr int V = Elem<T>::VEC;
  constexpr int XT = 4;
  const int cvn = C / V;
  const int cvb = min(cvn, 256);
  const int np = 256 / cvb;
  const int tp = threadIdx.x / cvb, tcv = threadIdx.x % cvb;
  if (tp >= np) return;
  const long ngroups = (PYR ? (long)B * 21 * ((H * W) >> 2) : (long)B * H * W) / XT;
  for (int cv = tcv; cv < cvn; cv += cvb) {
    const int c0 = cv * V;
    float wt[9][V], bs[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
      bs[j] = bias ? bias[c0 + j] : 0.f;
#pragma unroll
      for (int t = 0; t < 9; t++) wt[t][j] = w[(c0 + j) * 9 + (FLIP ? 8 - t : t)];
    }
    for (long g = (long)blockIdx.x * np + tp; g < ngroups; g += (long)gridDim.x * np) {
      const long p = g * XT;
      int xo, yo, b, Hs = H, Ws = W, s0 = 0;
      if constexpr (PYR) {
        const PyrPix q = pyr_decode(p, H, W);
        xo = q.xo; yo = q.yo; b = q.b; Hs = q.Hs; Ws = q.Ws; s0 = q.s0;
      } else {
        xo = (int)(p % W); long t2 = p / W;
        yo = (int)(t2 % H); b = (int)(t2 / H);
      }
      float acc[XT][V];
#pragma unroll
      for (int t = 0; t < XT; t++)
#pragma unroll
        for (int j = 0; j < V; j++) acc[t][j] = bs[j];
#pragma unroll
      for (int dy = 0; dy < 3; dy++) {
        const int yi = yo + dy - 1;
        if (yi < 0 || yi >= Hs) continue;
        const T* row = x + (long)b * xbs + (s0 + (long)yi * Ws) * ldx + c0;
        uint4 raw[XT + 2];
#pragma unroll
        for (int c = 0; c < XT + 2; c++) {
          const int xi = xo - 1 + c;
          raw[c] = (xi >= 0 && xi < Ws) ? *(const uint4*)(row + (long)xi * ldx) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < XT + 2; c++) {
          const Vec16<T> v = as_vec<T>(raw[c]);
          float f[V];
#pragma unroll
          for (int j = 0; j < V; j++) f[j] = to_f32(v.v[j]);
#pragma unroll
          for (int dx = 0; dx < 3; dx++) {
            const int t = c - dx;                      // column c is tap dx of output t = c - dx
            if (t >= 0 && t < XT) {
#pragma unroll
              for (int j = 0; j < V; j++) acc[t][j] += f[j] * wt[dy * 3 + dx][j];
            }
          }
        }
      }
      const long off = (long)b * ybs + (s0 + (long)yo * Ws + xo) * ldy + c0;
#pragma unroll
      for (int t = 0; t < XT; t++) {
        Vec16<T> o;
        if (z) {
#pragma unroll
          for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(acc[t][j]);
          *(uint4*)(z + off + (long)t * ldy) = as_u4(o);
        }
#pragma unroll
        for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(apply_act(acc[t][j], act));
        *(uint4*)(y + off + (long)t * ldy) = as_u4(o);
      }
    }
  }
}

// dw[c][tap] += sum_pix x[pix+tap][c] * dy[pix][c]; db[c] += sum dy.  Block = strip of pixels, thread = (pixel lane, cvec);
// per-thread register partials are reduced across the block's pixel lanes through LDS in two passes of five taps (every thread
// takes part in the column sums), so each block issues one store / atomic per (channel, tap).
// ROW4: the pixel loop walks groups of 4 consecutive pixels of one image row (strip % 4 == 0, every row width % 4 == 0): 4 dy loads + 18 x
// loads per group instead of 4 + 36, one pixel decode per group.
template <typename T, bool PYR, bool ROW4 = false>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, long ldx, long xbs,
                                                                const T* __restrict__ dy, long lddy, long dybs,
                                                                float* __restrict__ dw, float* __restrict__ db, int B, int H,
                                                                int W, int C, int strip, float* __restrict__ part) {
  constexpr int V = Elem<T>::VEC;
  __shared__ float red[5][256 * V];
  const int cvn = C / V;
  const int cvb = min(cvn, 256);
  const int np = 256 / cvb;
  const int tp = threadIdx.x / cvb, tcv = threadIdx.x % cvb;
  const int npix = PYR ? B * 21 * ((H * W) >> 2) : B * H * W;
  const int p0 = blockIdx.x * strip, p1 = min(npix, p0 + strip);
  for (int cv0 = 0; cv0 < cvn; cv0 += cvb) {
    const int cv = cv0 + tcv;
    const bool active = (cv < cvn) && (tp < np);
    const int c0 = cv * V;
    float aw[10][V];
#pragma unroll
    for (int k = 0; k < 10; k++)
#pragma unroll
      for (int j = 0; j < V; j++) aw[k][j] = 0.f;
    if constexpr (ROW4) {
      if (active) {
        for (int p = p0 + 4 * tp; p < p1; p += 4 * np) {
          int xo, yo, b, Hs = H, Ws = W, s0 = 0;
          if constexpr (PYR) {
            const PyrPix q = pyr_decode(p, H, W);
            xo = q.xo; yo = q.yo; b = q.b; Hs = q.Hs; Ws = q.Ws; s0 = q.s0;
          } else {
            xo = p % W; const int t = p / W; yo = t % H; b = t / H;
          }
          float gf[4][V];
          const T* gp = dy + (long)b * dybs + (s0 + (long)yo * Ws + xo) * lddy + c0;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const Vec16<T> g = as_vec<T>(*(const uint4*)(gp + (long)t * lddy));
#pragma unroll
            for (int j = 0; j < V; j++) { gf[t][j] = to_f32(g.v[j]); aw[9][j] += gf[t][j]; }
          }
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            const int yi = yo + ky - 1;
            if (yi < 0 || yi >= Hs) continue;
            const T* row = x + (long)b * xbs + (s0 + (long)yi * Ws) * ldx + c0;
            uint4 raw[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
              const int xi = xo - 1 + c;
              raw[c] = (xi >= 0 && xi < Ws) ? *(const uint4*)(row + (long)xi * ldx) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 6; c++) {
              const Vec16<T> v = as_vec<T>(raw[c]);
              float f[V];
#pragma unroll
              for (int j = 0; j < V; j++) f[j] = to_f32(v.v[j]);
#pragma unroll
              for (int kx = 0; kx < 3; kx++) {
                const int t = c - kx;                  // input column c is tap kx of output pixel t
                if (t >= 0 && t < 4) {
#pragma unroll
                  for (int j = 0; j < V; j++) aw[ky * 3 + kx][j] += f[j] * gf[t][j];
                }
              }
            }
          }
        }
      }
    } else
    if (active) {
      int p = p0 + tp;
      int xo = 0, yo = 0, b = 0, Hs = H, Ws = W, s0 = 0;
      if constexpr (!PYR) { xo = p % W; const int t = p / W; yo = t % H; b = t / H; }
      for (; p < p1; p += np) {
        if constexpr (PYR) {
          const PyrPix q = pyr_decode(p, H, W);
          xo = q.xo; yo = q.yo; b = q.b; Hs = q.Hs; Ws = q.Ws; s0 = q.s0;
        }
        Vec16<T> g = as_vec<T>(*(const uint4*)(dy + (long)b * dybs + (s0 + (long)yo * Ws + xo) * lddy + c0));
        float gf[V];
#pragma unroll
        for (int j = 0; j < V; j++) { gf[j] = to_f32(g.v[j]); aw[9][j] += gf[j]; }
        const T* xb = x + (long)b * xbs + c0;
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const int yi = yo + ky - 1;
          if (yi < 0 || yi >= Hs) continue;
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const int xi = xo + kx - 1;
            if (xi < 0 || xi >= Ws) continue;
            Vec16<T> v = as_vec<T>(*(const uint4*)(xb + (s0 + (long)yi * Ws + xi) * ldx));
#pragma unroll
            for (int j = 0; j < V; j++) aw[ky * 3 + kx][j] += to_f32(v.v[j]) * gf[j];
          }
        }
        if constexpr (!PYR) {
          xo += np;                                 // advance the pixel coordinate without divisions
          while (xo >= W) { xo -= W; if (++yo == H) { yo = 0; b++; } }
        }
      }
    }
    const int ncol = cvb * V;                        // channel columns this pass covers
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
      for (int k = 0; k < 5; k++)
#pragma unroll
        for (int j = 0; j < V; j++) red[k][(tp * cvb + tcv) * V + j] = aw[half * 5 + k][j];
      __syncthreads();
      for (int o = threadIdx.x; o < 5 * ncol; o += 256) {
        const int k5 = o / ncol, ci = o - k5 * ncol;
        const int c = cv0 * V + ci;
        if (c >= C) continue;
        float sacc = 0.f;
        for (int q = 0; q < np; q++) sacc += red[k5][q * ncol + ci];
        const int k = half * 5 + k5;
        if (part) part[((long)blockIdx.x * 10 + k) * C + c] = sacc;      // [strip][tap][C]; dwconv_wgrad_finalize_kernel sums the strips
        else if (k < 9) atomic_add_f32(dw + c * 9 + k, sacc);
        else if (db) atomic_add_f32(db + c, sacc);
      }
      __syncthreads();
    }
  }
}

// dw[c][k] (+)= sum_blocks part[blk][k][c] (k < 9), db[c] (+)= part[blk][9][c]; `accum` adds to the existing values (later image
// segments of the same depthwise kernel, dinov3_adapter.py:99-109)
__global__ __launch_bounds__(256) void dwconv_wgrad_finalize_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    float* __restrict__ db, int blocks, int C, int accum) {
  // workgroup = 32 columns (of the C*10 partial columns) x 8 block lanes
  __shared__ float red[8][33];
  const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col;
  float acc = 0.f;
  if (i < C * 10)
    for (int b = sl; b < blocks; b += 8) acc += part[(long)b * C * 10 + i];
  red[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && i < C * 10) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) t += red[q][col];
    const int k = i / C, c = i - k * C;
    if (k < 9) dw[c * 9 + k] = accum ? dw[c * 9 + k] + t : t;
    else if (db) db[c] = accum ? db[c] + t : t;
  }
}

// ---------------- max-pool 3x3 stride 2 pad 1 (dinov3_adapter.py:250); idx = winning tap (first max in scan order) -----------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int B, int H, int W, int C, int Ho, int Wo, long total) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V;
  GRID_STRIDE(i, total) {
    const int c0 = (int)(i % cvn) * V;
    long t = i / cvn;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float best[V]; int bi[V];
#pragma unroll
    for (int j = 0; j < V; j++) { best[j] = -INFINITY; bi[j] = 0; }
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
      const int yi = yo * 2 - 1 + dy;
      if (yi < 0 || yi >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const int xi = xo * 2 - 1 + dx;
        if (xi < 0 || xi >= W) continue;
        Vec16<T> v = as_vec<T>(*(const uint4*)(x + (((long)b * H + yi) * W + xi) * C + c0));
#pragma unroll
        for (int j = 0; j < V; j++) {
          float f = to_f32(v.v[j]);
          if (f > best[j]) { best[j] = f; bi[j] = dy * 3 + dx; }
        }
      }
    }
    const long off = (((long)b * Ho + yo) * Wo + xo) * C + c0;
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(best[j]);
    *(uint4*)(y + off) = as_u4(o);
    if (idx) {
#pragma unroll
      for (int j = 0; j < V; j++) idx[off + j] = (uint8_t)bi[j];
    }
  }
}

// gather form: each input pixel sums dy of the (<= 4) windows that selected it
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dy,
                                                          T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                          long total) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V;
  GRID_STRIDE(i, total) {
    const int c0 = (int)(i % cvn) * V;
    long t = i / cvn;
    const int xi = (int)(t % W); t /= W;
    const int yi = (int)(t % H);
    const int b = (int)(t / H);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) acc[j] = 0.f;
    // windows yo with yo*2-1+dy == yi, dy in 0..2
#pragma unroll
    for (int dy_ = 0; dy_ < 3; dy_++) {
      const int ty = yi + 1 - dy_;
      if (ty < 0 || (ty & 1)) continue;
      const int yo = ty >> 1;
      if (yo >= Ho) continue;
#pragma unroll
      for (int dx_ = 0; dx_ < 3; dx_++) {
        const int tx = xi + 1 - dx_;
        if (tx < 0 || (tx & 1)) continue;
        const int xo = tx >> 1;
        if (xo >= Wo) continue;
        const long off = (((long)b * Ho + yo) * Wo + xo) * C + c0;
        Vec16<T> g = as_vec<T>(*(const uint4*)(dy + off));
#pragma unroll
        for (int j = 0; j < V; j++)
          if (idx[off + j] == dy_ * 3 + dx_) acc[j] += to_f32(g.v[j]);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(acc[j]);
    *(uint4*)(dx + (((long)b * H + yi) * W + xi) * C + c0) = as_u4(o);
  }
}

// ---------------- bilinear (align_corners=False) upsample + add (dinov3_adapter.py:472-476) ----------------
// Thread = V consecutive channels of one output pixel; V = 8 (16-byte bf16 vectors) when C % 8 == 0, else 4.  (The first version read and
// wrote element by element: 1.3 TB/s on the 268 MB c1 level.)
template <typename TS, typename T, int V>
__global__ __launch_bounds__(256) void bilinear_add_kernel(const TS* __restrict__ src, long lds_, const T* __restrict__ base,
                                                           long ldb, T* __restrict__ out, long ldo, int B, int Hs, int Ws, int Ho,
                                                           int Wo, int C, long total) {
  const int cvn = C / V;
  const float sh = (float)Hs / (float)Ho, sw = (float)Ws / (float)Wo;
  auto load = [&](const TS* p, float* v) {
    if constexpr (sizeof(TS) == 2 && V == 8) {
      const bf16x8 t = __builtin_bit_cast(bf16x8, *(const uint4*)p);
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (float)t[j];
    } else if constexpr (sizeof(TS) == 4 && V % 4 == 0) {
#pragma unroll
      for (int j = 0; j < V; j += 4) { const float4 t = *(const float4*)(p + j); v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w; }
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) v[j] = to_f32(p[j]);
    }
  };
  GRID_STRIDE(i, total) {
    const int c0 = (int)(i % cvn) * V;
    long t = i / cvn;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    // PyTorch area_pixel_compute_source_index: max(0, scale*(dst+0.5)-0.5)
    float fy = fmaxf(0.f, sh * (yo + 0.5f) - 0.5f), fx = fmaxf(0.f, sw * (xo + 0.5f) - 0.5f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    float a00[V], a01[V], a10[V], a11[V], bs[V];
    load(src + (((long)b * Hs + y0) * Ws + x0) * lds_ + c0, a00);
    load(src + (((long)b * Hs + y0) * Ws + x1) * lds_ + c0, a01);
    load(src + (((long)b * Hs + y1) * Ws + x0) * lds_ + c0, a10);
    load(src + (((long)b * Hs + y1) * Ws + x1) * lds_ + c0, a11);
    const long po = ((long)b * Ho + yo) * Wo + xo;
    if (base) {
      if constexpr (sizeof(T) == 2 && V == 8) {
        const bf16x8 tb = __builtin_bit_cast(bf16x8, *(const uint4*)(base + po * ldb + c0));
#pragma unroll
        for (int j = 0; j < 8; j++) bs[j] = (float)tb[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; j++) bs[j] = to_f32(base[po * ldb + c0 + j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) bs[j] = 0.f;
    }
    float o[V];
#pragma unroll
    for (int j = 0; j < V; j++) o[j] = bs[j] + hy * (hx * a00[j] + lx * a01[j]) + ly * (hx * a10[j] + lx * a11[j]);
    if constexpr (sizeof(T) == 2 && V == 8) {
      bf16x8 r;
#pragma unroll
      for (int j = 0; j < 8; j++) r[j] = (bf16_t)o[j];
      *(uint4*)(out + po * ldo + c0) = __builtin_bit_cast(uint4, r);
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) out[po * ldo + c0 + j] = from_f32<T>(o[j]);
    }
  }
}

// adjoint of the bilinear resize above (the data gradient of F.interpolate(bilinear, align_corners=False), the tail of
// LearnableUpsampleBlock, dinounet_training.py:262-263), in GATHER form: an input pixel collects, from every output row / column whose
// two source taps include it, the same weights the forward used -- deterministic, no atomics.  For target sizes between 1x and 2x the
// source size (the only case the tail can see) that is at most 4 candidate rows x 4 candidate columns.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_resize_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx, long lddx, int B,
                                                                  int Hs, int Ws, int Ho, int Wo, int C, long total) {
  constexpr int V = 4;
  const int cvn = C / V;
  const float sh = (float)Hs / (float)Ho, sw = (float)Ws / (float)Wo;
  const float ih = (float)Ho / (float)Hs, iw = (float)Wo / (float)Ws;
  GRID_STRIDE(i, total) {
    const int c0 = (int)(i % cvn) * V;
    long t = i / cvn;
    const int xi = (int)(t % Ws); t /= Ws;
    const int yi = (int)(t % Hs);
    const int b = (int)(t / Hs);
    // outputs whose source coordinate lies in (yi - 1, yi + 1): yo in ((yi - 0.5) * ih - 0.5, (yi + 1.5) * ih - 0.5), widened by one
    const int ya = max(0, (int)floorf((yi - 0.5f) * ih - 0.5f) - 1), yb = min(Ho - 1, (int)ceilf((yi + 1.5f) * ih - 0.5f) + 1);
    const int xa = max(0, (int)floorf((xi - 0.5f) * iw - 0.5f) - 1), xb = min(Wo - 1, (int)ceilf((xi + 1.5f) * iw - 0.5f) + 1);
    float acc[V] = {0.f, 0.f, 0.f, 0.f};
    for (int yo = ya; yo <= yb; yo++) {
      const float fy = fmaxf(0.f, sh * (yo + 0.5f) - 0.5f);
      const int y0 = (int)fy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
      const float ly = fy - y0;
      const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int xo = xa; xo <= xb; xo++) {
        const float fx = fmaxf(0.f, sw * (xo + 0.5f) - 0.5f);
        const int x0 = (int)fx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float lx = fx - x0;
        const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
        if (wx == 0.f) continue;
        const T* g = dy + (((long)b * Ho + yo) * Wo + xo) * lddy + c0;
        const float wgt = wy * wx;
#pragma unroll
        for (int j = 0; j < V; j++) acc[j] += wgt * to_f32(g[j]);
      }
    }
    T* d = dx + (((long)b * Hs + yi) * Ws + xi) * lddx + c0;
#pragma unroll
    for (int j = 0; j < V; j++) d[j] = from_f32<T>(acc[j]);
  }
}

// ---------------- casts / layout ----------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
  GRID_STRIDE(i, n) d[i] = from_f32<TD>(to_f32(s[i]));
}

template <typename TD>
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ s, TD* __restrict__ d, int B, int C, int H,
                                                               int W, int Cpad, long total) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    long p = i / Cpad;
    const long hw = (long)H * W;
    const int b = (int)(p / hw);
    const long r = p % hw;
    d[i] = from_f32<TD>(c < C ? s[((long)b * C + c) * hw + r] : 0.f);
  }
}

template <typename TS>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const TS* __restrict__ s, long ld, float* __restrict__ d, int B, int C,
                                                           int H, int W, long total) {
  GRID_STRIDE(i, total) {
    const long hw = (long)H * W;
    const long r = i % hw;
    long t = i / hw;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    d[i] = to_f32(s[((long)b * hw + r) * ld + c]);
  }
}

// rows = (b, py, px), columns = (c, dy, dx): matches Conv2d(k16,s16).weight.flatten(1) (layers/patch_embed.py:61)
template <typename TD>
__global__ __launch_bounds__(256) void patchify16_kernel(const float* __restrict__ s, TD* __restrict__ d, int B, int C, int H,
                                                         int W, long total) {
  const int ph = H / 16, pw = W / 16;
  const int K = C * 256;
  GRID_STRIDE(i, total) {
    const int k = (int)(i % K);
    long r = i / K;
    const int px = (int)(r % pw); r /= pw;
    const int py = (int)(r % ph);
    const int b = (int)(r / ph);
    const int c = k >> 8, dy = (k >> 4) & 15, dx = k & 15;
    d[i] = from_f32<TD>(s[(((long)b * C + c) * H + py * 16 + dy) * W + px * 16 + dx]);
  }
}

// ---------------- MSDeformAttn: sampling locations + softmax of the attention logits (ms_deform_attn.py:188-197) --------------
// raw: (rows, M*P*2 + M*P) fp32/bf16 = [offsets | logits]; ref: (Lq, 2) reference points (x, y) shared by the batch;
// loc (rows, M, P, 2) = ref + off / (W, H); attn (rows, M, P) = softmax over P.   (single level)
template <typename T>
__global__ __launch_bounds__(256) void msda_prep_kernel(const T* __restrict__ raw, long ldr, const float* __restrict__ ref,
                                                        float* __restrict__ loc, float* __restrict__ attn, long rows, int Lq, int M,
                                                        int P, float invW, float invH, long total) {
  // P == 4 (the adapter's n_points) with 16-byte aligned rows: one thread = one (row, head) moved as whole vectors -- 8 offsets + 4 logits
  // in, 8 + 4 floats out (the element-by-element form ran at 1.8 TB/s)
  const bool vec4 = P == 4 && ((ldr * sizeof(T)) % 16 == 0) && ((((uintptr_t)raw) | ((uintptr_t)loc) | ((uintptr_t)attn)) & 15) == 0 &&
                    ((M * 8 * sizeof(T)) % 16 == 0);
  if (vec4) {
    GRID_STRIDE(i, total) {
      const int m = (int)(i % M);
      const long row = i / M;
      const int q = (int)(row % Lq);
      const float rx = ref[q * 2], ry = ref[q * 2 + 1];
      float off[8], lg[4];
      const T* po = raw + row * ldr + (long)m * 8;
      const T* pl = raw + row * ldr + (long)M * 8 + (long)m * 4;
      if constexpr (sizeof(T) == 2) {
        const bf16x8 t = __builtin_bit_cast(bf16x8, *(const uint4*)po);
        const bf16x4 u = __builtin_bit_cast(bf16x4, *(const uint2*)pl);
#pragma unroll
        for (int e = 0; e < 8; e++) off[e] = (float)t[e];
#pragma unroll
        for (int e = 0; e < 4; e++) lg[e] = (float)u[e];
      } else {
        const float4 t0 = *(const float4*)po, t1 = *(const float4*)(po + 4), u = *(const float4*)pl;
        off[0] = t0.x; off[1] = t0.y; off[2] = t0.z; off[3] = t0.w; off[4] = t1.x; off[5] = t1.y; off[6] = t1.z; off[7] = t1.w;
        lg[0] = u.x; lg[1] = u.y; lg[2] = u.z; lg[3] = u.w;
      }
      const float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
      float ex[4], sm = 0.f;
#pragma unroll
      for (int e = 0; e < 4; e++) { ex[e] = expf(lg[e] - mx); sm += ex[e]; }
      const float inv = 1.f / sm;
      float* lo = loc + (row * M + m) * 8;
      *(float4*)lo = make_float4(rx + off[0] * invW, ry + off[1] * invH, rx + off[2] * invW, ry + off[3] * invH);
      *(float4*)(lo + 4) = make_float4(rx + off[4] * invW, ry + off[5] * invH, rx + off[6] * invW, ry + off[7] * invH);
      *(float4*)(attn + (row * M + m) * 4) = make_float4(ex[0] * inv, ex[1] * inv, ex[2] * inv, ex[3] * inv);
    }
    return;
  }
  GRID_STRIDE(i, total) {
    const int m = (int)(i % M);
    const long row = i / M;
    const int q = (int)(row % Lq);
    const float rx = ref[q * 2], ry = ref[q * 2 + 1];
    const T* po = raw + row * ldr + (long)m * P * 2;
    const T* pl = raw + row * ldr + (long)M * P * 2 + (long)m * P;
    float* lo = loc + (row * M + m) * (long)P * 2;
    float* ao = attn + (row * M + m) * (long)P;
    float mx = -1e30f;
    for (int p = 0; p < P; p++) mx = fmaxf(mx, to_f32(pl[p]));
    float s = 0.f;
    for (int p = 0; p < P; p++) s += expf(to_f32(pl[p]) - mx);
    const float inv = 1.f / s;
    for (int p = 0; p < P; p++) {
      lo[p * 2] = rx + to_f32(po[p * 2]) * invW;
      lo[p * 2 + 1] = ry + to_f32(po[p * 2 + 1]) * invH;
      ao[p] = expf(to_f32(pl[p]) - mx) * inv;
    }
  }
}

// backward of the above: d_raw[off] = d_loc * (1/W, 1/H); d_raw[logit] = p * (g - sum_p p g)
template <typename T>
__global__ __launch_bounds__(256) void msda_prep_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ gloc,
                                                            const float* __restrict__ gattn, T* __restrict__ graw, long ldr,
                                                            long rows, int M, int P, float invW, float invH, long total) {
  const bool vec4 = P == 4 && ((ldr * sizeof(T)) % 16 == 0) && ((M * 8 * sizeof(T)) % 16 == 0) &&
                    ((((uintptr_t)graw) | ((uintptr_t)gloc) | ((uintptr_t)gattn) | ((uintptr_t)attn)) & 15) == 0;
  if (vec4) {
    GRID_STRIDE(i, total) {
      const int m = (int)(i % M);
      const long row = i / M;
      const long o = row * M + m;
      const float4 g0 = *(const float4*)(gloc + o * 8), g1 = *(const float4*)(gloc + o * 8 + 4);
      const float4 ga = *(const float4*)(gattn + o * 4), pa = *(const float4*)(attn + o * 4);
      const float dot = pa.x * ga.x + pa.y * ga.y + pa.z * ga.z + pa.w * ga.w;
      const float go[8] = {g0.x * invW, g0.y * invH, g0.z * invW, g0.w * invH, g1.x * invW, g1.y * invH, g1.z * invW, g1.w * invH};
      const float gg[4] = {pa.x * (ga.x - dot), pa.y * (ga.y - dot), pa.z * (ga.z - dot), pa.w * (ga.w - dot)};
      T* po = graw + row * ldr + (long)m * 8;
      T* pl = graw + row * ldr + (long)M * 8 + (long)m * 4;
      if constexpr (sizeof(T) == 2) {
        bf16x8 t; bf16x4 u;
#pragma unroll
        for (int e = 0; e < 8; e++) t[e] = (bf16_t)go[e];
#pragma unroll
        for (int e = 0; e < 4; e++) u[e] = (bf16_t)gg[e];
        *(uint4*)po = __builtin_bit_cast(uint4, t);
        *(uint2*)pl = __builtin_bit_cast(uint2, u);
      } else {
        *(float4*)po = make_float4(go[0], go[1], go[2], go[3]);
        *(float4*)(po + 4) = make_float4(go[4], go[5], go[6], go[7]);
        *(float4*)pl = make_float4(gg[0], gg[1], gg[2], gg[3]);
      }
    }
    return;
  }
  GRID_STRIDE(i, total) {
    const int m = (int)(i % M);
    const long row = i / M;
    const float* gl = gloc + (row * M + m) * (long)P * 2;
    const float* ga = gattn + (row * M + m) * (long)P;
    const float* pa = attn + (row * M + m) * (long)P;
    T* go = graw + row * ldr + (long)m * P * 2;
    T* gg = graw + row * ldr + (long)M * P * 2 + (long)m * P;
    float dot = 0.f;
    for (int p = 0; p < P; p++) dot += pa[p] * ga[p];
    for (int p = 0; p < P; p++) {
      go[p * 2] = from_f32<T>(gl[p * 2] * invW);
      go[p * 2 + 1] = from_f32<T>(gl[p * 2 + 1] * invH);
      gg[p] = from_f32<T>(pa[p] * (ga[p] - dot));
    }
  }
}

// dz = dy * act'(z)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dy, T* __restrict__ dz, long n,
                                                      int act) {
  constexpr int V = Elem<T>::VEC;
  if ((n % V) == 0 && ((((uintptr_t)z) | ((uintptr_t)dy) | ((uintptr_t)dz)) & 15) == 0) {      // 16-byte vectors
    GRID_STRIDE(i, n / V) {
      const Vec16<T> a = as_vec<T>(*(const uint4*)(z + i * V)), g = as_vec<T>(*(const uint4*)(dy + i * V));
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(to_f32(g.v[j]) * act_grad(to_f32(a.v[j]), act));
      *(uint4*)(dz + i * V) = as_u4(o);
    }
    return;
  }
  GRID_STRIDE(i, n) dz[i] = from_f32<T>(to_f32(dy[i]) * act_grad(to_f32(z[i]), act));
}


// ---------------- squeeze-excitation (dinounet_training.py:210-225) ----------------
// gate[b][c] = sigmoid(W2 relu(W1 pooled_b + b1) + b2), pooled = channel sums / P.  One workgroup per sample.
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float* __restrict__ sums /*(B,C,2)*/, float invP,
                                                          const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ b2,
                                                          float* __restrict__ hidden, float* __restrict__ gate, int C, int R) {
  extern __shared__ float sm[];   // pooled[C] | h[R]
  float* pooled = sm;
  float* h = sm + C;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) pooled[c] = sums[((long)b * C + c) * 2] * invP;
  __syncthreads();
  for (int r = tid; r < R; r += 256) {
    float acc = b1[r];
    for (int c = 0; c < C; c++) acc += W1[r * C + c] * pooled[c];
    acc = acc > 0.f ? acc : 0.f;
    h[r] = acc;
    hidden[(long)b * R + r] = acc;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc = b2[c];
    for (int r = 0; r < R; r++) acc += W2[c * R + r] * h[r];
    gate[(long)b * C + c] = 1.f / (1.f + __expf(-acc));
  }
}

// y = x * gate[b][c] (+ shortcut)
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gate,
                                                       const T* __restrict__ sc, long ldsc, T* __restrict__ y, long ldy, long P, int C,
                                                       long total) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V;
  GRID_STRIDE(i, total) {
    const int c0 = (int)(i % cvn) * V;
    const long pix = i / cvn;
    const int b = (int)(pix / P);
    Vec16<T> t = as_vec<T>(*(const uint4*)(x + pix * ldx + c0));
    Vec16<T> o;
    if (sc) {
      Vec16<T> s2 = as_vec<T>(*(const uint4*)(sc + pix * ldsc + c0));
#pragma unroll
      for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(to_f32(t.v[j]) * gate[(long)b * C + c0 + j] + to_f32(s2.v[j]));
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(to_f32(t.v[j]) * gate[(long)b * C + c0 + j]);
    }
    *(uint4*)(y + pix * ldy + c0) = as_u4(o);
  }
}

// gate MLP backward.  One 1024-thread workgroup holds a chunk of samples in LDS and every phase is parallel over (sample, unit);
// the parameter gradients are summed over the samples in a fixed order (deterministic, no atomics).
// dsum (B,C,2): [..,0] = sum_pix dy * x.   Outputs: dpool (B,C) = d loss / d x[b,p,c] through the pooling path (already / P);
// dW1 (R,C), db1 (R), dW2 (C,R), db2 (C) -- written (not accumulated).
__global__ __launch_bounds__(1024) void se_gate_bwd_kernel(const float* __restrict__ dsum, const float* __restrict__ sums, float invP,
                                                           const float* __restrict__ gate, const float* __restrict__ hidden,
                                                           const float* __restrict__ W1, const float* __restrict__ W2,
                                                           float* __restrict__ dpool, float* __restrict__ dW1, float* __restrict__ db1,
                                                           float* __restrict__ dW2, float* __restrict__ db2, int B, int C, int R, int Bc) {
  extern __shared__ float sm[];   // dgp[Bc][C] | pooled[Bc][C] | h[Bc][R] | dhp[Bc][R]
  float* dgp = sm;
  float* pooled = dgp + Bc * C;
  float* h = pooled + Bc * C;
  float* dhp = h + Bc * R;
  const int tid = threadIdx.x;
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int nb = min(Bc, B - b0);
    for (int i = tid; i < nb * C; i += 1024) {
      const long gi = (long)b0 * C + i;
      const float g = gate[gi];
      dgp[i] = dsum[gi * 2] * g * (1.f - g);
      pooled[i] = sums[gi * 2] * invP;
    }
    for (int i = tid; i < nb * R; i += 1024) h[i] = hidden[(long)b0 * R + i];
    __syncthreads();
    for (int i = tid; i < nb * R; i += 1024) {            // dhp[b][r] = relu'(h) * sum_c W2[c][r] dgp[b][c]
      const int bl = i / R, r = i - bl * R;
      float acc = 0.f;
      for (int c = 0; c < C; c++) acc += W2[c * R + r] * dgp[bl * C + c];
      dhp[i] = h[i] > 0.f ? acc : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < C * R; i += 1024) {
      float a2 = 0.f, a1 = 0.f;
      { const int c = i / R, r = i - c * R; for (int bl = 0; bl < nb; bl++) a2 += dgp[bl * C + c] * h[bl * R + r]; }        // W2 is (C,R)
      { const int r = i / C, c = i - r * C; for (int bl = 0; bl < nb; bl++) a1 += dhp[bl * R + r] * pooled[bl * C + c]; }   // W1 is (R,C)
      dW2[i] = b0 ? dW2[i] + a2 : a2;
      dW1[i] = b0 ? dW1[i] + a1 : a1;
    }
    for (int c = tid; c < C; c += 1024) {
      float a = 0.f;
      for (int bl = 0; bl < nb; bl++) a += dgp[bl * C + c];
      db2[c] = b0 ? db2[c] + a : a;
    }
    for (int r = tid; r < R; r += 1024) {
      float a = 0.f;
      for (int bl = 0; bl < nb; bl++) a += dhp[bl * R + r];
      db1[r] = b0 ? db1[r] + a : a;
    }
    for (int i = tid; i < nb * C; i += 1024) {            // dpool[b][c] = invP * sum_r W1[r][c] dhp[b][r]
      const int bl = i / C, c = i - bl * C;
      float acc = 0.f;
      for (int r = 0; r < R; r++) acc += W1[r * C + c] * dhp[bl * R + r];
      dpool[(long)b0 * C + i] = acc * invP;
    }
    __syncthreads();
  }
}

// dx = dy * gate[b][c] + dpool[b][c]
template <typename T>
__global__ __launch_bounds__(256) void se_scale_bwd_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ gate,
                                                           const float* __restrict__ dpool, T* __restrict__ dx, long lddx, long P,
                                                           int C, long total) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V;
  GRID_STRIDE(i, total) {
    const int c0 = (int)(i % cvn) * V;
    const long pix = i / cvn;
    const int b = (int)(pix / P);
    Vec16<T> t = as_vec<T>(*(const uint4*)(dy + pix * lddy + c0));
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(to_f32(t.v[j]) * gate[(long)b * C + c0 + j] + dpool[(long)b * C + c0 + j]);
    *(uint4*)(dx + pix * lddx + c0) = as_u4(o);
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  if ((dtype) == DU_BF16) { CALL_BF16; } else if ((dtype) == DU_F32) { CALL_F32; } else return DU_ERR_BAD_ARG;

static inline int dwconv_grid(long npix, int C, int v) {
  const int cvb = (C / v) < 256 ? (C / v) : 256;
  const int np = 256 / cvb;
  long blocks = (npix + np - 1) / np;
  if (blocks > 4096) blocks = 4096;          // >= 4 pixels per lane once the tensor is large: amortises the 9 x VEC tap loads
  return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int du_dwconv3x3_fwd(int dtype, const void* x, int64_t ldx, int64_t xbs, const float* w, const float* bias, void* y,
                                int64_t ldy, int64_t ybs, void* z, int B, int H, int W, int C, int act, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C % v || ldx % v || ldy % v || xbs % v || ybs % v) return DU_ERR_BAD_ARG;
  const int grid = dwconv_grid((long)B * H * W, C, v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dwconv_kernel<bf16_t, false, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, ldx, xbs, w, bias, (bf16_t*)y, ldy, ybs, (bf16_t*)z, B, H, W, C, act, 0),
             hipLaunchKernelGGL((dwconv_kernel<float, false, false>), dim3(grid), dim3(256), 0, st, (const float*)x, ldx, xbs, w, bias, (float*)y, ldy, ybs, (float*)z, B, H, W, C, act, 0));
  return du_check_launch();
}

extern "C" int du_dwconv3x3_bwd_data(int dtype, const void* dy, int64_t lddy, int64_t dybs, const float* w, void* dx, int64_t lddx,
                                     int64_t dxbs, int B, int H, int W, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!dy || !w || !dx || B <= 0 || C % v || lddy % v || lddx % v || dybs % v || dxbs % v) return DU_ERR_BAD_ARG;
  const int grid = dwconv_grid((long)B * H * W, C, v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dwconv_kernel<bf16_t, true, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, lddy, dybs, w, (const float*)nullptr, (bf16_t*)dx, lddx, dxbs, (bf16_t*)nullptr, B, H, W, C, DU_ACT_NONE, 0),
             hipLaunchKernelGGL((dwconv_kernel<float, true, false>), dim3(grid), dim3(256), 0, st, (const float*)dy, lddy, dybs, w, (const float*)nullptr, (float*)dx, lddx, dxbs, (float*)nullptr, B, H, W, C, DU_ACT_NONE, 0));
  return du_check_launch();
}

// the row forms (4 consecutive pixels of a row per thread) need every grid of the token pyramid -- widths 2W, W, W / 2 -- to be a multiple of 4
static inline bool dwconv_row4_ok(int W) {
  static const bool off = DU_GETENV("DU_DWCONV_NO_ROW4") != nullptr;      // debugging / A-B aid
  return !off && W % 8 == 0;
}

static inline int dwconv_wgrad_strip(long npix, int C, int v) {
  // ~512 workgroups per launch, but at least 4 pixels per pixel lane so the LDS reduction stays amortised
  const int cvb_ = (C / v) < 256 ? (C / v) : 256;
  const int np_ = 256 / cvb_;
  long strip_l = (npix + 511) / 512;        // 512 strips: the partial buffer the finalize kernel re-reads halves (measured vs 1024)
  if (strip_l < (long)np_ * 4) strip_l = (long)np_ * 4;
  strip_l = (strip_l + 3) / 4 * 4;          // whole 4-pixel groups (row form of the kernel)
  return (int)strip_l;
}

extern "C" int64_t du_dwconv_wgrad_ws_elems(int dtype, int B, int H, int W, int C) {
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % v) return 0;
  const long npix = (long)B * H * W;
  const int strip = dwconv_wgrad_strip(npix, C, v);
  return (int64_t)((npix + strip - 1) / strip) * C * 10;
}

extern "C" int du_dwconv3x3_bwd_weight(int dtype, const void* x, int64_t ldx, int64_t xbs, const void* dy, int64_t lddy, int64_t dybs,
                                       float* dw, float* db, int B, int H, int W, int C, float* ws, int64_t ws_elems, int accumulate,
                                       void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!x || !dy || !dw || B <= 0 || C % v || ldx % v || lddy % v || xbs % v || dybs % v) return DU_ERR_BAD_ARG;
  const long npix = (long)B * H * W;
  const int strip = dwconv_wgrad_strip(npix, C, v);
  long blocks = (npix + strip - 1) / strip;
  float* part = (ws && ws_elems >= blocks * C * 10) ? ws : nullptr;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dwconv_bwd_weight_kernel<bf16_t, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, ldx, xbs, (const bf16_t*)dy, lddy, dybs, dw, db, B, H, W, C, strip, part),
             hipLaunchKernelGGL((dwconv_bwd_weight_kernel<float, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, ldx, xbs, (const float*)dy, lddy, dybs, dw, db, B, H, W, C, strip, part));
  if (part)
    hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3((C * 10 + 31) / 32), dim3(256), 0, st, (const float*)part, dw, db, (int)blocks, C, accumulate);
  return du_check_launch();
}

// ---- the same three kernels over the (B, 21 n, C) token pyramid of ConvFFN (dinov3_adapter.py:99-109), one launch each ----
extern "C" int du_dwconv3x3_tokens_fwd(int dtype, const void* x, const float* w, const float* bias, void* y, void* z, int B, int H,
                                       int W, int C, int act, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % v) return DU_ERR_BAD_ARG;
  const long N = 21L * ((H * W) >> 2);
  if (dwconv_row4_ok(W) && dtype == DU_BF16) {       // W / 2 (the narrowest grid) % 4 == 0: 4 outputs per thread (dwconv_row4_kernel)
    const int grid4 = dwconv_grid((long)B * N / 4, C, v);
    hipLaunchKernelGGL((dwconv_row4_kernel<bf16_t, false, true>), dim3(grid4), dim3(256), 0, st, (const bf16_t*)x, (long)C, N * C, w, bias, (bf16_t*)y, (long)C, N * C, (bf16_t*)z, B, H, W, C, act);
    return du_check_launch();
  }
  const int grid = dwconv_grid((long)B * N, C, v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dwconv_kernel<bf16_t, false, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (long)C, N * C, w, bias, (bf16_t*)y, (long)C, N * C, (bf16_t*)z, B, H, W, C, act, 0),
             hipLaunchKernelGGL((dwconv_kernel<float, false, true>), dim3(grid), dim3(256), 0, st, (const float*)x, (long)C, N * C, w, bias, (float*)y, (long)C, N * C, (float*)z, B, H, W, C, act, 0));
  return du_check_launch();
}

extern "C" int du_dwconv3x3_tokens_bwd_data(int dtype, const void* dy, const float* w, void* dx, int B, int H, int W, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % v) return DU_ERR_BAD_ARG;
  const long N = 21L * ((H * W) >> 2);
  if (dwconv_row4_ok(W) && dtype == DU_BF16) {
    const int grid4 = dwconv_grid((long)B * N / 4, C, v);
    hipLaunchKernelGGL((dwconv_row4_kernel<bf16_t, true, true>), dim3(grid4), dim3(256), 0, st, (const bf16_t*)dy, (long)C, N * C, w, (const float*)nullptr, (bf16_t*)dx, (long)C, N * C, (bf16_t*)nullptr, B, H, W, C, DU_ACT_NONE);
    return du_check_launch();
  }
  const int grid = dwconv_grid((long)B * N, C, v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dwconv_kernel<bf16_t, true, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, (long)C, N * C, w, (const float*)nullptr, (bf16_t*)dx, (long)C, N * C, (bf16_t*)nullptr, B, H, W, C, DU_ACT_NONE, 0),
             hipLaunchKernelGGL((dwconv_kernel<float, true, true>), dim3(grid), dim3(256), 0, st, (const float*)dy, (long)C, N * C, w, (const float*)nullptr, (float*)dx, (long)C, N * C, (float*)nullptr, B, H, W, C, DU_ACT_NONE, 0));
  return du_check_launch();
}

// scratch: du_dwconv_wgrad_ws_elems(dtype, B, 21 * H * W / 4, 1, C) floats
extern "C" int du_dwconv3x3_tokens_bwd_weight(int dtype, const void* x, const void* dy, float* dw, float* db, int B, int H, int W, int C,
                                              float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!x || !dy || !dw || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % v) return DU_ERR_BAD_ARG;
  const long N = 21L * ((H * W) >> 2);
  const long npix = (long)B * N;
  const int strip = dwconv_wgrad_strip(npix, C, v);
  long blocks = (npix + strip - 1) / strip;
  float* part = (ws && ws_elems >= blocks * C * 10) ? ws : nullptr;
  if (!part) return DU_ERR_BAD_ARG;          // the partial + finalize form overwrites dw / db: no zero-fill contract on this entry point
  if (dwconv_row4_ok(W) && dtype == DU_BF16 && strip % 4 == 0) {
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<bf16_t, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (long)C, N * C, (const bf16_t*)dy, (long)C, N * C, dw, db, B, H, W, C, strip, part);
    hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3((C * 10 + 31) / 32), dim3(256), 0, st, (const float*)part, dw, db, (int)blocks, C, 0);
    return du_check_launch();
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dwconv_bwd_weight_kernel<bf16_t, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (long)C, N * C, (const bf16_t*)dy, (long)C, N * C, dw, db, B, H, W, C, strip, part),
             hipLaunchKernelGGL((dwconv_bwd_weight_kernel<float, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (long)C, N * C, (const float*)dy, (long)C, N * C, dw, db, B, H, W, C, strip, part));
  hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3((C * 10 + 31) / 32), dim3(256), 0, st, (const float*)part, dw, db, (int)blocks, C, 0);
  return du_check_launch();
}

extern "C" int du_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!x || !y || B <= 0 || C % v) return DU_ERR_BAD_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long total = (long)B * Ho * Wo * (C / v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, idx, B, H, W, C, Ho, Wo, total),
             hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)x, (float*)y, idx, B, H, W, C, Ho, Wo, total));
  return du_check_launch();
}

extern "C" int du_maxpool3x3s2_bwd(int dtype, const uint8_t* idx, const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!idx || !dy || !dx || B <= 0 || C % v) return DU_ERR_BAD_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long total = (long)B * H * W * (C / v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, idx, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C, Ho, Wo, total),
             hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, idx, (const float*)dy, (float*)dx, B, H, W, C, Ho, Wo, total));
  return du_check_launch();
}

extern "C" int du_bilinear_add_fwd(int src_dtype, int dtype, const void* src, int64_t lds_, const void* base, int64_t ldb, void* out,
                                   int64_t ldo, int B, int Hs, int Ws, int Ho, int Wo, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!src || !out || B <= 0 || C % 4) return DU_ERR_BAD_ARG;          // base == NULL: plain resize
  // 16-byte vectors (8 channels per thread) when every row pitch and pointer allows it
  const bool v8 = dtype == DU_BF16 && C % 8 == 0 && lds_ % 8 == 0 && ldo % 8 == 0 && (!base || ldb % 8 == 0) &&
                  ((((uintptr_t)src) | ((uintptr_t)out) | ((uintptr_t)base)) & 15) == 0;
  const int V = v8 ? 8 : 4;
  long total = (long)B * Ho * Wo * (C / V);
  dim3 g(grid_1d(total)), b(256);
#define BL_LAUNCH(TS, T, VV) hipLaunchKernelGGL((bilinear_add_kernel<TS, T, VV>), g, b, 0, st, (const TS*)src, lds_, (const T*)base, ldb, (T*)out, ldo, B, Hs, Ws, Ho, Wo, C, total)
  if (src_dtype == DU_F32 && dtype == DU_F32) BL_LAUNCH(float, float, 4);
  else if (src_dtype == DU_F32 && dtype == DU_BF16) { if (v8) BL_LAUNCH(float, bf16_t, 8); else BL_LAUNCH(float, bf16_t, 4); }
  else if (src_dtype == DU_BF16 && dtype == DU_BF16) { if (v8) BL_LAUNCH(bf16_t, bf16_t, 8); else BL_LAUNCH(bf16_t, bf16_t, 4); }
  else return DU_ERR_BAD_ARG;
#undef BL_LAUNCH
  return du_check_launch();
}

extern "C" int du_bilinear_resize_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int Hs, int Ws, int Ho,
                                      int Wo, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!dy || !dx || B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C % 4) return DU_ERR_BAD_ARG;
  long total = (long)B * Hs * Ws * (C / 4);
  dim3 g(grid_1d(total)), b(256);
  if (dtype == DU_F32)
    hipLaunchKernelGGL(bilinear_resize_bwd_kernel<float>, g, b, 0, st, (const float*)dy, lddy, (float*)dx, lddx, B, Hs, Ws, Ho, Wo, C, total);
  else if (dtype == DU_BF16)
    hipLaunchKernelGGL(bilinear_resize_bwd_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, B, Hs, Ws, Ho, Wo, C, total);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}

namespace {

// SwiGLU gate on an interleaved projection (layers/ffn_layers.py:73-77): u (rows, 2h) with columns (2j, 2j+1) = (w1 x + b1, w2 x + b2)[j]
// -> out (rows, h) = silu(u[2j]) * u[2j+1].  One 16-byte load = 8 / 4 inputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void swiglu_pairs_kernel(const T* __restrict__ u, T* __restrict__ out, long npairs_vec) {
  constexpr int V = Elem<T>::VEC;               // inputs per 16 bytes
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npairs_vec; i += (long)gridDim.x * 256) {
    const Vec16<T> x = as_vec<T>(*(const uint4*)(u + i * V));
    T o[V / 2];
#pragma unroll
    for (int j = 0; j < V / 2; j++) {
      const float a = to_f32(x.v[2 * j]), b = to_f32(x.v[2 * j + 1]);
      o[j] = from_f32<T>(a / (1.0f + __expf(-a)) * b);
    }
    if constexpr (sizeof(T) == 2) *(uint2*)(out + i * (V / 2)) = __builtin_bit_cast(uint2, o);
    else *(float2*)(out + i * (V / 2)) = __builtin_bit_cast(float2, o);
  }
}

// whole samples of a (B, n) fp32 tensor by index: gather dst[j] = src[idx[j]]  /  scatter dst[idx[j]] = src[j]
// (batch-subset stochastic depth of the ViT-7B blocks, layers/block.py:126-187)
__global__ __launch_bounds__(256) void sample_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, const long* __restrict__ idx,
                                                          long n4, int scatter) {
  const int j = blockIdx.y;
  const long s = idx[j];
  const float4* sp = (const float4*)src + (scatter ? (long)j : s) * n4;
  float4* dp = (float4*)dst + (scatter ? s : (long)j) * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dp[i] = sp[i];
}

}  // namespace

extern "C" int du_swiglu_pairs(int dtype, const void* u, void* out, int64_t rows, int64_t h, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == DU_BF16 ? 8 : 4;
  if (!u || !out || rows <= 0 || h <= 0 || (2 * h) % vec) return DU_ERR_BAD_ARG;
  const long nv = rows * (2 * h) / vec;
  long g = (nv + 255) / 256; if (g > 8192) g = 8192;
  if (dtype == DU_BF16) hipLaunchKernelGGL(swiglu_pairs_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)u, (bf16_t*)out, nv);
  else if (dtype == DU_F32) hipLaunchKernelGGL(swiglu_pairs_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const float*)u, (float*)out, nv);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}

extern "C" int du_sample_copy(const float* src, float* dst, const int64_t* idx, int k, int64_t n_per_sample, int scatter, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!src || !dst || !idx || k <= 0 || n_per_sample <= 0 || n_per_sample % 4) return DU_ERR_BAD_ARG;
  const long n4 = n_per_sample / 4;
  long g = (n4 + 255) / 256; if (g > 512) g = 512;
  hipLaunchKernelGGL(sample_copy_kernel, dim3((unsigned)g, k), dim3(256), 0, st, src, dst, (const long*)idx, n4, scatter);
  return du_check_launch();
}

// ordered reduction of split-K slabs (DU_STORE_SLABS) + bf16 rounding: 4 outputs per thread
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ slabs, bf16_t* __restrict__ out, int splits, long n4, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 acc = *(const float4*)(slabs + 4 * i);
    for (int s = 1; s < splits; s++) {
      const float4 t = *(const float4*)(slabs + (long)s * n + 4 * i);
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    bf16x4 r;
    r[0] = (bf16_t)acc.x; r[1] = (bf16_t)acc.y; r[2] = (bf16_t)acc.z; r[3] = (bf16_t)acc.w;
    *(uint2*)(out + 4 * i) = __builtin_bit_cast(uint2, r);
  }
}

extern "C" int du_splitk_reduce_bf16(const float* slabs, void* out, int splits, int64_t n, void* stream) {
  if (!slabs || !out || splits < 1 || n <= 0 || n % 4) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(grid_1d(n / 4)), dim3(256), 0, (hipStream_t)stream, slabs, (bf16_t*)out, splits, (long)(n / 4), (long)n);
  return du_check_launch();
}

extern "C" int du_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!src || !dst || n <= 0) return DU_ERR_BAD_ARG;
  dim3 g(grid_1d(n)), b(256);
  if (src_dtype == DU_F32 && dst_dtype == DU_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)src, (bf16_t*)dst, (long)n);
  else if (src_dtype == DU_BF16 && dst_dtype == DU_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)src, (float*)dst, (long)n);
  else if (src_dtype == DU_F32 && dst_dtype == DU_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, (long)n);
  else if (src_dtype == DU_BF16 && dst_dtype == DU_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)src, (bf16_t*)dst, (long)n);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}

extern "C" int du_nchw_to_nhwc_pad(int dst_dtype, const float* src, void* dst, int B, int C, int H, int W, int Cpad, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!src || !dst || B <= 0 || Cpad < C) return DU_ERR_BAD_ARG;
  long total = (long)B * H * W * Cpad;
  DISPATCH_T(dst_dtype,
             hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, src, (bf16_t*)dst, B, C, H, W, Cpad, total),
             hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, src, (float*)dst, B, C, H, W, Cpad, total));
  return du_check_launch();
}

extern "C" int du_nhwc_to_nchw_f32(int src_dtype, const void* src, int64_t ld, float* dst, int B, int C, int H, int W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!src || !dst || B <= 0) return DU_ERR_BAD_ARG;
  long total = (long)B * C * H * W;
  DISPATCH_T(src_dtype,
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)src, ld, dst, B, C, H, W, total),
             hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)src, ld, dst, B, C, H, W, total));
  return du_check_launch();
}

extern "C" int du_patchify16(int dst_dtype, const float* src, void* dst, int B, int C, int H, int W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!src || !dst || B <= 0 || H % 16 || W % 16) return DU_ERR_BAD_ARG;
  long total = (long)B * (H / 16) * (W / 16) * C * 256;
  DISPATCH_T(dst_dtype,
             hipLaunchKernelGGL(patchify16_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, src, (bf16_t*)dst, B, C, H, W, total),
             hipLaunchKernelGGL(patchify16_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, src, (float*)dst, B, C, H, W, total));
  return du_check_launch();
}

extern "C" int du_msda_prep(int dtype, const void* raw, int64_t ldr, const float* ref, float* loc, float* attn, int64_t rows, int Lq,
                            int M, int P, int Hs, int Ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!raw || !ref || !loc || !attn || rows <= 0 || Lq <= 0 || rows % Lq || M <= 0 || P <= 0 || P > 16) return DU_ERR_BAD_ARG;
  long total = rows * M;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(msda_prep_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)raw, ldr, ref, loc, attn, (long)rows, Lq, M, P, 1.f / Ws, 1.f / Hs, total),
             hipLaunchKernelGGL(msda_prep_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)raw, ldr, ref, loc, attn, (long)rows, Lq, M, P, 1.f / Ws, 1.f / Hs, total));
  return du_check_launch();
}

extern "C" int du_msda_prep_bwd(int dtype, const float* attn, const float* gloc, const float* gattn, void* graw, int64_t ldr, int64_t rows,
                                int M, int P, int Hs, int Ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!attn || !gloc || !gattn || !graw || rows <= 0 || M <= 0 || P <= 0) return DU_ERR_BAD_ARG;
  long total = rows * M;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(msda_prep_bwd_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, attn, gloc, gattn, (bf16_t*)graw, ldr, (long)rows, M, P, 1.f / Ws, 1.f / Hs, total),
             hipLaunchKernelGGL(msda_prep_bwd_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, attn, gloc, gattn, (float*)graw, ldr, (long)rows, M, P, 1.f / Ws, 1.f / Hs, total));
  return du_check_launch();
}

extern "C" int du_act_bwd(int dtype, const void* z, const void* dy, void* dz, int64_t n, int act, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!z || !dy || !dz || n <= 0) return DU_ERR_BAD_ARG;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid_1d(n)), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)dy, (bf16_t*)dz, (long)n, act),
             hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_1d(n)), dim3(256), 0, st, (const float*)z, (const float*)dy, (float*)dz, (long)n, act));
  return du_check_launch();
}

extern "C" int du_se_gate_fwd(const float* sums, float inv_count, const float* W1, const float* b1, const float* W2, const float* b2,
                              float* hidden, float* gate, int B, int C, int R, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!sums || !W1 || !b1 || !W2 || !b2 || !hidden || !gate || B <= 0 || C <= 0 || R <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(B), dim3(256), (C + R) * sizeof(float), st, sums, inv_count, W1, b1, W2, b2, hidden, gate, C, R);
  return du_check_launch();
}

extern "C" int du_se_scale_fwd(int dtype, const void* x, int64_t ldx, const float* gate, const void* shortcut, int64_t ldsc, void* y,
                               int64_t ldy, int B, int64_t P, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!x || !gate || !y || B <= 0 || P <= 0 || C % v || ldx % v || ldy % v || (shortcut && ldsc % v)) return DU_ERR_BAD_ARG;
  long total = (long)B * P * (C / v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(se_scale_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)x, ldx, gate, (const bf16_t*)shortcut, ldsc, (bf16_t*)y, ldy, (long)P, C, total),
             hipLaunchKernelGGL(se_scale_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)x, ldx, gate, (const float*)shortcut, ldsc, (float*)y, ldy, (long)P, C, total));
  return du_check_launch();
}

extern "C" int du_se_gate_bwd(const float* dsum, const float* sums, float inv_count, const float* gate, const float* hidden,
                              const float* W1, const float* W2, float* dpool, float* dW1, float* db1, float* dW2, float* db2, int B, int C,
                              int R, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!dsum || !sums || !gate || !hidden || !W1 || !W2 || !dpool || !dW1 || !db1 || !dW2 || !db2 || B <= 0 || C <= 0 || R <= 0)
    return DU_ERR_BAD_ARG;
  int Bc = (int)(8000 / (long)(C + R));               // samples resident in LDS at once (<= 64 KB)
  if (Bc < 1) return DU_ERR_UNSUPPORTED;
  if (Bc > B) Bc = B;
  hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(1), dim3(1024), (size_t)2 * Bc * (C + R) * sizeof(float), st, dsum, sums, inv_count, gate, hidden,
                     W1, W2, dpool, dW1, db1, dW2, db2, B, C, R, Bc);
  return du_check_launch();
}

extern "C" int du_se_scale_bwd(int dtype, const void* dy, int64_t lddy, const float* gate, const float* dpool, void* dx, int64_t lddx,
                               int B, int64_t P, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!dy || !gate || !dpool || !dx || B <= 0 || P <= 0 || C % v || lddy % v || lddx % v) return DU_ERR_BAD_ARG;
  long total = (long)B * P * (C / v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(se_scale_bwd_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)dy, lddy, gate, dpool, (bf16_t*)dx, lddx, (long)P, C, total),
             hipLaunchKernelGGL(se_scale_bwd_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)dy, lddy, gate, dpool, (float*)dx, lddx, (long)P, C, total));
  return du_check_launch();
}

// ---------------- sliding-window inference: Gaussian-weighted accumulation of window logits (predict_from_raw_data.py:607-610) --------
namespace {
// logits (nb, K, ph, pw) fp32; window i sits at slice coords[3i], rows coords[3i+1].., columns coords[3i+2]..; windows of one launch may
// overlap (50 % steps), hence fp32 atomics -- at most 4 windows meet in a pixel
__global__ __launch_bounds__(256) void window_accumulate_kernel(const float* __restrict__ logits, const float* __restrict__ gauss,
                                                                const int* __restrict__ coords, float* __restrict__ pred,
                                                                float* __restrict__ npred, int nb, int K, int ph, int pw, int D, int H,
                                                                int W) {
  const long per = (long)ph * pw, total = (long)nb * per;
  GRID_STRIDE(i, total) {
    const int b = (int)(i / per);
    const int r = (int)(i - (long)b * per);
    const int y = r / pw, x = r - y * pw;
    const int d = coords[b * 3], y0 = coords[b * 3 + 1], x0 = coords[b * 3 + 2];
    const float g = gauss[r];
    const long o = ((long)d * H + y0 + y) * W + x0 + x;
    atomic_add_f32(npred + o, g);
    for (int k = 0; k < K; k++) atomic_add_f32(pred + (long)k * D * H * W + o, logits[((long)b * K + k) * per + r] * g);
  }
}
__global__ __launch_bounds__(256) void window_normalize_kernel(float* __restrict__ pred, const float* __restrict__ npred, int K, long n) {
  GRID_STRIDE(i, n) {
    const float inv = 1.f / npred[i];
    for (int k = 0; k < K; k++) pred[(long)k * n + i] *= inv;
  }
}
}  // namespace

extern "C" int du_window_accumulate(const float* logits, const float* gauss, const int32_t* coords, float* pred, float* npred, int nb, int K,
                                    int ph, int pw, int D, int H, int W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!logits || !gauss || !coords || !pred || !npred || nb <= 0 || K <= 0 || ph <= 0 || pw <= 0 || D <= 0 || H < ph || W < pw) return DU_ERR_BAD_ARG;
  const long total = (long)nb * ph * pw;
  hipLaunchKernelGGL(window_accumulate_kernel, dim3(grid_1d(total)), dim3(256), 0, st, logits, gauss, (const int*)coords, pred, npred, nb, K, ph, pw,
                     D, H, W);
  return du_check_launch();
}

extern "C" int du_window_normalize(float* pred, const float* npred, int K, int64_t n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!pred || !npred || K <= 0 || n <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(window_normalize_kernel, dim3(grid_1d(n)), dim3(256), 0, st, pred, npred, K, (long)n);
  return du_check_launch();
}

extern "C" const char* du_version(void) { return "dinounet_hip 0.1 (gfx950)"; }

extern "C" int du_device_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  const char* a = p.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}

// ---------------- per-step weight packing: every trainable fp32 weight -> its kernel-ready bf16 / fp32 forms, ONE launch -------------
// Replaces ~300 tiny cast / permute / cat launches per train step (each >= 5 us inside a hipGraph).  Table row (8 x int64):
//   [src ptr, dst ptr, kind | dst_f32 << 8, A, B, T, Cp, n_out];  bprefix[i] = first workgroup of row i (4096 outputs per workgroup).
//   kind 0 CAST        dst[e] = src[e]
//        1 CONV_FWD    src (A=Cout, B=Cin, T taps) -> dst (Cout, T*Cp), (tap, ci) columns, ci >= Cin zero     (F.conv2d weight)
//        2 CONV_DGRAD  -> dst (Cin, T*Cout), [ci][(t, co)] = src[co][ci][t]
//        3 CONV_DGRAD_FLIP  same with t -> T-1-t (stride-1 data gradient as a convolution with the flipped filter)
//        4 CONVT_FWD   src (A=Cin, B=Cout, 2, 2) -> dst (4*Cout, Cin), [(q, co)][ci] = src[ci][co][q]       (ConvTranspose2d k2 s2)
//        5 CONVT_DGRAD -> dst (Cin, 4*Cout), [ci][(q, co)] = src[ci][co][q]
//        6 TRANSPOSE   src (A, B) -> dst (B, A), or with T > 0 a column block of a wider (B, T) matrix (transposed row-concatenation);
//                      64 x 64 tiles through LDS, ceil(A/64) * ceil(B/64) workgroups   (linear-layer data gradient as an "NT" product)
namespace {
constexpr int PACK_CHUNK = 4096;   // output elements per workgroup
__global__ __launch_bounds__(256) void pack_weights_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ bprefix, int n) {
  // workgroup -> (table row, chunk of the row): bprefix[i] = first workgroup of row i (uniform binary search, scalar loads)
  int lo = 0, hi = n;
  const long bid = blockIdx.x;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bprefix[mid] <= bid) lo = mid; else hi = mid; }
  const int64_t* r = table + (long)lo * 8;
  const float* src = (const float*)r[0];
  const int kind = (int)(r[2] & 0xff);
  const bool f32 = (r[2] >> 8) & 1;
  const int A = (int)r[3], B = (int)r[4], T = (int)r[5], Cp = (int)r[6];
  const int nout = (int)r[7];
  float* df = (float*)r[1];
  bf16_t* db = (bf16_t*)r[1];
  if (kind == 6) {
    // TRANSPOSE through LDS, one 64 x 64 tile per workgroup (the row owns ceil(A/64) * ceil(B/64) workgroups): 256-byte row segments in,
    // 128-byte (bf16) / 256-byte (fp32) row segments out.  (An element-per-thread transpose read src at a stride of B floats: the 40-odd
    // W^T packs of a dinounet_l step made this kernel 281 us instead of 66.)
    __shared__ float tile[64][65];
    const int tiles_a = (A + 63) >> 6;
    const int t = (int)(bid - bprefix[lo]);
    const int a0 = (t % tiles_a) * 64, b0 = (t / tiles_a) * 64;
    const long ldd = T > 0 ? T : A;                         // T > 0: dst is a column block of a wider (B, T) matrix
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int a = a0 + ty + 16 * i;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int b = b0 + tx * 4 + j;
        tile[ty + 16 * i][tx * 4 + j] = (a < A && b < B) ? src[(long)a * B + b] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int b = b0 + ty + 16 * i;
      if (b >= B) continue;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int a = a0 + tx * 4 + j;
        if (a >= A) continue;
        const float v = tile[tx * 4 + j][ty + 16 * i];
        if (f32) df[(long)b * ldd + a] = v; else db[(long)b * ldd + a] = (bf16_t)v;
      }
    }
    return;
  }
  const int e0 = (int)(bid - bprefix[lo]) * PACK_CHUNK;
  const int e1 = min(nout, e0 + PACK_CHUNK);
  if (kind == 0 && !f32 && (nout & 3) == 0 && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)db) & 7) == 0)) {
    for (int e = e0 + threadIdx.x * 4; e < e1; e += 1024) {
      const float4 v = *(const float4*)(src + e);
      bf16x4 o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
      *(uint2*)(db + e) = __builtin_bit_cast(uint2, o);
    }
    return;
  }
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    float v;
    long d = e;            // destination element
    switch (kind) {
      case 1: { const int co = e / (T * Cp); const int rr = e - co * (T * Cp); const int t = rr / Cp, ci = rr - t * Cp;
                v = ci < B ? src[((long)co * B + ci) * T + t] : 0.f; break; }
      case 2: case 3: { const int ci = e / (T * A); const int rr = e - ci * (T * A); int t = rr / A; const int co = rr - t * A;
                if (kind == 3) t = T - 1 - t;
                v = src[((long)co * B + ci) * T + t]; break; }
      case 4: { const int row = e / A; const int ci = e - row * A; const int q = row / B, co = row - q * B;
                v = src[((long)ci * B + co) * 4 + q]; break; }
      case 5: { const int ci = e / (4 * B); const int rr = e - ci * (4 * B); const int q = rr / B, co = rr - q * B;
                v = src[((long)ci * B + co) * 4 + q]; break; }
      default: v = src[e];
    }
    if (f32) df[d] = v; else db[d] = (bf16_t)v;
  }
}
}  // namespace

extern "C" int du_pack_weights(const int64_t* table, const int64_t* bprefix, int n, int64_t nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!table || !bprefix || n <= 0 || nblocks <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, table, bprefix, n);
  return du_check_launch();
}

// ---------------- FiLM modulation of FAPM (dinounet_training.py:427-429): z = gamma * z_specific + beta ----------------
// gb (rows, 2R) = [gamma | beta] (film generator output), z2 (rows, 2R) = [z_shared | z_specific] (the fused shared+specific GEMM);
// z (rows, R).  Backward writes the full dgb and the z_specific half of dz2 (the z_shared half is filled by the film-generator
// data gradient), so no slice / zero-fill / add launches of the autograd engine are needed.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void film_fwd_kernel(const T* __restrict__ gb, const T* __restrict__ z2, T* __restrict__ z, long rows, int R) {
  constexpr int V = Elem<T>::VEC;
  const int rv = R / V;
  const long total = rows * rv;
  GRID_STRIDE(i, total) {
    const long row = i / rv;
    const int c0 = (int)(i % rv) * V;
    Vec16<T> g = as_vec<T>(*(const uint4*)(gb + row * 2 * R + c0));
    Vec16<T> b = as_vec<T>(*(const uint4*)(gb + row * 2 * R + R + c0));
    Vec16<T> s = as_vec<T>(*(const uint4*)(z2 + row * 2 * R + R + c0));
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(to_f32(g.v[j]) * to_f32(s.v[j]) + to_f32(b.v[j]));
    *(uint4*)(z + row * R + c0) = as_u4(o);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void film_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ gb, const T* __restrict__ z2,
                                                       T* __restrict__ dgb, T* __restrict__ dz2, long rows, int R) {
  constexpr int V = Elem<T>::VEC;
  const int rv = R / V;
  const long total = rows * rv;
  GRID_STRIDE(i, total) {
    const long row = i / rv;
    const int c0 = (int)(i % rv) * V;
    Vec16<T> d = as_vec<T>(*(const uint4*)(dz + row * R + c0));
    Vec16<T> g = as_vec<T>(*(const uint4*)(gb + row * 2 * R + c0));
    Vec16<T> s = as_vec<T>(*(const uint4*)(z2 + row * 2 * R + R + c0));
    Vec16<T> og, os;
#pragma unroll
    for (int j = 0; j < V; j++) {
      og.v[j] = from_f32<T>(to_f32(d.v[j]) * to_f32(s.v[j]));     // d gamma
      os.v[j] = from_f32<T>(to_f32(d.v[j]) * to_f32(g.v[j]));     // d z_specific
    }
    *(uint4*)(dgb + row * 2 * R + c0) = as_u4(og);
    *(uint4*)(dgb + row * 2 * R + R + c0) = as_u4(d);             // d beta = dz
    *(uint4*)(dz2 + row * 2 * R + R + c0) = as_u4(os);
  }
}

// ------------------------------------------------------------------------------------------------------
// Segmentation head: the U-Net's last layer, a 1 x 1 convolution from 32 channels to K <= 4 classes on the full-resolution feature map
// (seg_layers[-1], dinounet_training.py:603-629 / nnU-Net UNetDecoder), logits in fp32 NCHW for the loss.  As a GEMM (M = 2 M pixels,
// N = K padded to 8, contraction 32) + an NHWC -> NCHW pass it ran at 1.3 TB/s (99 + ~25 us), its data / weight gradients as two more
// products (73 + 85 us) behind a pad-and-transpose of dlogits.  It is a stream: one thread per pixel.
//   forward : logits[b][k][p] = bias[k] + sum_c x[b][p][c] * bf16(w[k][c])          reads x once, writes K planes
//   backward: dx[b][p][c] = sum_k dl[b][k][p] * bf16(w[k][c]);  dw[k][c] = sum_p dl * x;  db[k] = sum_p dl        ONE pass over x
// (weights rounded to the activation dtype as autocast does; dlogits stay fp32).  Weight-gradient partials per workgroup -> du_strip_finalize.
// ------------------------------------------------------------------------------------------------------
constexpr int SEG_C = 32;
template <int K>
__global__ __launch_bounds__(256) void seg_head_fwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, long HW, long total) {
  __shared__ float ws[K * SEG_C + K];
  for (int i = threadIdx.x; i < K * SEG_C; i += 256) ws[i] = (float)(bf16_t)w[i];
  if (threadIdx.x < K) ws[K * SEG_C + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    uint4 r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = *(const uint4*)(x + p * ldx + j * 8);
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = ws[K * SEG_C + k];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const Vec16<bf16_t> v = as_vec<bf16_t>(r[j]);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xv = (float)v.v[e];
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = fmaf(xv, ws[k * SEG_C + j * 8 + e], acc[k]);
      }
    }
    const long b = p / HW, q = p - b * HW;
#pragma unroll
    for (int k = 0; k < K; k++) out[(b * K + k) * HW + q] = acc[k];
  }
}

template <int K>
__global__ __launch_bounds__(256) void seg_head_bwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ w,
                                                           const float* __restrict__ dl, bf16_t* __restrict__ dx, long lddx,
                                                           float* __restrict__ part, long HW, long total) {
  constexpr int NW = K * SEG_C + K, NWP = (NW + 1) & ~1;          // partial rows padded to an even length (du_strip_finalize sums pairs)
  __shared__ float ws[K * SEG_C];
  __shared__ float red[4][NWP];
  for (int i = threadIdx.x; i < K * SEG_C; i += 256) ws[i] = (float)(bf16_t)w[i];
  __syncthreads();
  float gw[K][SEG_C], gb[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    gb[k] = 0.f;
#pragma unroll
    for (int c = 0; c < SEG_C; c++) gw[k][c] = 0.f;
  }
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const long b = p / HW, q = p - b * HW;
    float g[K];
#pragma unroll
    for (int k = 0; k < K; k++) { g[k] = dl[(b * K + k) * HW + q]; gb[k] += g[k]; }
    uint4 r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = *(const uint4*)(x + p * ldx + j * 8);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const Vec16<bf16_t> v = as_vec<bf16_t>(r[j]);
      Vec16<bf16_t> o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xv = (float)v.v[e];
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
          d = fmaf(g[k], ws[k * SEG_C + j * 8 + e], d);
          gw[k][j * 8 + e] = fmaf(g[k], xv, gw[k][j * 8 + e]);
        }
        o.v[e] = (bf16_t)d;
      }
      if (dx) *(uint4*)(dx + p * lddx + j * 8) = as_u4(o);
    }
  }
  // workgroup partial of (dw, db): wave sums, then the 4 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int c = 0; c < SEG_C; c++) {
      const float t = wave_sum(gw[k][c]);
      if (lane == 0) red[wave][k * SEG_C + c] = t;
    }
    const float tb = wave_sum(gb[k]);
    if (lane == 0) red[wave][K * SEG_C + k] = tb;
  }
  if (NWP != NW && threadIdx.x < 4) red[threadIdx.x][NW] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < NWP; i += 256) part[(long)blockIdx.x * NWP + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
}  // namespace

// x (B, HW, 32) bf16 NHWC with pixel stride ldx; w (K, 32) fp32, bias (K) fp32 or null; out (B, K, HW) fp32.  K in 1..4.
extern "C" int du_seg_head_fwd(const void* x, int64_t ldx, const float* w, const float* bias, float* out, int B, int64_t HW, int C, int K,
                               void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || !w || !out || B <= 0 || HW <= 0) return DU_ERR_BAD_ARG;
  if (C != SEG_C || K < 1 || K > 4 || ldx % 8 || (((uintptr_t)x) & 15)) return DU_ERR_UNSUPPORTED;
  const long total = (long)B * HW;
  long g = (total + 255) / 256; if (g > 256 * 16) g = 256 * 16;
#define SEG_FWD(K_) hipLaunchKernelGGL((seg_head_fwd_kernel<K_>), dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)x, (long)ldx, w, bias, out, (long)HW, total)
  switch (K) { case 1: SEG_FWD(1); break; case 2: SEG_FWD(2); break; case 3: SEG_FWD(3); break; default: SEG_FWD(4); break; }
#undef SEG_FWD
  return du_check_launch();
}
// number of fp32 partial rows du_seg_head_bwd writes (each K * 32 + K floats rounded up to even): the caller's scratch must hold that many
extern "C" int du_seg_head_bwd_blocks(int B, int64_t HW) {
  const long total = (long)B * HW;
  long g = (total + 255) / 256; if (g > 1024) g = 1024;
  return (int)g;
}
// dl (B, K, HW) fp32; dx (B, HW, 32) bf16 with pixel stride lddx (nullable: weight / bias gradients only); part: du_seg_head_bwd_blocks x
// (K * 33 rounded up to even) fp32 scratch; dwb (K * 33 rounded up to even) fp32 OVERWRITTEN: dw (K, 32) then db (K).
extern "C" int du_seg_head_bwd(const void* x, int64_t ldx, const float* w, const float* dl, void* dx, int64_t lddx, float* part, float* dwb,
                               int B, int64_t HW, int C, int K, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || !w || !dl || !part || !dwb || B <= 0 || HW <= 0) return DU_ERR_BAD_ARG;
  if (C != SEG_C || K < 1 || K > 4 || ldx % 8 || (dx && lddx % 8) || ((((uintptr_t)x) | ((uintptr_t)dx)) & 15)) return DU_ERR_UNSUPPORTED;
  const long total = (long)B * HW;
  const int g = du_seg_head_bwd_blocks(B, HW);
#define SEG_BWD(K_) hipLaunchKernelGGL((seg_head_bwd_kernel<K_>), dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)x, (long)ldx, w, dl, (bf16_t*)dx, (long)lddx, part, (long)HW, total)
  switch (K) { case 1: SEG_BWD(1); break; case 2: SEG_BWD(2); break; case 3: SEG_BWD(3); break; default: SEG_BWD(4); break; }
#undef SEG_BWD
  int rc = du_check_launch();
  if (rc != DU_OK) return rc;
  const int n = (K * SEG_C + K + 1) & ~1;
  return du_strip_finalize(part, dwb, 1, g, n / 2, stream);
}


extern "C" int du_film_fwd(int dtype, const void* gb, const void* z2, void* z, int64_t rows, int R, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!gb || !z2 || !z || rows <= 0 || R <= 0 || R % v) return DU_ERR_BAD_ARG;
  const long total = rows * (R / v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(film_fwd_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)gb, (const bf16_t*)z2, (bf16_t*)z, (long)rows, R),
             hipLaunchKernelGGL(film_fwd_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)gb, (const float*)z2, (float*)z, (long)rows, R));
  return du_check_launch();
}

extern "C" int du_film_bwd(int dtype, const void* dz, const void* gb, const void* z2, void* dgb, void* dz2, int64_t rows, int R, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (!dz || !gb || !z2 || !dgb || !dz2 || rows <= 0 || R <= 0 || R % v) return DU_ERR_BAD_ARG;
  const long total = rows * (R / v);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(film_bwd_kernel<bf16_t>, dim3(grid_1d(total)), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)gb, (const bf16_t*)z2, (bf16_t*)dgb, (bf16_t*)dz2, (long)rows, R),
             hipLaunchKernelGGL(film_bwd_kernel<float>, dim3(grid_1d(total)), dim3(256), 0, st, (const float*)dz, (const float*)gb, (const float*)z2, (float*)dgb, (float*)dz2, (long)rows, R));
  return du_check_launch();
}
