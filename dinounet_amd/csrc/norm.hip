// LayerNorm (token rows) and channel-statistics norms (InstanceNorm2d / BatchNorm2d on NHWC) for gfx950.
// All are HBM-bound: 16-byte vector loads, fp32 statistics, wave64 shuffle reductions, one pass per tensor.
#include "common.h"

int g_ln_rows2 = 1;      // du_set_option key 18: LayerNorm forward with two rows per wave: 0 never, 1 where one round of waves overflows by < 2x (default), 2 always

namespace {

// ------------------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, the row lives in registers (MAXV 16-byte vectors per lane).
// ------------------------------------------------------------------------------------------------------
template <typename TI, typename TO, int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TI* __restrict__ x, long ldx, const float* __restrict__ w,
                                                            const float* __restrict__ b, TO* __restrict__ y, long ldy,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, int D, float eps) {
  constexpr int VI = Elem<TI>::VEC;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = D / VI;
  float v[MAXV][VI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int vi = lane + i * 64;
    if (vi < nvec) {
      Vec16<TI> t = as_vec<TI>(*(const uint4*)(x + row * ldx + (long)vi * VI));
#pragma unroll
      for (int j = 0; j < VI; j++) { v[i][j] = to_f32(t.v[j]); s += v[i][j]; }
    } else {
#pragma unroll
      for (int j = 0; j < VI; j++) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int vi = lane + i * 64;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < VI; j++) { float d = v[i][j] - mean; s2 += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)D + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int vi = lane + i * 64;
    if (vi < nvec) {
      const int c0 = vi * VI;
#pragma unroll
      for (int j = 0; j < VI; j++) y[row * ldy + c0 + j] = from_f32<TO>((v[i][j] - mean) * rstd * w[c0 + j] + b[c0 + j]);
    }
  }
}

// Two rows per wave, both rows' loads in flight (round 6).  With one row per wave the ViT's 8232 rows are 8232 waves on a chip that holds
// 8192 (32 per CU): forty waves form a round of their own behind the first -- a whole load / reduce / store latency chain for 0.5 % of
// the rows, 52 times per step.  Half the waves, the same bytes in flight.
template <typename TI, typename TO, int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd2_kernel(const TI* __restrict__ x, long ldx, const float* __restrict__ w,
                                                             const float* __restrict__ b, TO* __restrict__ y, long ldy,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             long rows, int D, float eps) {
  constexpr int VI = Elem<TI>::VEC;
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (row0 >= rows) return;
  const bool two = row0 + 1 < rows;
  const int nvec = D / VI;
  float v[2][MAXV][VI];
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; u++)
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
      const int vi = lane + i * 64;
      if (vi < nvec && (u == 0 || two)) {
        Vec16<TI> t = as_vec<TI>(*(const uint4*)(x + (row0 + u) * ldx + (long)vi * VI));
#pragma unroll
        for (int j = 0; j < VI; j++) { v[u][i][j] = to_f32(t.v[j]); s[u] += v[u][i][j]; }
      } else {
#pragma unroll
        for (int j = 0; j < VI; j++) v[u][i][j] = 0.f;
      }
    }
  float mean[2], rstd[2];
#pragma unroll
  for (int u = 0; u < 2; u++) mean[u] = wave_sum(s[u]) / (float)D;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VI; j++) { const float d = v[u][i][j] - mean[u]; s2 += d * d; }
      }
    }
    rstd[u] = rsqrtf(wave_sum(s2) / (float)D + eps);
  }
  if (lane == 0) {
    if (mean_out) { mean_out[row0] = mean[0]; if (two) mean_out[row0 + 1] = mean[1]; }
    if (rstd_out) { rstd_out[row0] = rstd[0]; if (two) rstd_out[row0 + 1] = rstd[1]; }
  }
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      const int c0 = vi * VI;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (u == 1 && !two) break;
#pragma unroll
        for (int j = 0; j < VI; j++) y[(row0 + u) * ldy + c0 + j] = from_f32<TO>((v[u][i][j] - mean[u]) * rstd[u] * w[c0 + j] + b[c0 + j]);
      }
    }
  }
}

// LayerNorm backward, part 1: dx, one wave per row (same shape as the forward).
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const float* __restrict__ w, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, T* __restrict__ dx, long rows, int D,
                                                               const T* __restrict__ dres) {
  // dres (nullable): gradient arriving on the residual branch that by-passes the norm (y = x + f(LN(x))): dx = LN-path + dres,
  // fused here instead of a separate full-size add in the autograd engine
  constexpr int VI = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = D / VI;
  const float mu = mean[row], rs = rstd[row];
  float xh[MAXV][VI], g[MAXV][VI];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int vi = lane + i * 64;
    if (vi < nvec) {
      Vec16<T> tx = as_vec<T>(*(const uint4*)(x + row * (long)D + (long)vi * VI));
      Vec16<T> tg = as_vec<T>(*(const uint4*)(dy + row * (long)D + (long)vi * VI));
#pragma unroll
      for (int j = 0; j < VI; j++) {
        xh[i][j] = (to_f32(tx.v[j]) - mu) * rs;
        g[i][j] = to_f32(tg.v[j]) * w[vi * VI + j];
        c1 += g[i][j];
        c2 += g[i][j] * xh[i][j];
      }
    }
  }
  c1 = wave_sum(c1) / (float)D;
  c2 = wave_sum(c2) / (float)D;
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int vi = lane + i * 64;
    if (vi < nvec) {
      Vec16<T> o;
      if (dres) {
        Vec16<T> tr = as_vec<T>(*(const uint4*)(dres + row * (long)D + (long)vi * VI));
#pragma unroll
        for (int j = 0; j < VI; j++) o.v[j] = from_f32<T>(rs * (g[i][j] - c1 - xh[i][j] * c2) + to_f32(tr.v[j]));
      } else {
#pragma unroll
        for (int j = 0; j < VI; j++) o.v[j] = from_f32<T>(rs * (g[i][j] - c1 - xh[i][j] * c2));
      }
      *(uint4*)(dx + row * (long)D + (long)vi * VI) = as_u4(o);
    }
  }
}

// LayerNorm backward in ONE pass over x / dy (round 4; the two kernels above and below read both tensors twice: 0.74 + 0.44 ms per step
// on the adapter's 43008 x 1024 LayerNorms).  A workgroup owns a strip of rows, one wave per row as in layernorm_bwd_dx_kernel; on the way
// every lane also accumulates the weight / bias gradient terms of ITS columns (dw += dy * xhat, db += dy) over the rows its wave visits.
// The four waves' sums are combined through LDS and written as one partial row per workgroup: part[strip][D][2] -> finalize_kernel.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_fused_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  const float* __restrict__ w, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, T* __restrict__ dx, long rows, int D,
                                                                  const T* __restrict__ dres, int strip, float* __restrict__ part) {
  constexpr int VI = Elem<T>::VEC;
  extern __shared__ float ln_red[];                     // [3][D][2]: the sums of waves 1-3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = D / VI;
  float wv[MAXV][VI], sa[MAXV][VI], sb[MAXV][VI];
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    const int vi = lane + i * 64;
#pragma unroll
    for (int j = 0; j < VI; j++) { wv[i][j] = vi < nvec ? w[vi * VI + j] : 0.f; sa[i][j] = 0.f; sb[i][j] = 0.f; }
  }
  const long r0 = (long)blockIdx.x * strip;
  const long r1 = r0 + strip < rows ? r0 + strip : rows;
  // two rows of the wave in flight per iteration (a strip gives a wave ~20 rows: one row at a time left the loads latency-bound)
  for (long row0 = r0 + wave; row0 < r1; row0 += 8) {
    uint4 rx[2][MAXV], rg[2][MAXV];
    float mu[2], rs[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const long row = row0 + 4 * u;
      const bool live = row < r1;
      mu[u] = live ? mean[row] : 0.f; rs[u] = live ? rstd[row] : 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; i++) {
        const int vi = lane + i * 64;
        rx[u][i] = make_uint4(0, 0, 0, 0); rg[u][i] = make_uint4(0, 0, 0, 0);
        if (live && vi < nvec) {
          rx[u][i] = *(const uint4*)(x + row * (long)D + (long)vi * VI);
          rg[u][i] = *(const uint4*)(dy + row * (long)D + (long)vi * VI);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const long row = row0 + 4 * u;
      if (row >= r1) break;
      float xh[MAXV][VI], g[MAXV][VI];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; i++) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
          Vec16<T> tx = as_vec<T>(rx[u][i]);
          Vec16<T> tg = as_vec<T>(rg[u][i]);
#pragma unroll
          for (int j = 0; j < VI; j++) {
            const float gy = to_f32(tg.v[j]);
            xh[i][j] = (to_f32(tx.v[j]) - mu[u]) * rs[u];
            g[i][j] = gy * wv[i][j];
            c1 += g[i][j];
            c2 += g[i][j] * xh[i][j];
            sa[i][j] += gy * xh[i][j];
            sb[i][j] += gy;
          }
        }
      }
      c1 = wave_sum(c1) / (float)D;
      c2 = wave_sum(c2) / (float)D;
#pragma unroll
      for (int i = 0; i < MAXV; i++) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
          Vec16<T> o;
          if (dres) {
            Vec16<T> tr = as_vec<T>(*(const uint4*)(dres + row * (long)D + (long)vi * VI));
#pragma unroll
            for (int j = 0; j < VI; j++) o.v[j] = from_f32<T>(rs[u] * (g[i][j] - c1 - xh[i][j] * c2) + to_f32(tr.v[j]));
          } else {
#pragma unroll
            for (int j = 0; j < VI; j++) o.v[j] = from_f32<T>(rs[u] * (g[i][j] - c1 - xh[i][j] * c2));
          }
          *(uint4*)(dx + row * (long)D + (long)vi * VI) = as_u4(o);
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VI; j += 2)
          *(float4*)(ln_red + ((long)(wave - 1) * D + vi * VI + j) * 2) = make_float4(sa[i][j], sb[i][j], sa[i][j + 1], sb[i][j + 1]);
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VI; j += 2) {
          float4 t = make_float4(sa[i][j], sb[i][j], sa[i][j + 1], sb[i][j + 1]);
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const float4 u = *(const float4*)(ln_red + ((long)q * D + vi * VI + j) * 2);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
          }
          *(float4*)(part + ((long)blockIdx.x * D + vi * VI + j) * 2) = t;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Channel statistics over NHWC: thread = (pixel lane, channel vector); block = one strip of one group.
// ------------------------------------------------------------------------------------------------------
// pixels per block: sized so a launch has ~512 workgroups (256 CUs x 2: fewer, longer-lived workgroups measured faster in the step than
// 2048 short ones, and the partial buffer the finalize kernel reads is 4x smaller) but every pixel lane still runs >= 4 iterations
static inline int pick_strip(int G, long P, int C, int vec) {
  const int cvb = (C / vec) < 256 ? (C / vec) : 256;
  const int np = 256 / (cvb > 0 ? cvb : 1);
  static const long tmax = DU_GETENV("DU_STRIP_TARGET") ? atol(DU_GETENV("DU_STRIP_TARGET")) : 512;    // tuning aid (2048: +10 % time in the step)
  long target = 65536L * 8 / C;              // wide rows: fewer, longer strips keep the partial buffer (strips x C x 2) small
  if (target > tmax) target = tmax;
  if (target < 256) target = 256;
  long per_group = target / (G > 0 ? G : 1);
  if (per_group < 1) per_group = 1;
  long s = (P + per_group - 1) / per_group;
  const long smin = (long)np * 4;
  if (s < smin) s = smin;
  if (s > P) s = P;
  return (int)s;
}

template <typename T, int KIND>
struct StatOp;

// Sum the per-thread partials a[V], b[V] over the pixel lanes of a workgroup (thread = tp * cvb + tcv).  When several pixel lanes
// share a wave (cvb < 64, power of two) they are first combined with xor shuffles, so at most 4 values per channel reach LDS;
// the first version let cvb*V threads walk np <= 64 LDS entries serially, which dominated the small-C statistics kernels.
// emit(cv_local_channel_index j, sum_a, sum_b) is called by the threads holding final sums (tp == 0 lanes).
template <int V, typename E>
__device__ __forceinline__ void lane_reduce(float* a, float* b, int cvb, int np, int tp, int tcv, float (*red)[256 * V], bool valid, E emit) {
  const bool pow2 = (cvb & (cvb - 1)) == 0;
  if (pow2 && cvb < 64) {
    // lanes of one wave: tp spans 64 / cvb values; xor offsets cvb, 2 cvb, ... < 64 stay inside the wave
#pragma unroll
    for (int j = 0; j < V; j++) {
      for (int o = cvb; o < 64; o <<= 1) { a[j] += __shfl_xor(a[j], o, 64); b[j] += __shfl_xor(b[j], o, 64); }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < cvb) {
#pragma unroll
      for (int j = 0; j < V; j++) { red[0][(wave * cvb + lane) * V + j] = a[j]; red[1][(wave * cvb + lane) * V + j] = b[j]; }
    }
    __syncthreads();
    if (threadIdx.x < cvb && valid) {
#pragma unroll
      for (int j = 0; j < V; j++) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) { sa += red[0][(q * cvb + threadIdx.x) * V + j]; sb += red[1][(q * cvb + threadIdx.x) * V + j]; }
        emit(j, sa, sb);
      }
    }
    __syncthreads();
    return;
  }
#pragma unroll
  for (int j = 0; j < V; j++) { red[0][threadIdx.x * V + j] = a[j]; red[1][threadIdx.x * V + j] = b[j]; }
  __syncthreads();
  if (tp == 0 && valid) {
#pragma unroll
    for (int j = 0; j < V; j++) {
      float sa = 0.f, sb = 0.f;
      for (int q = 0; q < np; q++) { sa += red[0][(q * cvb + tcv) * V + j]; sb += red[1][(q * cvb + tcv) * V + j]; }
      emit(j, sa, sb);
    }
  }
  __syncthreads();
}

// generic strip reducer: F(xvec, dyvec, g, c0) -> (a[VEC], b[VEC]) accumulated per channel, then written with atomics
template <typename T, typename F>
__device__ __forceinline__ void strip_reduce(int G, long P, int C, int STRIP, float* out /*[G][C][2]*/, float* part, F f) {
  constexpr int V = Elem<T>::VEC;
  __shared__ float red[2][256 * V];
  const int cv_total = C / V;
  const long strips = (P + STRIP - 1) / STRIP;
  const int g = blockIdx.x / strips;
  const long p0 = (long)(blockIdx.x % strips) * STRIP;
  const long p1 = min(P, p0 + STRIP);
  const int cvb = min(cv_total, 256);        // channel vectors handled per pass
  const int tp = threadIdx.x / cvb;          // pixel lane
  const int tcv = threadIdx.x % cvb;
  const int np = 256 / cvb;                  // pixel lanes
  for (int cv0 = 0; cv0 < cv_total; cv0 += cvb) {
    const int cv = cv0 + tcv;
    float a[V], b[V];
#pragma unroll
    for (int j = 0; j < V; j++) { a[j] = 0.f; b[j] = 0.f; }
    if (cv < cv_total && tp < np) {
#pragma unroll 4
      for (long p = p0 + tp; p < p1; p += np) f((long)g * P + p, g, cv * V, a, b);
    }
    lane_reduce<V>(a, b, cvb, np, tp, tcv, red, cv < cv_total, [&](int j, float sa, float sb) {
      if (part) {   // two-stage: this strip's partial goes to part[blockIdx.x][C][2]; finalize_kernel sums the strips
        part[((long)blockIdx.x * C + cv * V + j) * 2 + 0] = sa;
        part[((long)blockIdx.x * C + cv * V + j) * 2 + 1] = sb;
      } else {
        atomic_add_f32(out + ((long)g * C + cv * V + j) * 2 + 0, sa);
        atomic_add_f32(out + ((long)g * C + cv * V + j) * 2 + 1, sb);
      }
    });
  }
}

// LayerNorm backward, part 2: dw[c] += sum_rows dy*xhat, db[c] += sum_rows dy  (column reduction over row strips),
// written interleaved as out[c][0..1] = (dw, db).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_wb_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ out, long rows, int D, int strip, float* part) {
  constexpr int V = Elem<T>::VEC;
  strip_reduce<T>(1, rows, D, strip, out, part, [&](long row, int g, int c0, float* a, float* b) {
    Vec16<T> tx = as_vec<T>(*(const uint4*)(x + row * (long)D + c0));
    Vec16<T> tg = as_vec<T>(*(const uint4*)(dy + row * (long)D + c0));
    const float mu = mean[row], rs = rstd[row];
#pragma unroll
    for (int j = 0; j < V; j++) {
      float gy = to_f32(tg.v[j]);
      a[j] += gy * (to_f32(tx.v[j]) - mu) * rs;
      b[j] += gy;
    }
  });
}

template <typename T>
__global__ __launch_bounds__(256) void chan_stats_kernel(const T* __restrict__ x, long ldx, float* __restrict__ sums, int G,
                                                         long P, int C, int strip, float* part) {
  constexpr int V = Elem<T>::VEC;
  strip_reduce<T>(G, P, C, strip, sums, part, [&](long pix, int g, int c0, float* a, float* b) {
    Vec16<T> t = as_vec<T>(*(const uint4*)(x + pix * ldx + c0));
#pragma unroll
    for (int j = 0; j < V; j++) { float v = to_f32(t.v[j]); a[j] += v; b[j] += v * v; }
  });
}

// sums[g][c] += (sum_pix a*b, sum_pix a)   (squeeze-excitation gate gradient: a = dy, b = x)
template <typename T>
__global__ __launch_bounds__(256) void chan_dot_kernel(const T* __restrict__ a_, long lda, const T* __restrict__ b_, long ldb,
                                                       float* __restrict__ sums, int G, long P, int C, int strip, float* part) {
  constexpr int V = Elem<T>::VEC;
  strip_reduce<T>(G, P, C, strip, sums, part, [&](long pix, int g, int c0, float* a, float* b) {
    Vec16<T> ta = as_vec<T>(*(const uint4*)(a_ + pix * lda + c0));
    Vec16<T> tb = as_vec<T>(*(const uint4*)(b_ + pix * ldb + c0));
#pragma unroll
    for (int j = 0; j < V; j++) { float va = to_f32(ta.v[j]); a[j] += va * to_f32(tb.v[j]); b[j] += va; }
  });
}

// The three norm+activation kernels share one decomposition: workgroup (blockIdx.y = group g, blockIdx.x = pixel-lane slot),
// thread = (pixel lane, channel vector).  A thread's per-channel constants (scale / shift / sums) live in registers and it walks the
// pixels of its group, so the inner loop is 1-2 vector loads + VEC FMAs (+ activation) + 1 vector store -- the first version
// re-read mean/rstd/w/b per element (32 scalar loads per 16-byte vector) and ran at ~0.9 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ w, const float* __restrict__ b, long P,
                                                           int C, int act) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V, cvb = min(cvn, 256), np = 256 / cvb;
  const int tp = threadIdx.x / cvb, tcv = threadIdx.x % cvb;
  if (tp >= np) return;
  const int g = blockIdx.y;
  for (int cv = tcv; cv < cvn; cv += cvb) {
    const int c0 = cv * V;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
      sc[j] = rstd[(long)g * C + c0 + j] * w[c0 + j];
      sh[j] = b[c0 + j] - mean[(long)g * C + c0 + j] * sc[j];
    }
    // four independent 16-byte loads in flight per thread (one load per trip left the 134 MB decoder tensors at ~2.8 TB/s)
    const long step = (long)gridDim.x * np;
    long p = (long)blockIdx.x * np + tp;
    for (; p + 3 * step < P; p += 4 * step) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; u++) r[u] = *(const uint4*)(x + ((long)g * P + p + u * step) * ldx + c0);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        Vec16<T> t = as_vec<T>(r[u]);
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(apply_act(to_f32(t.v[j]) * sc[j] + sh[j], act));
        *(uint4*)(y + ((long)g * P + p + u * step) * ldy + c0) = as_u4(o);
      }
    }
    for (; p < P; p += step) {
      const long pix = (long)g * P + p;
      Vec16<T> t = as_vec<T>(*(const uint4*)(x + pix * ldx + c0));
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; j++) o.v[j] = from_f32<T>(apply_act(to_f32(t.v[j]) * sc[j] + sh[j], act));
      *(uint4*)(y + pix * ldy + c0) = as_u4(o);
    }
  }
}

// backward pass 1: per-(g,c) sums of dz = dy * act'(z) and dz * xhat over this workgroup's pixels -> partial (or atomics)
// U pixels of this thread are requested together (2 U 16-byte loads in flight per thread): with 2 workgroups per CU the kernel is
// latency x concurrency bound, not bandwidth bound (U = 2: 2.6 TB/s over the decoder's layers in the dinounet_l step)
template <typename T, int U>
__global__ __launch_bounds__(256) void norm_act_bwd_stats_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy,
                                                                 long lddy, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ bsums, int G,
                                                                 long P, int C, int act, int STRIP, float* part) {
  constexpr int V = Elem<T>::VEC;
  __shared__ float red[2][256 * V];
  const int cv_total = C / V;
  const long strips = (P + STRIP - 1) / STRIP;
  const int g = blockIdx.x / strips;
  const long p0 = (long)(blockIdx.x % strips) * STRIP;
  const long p1 = min(P, p0 + STRIP);
  const int cvb = min(cv_total, 256);
  const int tp = threadIdx.x / cvb, tcv = threadIdx.x % cvb;
  const int np = 256 / cvb;
  for (int cv0 = 0; cv0 < cv_total; cv0 += cvb) {
    const int cv = cv0 + tcv;
    float a[V], bb[V];
#pragma unroll
    for (int j = 0; j < V; j++) { a[j] = 0.f; bb[j] = 0.f; }
    if (cv < cv_total && tp < np) {
      const int c0 = cv * V;
      float mu[V], rs[V], wc[V], bc[V];
#pragma unroll
      for (int j = 0; j < V; j++) { mu[j] = mean[(long)g * C + c0 + j]; rs[j] = rstd[(long)g * C + c0 + j]; wc[j] = w[c0 + j]; bc[j] = b[c0 + j]; }
      auto one = [&](const uint4& rx, const uint4& rg) {
        Vec16<T> tx = as_vec<T>(rx), tg = as_vec<T>(rg);
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float xh = (to_f32(tx.v[j]) - mu[j]) * rs[j];
          const float dz = to_f32(tg.v[j]) * act_grad(xh * wc[j] + bc[j], act);
          a[j] += dz;
          bb[j] += dz * xh;
        }
      };
      long p = p0 + tp;
      for (; p + (long)(U - 1) * np < p1; p += (long)U * np) {
        uint4 rx[U], rg[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const long pix = (long)g * P + p + (long)u * np;
          rx[u] = *(const uint4*)(x + pix * ldx + c0);
          rg[u] = *(const uint4*)(dy + pix * lddy + c0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) one(rx[u], rg[u]);
      }
      for (; p < p1; p += np) {
        const long pix = (long)g * P + p;
        one(*(const uint4*)(x + pix * ldx + c0), *(const uint4*)(dy + pix * lddy + c0));
      }
    }
    lane_reduce<V>(a, bb, cvb, np, tp, tcv, red, cv < cv_total, [&](int j, float sa, float sb) {
      if (part) {
        part[((long)blockIdx.x * C + cv * V + j) * 2 + 0] = sa;
        part[((long)blockIdx.x * C + cv * V + j) * 2 + 1] = sb;
      } else {
        atomic_add_f32(bsums + ((long)g * C + cv * V + j) * 2 + 0, sa);
        atomic_add_f32(bsums + ((long)g * C + cv * V + j) * 2 + 1, sb);
      }
    });
  }
}

// backward pass 2: dx = w*rstd*(dz - s1/n - xhat*s2/n)   (use_batch_stats = 0: dx = dz*w*rstd)
template <typename T>
__global__ __launch_bounds__(256) void norm_act_bwd_dx_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy,
                                                              long lddy, T* __restrict__ dx, long lddx,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              const float* __restrict__ bsums, long P, int C, int act,
                                                              float inv_count, int use_batch_stats) {
  constexpr int V = Elem<T>::VEC;
  const int cvn = C / V, cvb = min(cvn, 256), np = 256 / cvb;
  const int tp = threadIdx.x / cvb, tcv = threadIdx.x % cvb;
  if (tp >= np) return;
  const int g = blockIdx.y;
  for (int cv = tcv; cv < cvn; cv += cvb) {
    const int c0 = cv * V;
    float mu[V], rs[V], wc[V], bc[V], k1[V], k2[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
      mu[j] = mean[(long)g * C + c0 + j]; rs[j] = rstd[(long)g * C + c0 + j]; wc[j] = w[c0 + j]; bc[j] = b[c0 + j];
      k1[j] = use_batch_stats ? bsums[((long)g * C + c0 + j) * 2 + 0] * inv_count : 0.f;
      k2[j] = use_batch_stats ? bsums[((long)g * C + c0 + j) * 2 + 1] * inv_count : 0.f;
    }
    auto one = [&](const uint4& rx, const uint4& rg, long pix) {
      Vec16<T> tx = as_vec<T>(rx), tg = as_vec<T>(rg);
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; j++) {
        const float xh = (to_f32(tx.v[j]) - mu[j]) * rs[j];
        const float dz = to_f32(tg.v[j]) * act_grad(xh * wc[j] + bc[j], act);
        o.v[j] = from_f32<T>(wc[j] * rs[j] * (dz - k1[j] - xh * k2[j]));
      }
      *(uint4*)(dx + pix * lddx + c0) = as_u4(o);
    };
    // 2 x 2 independent 16-byte loads in flight per thread
    const long step = (long)gridDim.x * np;
    long p = (long)blockIdx.x * np + tp;
    for (; p + 3 * step < P; p += 4 * step) {          // 8 loads in flight per thread (2 workgroups per CU: latency x concurrency bound)
      uint4 rx[4], rg[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const long pu = (long)g * P + p + u * step;
        rx[u] = *(const uint4*)(x + pu * ldx + c0);
        rg[u] = *(const uint4*)(dy + pu * lddy + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) one(rx[u], rg[u], (long)g * P + p + u * step);
    }
    for (; p + step < P; p += 2 * step) {
      const long pa = (long)g * P + p, pb = pa + step;
      const uint4 xa = *(const uint4*)(x + pa * ldx + c0), xb = *(const uint4*)(x + pb * ldx + c0);
      const uint4 ga = *(const uint4*)(dy + pa * lddy + c0), gb = *(const uint4*)(dy + pb * lddy + c0);
      one(xa, ga, pa);
      one(xb, gb, pb);
    }
    for (; p < P; p += step) {
      const long pix = (long)g * P + p;
      one(*(const uint4*)(x + pix * ldx + c0), *(const uint4*)(dy + pix * lddy + c0), pix);
    }
  }
}

// pixel-lane slots per group for the two elementwise kernels: ~4096 workgroups in total, >= 4 pixels per lane
static inline int norm_slots(int G, long P, int C, int vec, bool bwd = false) {
  // measured in the step (rocprofv3, tools/_normsweep.sh): the forward kernel (1 load + 1 store per pixel) is best with many short
  // workgroups, the backward-dx kernel (2 loads + 1 store, 6 channel constants per thread) with few long ones: 42 -> 31 us average
  static const long total_f = DU_GETENV("DU_NORM_SLOTS") ? atol(DU_GETENV("DU_NORM_SLOTS")) : 4096;     // tuning aids
  static const long total_b = DU_GETENV("DU_NORM_SLOTS_BWD") ? atol(DU_GETENV("DU_NORM_SLOTS_BWD")) : 512;
  const long total = bwd ? total_b : total_f;
  const int cvb = (C / vec) < 256 ? (C / vec) : 256;
  const int np = 256 / (cvb > 0 ? cvb : 1);
  long s = (P + (long)np * 4 - 1) / ((long)np * 4);
  long cap = total / (G > 0 ? G : 1);
  if (cap < 1) cap = 1;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (int)s;
}

// out[g][c][j] = sum over the strips of group g of part[g*strips + s][c][j].  Workgroup = COLS columns x (256 / COLS) strip
// lanes; 4 independent partial sums per lane keep 4 loads in flight.
template <int COLS>
__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int G, int strips, int C2,
                                                       int first_only) {
  constexpr int LANES = 256 / COLS;
  __shared__ float red[LANES][COLS + 1];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int chunks = (C2 + COLS - 1) / COLS;
  const int g = blockIdx.x / chunks, c = (blockIdx.x % chunks) * COLS + col;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C2) {
    const float* p = part + (long)g * strips * C2 + c;
    int s = sl;
    for (; s + 3 * LANES < strips; s += 4 * LANES) {
      a0 += p[(long)s * C2]; a1 += p[(long)(s + LANES) * C2]; a2 += p[(long)(s + 2 * LANES) * C2]; a3 += p[(long)(s + 3 * LANES) * C2];
    }
    for (; s < strips; s += LANES) a0 += p[(long)s * C2];
  }
  red[sl][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && c < C2) {
    float t = 0.f;
    for (int q = 0; q < LANES; q++) t += red[q][col];
    if (!first_only) out[(long)g * C2 + c] = t;
    else if (first_only == 2) out[(long)g * C2 + (c & 1) * (C2 >> 1) + (c >> 1)] = t;   // planar: all first components, then all second
    else if ((c & 1) == 0) out[((long)g * C2 + c) >> 1] = t;      // only the first component of every (sum, sum2) pair, compacted
  }
}

// finalize + what the next tiny kernel would do, in one launch (every launch of this size costs ~5 us inside the replayed graph):
//  STATS: out[g][c] = (sum, sum of squares) as above, then mean / rstd (biased variance) and the optional BatchNorm running-statistics
//         update -- what norm_stats_finalize_kernel does.  A workgroup owns COLS consecutive columns = COLS / 2 channels of one group.
template <int COLS>
__global__ __launch_bounds__(256) void finalize_stats_kernel(const float* __restrict__ part, float* __restrict__ out, int G, int strips, int C2,
                                                             float inv_count, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                             float unbias) {
  constexpr int LANES = 256 / COLS;
  __shared__ float red[LANES][COLS + 1];
  __shared__ float tot[COLS];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int chunks = (C2 + COLS - 1) / COLS;
  const int g = blockIdx.x / chunks, c = (blockIdx.x % chunks) * COLS + col;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C2) {
    const float* p = part + (long)g * strips * C2 + c;
    int s = sl;
    for (; s + 3 * LANES < strips; s += 4 * LANES) {
      a0 += p[(long)s * C2]; a1 += p[(long)(s + LANES) * C2]; a2 += p[(long)(s + 2 * LANES) * C2]; a3 += p[(long)(s + 3 * LANES) * C2];
    }
    for (; s < strips; s += LANES) a0 += p[(long)s * C2];
  }
  red[sl][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0) {
    float t = 0.f;
    for (int q = 0; q < LANES; q++) t += red[q][col];
    tot[col] = t;
    if (c < C2 && out) out[(long)g * C2 + c] = t;
  }
  __syncthreads();
  if (sl == 0 && (col & 1) == 0 && c + 1 < C2) {
    const long i = ((long)g * C2 + c) >> 1;                 // (group, channel)
    const float m = tot[col] * inv_count;
    float var = tot[col + 1] * inv_count - m * m;
    var = var > 0.f ? var : 0.f;
    mean[i] = m;
    rstd[i] = rsqrtf(var + eps);
    if (run_mean) {                                         // G == 1 (BatchNorm)
      run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * m;
      run_var[i] = (1.f - momentum) * run_var[i] + momentum * var * unbias;
    }
  }
}

//  GRADS: bs[g][c] = (sum dz, sum dz * xhat) as above for EVERY group, then dw[ch] = sum_g bs[g][ch][1], db[ch] = sum_g bs[g][ch][0] --
//         what norm_param_grads_kernel does.  A workgroup owns COLS consecutive columns of ALL groups.
template <int COLS>
__global__ __launch_bounds__(256) void finalize_grads_kernel(const float* __restrict__ part, float* __restrict__ bs, int G, int strips, int C2,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  constexpr int LANES = 256 / COLS;
  __shared__ float red[LANES][COLS + 1];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + col;
  float over_g = 0.f;
  for (int g = 0; g < G; g++) {
    float a0 = 0.f, a1 = 0.f;
    if (c < C2) {
      const float* p = part + (long)g * strips * C2 + c;
      int s = sl;
      for (; s + LANES < strips; s += 2 * LANES) { a0 += p[(long)s * C2]; a1 += p[(long)(s + LANES) * C2]; }
      for (; s < strips; s += LANES) a0 += p[(long)s * C2];
    }
    red[sl][col] = a0 + a1;
    __syncthreads();
    if (sl == 0 && c < C2) {
      float t = 0.f;
      for (int q = 0; q < LANES; q++) t += red[q][col];
      bs[(long)g * C2 + c] = t;
      over_g += t;
    }
    __syncthreads();
  }
  if (sl == 0 && c < C2) {
    if (c & 1) dw[c >> 1] = over_g; else db[c >> 1] = over_g;
  }
}

// launches the strip kernel through `launch(part)` and, in two-stage mode, the finalize kernel
template <typename L>
int strip_launch(float* out, float* ws, long ws_elems, int G, long strips, int C, hipStream_t st, L launch, int first_only = 0) {
  const long need = (long)G * strips * C * 2;
  float* part = (ws && ws_elems >= need) ? ws : nullptr;   // with scratch the result is always OVERWRITTEN (no zero-fill needed)
  launch(part);
  if (part) {
    if ((long)G * C * 2 >= 4096) {
      const int chunks = (C * 2 + 31) / 32;
      hipLaunchKernelGGL(finalize_kernel<32>, dim3((unsigned)(G * chunks)), dim3(256), 0, st, (const float*)part, out, G, (int)strips, C * 2, first_only);
    } else {
      const int chunks = (C * 2 + 3) / 4;
      hipLaunchKernelGGL(finalize_kernel<4>, dim3((unsigned)(G * chunks)), dim3(256), 0, st, (const float*)part, out, G, (int)strips, C * 2, first_only);
    }
  } else if (first_only) {
    return DU_ERR_BAD_ARG;      // the compacted form needs the two-stage path (scratch)
  }
  return du_check_launch();
}

// mean / rstd from channel sums (biased variance, like InstanceNorm2d / BatchNorm2d forward); optional running-statistics
// update of BatchNorm (momentum m, unbiased variance), G == 1 only.
__global__ __launch_bounds__(256) void norm_stats_finalize_kernel(const float* __restrict__ sums, float inv_count, float eps,
                                                                  float* __restrict__ mean, float* __restrict__ rstd, long n,
                                                                  float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                  float momentum, float unbias) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float m = sums[i * 2] * inv_count;
  float var = sums[i * 2 + 1] * inv_count - m * m;
  var = var > 0.f ? var : 0.f;
  mean[i] = m;
  rstd[i] = rsqrtf(var + eps);
  if (run_mean) {
    run_mean[i] = (1.f - momentum) * run_mean[i] + momentum * m;
    run_var[i] = (1.f - momentum) * run_var[i] + momentum * var * unbias;
  }
}

// dw[c] = sum_g bs[g][c][1], db[c] = sum_g bs[g][c][0]
__global__ __launch_bounds__(256) void norm_param_grads_kernel(const float* __restrict__ bs, float* __restrict__ dw, float* __restrict__ db,
                                                               int G, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < G; g++) { b += bs[((long)g * C + c) * 2]; a += bs[((long)g * C + c) * 2 + 1]; }
  dw[c] = a; db[c] = b;
}

int grid_for(long total) { long g = (total + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

template <typename TI, typename TO>
int ln_fwd_dispatch(const void* x, long ldx, const float* w, const float* b, void* y, long ldy, float* m, float* r, long rows,
                    int D, float eps, hipStream_t st) {
  const int nvec = D / Elem<TI>::VEC;
  const int maxv = (nvec + 63) / 64;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  // rows that overflow ONE round of waves (32 per CU) by a little: two rows per wave (see layernorm_fwd2_kernel)
  static int wave_slots = 0;
  if (!wave_slots) {
    int dev = 0; hipDeviceProp_t prop;
    wave_slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? 32 * prop.multiProcessorCount : 8192;
  }
  if (g_ln_rows2 && maxv <= 4 && (g_ln_rows2 > 1 || (rows > wave_slots && rows <= 2L * wave_slots))) {
    dim3 grid2((unsigned)((rows + 7) / 8));
#define LN2_LAUNCH(MV) hipLaunchKernelGGL((layernorm_fwd2_kernel<TI, TO, MV>), grid2, block, 0, st, (const TI*)x, ldx, w, b, (TO*)y, ldy, m, r, rows, D, eps)
    if (maxv <= 1) LN2_LAUNCH(1); else if (maxv <= 2) LN2_LAUNCH(2); else LN2_LAUNCH(4);
#undef LN2_LAUNCH
    return du_check_launch();
  }
#define LN_LAUNCH(MV) hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO, MV>), grid, block, 0, st, (const TI*)x, ldx, w, b, (TO*)y, ldy, m, r, rows, D, eps)
  if (maxv <= 1) LN_LAUNCH(1); else if (maxv <= 2) LN_LAUNCH(2); else if (maxv <= 4) LN_LAUNCH(4);
  else if (maxv <= 8) LN_LAUNCH(8); else if (maxv <= 16) LN_LAUNCH(16); else return DU_ERR_UNSUPPORTED;
#undef LN_LAUNCH
  return du_check_launch();
}

template <typename T>
int ln_bwd_dispatch(const void* x, const void* dy, const float* w, const float* mean, const float* rstd, void* dx, float* dwdb,
                    long rows, int D, float* ws, long ws_elems, const void* dres, hipStream_t st) {
  const int nvec = D / Elem<T>::VEC;
  const int maxv = (nvec + 63) / 64;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  {
    // one pass (dx + the weight / bias gradient partials) when the scratch for the partials is there and a lane's columns fit in registers
    static const bool off = DU_GETENV("DU_LN_BWD_FUSED") && atoi(DU_GETENV("DU_LN_BWD_FUSED")) == 0;
    const int STRIP = pick_strip(1, rows, D, Elem<T>::VEC);
    const long strips = (rows + STRIP - 1) / STRIP;
    const size_t lds = (size_t)3 * D * 2 * sizeof(float);
    if (!off && maxv <= 4 && ws && ws_elems >= strips * D * 2 && lds <= 64 * 1024) {
      return strip_launch(dwdb, ws, ws_elems, 1, strips, D, st, [&](float* part) {
#define LNF_LAUNCH(MV) hipLaunchKernelGGL((layernorm_bwd_fused_kernel<T, MV>), dim3((unsigned)strips), block, lds, st, (const T*)x, (const T*)dy, w, mean, rstd, (T*)dx, rows, D, (const T*)dres, STRIP, part)
        if (maxv <= 1) LNF_LAUNCH(1); else if (maxv <= 2) LNF_LAUNCH(2); else LNF_LAUNCH(4);
#undef LNF_LAUNCH
      }, 2);
    }
  }
#define LNB_LAUNCH(MV) hipLaunchKernelGGL((layernorm_bwd_dx_kernel<T, MV>), grid, block, 0, st, (const T*)x, (const T*)dy, w, mean, rstd, (T*)dx, rows, D, (const T*)dres)
  if (maxv <= 1) LNB_LAUNCH(1); else if (maxv <= 2) LNB_LAUNCH(2); else if (maxv <= 4) LNB_LAUNCH(4);
  else if (maxv <= 8) LNB_LAUNCH(8); else if (maxv <= 16) LNB_LAUNCH(16); else return DU_ERR_UNSUPPORTED;
#undef LNB_LAUNCH
  const int STRIP = pick_strip(1, rows, D, Elem<T>::VEC);
  long strips = (rows + STRIP - 1) / STRIP;
  // with scratch the result is PLANAR (dw[D] then db[D]) so both gradients are contiguous tensors; the atomics fallback keeps the
  // interleaved (D, 2) layout
  return strip_launch(dwdb, ws, ws_elems, 1, strips, D, st, [&](float* part) {
    hipLaunchKernelGGL(layernorm_bwd_wb_kernel<T>, dim3((unsigned)strips), block, 0, st, (const T*)x, (const T*)dy, mean, rstd, dwdb, rows, D, STRIP, part);
  }, (ws && ws_elems >= (long)strips * D * 2) ? 2 : 0);
}

}  // namespace

extern "C" int du_layernorm_fwd(int in_dtype, int out_dtype, const void* x, int64_t ldx, const float* w, const float* b, void* y,
                                int64_t ldy, float* mean_out, float* rstd_out, int64_t rows, int D, float eps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 0 || D <= 0 || !x || !y || !w || !b) return DU_ERR_BAD_ARG;
  const int vi = in_dtype == DU_BF16 ? 8 : 4;
  if (D % vi || ldx % vi) return DU_ERR_BAD_ARG;
  if (in_dtype == DU_F32 && out_dtype == DU_F32) return ln_fwd_dispatch<float, float>(x, ldx, w, b, y, ldy, mean_out, rstd_out, rows, D, eps, st);
  if (in_dtype == DU_F32 && out_dtype == DU_BF16) return ln_fwd_dispatch<float, bf16_t>(x, ldx, w, b, y, ldy, mean_out, rstd_out, rows, D, eps, st);
  if (in_dtype == DU_BF16 && out_dtype == DU_BF16) return ln_fwd_dispatch<bf16_t, bf16_t>(x, ldx, w, b, y, ldy, mean_out, rstd_out, rows, D, eps, st);
  if (in_dtype == DU_BF16 && out_dtype == DU_F32) return ln_fwd_dispatch<bf16_t, float>(x, ldx, w, b, y, ldy, mean_out, rstd_out, rows, D, eps, st);
  return DU_ERR_BAD_ARG;
}

extern "C" int du_strip_finalize(const float* part, float* out, int G, int strips, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!part || !out || G <= 0 || strips <= 0 || C <= 0) return DU_ERR_BAD_ARG;
  if ((long)G * C * 2 >= 4096) {
    const int chunks = (C * 2 + 31) / 32;
    hipLaunchKernelGGL(finalize_kernel<32>, dim3((unsigned)(G * chunks)), dim3(256), 0, st, part, out, G, strips, C * 2, 0);
  } else {
    const int chunks = (C * 2 + 3) / 4;
    hipLaunchKernelGGL(finalize_kernel<4>, dim3((unsigned)(G * chunks)), dim3(256), 0, st, part, out, G, strips, C * 2, 0);
  }
  return du_check_launch();
}

extern "C" int64_t du_reduce_ws_elems(int dtype, int G, int64_t P, int C) {
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C <= 0 || C % v) return 0;
  const int STRIP = pick_strip(G, P, C, v);
  return (int64_t)G * ((P + STRIP - 1) / STRIP) * C * 2;
}

extern "C" int du_layernorm_bwd(int dtype, const void* x, const void* dy, const float* w, const float* mean, const float* rstd,
                                void* dx, float* dwdb, int64_t rows, int D, float* ws, int64_t ws_elems, const void* dres, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 0 || D <= 0 || !x || !dy || !dx || !dwdb) return DU_ERR_BAD_ARG;
  const int vi = dtype == DU_BF16 ? 8 : 4;
  if (D % vi) return DU_ERR_BAD_ARG;
  if (dtype == DU_F32) return ln_bwd_dispatch<float>(x, dy, w, mean, rstd, dx, dwdb, rows, D, ws, ws_elems, dres, st);
  if (dtype == DU_BF16) return ln_bwd_dispatch<bf16_t>(x, dy, w, mean, rstd, dx, dwdb, rows, D, ws, ws_elems, dres, st);
  return DU_ERR_BAD_ARG;
}

extern "C" int du_chan_stats(int dtype, const void* x, int64_t ldx, float* sums, int G, int64_t P, int C, float* ws, int64_t ws_elems,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C <= 0 || C % v || ldx % v || !x || !sums) return DU_ERR_BAD_ARG;
  const int STRIP = pick_strip(G, P, C, v);
  long strips = (P + STRIP - 1) / STRIP;
  dim3 grid((unsigned)(G * strips)), block(256);
  if (dtype != DU_BF16 && dtype != DU_F32) return DU_ERR_BAD_ARG;
  return strip_launch(sums, ws, ws_elems, G, strips, C, st, [&](float* part) {
    if (dtype == DU_BF16) hipLaunchKernelGGL(chan_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, sums, G, P, C, STRIP, part);
    else hipLaunchKernelGGL(chan_stats_kernel<float>, grid, block, 0, st, (const float*)x, ldx, sums, G, P, C, STRIP, part);
  });
}

// column sums only: out[c] = sum_rows x[row][c]  (bias gradients); scratch du_reduce_ws_elems(dtype, 1, rows, C) is required
extern "C" int du_colsum(int dtype, const void* x, int64_t ldx, float* out, int64_t rows, int C, float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (rows <= 0 || C <= 0 || C % v || ldx % v || !x || !out || !ws) return DU_ERR_BAD_ARG;
  if (dtype != DU_BF16 && dtype != DU_F32) return DU_ERR_BAD_ARG;
  const int STRIP = pick_strip(1, rows, C, v);
  long strips = (rows + STRIP - 1) / STRIP;
  dim3 grid((unsigned)strips), block(256);
  return strip_launch(out, ws, ws_elems, 1, strips, C, st, [&](float* part) {
    if (dtype == DU_BF16) hipLaunchKernelGGL(chan_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, out, 1, (long)rows, C, STRIP, part);
    else hipLaunchKernelGGL(chan_stats_kernel<float>, grid, block, 0, st, (const float*)x, ldx, out, 1, (long)rows, C, STRIP, part);
  }, 1);
}

extern "C" int du_chan_dot(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, float* sums, int G, int64_t P, int C,
                           float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C <= 0 || C % v || lda % v || ldb % v || !a || !b || !sums) return DU_ERR_BAD_ARG;
  const int STRIP = pick_strip(G, P, C, v);
  long strips = (P + STRIP - 1) / STRIP;
  dim3 grid((unsigned)(G * strips)), block(256);
  if (dtype != DU_BF16 && dtype != DU_F32) return DU_ERR_BAD_ARG;
  return strip_launch(sums, ws, ws_elems, G, strips, C, st, [&](float* part) {
    if (dtype == DU_BF16) hipLaunchKernelGGL(chan_dot_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, sums, G, P, C, STRIP, part);
    else hipLaunchKernelGGL(chan_dot_kernel<float>, grid, block, 0, st, (const float*)a, lda, (const float*)b, ldb, sums, G, P, C, STRIP, part);
  });
}

extern "C" int du_norm_stats_finalize(const float* sums, float count, float eps, float* mean, float* rstd, int G, int C, float* run_mean,
                                      float* run_var, float momentum, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!sums || !mean || !rstd || G <= 0 || C <= 0 || count <= 0.f || (run_mean && G != 1) || (!run_mean != !run_var)) return DU_ERR_BAD_ARG;
  const long n = (long)G * C;
  const float unbias = count > 1.f ? count / (count - 1.f) : 1.f;
  hipLaunchKernelGGL(norm_stats_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sums, 1.f / count, eps, mean, rstd, n,
                     run_mean, run_var, momentum, unbias);
  return du_check_launch();
}

extern "C" int du_norm_param_grads(const float* bsums, float* dw, float* db, int G, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!bsums || !dw || !db || G <= 0 || C <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(norm_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, bsums, dw, db, G, C);
  return du_check_launch();
}

extern "C" int du_norm_act_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* mean, const float* rstd,
                               const float* w, const float* b, int G, int64_t P, int C, int act, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C % v || ldx % v || ldy % v) return DU_ERR_BAD_ARG;
  dim3 grid(norm_slots(G, P, C, v), G), block(256);
  if (dtype == DU_BF16) hipLaunchKernelGGL(norm_act_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, mean, rstd, w, b, (long)P, C, act);
  else if (dtype == DU_F32) hipLaunchKernelGGL(norm_act_fwd_kernel<float>, grid, block, 0, st, (const float*)x, ldx, (float*)y, ldy, mean, rstd, w, b, (long)P, C, act);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}

extern "C" int du_norm_act_bwd_stats(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* mean,
                                     const float* rstd, const float* w, const float* b, float* bsums, int G, int64_t P, int C,
                                     int act, float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C % v || ldx % v || lddy % v) return DU_ERR_BAD_ARG;
  const int STRIP = pick_strip(G, P, C, v);
  long strips = (P + STRIP - 1) / STRIP;
  dim3 grid((unsigned)(G * strips)), block(256);
  if (dtype != DU_BF16 && dtype != DU_F32) return DU_ERR_BAD_ARG;
  return strip_launch(bsums, ws, ws_elems, G, strips, C, st, [&](float* part) {
    static const int deep = DU_GETENV("DU_NORM_BWD_UNROLL") ? atoi(DU_GETENV("DU_NORM_BWD_UNROLL")) : 4;     // A-B aid: 2 = the round-2 loop
    if (dtype == DU_BF16) {
      if (deep >= 4) hipLaunchKernelGGL((norm_act_bwd_stats_kernel<bf16_t, 4>), grid, block, 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, mean, rstd, w, b, bsums, G, P, C, act, STRIP, part);
      else hipLaunchKernelGGL((norm_act_bwd_stats_kernel<bf16_t, 2>), grid, block, 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, mean, rstd, w, b, bsums, G, P, C, act, STRIP, part);
    } else hipLaunchKernelGGL((norm_act_bwd_stats_kernel<float, 2>), grid, block, 0, st, (const float*)x, ldx, (const float*)dy, lddy, mean, rstd, w, b, bsums, G, P, C, act, STRIP, part);
  });
}

namespace {
int launch_finalize_stats(const float* part, float* sums, int G, long strips, int C, float count, float eps, float* mean, float* rstd,
                          float* run_mean, float* run_var, float momentum, hipStream_t st) {
  const float unbias = count > 1.f ? count / (count - 1.f) : 1.f;
  if ((long)G * C * 2 >= 4096) {
    const int chunks = (C * 2 + 31) / 32;
    hipLaunchKernelGGL(finalize_stats_kernel<32>, dim3((unsigned)(G * chunks)), dim3(256), 0, st, part, sums, G, (int)strips, C * 2, 1.f / count, eps,
                       mean, rstd, run_mean, run_var, momentum, unbias);
  } else {
    const int chunks = (C * 2 + 3) / 4;
    hipLaunchKernelGGL(finalize_stats_kernel<4>, dim3((unsigned)(G * chunks)), dim3(256), 0, st, part, sums, G, (int)strips, C * 2, 1.f / count, eps,
                       mean, rstd, run_mean, run_var, momentum, unbias);
  }
  return du_check_launch();
}
}  // namespace

// du_strip_finalize + du_norm_stats_finalize in one launch (statistics partials from a convolution epilogue); sums (nullable) also gets
// the (G, C, 2) totals
extern "C" int du_strip_finalize_norm(const float* part, float* sums, int G, int strips, int C, float count, float eps, float* mean,
                                      float* rstd, float* run_mean, float* run_var, float momentum, void* stream) {
  if (!part || !mean || !rstd || G <= 0 || strips <= 0 || C <= 0 || count <= 0.f) return DU_ERR_BAD_ARG;
  if (run_mean && (G != 1 || !run_var)) return DU_ERR_BAD_ARG;
  return launch_finalize_stats(part, sums, G, strips, C, count, eps, mean, rstd, run_mean, run_var, momentum, (hipStream_t)stream);
}

// du_chan_stats + du_norm_stats_finalize: strip partials, then ONE kernel for totals, mean / rstd and the running statistics.  Needs the
// scratch (du_reduce_ws_elems); count = pixels per group the statistics are taken over
extern "C" int du_chan_stats_norm(int dtype, const void* x, int64_t ldx, float* sums, int G, int64_t P, int C, float* ws, int64_t ws_elems,
                                  float count, float eps, float* mean, float* rstd, float* run_mean, float* run_var, float momentum,
                                  void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C <= 0 || C % v || ldx % v || !x || !mean || !rstd || count <= 0.f) return DU_ERR_BAD_ARG;
  if (dtype != DU_BF16 && dtype != DU_F32) return DU_ERR_BAD_ARG;
  if (run_mean && (G != 1 || !run_var)) return DU_ERR_BAD_ARG;
  const int STRIP = pick_strip(G, P, C, v);
  const long strips = (P + STRIP - 1) / STRIP;
  if (!ws || ws_elems < (long)G * strips * C * 2) return DU_ERR_BAD_ARG;
  dim3 grid((unsigned)(G * strips)), block(256);
  if (dtype == DU_BF16) hipLaunchKernelGGL(chan_stats_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, sums, G, P, C, STRIP, ws);
  else hipLaunchKernelGGL(chan_stats_kernel<float>, grid, block, 0, st, (const float*)x, ldx, sums, G, P, C, STRIP, ws);
  return launch_finalize_stats(ws, sums, G, strips, C, count, eps, mean, rstd, run_mean, run_var, momentum, st);
}

// du_norm_act_bwd_stats + du_norm_param_grads: strip partials, then ONE kernel for bsums (G, C, 2), dw (C) and db (C).  Needs the scratch
extern "C" int du_norm_act_bwd_stats_grads(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* mean,
                                           const float* rstd, const float* w, const float* b, float* bsums, float* dw, float* db, int G,
                                           int64_t P, int C, int act, float* ws, int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C <= 0 || C % v || ldx % v || lddy % v || !bsums || !dw || !db) return DU_ERR_BAD_ARG;
  if (dtype != DU_BF16 && dtype != DU_F32) return DU_ERR_BAD_ARG;
  const int STRIP = pick_strip(G, P, C, v);
  const long strips = (P + STRIP - 1) / STRIP;
  if (!ws || ws_elems < (long)G * strips * C * 2) return DU_ERR_BAD_ARG;
  dim3 grid((unsigned)(G * strips)), block(256);
  static const int deep = DU_GETENV("DU_NORM_BWD_UNROLL") ? atoi(DU_GETENV("DU_NORM_BWD_UNROLL")) : 4;     // A-B aid: 2 = the round-2 loop
  if (dtype == DU_BF16) {
    if (deep >= 4) hipLaunchKernelGGL((norm_act_bwd_stats_kernel<bf16_t, 4>), grid, block, 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, mean, rstd, w, b, bsums, G, P, C, act, STRIP, ws);
    else hipLaunchKernelGGL((norm_act_bwd_stats_kernel<bf16_t, 2>), grid, block, 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, mean, rstd, w, b, bsums, G, P, C, act, STRIP, ws);
  } else hipLaunchKernelGGL((norm_act_bwd_stats_kernel<float, 2>), grid, block, 0, st, (const float*)x, ldx, (const float*)dy, lddy, mean, rstd, w, b, bsums, G, P, C, act, STRIP, ws);
  if ((long)C * 2 >= 512) hipLaunchKernelGGL(finalize_grads_kernel<32>, dim3((unsigned)((C * 2 + 31) / 32)), dim3(256), 0, st, (const float*)ws, bsums, G, (int)strips, C * 2, dw, db);
  else hipLaunchKernelGGL(finalize_grads_kernel<4>, dim3((unsigned)((C * 2 + 3) / 4)), dim3(256), 0, st, (const float*)ws, bsums, G, (int)strips, C * 2, dw, db);
  return du_check_launch();
}

extern "C" int du_norm_act_bwd_dx(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx,
                                  const float* mean, const float* rstd, const float* w, const float* b, const float* bsums, int G,
                                  int64_t P, int C, int act, float count, int use_batch_stats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int v = dtype == DU_BF16 ? 8 : 4;
  if (G <= 0 || P <= 0 || C % v || ldx % v || lddy % v || lddx % v) return DU_ERR_BAD_ARG;
  dim3 grid(norm_slots(G, P, C, v, true), G), block(256);
  const float inv = count > 0 ? 1.0f / count : 0.f;
  if (dtype == DU_BF16) hipLaunchKernelGGL(norm_act_bwd_dx_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, mean, rstd, w, b, bsums, (long)P, C, act, inv, use_batch_stats);
  else if (dtype == DU_F32) hipLaunchKernelGGL(norm_act_bwd_dx_kernel<float>, grid, block, 0, st, (const float*)x, ldx, (const float*)dy, lddy, (float*)dx, lddx, mean, rstd, w, b, bsums, (long)P, C, act, inv, use_batch_stats);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}
