// ViT self-attention for gfx950: RoPE + head split, and a flash-style fused softmax(QK^T)V forward on MFMA.
//
// The attention kernel computes TRANSPOSED products so that the query index lives on the lane axis of every
// accumulator (wave64, v_mfma_f32_32x32x16_bf16):
//     S^T[key][q] = sum_d K[key][d] * Q[q][d]       A = K tile (LDS),           B = Q (registers, loaded once)
//     O^T[dv][q]  = sum_key V^T[dv][key] * P^T[key][q]   A = V^T tile (LDS),     B = P^T = the S^T accumulators
// The S^T accumulator register r of lane (q = lane&31, half = lane>>5) holds key (r&3) + 8*(r>>2) + 4*half of a
// 32-key block; exponentiated and packed to bf16, registers 8s..8s+7 ARE the B fragment of k-step s, so P never
// moves between lanes or through LDS.  The contraction order over keys is permuted accordingly on the V^T side
// (two 8-byte LDS reads per fragment).  Row max needs one cross-half exchange per tile; the row sum is kept as
// a per-lane partial and combined once at the end; the O rescale factor is lane-local.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// qkv (B, N, 3, H, Dh) -> q, k, v (B, H, Npad, Dh) with RoPE on q, k for tokens >= prefix (layers/attention.py:66-85)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_split_kernel(const T* __restrict__ qkv, T* __restrict__ q, T* __restrict__ k,
                                                             T* __restrict__ v, const float* __restrict__ sin_t,
                                                             const float* __restrict__ cos_t, int B, int N, int Npad, int H,
                                                             int Dh, int prefix, float qscale, long total) {
  constexpr int V = Elem<T>::VEC;
  const int dv = Dh / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int d0 = (int)(i % dv) * V;
    long t = i / dv;
    const int h = (int)(t % H); t /= H;
    const int which = (int)(t % 3); t /= 3;
    const int n = (int)(t % N);
    const int b = (int)(t / N);
    const T* src = qkv + ((((long)b * N + n) * 3 + which) * H + h) * Dh;
    Vec16<T> x = as_vec<T>(*(const uint4*)(src + d0));
    float o[V];
#pragma unroll
    for (int j = 0; j < V; j++) o[j] = to_f32(x.v[j]);
    if (which < 2 && n >= prefix) {
      const int half = Dh / 2;
      const int dp = d0 < half ? d0 + half : d0 - half;
      const float sgn = d0 < half ? -1.f : 1.f;
      Vec16<T> y = as_vec<T>(*(const uint4*)(src + dp));
      const float* sp = sin_t + (long)(n - prefix) * Dh + d0;
      const float* cp = cos_t + (long)(n - prefix) * Dh + d0;
#pragma unroll
      for (int j = 0; j < V; j++) o[j] = o[j] * cp[j] + sgn * to_f32(y.v[j]) * sp[j];
    }
    if (which == 0) {
#pragma unroll
      for (int j = 0; j < V; j++) o[j] *= qscale;
    }
    T* dst = (which == 0 ? q : (which == 1 ? k : v)) + (((long)b * H + h) * Npad + n) * Dh + d0;
    Vec16<T> r;
#pragma unroll
    for (int j = 0; j < V; j++) r.v[j] = from_f32<T>(o[j]);
    *(uint4*)dst = as_u4(r);
  }
}

// ------------------------------------------------------------------------------------------------------
// flash attention forward, bf16 in / fp32 accumulate / bf16 out
// ------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int N,
                                                       int Npad) {
  constexpr int KT = 64;             // keys per tile
  constexpr int KLD = DH + 8;        // K image [key][DH + 8]: 16-byte-padded rows, conflict-free ds_read_b128 fragments
  constexpr int VLD = DH + 32;       // V image [key][DH + 32]: row pitch = 64 B (mod 256 B) so the 4 key rows of one LDS transpose
                                     // read (ds_read_b64_tr_b16) fall on disjoint bank groups; V is stored as it lies in HBM
  constexpr int NKK = DH / 16;       // k-steps of the S^T product
  constexpr int NDB = DH / 32;       // 32-wide dv blocks of O^T
  constexpr int BUF = KT * KLD + KT * VLD;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BUF];   // two {K, V} buffers: one barrier per key tile
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_v4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, ql = lane & 31;
  // XCD-aware order: workgroup b lands on XCD b % 8, so with the plain (q tile, head) grid the 9 query tiles of one head sat on 8
  // different XCDs and every XCD pulled every head's K / V through its own L2 (FETCH_SIZE 5.5x the algorithmic bytes).  All query
  // tiles of a head now share an XCD: heads are dealt round-robin to the XCDs, the tiles of a head are consecutive local indices.
  int bh = blockIdx.y, qt = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    bh = (idx / (int)gridDim.x) * 8 + xcd;
    qt = idx % (int)gridDim.x;
  }
  const int q0 = qt * 128 + wave * 32;
  const bool active = q0 < N;        // wave-uniform: the last query tile of N = 1029 keeps only one wave busy
  const bf16_t* Qb = Q + (long)bh * Npad * DH;
  const bf16_t* Kb = K + (long)bh * Npad * DH;
  const bf16_t* Vb = V + (long)bh * Npad * DH;

  // Q fragments (B operand): lane (q, half) holds Q[q][kk*16 + half*8 .. +8]
  bf16x8 qf[NKK];
  {
    int qr = q0 + ql; if (qr > N - 1) qr = N - 1;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) qf[kk] = *(const bf16x8*)(Qb + (long)qr * DH + kk * 16 + half * 8);
  }
  f32x16 acc_o[NDB];
#pragma unroll
  for (int d = 0; d < NDB; d++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc_o[d][r] = 0.f;
  float m_run = -1e30f, l_part = 0.f;

  constexpr int NVEC = KT * DH / 8 / 256;   // 16-byte vectors per thread per tile (2 for DH=64, 4 for DH=128)
  uint4 kreg[NVEC], vreg[NVEC];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NVEC; i++) {
      int v = tid + i * 256;
      int key = v / (DH / 8), d8 = v % (DH / 8);
      int kg = kt * KT + key;
      if (kg < N) {
        kreg[i] = *(const uint4*)(Kb + (long)kg * DH + d8 * 8);
        vreg[i] = *(const uint4*)(Vb + (long)kg * DH + d8 * 8);
      } else {
        kreg[i] = make_uint4(0, 0, 0, 0);
        vreg[i] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* Ks = smem + buf * BUF;
    bf16_t* Vs = Ks + KT * KLD;
#pragma unroll
    for (int i = 0; i < NVEC; i++) {
      int v = tid + i * 256;
      int key = v / (DH / 8), d8 = v % (DH / 8);
      *(uint4*)(Ks + key * KLD + d8 * 8) = kreg[i];
      *(uint4*)(Vs + key * VLD + d8 * 8) = vreg[i];
    }
  };

  const int ntiles = (N + KT - 1) / KT;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; kt++) {
    const bool more = kt + 1 < ntiles;
    if (more) gload(kt + 1);
    const bf16_t* Ks = smem + (kt & 1) * BUF;
    const bf16_t* Vs = Ks + KT * KLD;
    if (active) {
      // ---- S^T = K Q^T for the two 32-key blocks ----
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
#pragma unroll
        for (int r = 0; r < 16; r++) s[kb][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) {
          bf16x8 kf = *(const bf16x8*)(Ks + (kb * 32 + ql) * KLD + kk * 16 + half * 8);
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[kb], 0, 0, 0);
        }
      }
      // ---- mask the tail keys (last tile only), online softmax in base 2 (q was pre-scaled by Dh^-1/2 * log2 e) ----
      if (!more) {
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int key = kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= N) s[kb][r] = -1e30f;
          }
      }
      float mx = s[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // deferred rescale: keep the old running max while the tile's max exceeds it by <= 8 (P <= 2^8, exact in the fp32
      // accumulators, bf16 P keeps its relative precision); the branch is wave-uniform
      if (!__all(mx - m_run <= 8.0f)) {
        const float mn = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - mn);
        m_run = mn;
        l_part *= alpha;
#pragma unroll
        for (int d = 0; d < NDB; d++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc_o[d][r] *= alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float p = __builtin_amdgcn_exp2f(s[kb][r] - m_run);
          s[kb][r] = p;
          psum += p;
        }
      l_part += psum;

      // ---- O^T += V^T P^T: B fragment of k-step st = accumulator registers 8st..8st+7 of s[kb] (keys kb*32 + 16st + 4half + {0..3, 8..11});
      //      A fragment = the matching V rows, fetched with two LDS transpose reads ----
      const int g = lane >> 4, p16 = lane & 15;
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
#pragma unroll
        for (int st = 0; st < 2; st++) {
          bf16x8 pf;
#pragma unroll
          for (int e = 0; e < 8; e++) pf[e] = (bf16_t)s[kb][8 * st + e];
#pragma unroll
          for (int d = 0; d < NDB; d++) {
            const bf16_t* vp = Vs + (kb * 32 + 16 * st + 4 * (g >> 1) + (p16 >> 2)) * VLD + d * 32 + 16 * (g & 1) + 4 * (p16 & 3);
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)vp);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(vp + 8 * VLD));
            s16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf, acc_o[d], 0, 0, 0);
          }
        }
      }
    }
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }

  // ---- finalize: O[q][dv] = O^T[dv][q] / l ----
  const float l = l_part + __shfl_xor(l_part, 32, 64);
  const float inv = 1.f / l;
  const int q = q0 + ql;
  if (q < N) {
    const int b = bh / H, h = bh % H;
    bf16_t* op = O + ((long)b * N + q) * ((long)H * DH) + (long)h * DH;
#pragma unroll
    for (int d = 0; d < NDB; d++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; e++) o4[e] = (bf16_t)(acc_o[d][g4 * 4 + e] * inv);
        *(bf16x4*)(op + d * 32 + 8 * g4 + 4 * half) = o4;
      }
  }
}

// row softmax, fp32, in place; pad columns [cols, ld) are zeroed (parity-mode attention: scores materialised)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long rows, int cols, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = x + row * ld;
  float mx = -1e30f;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) { float e = expf(p[c] - mx); p[c] = e; s += e; }
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < (int)ld; c += 64) p[c] = c < cols ? p[c] * inv : 0.f;
}

}  // namespace

extern "C" int du_qkv_rope_split(int dtype, const void* qkv, void* q, void* k, void* v, const float* sin_t, const float* cos_t,
                                 int B, int N, int Npad, int H, int Dh, int prefix, float qscale, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == DU_BF16 ? 8 : 4;
  if (!qkv || !q || !k || !v || !sin_t || !cos_t || B <= 0 || N <= 0 || Npad < N || H <= 0 || Dh <= 0 || (Dh / 2) % vec || prefix < 0 ||
      prefix > N)
    return DU_ERR_BAD_ARG;
  long total = (long)B * N * 3 * H * (Dh / vec);
  long g = (total + 255) / 256; if (g > 65535 * 4) g = 65535 * 4;
  if (dtype == DU_BF16)
    hipLaunchKernelGGL(qkv_rope_split_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)q, (bf16_t*)k,
                       (bf16_t*)v, sin_t, cos_t, B, N, Npad, H, Dh, prefix, qscale, total);
  else if (dtype == DU_F32)
    hipLaunchKernelGGL(qkv_rope_split_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const float*)qkv, (float*)q, (float*)k,
                       (float*)v, sin_t, cos_t, B, N, Npad, H, Dh, prefix, qscale, total);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}

extern "C" int du_attention_fwd(const void* q, const void* k, const void* v, void* out, int B, int H, int N, int Npad, int Dh,
                                void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || N <= 0 || Npad < N) return DU_ERR_BAD_ARG;
  dim3 grid((N + 127) / 128, B * H), block(256);
  if (Dh == 64)
    hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, H, N, Npad);
  else if (Dh == 128)
    hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, H, N, Npad);
  else return DU_ERR_UNSUPPORTED;
  return du_check_launch();
}

extern "C" int du_softmax_rows_f32(float* x, int64_t rows, int cols, int64_t ld, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || rows <= 0 || cols <= 0 || ld < cols) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, (long)rows, cols, (long)ld);
  return du_check_launch();
}
