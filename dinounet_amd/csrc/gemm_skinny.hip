// Skinny-M "NT" product for the ragged last rows of a tall bf16 GEMM (gfx950).
//
// Why: the ViT products of dinounet_l have M = 8 * 1029 = 64 * 128 + 40 rows.  With 128 x 128 tiles the 65th tile row adds N/128
// workgroups to a grid that otherwise fills the 512 resident slots (2 per CU) a whole number of times, and those few workgroups run
// a full K loop alone: measured +40 % (N 1024, K 1024), +34 % (N 1024, K 4096), +13 % (N 3072, K 1024) over the same product with
// M = 8192 (tools/gemm_ragged.py).  gemm_bf16.hip therefore hands the first M - r rows to the tile kernel and the last r <= 64 rows
// to this pair of kernels, which parallelise over N AND K instead of M:
//   (1) gemm_skinny_partial_kernel: workgroup = (32 output columns, one K slice), its 4 waves take interleaved 64-wide K chunks.
//       No LDS staging: a lane loads 64 contiguous bytes (32 k) of "its" row of A (m = lane & 31 of each 32-row block) and of B
//       (n = lane & 31) straight into MFMA fragments -- lanes 0..31 carry k [0,32), lanes 32..63 k [32,64) of the chunk, and the j-th
//       v_mfma_f32_32x32x16_bf16 of the chunk consumes bytes [16j, 16j+16) of both, so A and B agree on the contraction index.
//       The 4 wave accumulators are summed through an 8 KB LDS tile and written as one fp32 partial [slice][64][N].
//   (2) gemm_skinny_finish_kernel: sums the slices and applies the du_gemm epilogue (alpha, bias, act, gamma, row_scale, residual).
// Scratch: du_gemm_ws_elems() floats lent by the caller (du_gemm_args.ws).
#include "gemm_params.h"
#include "gemm_skinny_body.h"

namespace {

__global__ __launch_bounds__(256) void gemm_skinny_partial_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                                  long ldb, float* __restrict__ part, int M, int N, int K,
                                                                  int k_per_slice, SkinnyEpi P, int fused) {
  __shared__ float red[64][SK_BN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * SK_BN;
  const int k0 = blockIdx.y * k_per_slice, k1 = min(K, k0 + k_per_slice);
  const int nrb = (M + 31) >> 5;                       // 32-row blocks in use (1 or 2)
  const int kh = (lane >> 5) * 32;                     // this half-wave's k offset inside a chunk
  const int n = n0 + (lane & 31);
  const bf16_t* bp = B + (long)min(n, N - 1) * ldb + kh;
  const bf16_t* ap0 = A + (long)min(lane & 31, M - 1) * lda + kh;
  const bf16_t* ap1 = A + (long)min(32 + (lane & 31), M - 1) * lda + kh;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll 2
  for (int k = k0 + wave * SK_CHUNK; k < k1; k += 4 * SK_CHUNK) {
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
  // sum the 4 waves: wave w adds its tile in round w (D layout: column lane & 31, row (r&3) + 8*(r>>2) + 4*(lane>>5))
  for (int w = 0; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float* d0 = &red[row][lane & 31];
        float* d1 = &red[32 + row][lane & 31];
        *d0 = w ? *d0 + acc0[r] : acc0[r];
        *d1 = w ? *d1 + acc1[r] : acc1[r];
      }
    }
    __syncthreads();
  }
  if (fused) {
    // single K slice: the du_gemm epilogue right here (alpha, bias, act, gamma, row_scale, residual), no partial buffer, no second launch
    for (int i = tid; i < 64 * (SK_BN / 4); i += 256) {
      const int m = i / (SK_BN / 4), c = (i % (SK_BN / 4)) * 4;
      const int n = n0 + c;
      if (m >= M || n >= N) continue;
      float o[4] = {red[m][c] * P.alpha, red[m][c + 1] * P.alpha, red[m][c + 2] * P.alpha, red[m][c + 3] * P.alpha};
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
      if (P.out_bf16) {
        if (P.residual) {
          const bf16_t* rp = (const bf16_t*)P.residual + (long)m * P.ldr + n;
#pragma unroll
          for (int e = 0; e < 4; e++) o[e] += (float)rp[e];
        }
        bf16x4 t;
#pragma unroll
        for (int e = 0; e < 4; e++) t[e] = (bf16_t)o[e];
        *(uint2*)((bf16_t*)P.C + (long)m * P.ldc + n) = __builtin_bit_cast(uint2, t);
      } else {
        if (P.residual) {
          const float4 rr = *(const float4*)((const float*)P.residual + (long)m * P.ldr + n);
          o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
        }
        *(float4*)((float*)P.C + (long)m * P.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    return;
  }
  float* dst = part + (long)blockIdx.y * 64 * N;
  for (int i = tid; i < 64 * SK_BN; i += 256) {
    const int m = i >> 5, c = i & 31;
    if (m < M && n0 + c < N) dst[(long)m * N + n0 + c] = red[m][c];
  }
}

// Single-launch form: one workgroup = 32 output columns x the WHOLE contraction, NW waves each taking every NW-th 64-wide K chunk (so a
// K = 4096 product keeps 16 waves x 12 sixteen-byte loads per lane in flight instead of 4 waves walking 16 chunks each: the 40-row
// tails are latency-bound, 96 of them per dinounet_l step).  Every wave parks its 64 x 32 fp32 tile in its own LDS slot; one barrier;
// the epilogue threads sum the NW slots for their 4 columns and apply the du_gemm epilogue (alpha, bias, act, gamma, row_scale,
// residual) -- no partial buffer, no second launch.
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_fused_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                                    long ldb, int M, int N, int K, SkinnyEpi P) {
  extern __shared__ __attribute__((aligned(16))) float sk_red[];      // [NW][64][SK_BN + 1]
  skinny_fused_body<NW, (NW >= 16 ? 1 : 2)>(A, lda, B, ldb, M, N, K, P, sk_red, (int)blockIdx.x);      // (16 waves: 128 registers per lane)
}

template <int NW>
int launch_skinny_fused(const du_gemm_args& a, const SkinnyEpi& E, hipStream_t st) {
  constexpr int LDS = NW * 64 * (SK_BN + 1) * 4;
  auto kfn = gemm_skinny_fused_kernel<NW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3((a.N + SK_BN - 1) / SK_BN), dim3(NW * 64), LDS, st, (const bf16_t*)a.A, (long)a.lda, (const bf16_t*)a.B,
                     (long)a.ldb, a.M, a.N, a.K, E);
  return du_check_launch();
}

template <typename TC>
__global__ __launch_bounds__(256) void gemm_skinny_finish_kernel(const float* __restrict__ part, int slices, GemmParams P) {
  const int n4 = P.N >> 2;
  const long total = (long)P.M * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    float4 t = *(const float4*)(part + (long)m * P.N + n);
    for (int s = 1; s < slices; s++) {
      const float4 u = *(const float4*)(part + ((long)s * 64 + m) * P.N + n);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float o[4] = {t.x * P.alpha, t.y * P.alpha, t.z * P.alpha, t.w * P.alpha};
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
    if (P.residual) {
      const TC* rp = (const TC*)P.residual + (long)m * P.ldr + n;
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] += to_f32(rp[e]);
    }
    TC* cp = (TC*)P.C + (long)m * P.ldc + n;
#pragma unroll
    for (int e = 0; e < 4; e++) cp[e] = from_f32<TC>(o[e]);
  }
}

inline int skinny_slices(int N, int K) {
  // ~1024 workgroups, at least one 64-wide chunk per wave
  const int ntiles = (N + SK_BN - 1) / SK_BN;
  int s = 1024 / ntiles;
  const int smax = K / (4 * SK_CHUNK);
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

}  // namespace

// scratch floats for du_gemm_skinny on an (M <= 64) x N x K product
int64_t du_gemm_skinny_ws_elems(int N, int K) { return (int64_t)skinny_slices(N, K) * 64 * N; }

// C[m][n] = epilogue(sum_k A[m][k] B[n][k]) for M <= 64 rows; plain row-major A, B (k contiguous), plain store.
// Returns DU_ERR_UNSUPPORTED when the shape is not served (caller falls back to the tile kernels).
int du_gemm_skinny(const du_gemm_args& a, hipStream_t st) {
  if (a.dtype != DU_BF16 || a.a_mode != DU_PLAIN_ROW || a.b_mode != DU_PLAIN_ROW || a.store_mode != DU_STORE_PLAIN) return DU_ERR_UNSUPPORTED;
  if (a.M < 1 || a.M > 64 || a.K % SK_CHUNK || a.N % 4 || a.batch > 1 || a.split_k > 1 || a.lda % 8 || a.ldb % 8) return DU_ERR_UNSUPPORTED;
  if ((((uintptr_t)a.A) | ((uintptr_t)a.B)) & 15) return DU_ERR_UNSUPPORTED;
  SkinnyEpi E{};
  E.C = a.C; E.ldc = a.ldc; E.residual = a.residual; E.ldr = a.ldr; E.bias = a.bias; E.gamma = a.gamma; E.row_scale = a.row_scale;
  E.alpha = a.alpha; E.act = a.act; E.rs_rows = a.rs_rows > 0 ? a.rs_rows : 1; E.out_bf16 = a.out_dtype == DU_BF16;
  static const bool no_fuse = DU_GETENV("DU_SKINNY_NO_FUSE") != nullptr;     // A-B aid
  // K > 2048 (fc2 of the ViT, K = 4096): a fragment load touches 32 rows at the SAME column offset, 8 KB apart -- every request of the
  // launch lands on the same few memory channels and the fused form (all workgroups walk K in step) takes 21 us against 14 us for the
  // split-K pair below, whose slices sit at different column offsets (tools/gemm_ragged.py)
  static const int fuse_kmax = DU_GETENV("DU_SKINNY_FUSE_KMAX") ? atoi(DU_GETENV("DU_SKINNY_FUSE_KMAX")) : 2048;      // A-B aid
  if (!no_fuse && a.K <= fuse_kmax) {
    // one launch: every workgroup runs the whole contraction of its 32 columns (4 waves x K/4) and applies the epilogue itself.  The
    // split-K pair below costs two launches + a partial round trip (12 us for 40 rows, 96 times per dinounet_l step)
    // waves per workgroup: enough that a wave walks at most ~4 chunks (K = 1024: 16 waves x 1 chunk, 4096: 16 x 4)
    static const int nw_env = DU_GETENV("DU_SKINNY_WAVES") ? atoi(DU_GETENV("DU_SKINNY_WAVES")) : 0;      // A-B aid: 4 / 8 / 16
    const int chunks = a.K / SK_CHUNK;
    const int nw = nw_env ? nw_env : (chunks >= 16 ? 16 : chunks >= 8 ? 8 : 4);
    if (nw >= 16) return launch_skinny_fused<16>(a, E, st);
    if (nw >= 8) return launch_skinny_fused<8>(a, E, st);
    return launch_skinny_fused<4>(a, E, st);
  }
  const int slices = skinny_slices(a.N, a.K);
  if (!a.ws || a.ws_elems < (int64_t)slices * 64 * a.N) return DU_ERR_UNSUPPORTED;
  int kps = (a.K + slices - 1) / slices;
  kps = ((kps + SK_CHUNK - 1) / SK_CHUNK) * SK_CHUNK;
  const int used = (a.K + kps - 1) / kps;               // empty trailing slices are dropped
  dim3 grid((a.N + SK_BN - 1) / SK_BN, used);
  hipLaunchKernelGGL(gemm_skinny_partial_kernel, grid, dim3(256), 0, st, (const bf16_t*)a.A, (long)a.lda, (const bf16_t*)a.B, (long)a.ldb,
                     a.ws, a.M, a.N, a.K, kps, E, 0);
  GemmParams P = make_params(a, DU_PLAIN_ROW, DU_PLAIN_ROW, 64, SK_BN, SK_CHUNK);
  const long total = (long)a.M * (a.N / 4);
  const int fg = (int)((total + 255) / 256);
  if (a.out_dtype == DU_BF16) hipLaunchKernelGGL(gemm_skinny_finish_kernel<bf16_t>, dim3(fg), dim3(256), 0, st, (const float*)a.ws, used, P);
  else hipLaunchKernelGGL(gemm_skinny_finish_kernel<float>, dim3(fg), dim3(256), 0, st, (const float*)a.ws, used, P);
  return du_check_launch();
}
