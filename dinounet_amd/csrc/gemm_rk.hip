// bf16 NT products with a SHORT contraction (K <= 256) and a tall M: the adapter / FAPM / decoder 1x1 projections, ConvTranspose2d k2 s2
// forward (pixel-shuffle store) and data gradient (2 x 2 patch gather), the dgrad of the MSDA offsets + weights linear (K = 192), the
// SPM's fc1 / fc2 (dinov3_adapter.py:84-91,289-294; dinounet_training.py:255-264,419-441,558).
//
//   C[m][n] = act( sum_k A[m][k] * W(n, k) + bias[n] )        bf16 in, fp32 accumulate, bf16 out
//
// These are STREAMING problems: (K + N) * 2 bytes per row against 2 N K flops -- at K = 64, N = 1024 the output alone is 2 KB per row
// and the product 131 kFLOP: HBM-bound by a factor of ten.  The tile kernels run them at a third to a half of the copy rate
// (profiles/r05_step_detail_per_shape_v1.txt: 131072 x 1024 x 64 in 112 us = 2.5 TB/s, 524288 x 128 x 32 in 96 us = 1.7 TB/s): a 128 x 128
// or 256 x 256 tile with one to four K-steps is all prologue and epilogue, operands staged through LDS for a handful of MFMAs, the
// result staged through LDS again.  Here nothing but the weights touches LDS:
//   * a workgroup owns a CHUNK of <= 256 output columns and keeps that slice of W in LDS for its whole life ([n][K] rows padded by 16
//     bytes: the 16-lane groups of a ds_read_b128 fragment read hit 16 different 16-byte slots), filled once from L2;
//   * a WAVE owns a block of 32 rows at a time and walks down the matrix on its own -- no barrier after the weight fill.  The block's A
//     fragments (K / 16 x 16 bytes per lane: row lane & 31, k chunk lane >> 5) are loaded straight from global memory into registers,
//     the next block's while this one computes; they are the MFMA's second operand for every 32-column block of the chunk
//     (v_mfma_f32_32x32x16_bf16 fed (W fragment, A fragment): a lane then holds one output row and 4 consecutive columns per register
//     quad), K / 16 MFMAs + K / 16 LDS reads per 32 x 32 block;
//   * the block is stored from registers: bias (LDS image of the chunk's slice), activation, bf16 pairs, two v_permlane32_swap per 16
//     columns (a row's two lanes then hold 32 contiguous bytes), two 16-byte buffer stores per lane and block -- the persistent GEMM's
//     drain (gemm_p8.hip) without a tile to wait for;
//   * plain rows, the ConvTranspose pixel-shuffle store (a 32-column block lies inside one tap: a per-lane base pixel + a scalar tap
//     offset), the 2 x 2 patch gather of the ConvTranspose data gradient as A (a k-step lies inside one tap: a scalar offset per k-step),
//     and W given row-major [N][K] or as [K][N] (the data-gradient form: transposed once while the LDS image is filled).
// Roofline: HBM.  Algorithmic bytes per row = (K + N) * 2.
#include <stdlib.h>
#include "common.h"
#include "gemm_params.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned RK_OOR = 0x80000000u;

struct RkParams {
  const bf16_t* A; long lda;
  int gather;                       // 0: plain rows; 1: ConvTranspose k2 s2 data gradient -- row m = input pixel (b, y, x), k = (tap, c)
  int gWi, gHo, gWo, gC;            //    dy is (B, 2 Ho, 2 Wo = Wi, ld = lda) and A(m, tap * C + c) = dy[b][2 y + tap / 2][2 x + tap % 2][c]
  const bf16_t* W; long ldb; int b_col;     // b_col: element (n, k) at W[k * ldb + n] (else W[n * ldb + k])
  bf16_t* C; long ldc;
  int store_mode, ps_H, ps_W, ps_C;
  const float* bias; int act;
  int M, N, K;
  int nchunks, chunk_cols, wgs_per_chunk;
  unsigned a_bytes, c_bytes;
};

__device__ __forceinline__ void rk_swap_halves(unsigned& a, unsigned& b) {      // a's lanes 32-63 <-> b's lanes 0-31
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// KS = K / 16 (k-steps), NW = waves per workgroup
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_nt_rk_kernel(RkParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int K = KS * 16;
  constexpr int PITCH = K * 2 + 16;                   // bytes per W row in LDS
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x % P.nchunks, g = blockIdx.x / P.nchunks;
  const int n0 = chunk * P.chunk_cols;
  const int ncols = min(P.chunk_cols, P.N - n0);      // multiple of 32
  float* bias_l = (float*)(smem + (long)P.chunk_cols * PITCH);

  // ---- the chunk's weights -> LDS image [n][K] (+ 16 bytes of padding per row), its bias slice behind it ----
  if (!P.b_col) {
    constexpr int C8 = K / 8;                         // 16-byte pieces per row
    for (int v = tid; v < ncols * C8; v += 64 * NW) {
      const int n = v / C8, c8 = v - n * C8;
      *(uint4*)(smem + n * PITCH + c8 * 16) = *(const uint4*)(P.W + (long)(n0 + n) * P.ldb + c8 * 8);
    }
  } else {
    const int n8 = ncols / 8;                         // W[k][n0 + 8 j .. + 7] -> column j's eight rows, element k
    for (int v = tid; v < K * n8; v += 64 * NW) {
      const int k = v / n8, j = v - k * n8;
      const bf16x8 t = __builtin_bit_cast(bf16x8, *(const uint4*)(P.W + (long)k * P.ldb + n0 + j * 8));
#pragma unroll
      for (int e = 0; e < 8; e++) *(bf16_t*)(smem + (j * 8 + e) * PITCH + k * 2) = t[e];
    }
  }
  for (int v = tid; v < ncols; v += 64 * NW) bias_l[v] = P.bias ? P.bias[n0 + v] : 0.f;
  __syncthreads();

  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)P.a_bytes, 0x00020000);
  const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)P.C, 0, (int)P.c_bytes, 0x00020000);
  const int nrb = (P.M + 31) / 32;
  const int stride = P.wgs_per_chunk * NW;
  const int logC = P.gather ? __builtin_ctz(P.gC) : 0;

  // byte offset of this lane's A row of block rb at k = 8 hi (out of range past M)
  auto a_row = [&](int rb) -> unsigned {
    const int m = rb * 32 + (lane & 31);
    if (m >= P.M) return RK_OOR;
    if (!P.gather) return (unsigned)(((long)m * P.lda + 8 * hi) * 2);
    const int x = m % P.gWo, t = m / P.gWo, y = t % P.gHo, b = t / P.gHo;
    return (unsigned)(((((long)b * 2 * P.gHo + 2 * y) * P.gWi + 2 * x) * P.lda + 8 * hi) * 2);
  };
  // scalar byte offset of k-step s inside a row
  auto a_koff = [&](int s) -> unsigned {
    if (!P.gather) return (unsigned)(s * 32);
    const int k = s * 16, tap = k >> logC, c = k - (tap << logC);
    return (unsigned)((((long)(tap >> 1) * P.gWi + (tap & 1)) * P.lda + c) * 2);
  };
  unsigned koff[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) koff[s] = __builtin_amdgcn_readfirstlane(a_koff(s));

  u32x4_t Af[KS], An[KS];
  auto load_a = [&](u32x4_t (&dst)[KS], int rb) {
    const unsigned base = rb < nrb ? a_row(rb) : RK_OOR;
#pragma unroll
    for (int s = 0; s < KS; s++) dst[s] = __builtin_amdgcn_raw_buffer_load_b128(ra, base, koff[s], 0);
  };
  // W fragment of 32-column block cb, k-step s: lane -> row n = cb * 32 + (lane & 31), k chunk 16 s + 8 hi
  const unsigned char* wl = smem + (lane & 31) * PITCH + hi * 16;

  int rb = g * NW + wave;
  load_a(Af, rb);
  for (; rb < nrb; rb += stride) {
    load_a(An, rb + stride);                          // the next block's rows travel while this one computes
    // this lane's output row: byte offset of column 8 hi of the chunk (plain), or of the base pixel (pixel shuffle)
    const int m = rb * 32 + (lane & 31);
    unsigned crow = RK_OOR;
    if (m < P.M) {
      if (P.store_mode == DU_STORE_PIXEL_SHUFFLE2) {
        const int x = m % P.ps_W, t = m / P.ps_W, y = t % P.ps_H, b = t / P.ps_H;
        crow = (unsigned)(((((long)b * 2 * P.ps_H + 2 * y) * (2 * P.ps_W) + 2 * x) * P.ldc + 8 * hi) * 2);
      } else {
        crow = (unsigned)(((long)m * P.ldc + n0 + 8 * hi) * 2);
      }
    }
    const int nblocks = ncols >> 5;
    for (int cb = 0; cb < nblocks; cb++) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const bf16x8 wf = *(const bf16x8*)(wl + cb * 32 * PITCH + s * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, Af[s]), acc, 0, 0, 0);
      }
      // accumulator register 4 g + e = column 8 g + 4 hi + e of this lane's row
      unsigned coff;                                   // scalar byte offset of the block's first column relative to crow
      if (P.store_mode == DU_STORE_PIXEL_SHUFFLE2) {
        const int n = n0 + cb * 32, q = n / P.ps_C, co = n - q * P.ps_C;
        coff = (unsigned)((((long)(q >> 1) * (2 * P.ps_W) + (q & 1)) * P.ldc + co) * 2);
      } else {
        coff = (unsigned)(cb * 64);
      }
      coff = __builtin_amdgcn_readfirstlane(coff);
      const float* bl = bias_l + cb * 32 + 4 * hi;
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {                 // columns 16 h2 .. + 15: groups 2 h2 and 2 h2 + 1
        unsigned kk[4];
#pragma unroll
        for (int part = 0; part < 2; part++) {
          const int gq = 2 * h2 + part;
          const float4 bv = *(const float4*)(bl + 8 * gq);
          float o[4] = {acc[4 * gq] + bv.x, acc[4 * gq + 1] + bv.y, acc[4 * gq + 2] + bv.z, acc[4 * gq + 3] + bv.w};
          if (P.act != DU_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = apply_act(o[e], P.act);
          }
          const bf16x2 t0 = {(bf16_t)o[0], (bf16_t)o[1]}, t1 = {(bf16_t)o[2], (bf16_t)o[3]};
          kk[2 * part] = __builtin_bit_cast(unsigned, t0);
          kk[2 * part + 1] = __builtin_bit_cast(unsigned, t1);
        }
        rk_swap_halves(kk[0], kk[2]);                  // lanes 0-31: columns 16 h2 .. + 7, lanes 32-63: 16 h2 + 8 .. + 15
        rk_swap_halves(kk[1], kk[3]);
        const u32x4_t v = {kk[0], kk[1], kk[2], kk[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, rc, crow == RK_OOR ? RK_OOR : crow + (unsigned)(h2 * 32), coff, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; s++) Af[s] = An[s];
  }
}

template <int KS, int NW>
int rk_launch(const RkParams& P, hipStream_t st, size_t lds) {
  void (*kfn)(RkParams) = gemm_nt_rk_kernel<KS, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(P.nchunks * P.wgs_per_chunk), dim3(64 * NW), lds, st, P);
  return du_check_launch();
}

}  // namespace

int g_rk_mode = 1;       // du_set_option key 12: 0 = never, 1 = where it pays (default: K <= 192, K = 256 up to 512 columns), 2 = every K = 256 too, 3 = wherever legal (tests)

// the ConvTranspose2d k2 s2 data gradient as a gathered A operand (as gemm_p8.hip's convt_gather_geom, any channel count that is a power of two >= 16)
static bool rk_gather_ok(const du_gemm_args& a) {
  const du_conv_geom& g = a.geom;
  if (g.KH != 2 || g.KW != 2 || g.stride != 2 || g.pad != 0 || g.transposed || g.p2) return false;
  if (g.Hi != 2 * g.Ho || g.Wi != 2 * g.Wo || g.Ho <= 0 || g.Wo <= 0) return false;
  if (g.C < 16 || (g.C & (g.C - 1)) || a.K != 4 * g.C) return false;
  return a.M % (g.Ho * g.Wo) == 0;
}

bool du_gemm_rk_serves(const du_gemm_args& a) {
  if (!g_rk_mode || a.dtype != DU_BF16 || a.out_dtype != DU_BF16) return false;
  if (a.a_mode != DU_PLAIN_ROW && !(a.a_mode == DU_IM2COL_ROW && rk_gather_ok(a))) return false;
  if (a.b_mode != DU_PLAIN_ROW && a.b_mode != DU_PLAIN_COL) return false;
  if (a.K != 32 && a.K != 64 && a.K != 128 && a.K != 192 && a.K != 256) return false;
  // K = 256: up to 512 columns (two chunks); at N = 1024 the four column chunks read A four times and the multi-phase kernel ties
  // (43008 x 1024 x 256: 43.9 vs 47.0 us alone, no difference in the step, profiles/r05_ab_rk_splitk_v1.txt)
  if (a.K == 256 && g_rk_mode < 2 && a.N > 512) return false;
  if (a.N < 32 || a.N % 32 || a.M < 4096 || a.batch > 1 || a.split_k > 1) return false;
  // where it pays (profiles/r05_gemm_rk_table_v2.txt): streams of >= 2^24 output elements -- below that the weight fill and the first
  // rows' latency are most of the launch (32768 x 256 x 64: 15.1 us against 13.6 on the 128 x 128 kernel, 8192 x 128 x 128: 11.6 against 10.6)
  if (g_rk_mode < 3 && (long)a.M * a.N < (1L << 24)) return false;
  if (a.alpha != 1.0f || a.gamma || a.row_scale || a.residual || a.act == DU_ACT_SWIGLU) return false;
  if (a.store_mode != DU_STORE_PLAIN && !(a.store_mode == DU_STORE_PIXEL_SHUFFLE2 && a.ps_C > 0 && a.ps_C % 32 == 0 && a.N == 4 * a.ps_C &&
                                         a.ps_H > 0 && a.ps_W > 0 && a.M % (a.ps_H * a.ps_W) == 0)) return false;
  if (a.lda % 8 || a.ldb % 8 || a.ldc % 8) return false;
  if ((((uintptr_t)a.A) | ((uintptr_t)a.B) | ((uintptr_t)a.C)) & 15) return false;
  const long rows_a = a.a_mode == DU_IM2COL_ROW ? 4L * a.M : a.M;           // dy has four pixels per row of the gathered operand
  const long rows_c = a.store_mode == DU_STORE_PIXEL_SHUFFLE2 ? 4L * a.M : a.M;
  if (rows_a * a.lda * 2 >= 0x7fffffffL || rows_c * a.ldc * 2 >= 0x7fffffffL) return false;
  return true;
}

int du_gemm_nt_rk(const du_gemm_args& a, hipStream_t st) {
  if (!du_gemm_rk_serves(a)) return DU_ERR_UNSUPPORTED;
  RkParams P{};
  P.A = (const bf16_t*)a.A; P.lda = a.lda;
  if (a.a_mode == DU_IM2COL_ROW) { P.gather = 1; P.gWi = a.geom.Wi; P.gHo = a.geom.Ho; P.gWo = a.geom.Wo; P.gC = a.geom.C; }
  P.W = (const bf16_t*)a.B; P.ldb = a.ldb; P.b_col = a.b_mode == DU_PLAIN_COL ? 1 : 0;
  P.C = (bf16_t*)a.C; P.ldc = a.ldc;
  P.store_mode = a.store_mode; P.ps_H = a.ps_H; P.ps_W = a.ps_W; P.ps_C = a.ps_C;
  P.bias = a.bias; P.act = a.act;
  P.M = a.M; P.N = a.N; P.K = a.K;
  const long rows_a = P.gather ? 4L * a.M : a.M, rows_c = a.store_mode == DU_STORE_PIXEL_SHUFFLE2 ? 4L * a.M : a.M;
  P.a_bytes = (unsigned)(rows_a * a.lda * 2);
  P.c_bytes = (unsigned)(rows_c * a.ldc * 2);
  // chunks of <= 256 columns (<= 128 at K = 256... the image must leave room: 256 x 528 B = 132 KB fits); equal chunks
  const int maxc = 256;
  P.nchunks = (a.N + maxc - 1) / maxc;
  P.chunk_cols = ((a.N / 32 + P.nchunks - 1) / P.nchunks) * 32;
  const size_t lds = (size_t)P.chunk_cols * (a.K * 2 + 16) + (size_t)P.chunk_cols * 4;
  // workgroups: as many as stay resident (LDS-bound), at least 4 row blocks per wave
  const bool big = a.K >= 192;                        // 8 waves per workgroup where one workgroup fills a CU's LDS
  const int nw = big ? 8 : 4;
  int per_cu = (int)((160 * 1024) / (lds + 512));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > (big ? 1 : 4)) per_cu = big ? 1 : 4;
  long wgs = 256L * per_cu / P.nchunks;
  const long nrb = (a.M + 31) / 32;
  const long need = (nrb + nw * 2 - 1) / (nw * 2);     // at least two blocks of rows per wave
  if (wgs > need) wgs = need;
  if (wgs < 1) wgs = 1;
  P.wgs_per_chunk = (int)wgs;
  switch (a.K) {
    case 32: return rk_launch<2, 4>(P, st, lds);
    case 64: return rk_launch<4, 4>(P, st, lds);
    case 128: return rk_launch<8, 4>(P, st, lds);
    case 192: return rk_launch<12, 8>(P, st, lds);
    case 256: return rk_launch<16, 8>(P, st, lds);
  }
  return DU_ERR_UNSUPPORTED;
}
