// bf16 MFMA GEMM / implicit-convolution engine for gfx950 (throughput path of du_gemm; the fp32 parity mode and the rarely
// used operand combinations stay on the generic kernel in gemm.hip).
//
//   C[m][n] = epilogue( alpha * sum_k A(m,k) * B(n,k) ),  v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// What is different from the generic kernel, and why (rocprofv3 per-shape table, profiles/r01_*):
//  * operands whose CONTRACTION index is the slow one in memory (weight gradients: A = dY^T, B = X^T / im2col(X)^T) are staged
//    as they lie -- LDS image [k][outer], 16-byte copies, no element-wise transposing stores -- and the MFMA fragments are
//    fetched with the LDS transpose read ds_read_b64_tr_b16 (two per fragment).  Row pitch = 64 B (mod 256 B) keeps the four
//    k-rows of one transpose read on disjoint bank groups.
//  * two LDS buffers and ONE barrier per K step: tile t+1 is fetched into registers before the MFMAs of tile t are issued and
//    written to the other buffer after them.
//  * the epilogue goes through LDS: accumulators are parked as fp32 rows, then every thread applies bias / activation /
//    LayerScale / DropPath scale / residual to 4 consecutive columns and stores 8 B (bf16) or 16 B (fp32) so each output row is
//    written in full cache lines (the per-lane scalar stores of the generic kernel ran the HBM-bound small-K shapes at < 1 TB/s).
//  * workgroup -> tile mapping is XCD-aware: the 8 XCDs each walk a contiguous range of tiles, so tiles that share an A row
//    panel hit the same L2.
#include "common.h"
#include "gemm_params.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int ROW_LD = BK + 8;   // ROW image: S[outer][BK + 8]   (144-B pitch: conflict-free ds_read_b128 fragments)

__host__ __device__ constexpr int col_pitch(int outer) { return (outer * 2) % 256 == 64 || (outer * 2) % 256 == 192 ? outer : outer + 32; }
__host__ __device__ constexpr bool is_row(int mode) { return mode == DU_PLAIN_ROW || mode == DU_IM2COL_ROW; }
__host__ __device__ constexpr int tile_elems(int mode, int outer) { return is_row(mode) ? outer * ROW_LD : BK * col_pitch(outer); }

// ---------------------------------------------------------------------------------------------------------------
// global -> registers -> LDS tile loader (OUTER x BK of one operand), 16-byte vectors
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, int OUTER>
struct Loader {
  static constexpr int NT = 256, VEC = 8;
  static constexpr bool ROWMODE = is_row(MODE);
  static constexpr int PITCH = ROWMODE ? ROW_LD : col_pitch(OUTER);
  static constexpr int TOTAL = OUTER * BK / VEC;
  static constexpr int NV = (TOTAL + NT - 1) / NT;
  static constexpr int KV = BK / VEC;      // vectors per row (ROW modes)
  static constexpr int OV = OUTER / VEC;   // vectors per k-row (COL modes)
  uint4 regs[NV];
  int pb[NV], py[NV], px[NV];

  __device__ __forceinline__ void init(const Operand& op, int tid, int o0, int outer_dim, int kbeg) {
    if constexpr (MODE == DU_IM2COL_ROW) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        int v = tid + i * NT;
        int o = o0 + v / KV;
        if (o >= outer_dim) o = outer_dim - 1;
        int xo = o % op.Wo; int t = o / op.Wo;
        px[i] = xo; py[i] = t % op.Ho; pb[i] = t / op.Ho;
      }
    } else if constexpr (MODE == DU_IM2COL_COL) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        int v = tid + i * NT;
        long pix = (long)kbeg + v / OV;
        int xo = (int)(pix % op.Wo); long t = pix / op.Wo;
        px[i] = xo; py[i] = (int)(t % op.Ho); pb[i] = (int)(t / op.Ho);
      }
    }
  }
  __device__ __forceinline__ void advance(const Operand& op) {
    if constexpr (MODE == DU_IM2COL_COL) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        px[i] += BK;
        while (px[i] >= op.Wo) { px[i] -= op.Wo; py[i]++; }
        while (py[i] >= op.Ho) { py[i] -= op.Ho; pb[i]++; }
      }
    }
  }
  __device__ __forceinline__ void load(const Operand& op, int tid, int o0, int outer_dim, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int v = tid + i * NT;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (TOTAL % NT == 0 || v < TOTAL) {
        if constexpr (MODE == DU_PLAIN_ROW) {
          int o = o0 + v / KV, k = k0 + (v % KV) * VEC;
          if (o < outer_dim && k < kend) r = *(const uint4*)((const bf16_t*)op.p + (long)o * op.ld + k);
        } else if constexpr (MODE == DU_PLAIN_COL) {
          int k = k0 + v / OV, o = o0 + (v % OV) * VEC;
          if (o < outer_dim && k < kend) r = *(const uint4*)((const bf16_t*)op.p + (long)k * op.ld + o);
        } else if constexpr (MODE == DU_IM2COL_ROW) {
          int o = o0 + v / KV, k = k0 + (v % KV) * VEC;
          if (o < outer_dim && k < kend) {
            const bf16_t* q = im2col_ptr<bf16_t>(op, pb[i], py[i], px[i], k);
            if (q) r = *(const uint4*)q;
          }
        } else {
          int k = k0 + v / OV, o = o0 + (v % OV) * VEC;
          if (o < outer_dim && k < kend) {
            const bf16_t* q = im2col_ptr<bf16_t>(op, pb[i], py[i], px[i], o);
            if (q) r = *(const uint4*)q;
          }
        }
      }
      regs[i] = r;
    }
  }
  __device__ __forceinline__ void store(bf16_t* S, int tid) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int v = tid + i * NT;
      if (TOTAL % NT == 0 || v < TOTAL) {
        if constexpr (ROWMODE) *(uint4*)(S + (v / KV) * PITCH + (v % KV) * VEC) = regs[i];
        else *(uint4*)(S + (v / OV) * PITCH + (v % OV) * VEC) = regs[i];
      }
    }
  }
  // MFMA operand fragment of the 32-row block starting at outer index i0 (tile-local), k-step kk (16 contraction elements):
  // lane l <-> outer i0 + (l & 31), contraction kk*16 + 8*(l >> 5) + 0..7
  static __device__ __forceinline__ bf16x8 frag(const bf16_t* S, int i0, int kk, int lane) {
    if constexpr (ROWMODE) {
      return *(const bf16x8*)(S + (i0 + (lane & 31)) * PITCH + kk * 16 + (lane >> 5) * 8);
    } else {
      const int g = lane >> 4, p = lane & 15;
      const bf16_t* q = S + (kk * 16 + 8 * (g >> 1) + (p >> 2)) * PITCH + i0 + 16 * (g & 1) + 4 * (p & 3);
      typedef __attribute__((address_space(3))) s16x4 lds_v4;
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)q);
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(q + 4 * PITCH));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8, r);
    }
  }
};

template <typename TC> struct Out4;
template <> struct Out4<float> {
  static __device__ __forceinline__ void load(const float* p, float* v) { float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Out4<bf16_t> {
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

// KSCALE (weight-gradient form only): DropPath's per-sample scale on the contraction rows, dropped samples' K tiles skipped.  A template
// parameter, not a run-time test of P.k_scale: with the `if (s_kt != 0)` skip branch in the loop the register allocator parks the
// accumulators in AGPRs and copies all of them to VGPRs and back EVERY K step (64 v_accvgpr_read + 64 v_accvgpr_write per 16 MFMAs,
// tools/isa_budget.py), in scaled and unscaled launches alike: +6 us per launch, DESIGN 6.27.  Unscaled launches now compile without it.
template <int AMODE, int BMODE, typename TC, int WM, int WN, int TM, int TN, bool KSCALE = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams P) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_EL = tile_elems(AMODE, BM), B_EL = tile_elems(BMODE, BN);
  constexpr int STG_LD = BN + 4;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;
  constexpr int BUF_EL = A_EL + B_EL;       // buffer b: A tile at smem + b*BUF_EL, B tile right behind it

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: hardware places workgroup b on XCD b % 8; give each XCD a contiguous range of tiles
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;                                   // grouped tile order, see gemm_glds.hip
  if (P.group_m > 1) {
    const int band = P.group_m * P.tiles_n;
    const int g = tile / band, l = tile - g * band;
    const int first = g * P.group_m;
    const int gsz = min(P.tiles_m - first, P.group_m);
    tn = l / gsz; tm = first + (l - tn * gsz);
  } else {
    tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int batch = z / P.split_k, split = z % P.split_k;
  const int kbeg = split * P.k_per_split;
  const int kend = min(P.K, kbeg + P.k_per_split);

  Operand opa = P.a, opb = P.b;
  opa.p = (const bf16_t*)opa.p + (long)batch * opa.bstride;
  opb.p = (const bf16_t*)opb.p + (long)batch * opb.bstride;

  Loader<AMODE, BM> la;
  Loader<BMODE, BN> lb;
  la.init(opa, tid, m0, P.M, kbeg);
  lb.init(opb, tid, n0, P.N, kbeg);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // bias-gradient side sum (du_gemm_args.a_colsum, weight-gradient products): sum_k A(m, k) for this tile's rows, taken by ONE tile column
  // per (tile row, split) -- tn == split % tiles_n rotates that duty over the columns -- from the A fragments the MFMAs consume anyway
  // (lane l holds row (l & 31), 8 consecutive k per k-step: every k of the tile exactly once across the two half-waves).  The owner
  // test is loop-invariant on purpose: the compiler unswitches the K loop, so only the owners run the slower variant (+10..25 % on their
  // launch).  Dealing the K tiles out over ALL columns (a per-iteration test, tried with a modulo and with a counter) slowed every
  // workgroup's loop and lost the whole gain: 35.2 vs 33.9 ms per step.
  const bool colsum_on = (AMODE == DU_PLAIN_COL) && P.a_colsum && wn == 0 && tn == split % P.tiles_n;
  float csum[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) csum[i] = 0.f;
  // the same for the gathered dY operand of a ConvTranspose weight gradient (du_gemm_args.b_colsum): one tile ROW per (tile column, split)
  const bool bsum_on = (BMODE == DU_IM2COL_COL) && P.b_colsum && wm == 0 && tm == split % P.tiles_m;
  float bsum[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) bsum[j] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    la.load(opa, tid, m0, P.M, kbeg, kend);
    lb.load(opb, tid, n0, P.N, kbeg, kend);
    la.store(smem, tid);
    lb.store(smem + A_EL, tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < nk);
    if (more) {
      la.advance(opa); lb.advance(opb);
      la.load(opa, tid, m0, P.M, kbeg + (kt + 1) * BK, kend);
      lb.load(opb, tid, n0, P.N, kbeg + (kt + 1) * BK, kend);
    }
    const bf16_t* Ac = smem + cur * BUF_EL;
    const bf16_t* Bc = Ac + A_EL;
    // weight-gradient form with a per-sample scale on the contraction rows (DropPath in backward: dW = (s . dY)^T X): the scale is uniform
    // over a K tile (rs_rows % BK == 0, checked by the host); a zero scale (dropped sample) skips the tile's MFMAs altogether
    float s_kt = 1.f;
    if constexpr (KSCALE) s_kt = P.row_scale[(kbeg + kt * BK) / P.rs_rows];
    if (!KSCALE || s_kt != 0.f) {
#pragma unroll
    for (int kk = 0; kk < BK / 16; kk++) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) fa[i] = Loader<AMODE, BM>::frag(Ac, (wm * TM + i) * 32, kk, lane);
      if constexpr (KSCALE) {
        if (s_kt != 1.f) {
#pragma unroll
          for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) fa[i][e] = (bf16_t)((float)fa[i][e] * s_kt);      // rounds like the materialised s . dY did
        }
      }
#pragma unroll
      for (int j = 0; j < TN; j++) fb[j] = Loader<BMODE, BN>::frag(Bc, (wn * TN + j) * 32, kk, lane);
      if constexpr (AMODE == DU_PLAIN_COL) {
        if (colsum_on) {
#pragma unroll
          for (int i = 0; i < TM; i++)
            csum[i] = frag_sum8(fa[i], csum[i]);
        }
      }
      if constexpr (BMODE == DU_IM2COL_COL) {
        if (bsum_on) {
#pragma unroll
          for (int j = 0; j < TN; j++)
            bsum[j] = frag_sum8(fb[j], bsum[j]);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    }
    if (more) {
      la.store(smem + (cur ^ 1) * BUF_EL, tid);
      lb.store(smem + (cur ^ 1) * BUF_EL + A_EL, tid);
    }
    __syncthreads();
  }

  if constexpr (AMODE == DU_PLAIN_COL) {
    if (colsum_on) {
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int m = m0 + (wm * TM + i) * 32 + (lane & 31);
        if (m < P.M) atomic_add_f32(P.a_colsum + m, csum[i]);
      }
    }
  }
  if constexpr (BMODE == DU_IM2COL_COL) {
    if (bsum_on) {
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (n < P.N) atomic_add_f32(P.b_colsum + n % P.b.C, bsum[j]);
      }
    }
  }
  TC* Cb = (TC*)P.C + (long)batch * P.cbs;
  // ---- split-K: fp32 atomics straight from the accumulators (lanes 0..31 = 32 consecutive columns) ----
  if (P.split_k > 1) {
    // DU_STORE_SLABS: this split's partial tile goes to its own slab with plain stores (reduced in a fixed order afterwards): the
    // forward / data-gradient uses of split-K must be bit-reproducible run to run, which fp32 atomics are not
    const bool slabs = P.store_mode == DU_STORE_SLABS;
    float* Cs = (float*)Cb + (slabs ? (long)split * P.M * P.ldc : 0L);
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (n >= P.N) continue;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < P.M) {
            if (slabs) Cs[(long)m * P.ldc + n] = acc[i][j][r] * P.alpha;
            else atomic_add_f32(Cs + (long)m * P.ldc + n, acc[i][j][r] * P.alpha);
          }
        }
    }
    return;
  }
  // ---- LDS-staged epilogue: WM*32 rows x BN columns of fp32 per pass ----
  float* stg = (float*)smem_raw;
  const TC* Rb = (const TC*)P.residual;
  if (Rb) Rb += (long)batch * P.cbs;
  constexpr int C4 = BN / 4;                 // float4 groups per staged row
  constexpr int NVEC = WM * 32 * C4;
#pragma unroll
  for (int i = 0; i < TM; i++) {
    if (i > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        stg[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * STG_LD + (wn * TN + j) * 32 + (lane & 31)] = acc[i][j][r];
    __syncthreads();
    for (int v = tid; v < NVEC; v += 256) {
      const int row = v / C4, c4 = v % C4;
      const int m = m0 + ((row >> 5) * TM + i) * 32 + (row & 31);
      const int n = n0 + c4 * 4;
      if (m >= P.M || n >= P.N) continue;
      float4 t = *(const float4*)(stg + row * STG_LD + c4 * 4);
      float o[4] = {t.x * P.alpha, t.y * P.alpha, t.z * P.alpha, t.w * P.alpha};
      if (P.bias) {
        float4 bb = *(const float4*)(P.bias + n);
        o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
      }
      if (P.act != DU_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = apply_act(o[e], P.act);
      }
      if (P.gamma) {
        float4 gg = *(const float4*)(P.gamma + n);
        o[0] *= gg.x; o[1] *= gg.y; o[2] *= gg.z; o[3] *= gg.w;
      }
      if (P.row_scale && !P.k_scale) {      // (k_scale: the scale was applied to the contraction rows inside the loop)
        const float rs = P.row_scale[m / P.rs_rows];
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] *= rs;
      }
      long off, offr;
      if (P.store_mode == DU_STORE_PIXEL_SHUFFLE2) {
        const int ps_q = n / P.ps_C, ps_co = n - ps_q * P.ps_C;
        int x = m % P.ps_W; int t2 = m / P.ps_W; int y = t2 % P.ps_H; int b = t2 / P.ps_H;
        const long opix = ((long)b * 2 * P.ps_H + 2 * y + (ps_q >> 1)) * (2 * P.ps_W) + 2 * x + (ps_q & 1);
        off = opix * P.ldc + ps_co;
        offr = opix * P.ldr + ps_co;      // the residual is laid out like the OUTPUT (pixel-shuffled), not like the GEMM
      } else {
        off = (long)m * P.ldc + n;
        offr = (long)m * P.ldr + n;
      }
      if (Rb) {
        float rr[4];
        Out4<TC>::load(Rb + offr, rr);
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] += rr[e];
      }
      Out4<TC>::store(Cb + off, o);
    }
  }
}

template <int AMODE, int BMODE, typename TC, int WM, int WN, int TM, int TN, bool KSCALE = false>
int launch_cfg(const du_gemm_args& a, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_EL = tile_elems(AMODE, BM), B_EL = tile_elems(BMODE, BN);
  constexpr int MAIN_BYTES = 2 * (A_EL + B_EL) * 2;
  constexpr int STG_BYTES = WM * 32 * (BN + 4) * 4;
  constexpr int LDS_BYTES = MAIN_BYTES > STG_BYTES ? MAIN_BYTES : STG_BYTES;
  GemmParams P = make_params(a, AMODE, BMODE, BM, BN, BK);
  P.tiles_m = (a.M + BM - 1) / BM;
  static const int group_env = DU_GETENV("DU_GEMM_GROUP_M") ? atoi(DU_GETENV("DU_GEMM_GROUP_M")) : 8;    // 0 / 1: row-major tile order
  P.group_m = group_env;
  dim3 grid(P.tiles_m * P.tiles_n, (a.batch < 1 ? 1 : a.batch) * P.split_k);
  if constexpr (AMODE == DU_PLAIN_COL && !KSCALE) {
    if (P.k_scale) return launch_cfg<AMODE, BMODE, TC, WM, WN, TM, TN, true>(a, st);
  }
  auto kfn = gemm_bf16_kernel<AMODE, BMODE, TC, WM, WN, TM, TN, KSCALE>;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(256), LDS_BYTES, st, P);
  return du_check_launch();
}

template <int AMODE, int BMODE, typename TC>
int launch_shape(const du_gemm_args& a, hipStream_t st) {
  // tile shapes: 128x128 default; narrow-N and narrow-M variants keep the MFMA work close to the useful work for the
  // 32/64-channel decoder convolutions (N small) and their weight gradients (M small)
  if (a.N <= 32) return launch_cfg<AMODE, BMODE, TC, 4, 1, 1, 1>(a, st);                  // 128 x 32
  if (a.M <= 32) return launch_cfg<AMODE, BMODE, TC, 1, 4, 1, 1>(a, st);                  // 32 x 128
  if (a.N <= 64) return launch_cfg<AMODE, BMODE, TC, 2, 2, 2, 1>(a, st);                  // 128 x 64
  if (a.M <= 64) return launch_cfg<AMODE, BMODE, TC, 2, 2, 1, 2>(a, st);                  // 64 x 128
  return launch_cfg<AMODE, BMODE, TC, 2, 2, 2, 2>(a, st);                                  // 128 x 128
}

template <typename TC>
int launch_modes(const du_gemm_args& a, hipStream_t st) {
  const int am = a.a_mode, bm = a.b_mode;
  if (am == DU_PLAIN_ROW && bm == DU_PLAIN_ROW) return launch_shape<DU_PLAIN_ROW, DU_PLAIN_ROW, TC>(a, st);
  if (am == DU_IM2COL_ROW && bm == DU_PLAIN_ROW) return launch_shape<DU_IM2COL_ROW, DU_PLAIN_ROW, TC>(a, st);
  if (am == DU_PLAIN_COL && bm == DU_PLAIN_COL) return launch_shape<DU_PLAIN_COL, DU_PLAIN_COL, TC>(a, st);
  if (am == DU_PLAIN_COL && bm == DU_IM2COL_COL) return launch_shape<DU_PLAIN_COL, DU_IM2COL_COL, TC>(a, st);
  if (am == DU_PLAIN_ROW && bm == DU_PLAIN_COL) return launch_shape<DU_PLAIN_ROW, DU_PLAIN_COL, TC>(a, st);
  return DU_ERR_UNSUPPORTED;
}

}  // namespace

// slabs a DU_STORE_SLABS product writes: make_params' rounding of the K range per split to whole K tiles of THIS engine (BK), empty ranges dropped
extern "C" int du_gemm_slab_count(int K, int split_k) {
  if (K <= 0) return 0;
  if (split_k < 1) split_k = 1;
  int kps = (K + split_k - 1) / split_k;
  kps = ((kps + BK - 1) / BK) * BK;
  return (K + kps - 1) / kps;
}

int du_gemm_nt_glds(const du_gemm_args& a, hipStream_t st);   // gemm_glds.hip
int du_gemm_skinny(const du_gemm_args& a, hipStream_t st);    // gemm_skinny.hip
int64_t du_gemm_skinny_ws_elems(int N, int K);

int du_gemm_nt_p8(const du_gemm_args& a, hipStream_t st, int tail_rows = 0);     // gemm_p8.hip
bool du_gemm_p8_tail_ok(const du_gemm_args& whole, int r);
bool du_gemm_p8_wants(const du_gemm_args& a);
int du_gemm_p8_choice(const du_gemm_args& a);
bool du_gemm_glds_serves(const du_gemm_args& a);              // gemm_glds.hip
bool du_gemm_rk_serves(const du_gemm_args& a);                // gemm_rk.hip (short contractions, weights resident in LDS)
int du_gemm_nt_rk(const du_gemm_args& a, hipStream_t st);
int du_gemm_tn_p8(const du_gemm_args& a, hipStream_t st);     // gemm_p8.hip (weight gradients)
int du_gemm_tn_p8_splits(const du_gemm_args& a);

// Rows of a tall bf16 NT product that should leave the tile grid for the K-parallel skinny kernels (gemm_skinny.hip).
//  * products served by the 256 x 256 multi-phase kernel (gemm_p8.hip): r = M % 256 when 0 < r <= 64 (the ViT: M = 8 * 1029 =
//    32 * 256 + 40) -- a ragged 33rd tile row would run a full K loop for 40 live rows;
//  * products on the 128 x 128 kernel: r = M % 128 when 0 < r <= 64 and the full tile rows fill the resident workgroup slots (2 per
//    CU) a whole number of times, at most twice (measured, tools/gemm_ragged.py: one round -- proj 39 -> 36 us, fc2 122 -> 95 us;
//    three rounds (qkv) 82 -> 85 us and four (fc1) no penalty to remove: the ragged tiles hide behind the spread of finish times).
// 0 = leave the product alone.
int du_gemm_ragged_rows(const du_gemm_args& a) {
  static const bool off = DU_GETENV("DU_GEMM_NO_RAGGED_SPLIT") != nullptr;      // debugging / A-B aid
  if (off || a.dtype != DU_BF16 || a.a_mode != DU_PLAIN_ROW || a.b_mode != DU_PLAIN_ROW) return 0;
  if (a.store_mode == DU_STORE_QKV_HEADS) {      // head-major qkv planes: the ragged rows ride in the persistent kernel's launch or the product is declined
    const int r = a.M % 256;
    if (r < 1 || r > 64 || a.batch > 1 || a.split_k > 1) return 0;
    du_gemm_args head = a;
    head.M = a.M - r;
    return du_gemm_p8_wants(head) && du_gemm_p8_tail_ok(a, r) ? r : 0;
  }
  if (a.store_mode != DU_STORE_PLAIN) return 0;
  if (a.act == DU_ACT_SWIGLU) return 0;      // the skinny kernels have no gate epilogue: the multi-phase kernel keeps the ragged rows
  if (a.batch > 1 || a.split_k > 1 || a.K % 64 || a.N < 96 || a.N % 4 || a.M < 1024 || a.lda % 8 || a.ldb % 8) return 0;
  {
    const int r = a.M % 256;
    if (r > 0 && r <= 64 && !(a.row_scale && (a.rs_rows < 1 || (a.M - r) % a.rs_rows))) {
      du_gemm_args head = a;
      head.M = a.M - r;
      if (du_gemm_p8_wants(head)) return r;
    }
    if (du_gemm_p8_wants(a)) return 0;
  }
  const int r = a.M % 128;
  if (r == 0 || r > 64) return 0;
  if (a.row_scale && (a.rs_rows < 1 || (a.M - r) % a.rs_rows)) return 0;
  static int slots = 0;
  if (!slots) {
    int dev = 0; hipDeviceProp_t prop;
    slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? 2 * prop.multiProcessorCount : 512;
  }
  const long full = (long)(a.M / 128) * ((a.N + 127) / 128);
  if (full % slots || full / slots > 2) return 0;
  return r;
}

long du_gemm_p8_ks_bytes(const du_gemm_args& a);               // gemm_p8.hip
long du_gemm_p8_tail_bytes(const du_gemm_args& whole);
long du_gemm_ks_bytes_bf16(const du_gemm_args& a) {
  if (a.dtype != DU_BF16 || a.a_mode != DU_PLAIN_ROW || a.b_mode != DU_PLAIN_ROW) return 0;
  du_gemm_args head = a;
  head.ks_ws = nullptr; head.ks_ws_bytes = 0;      // (the ragged-row rule must not depend on the scratch being asked about)
  int r = a.M % 256;
  if (r < 1 || r > 64 || !du_gemm_p8_tail_ok(a, r)) r = 0;
  head.M = a.M - r;
  const long pair = du_gemm_p8_ks_bytes(head);
  // (the sliced units exist in the one-shot tile kernels' launches: choice 1 / 2)
  const int c = r > 0 ? du_gemm_p8_choice(head) : 0;
  const long tail = (c == 1 || c == 2) ? du_gemm_p8_tail_bytes(a) : 0;
  return pair > tail ? pair : tail;
}

// large contraction-contiguous products: the 256 x 256 multi-phase kernel where it pays, else the 128 x 128 direct-to-LDS kernel
static int nt_tiles(const du_gemm_args& a, hipStream_t st) {
  if (du_gemm_p8_wants(a)) {
    int rc = du_gemm_nt_p8(a, st);
    if (rc != DU_ERR_UNSUPPORTED) return rc;
  }
  return du_gemm_nt_glds(a, st);
}

// returns DU_ERR_UNSUPPORTED when the generic kernel must be used instead
int du_gemm_bf16_fast(const du_gemm_args& a, hipStream_t st) {
  if (a.dtype != DU_BF16) return DU_ERR_UNSUPPORTED;
  if (a.N % 4 || a.ldc % 4 || (((uintptr_t)a.C) & 15)) return DU_ERR_UNSUPPORTED;
  if (a.bias && (((uintptr_t)a.bias) & 15)) return DU_ERR_UNSUPPORTED;
  if (a.gamma && (((uintptr_t)a.gamma) & 15)) return DU_ERR_UNSUPPORTED;
  if (a.residual && (a.ldr % 4 || (((uintptr_t)a.residual) & 15))) return DU_ERR_UNSUPPORTED;
  if (a.store_mode == DU_STORE_PIXEL_SHUFFLE2 && a.ps_C % 4) return DU_ERR_UNSUPPORTED;
  if (a.c_batch_stride % 4) return DU_ERR_UNSUPPORTED;
  if (a.a_mode == DU_PLAIN_COL && (a.b_mode == DU_PLAIN_COL || a.b_mode == DU_IM2COL_COL)) {     // weight gradients: the multi-phase kernel where it is legal
    int rc = du_gemm_tn_p8(a, st);
    if (rc != DU_ERR_UNSUPPORTED) return rc;
  }
  if (du_gemm_rk_serves(a)) {        // K <= 256, tall M: the streaming kernel (any M: rows past the last full block are masked)
    int rc = du_gemm_nt_rk(a, st);
    if (rc != DU_ERR_UNSUPPORTED) return rc;
  }
  {
    const int r = du_gemm_ragged_rows(a);
    if (r > 0 && a.ws && a.ws_elems >= du_gemm_skinny_ws_elems(a.N, a.K)) {
      // exact part on the tile kernel, the short ragged tail on the K-parallel skinny kernels (gemm_skinny.hip)
      const long m0 = a.M - r;
      const long osz = a.out_dtype == DU_BF16 ? 2 : 4;
      du_gemm_args tail = a;
      tail.M = r;
      tail.A = (const char*)a.A + m0 * a.lda * 2;
      tail.C = (char*)a.C + m0 * a.ldc * osz;
      if (a.residual) tail.residual = (const char*)a.residual + m0 * a.ldr * osz;
      if (a.row_scale) tail.row_scale = a.row_scale + m0 / a.rs_rows;
      du_gemm_args head = a;
      head.M = (int)m0;
      // round 3: where the head runs on a multi-phase kernel, the tail rides in the SAME launch (extra workgroups behind the tiles)
      if (du_gemm_p8_wants(head) && du_gemm_p8_tail_ok(a, r)) {
        const int rc_m = du_gemm_nt_p8(head, st, r);
        if (rc_m != DU_ERR_UNSUPPORTED) return rc_m;
      }
      // (forking the tail onto a side stream with event edges measured slower both times it was tried, inside the hipGraph: round 1 beside
      // the 128 x 128 kernels 191.6 vs 194.0 slices/s; round 2 beside the multi-phase kernels, 8 KB-LDS tail form, 33.9 vs 33.2 ms per step)
      int rc = nt_tiles(head, st);
      if (rc == DU_OK) {
        rc = du_gemm_skinny(tail, st);
        if (rc == DU_ERR_UNSUPPORTED) {
          tail.ws = nullptr; tail.ws_elems = 0;
          return a.out_dtype == DU_BF16 ? launch_modes<bf16_t>(tail, st) : launch_modes<float>(tail, st);
        }
        return rc;
      }
      if (rc != DU_ERR_UNSUPPORTED) return rc;
    }
    int rc = nt_tiles(a, st);
    if (rc != DU_ERR_UNSUPPORTED) return rc;
  }
  if (a.out_dtype == DU_BF16) return launch_modes<bf16_t>(a, st);
  return launch_modes<float>(a, st);
}

// which kernel family du_gemm runs for the bulk of this product (measurement tools name the kernel from this, not from a mirror of the
// dispatch): 0 generic (gemm.hip), 1 bf16 tile engine (this file), 2 128 x 128 direct-to-LDS (gemm_glds.hip), 3 / 4 the 256 x 256 /
// 256 x 128 multi-phase kernels (gemm_p8.hip), 5 the multi-phase weight-gradient kernel (gemm_p8.hip, TN form), 6 the persistent 256 x 128
// kernel (gemm_p8.hip), 7 the resident-weights streaming kernel for K <= 256 (gemm_rk.hip)
int du_gemm_route_bf16(const du_gemm_args& a) {
  if (a.dtype != DU_BF16) return 0;
  if (a.N % 4 || a.ldc % 4 || (((uintptr_t)a.C) & 15)) return 0;
  if (a.a_mode == DU_PLAIN_COL && du_gemm_tn_p8_splits(a)) return 5;
  if (du_gemm_rk_serves(a)) return 7;
  du_gemm_args head = a;
  const int r = du_gemm_ragged_rows(a);
  if (r > 0) head.M = a.M - r;
  const int c = du_gemm_p8_choice(head);
  if (c == 5) return 8;      // (2 + c = 7 is the resident-weights kernel)
  if (c) return 2 + c;
  return du_gemm_glds_serves(head) ? 2 : 1;
}
