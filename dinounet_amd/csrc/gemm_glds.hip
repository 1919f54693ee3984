// bf16 "NT" GEMM (both operands contraction-contiguous: linear layers, 1x1 convs, the frozen ViT) with direct-to-LDS loads.
//
//   C[m][n] = epilogue( alpha * sum_k A[m][k] * B[n][k] ),  128 x 128 x 64 tiles, 4 waves (2 x 2), v_mfma_f32_32x32x16_bf16.
//
// global_load_lds_dwordx4 moves each operand tile HBM/L2 -> LDS without passing through VGPRs (no staging registers, no
// ds_write pass); the LDS image of a tile is lane-linear ([128 rows][8 x 16-byte chunks], 128-B rows, no padding possible), so
// the bank-conflict swizzle is applied to the per-lane SOURCE address and undone in the fragment read (cdna guide 5.4 rule 21):
// physical chunk pc of row r holds logical k-chunk pc ^ ((r >> 1) & 7), which puts the 16 rows of every ds_read_b128 lane group
// on 16 distinct 16-byte bank slots.  Two LDS buffers, one barrier per K step; tile t+1 is in flight while tile t is multiplied.
// Rows past M / N are clamped (their products are never stored); K must be a multiple of 64.
#include "common.h"
#include "gemm_params.h"

namespace {

constexpr int BM = 128;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <typename TC> struct Out4g;
template <> struct Out4g<float> {
  static __device__ __forceinline__ void load(const float* p, float* v) { float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Out4g<bf16_t> {
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

// BK = 64, NST = 2: two buffers, vmcnt(0) + barrier per K step (64 KB LDS, 2 workgroups / CU).
// BK = 32, NST = 3: three buffers, tile t+2 is issued while tile t is multiplied and only tile t+1 is waited for (counted
//                   vmcnt, raw s_barrier: a plain __syncthreads() would drain the DMA queue); 48 KB LDS, 3 workgroups / CU.
// WNW = 2: 128 x 128 tile, 4 waves (2 x 2).  WNW = 4: 128 x 256 tile, 8 waves (2 x 4), 27 % less operand traffic per flop (the
// 128 x 128 kernel moves 32 KB L2->LDS per 2.1 MFLOP, as many cycles as its MFMAs), one 96 KB workgroup per CU.
template <typename TC, int BK, int NST, int WNW>
__global__ __launch_bounds__(WNW * 128) void gemm_nt_glds_kernel(GemmParams P) {
  constexpr int BN = WNW * 64, NW = 2 * WNW, NT = NW * 64, STG_LD = BN + 4;
  constexpr int A_EL = BM * BK, B_EL = BN * BK, BUF_EL = A_EL + B_EL;   // elements per operand tile / per buffer
  constexpr int CH = BK / 8;              // 16-byte chunks per tile row
  constexpr int RPW = 64 / CH;            // tile rows covered by one wave-level global_load_lds (64 lanes x 16 B)
  constexpr int RA = BM / (NW * RPW), RB = BN / (NW * RPW);   // staging rounds per operand
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;       // buffer b: A tile at b*BUF_EL, B tile right behind it

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Tile order inside an XCD's run: bands of group_m tile rows walked column by column, so the ~64 workgroups resident on an XCD
  // cover a square-ish (8 x 8) patch of C and share 8 A row panels + 8 B column panels through its L2.  Row-major order made them
  // 2-3 rows x all columns: every round re-fetched the whole B matrix from the fabric (FETCH_SIZE 3.2x the algorithmic bytes).
  int tm, tn;
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
  const int batch = blockIdx.y;
  const bf16_t* Ag = (const bf16_t*)P.a.p + (long)batch * P.a.bstride;
  const bf16_t* Bg = (const bf16_t*)P.b.p + (long)batch * P.b.bstride;

  // per-lane source rows of the staging rounds: round r covers tile rows (r*4 + wave)*RPW + lane/CH, physical chunk lane%CH.
  // Swizzle: physical chunk pc of row r holds logical chunk pc ^ swz(r); swz spreads the 16 rows of a ds_read_b128 lane group over
  // the 16 16-byte slots of a 256-byte bank row (BK = 64: 2 rows per bank row, BK = 32: 4 rows per bank row).
  const bf16_t* asrc[RA];
  const bf16_t* bsrc[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) {
    const int row = (r * NW + wave) * RPW + lane / CH;
    const int lc = (lane % CH) ^ (BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3));
    if (r < RA) {
      int ma = m0 + row; if (ma > P.M - 1) ma = P.M - 1;
      asrc[r] = Ag + (long)ma * P.a.ld + lc * 8;
    }
    int nb = n0 + row; if (nb > P.N - 1) nb = P.N - 1;
    bsrc[r] = Bg + (long)nb * P.b.ld + lc * 8;
  }
  auto stage = [&](int buf, int kt) {
    bf16_t* At = smem + buf * BUF_EL;
    bf16_t* Bt = At + A_EL;
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int rowbase = (r * NW + wave) * RPW;       // wave-uniform LDS destination (lane l lands at + l * 16 B)
      if (r < RA) __builtin_amdgcn_global_load_lds((gbl_void*)(asrc[r] + kt * BK), (lds_void*)(At + rowbase * BK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(bsrc[r] + kt * BK), (lds_void*)(Bt + rowbase * BK), 16, 0, 0);
    }
  };
  // fragment of the 32-row block at tile row i0, k-step kk: lane l <-> row i0 + (l & 31), logical chunk kk*2 + (l >> 5)
  auto frag = [&](const bf16_t* T, int i0, int kk) -> bf16x8 {
    const int row = i0 + (lane & 31);
    const int pc = (kk * 2 + (lane >> 5)) ^ (BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3));
    return *(const bf16x8*)(T + row * BK + pc * 8);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nk = P.K / BK;
  auto compute = [&](int buf) {
    const bf16_t* Ac = smem + buf * BUF_EL;
    const bf16_t* Bc = Ac + A_EL;
#pragma unroll
    for (int kk = 0; kk < BK / 16; kk++) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) fa[i] = frag(Ac, (wm * 2 + i) * 32, kk);
#pragma unroll
      for (int j = 0; j < 2; j++) fb[j] = frag(Bc, (wn * 2 + j) * 32, kk);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (NST == 2) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of tile 0 has landed (this wave's share); barrier = everyone's
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // 3-stage ring: RA + RB DMA instructions per tile per thread; "vmcnt(RA + RB)" = everything but the newest tile has landed
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RA + RB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
      int nxt2 = cur + 2; if (nxt2 >= 3) nxt2 -= 3;
      if (kt + 2 < nk) stage(nxt2, kt + 2);      // buffer last read in step kt-1 (everyone passed that step's barrier)
      compute(cur);
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RA + RB) : "memory");       // tile kt+1 landed, kt+2 may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1; if (cur >= 3) cur -= 3;
    }
    __syncthreads();
  }

  // ---- LDS-staged epilogue (same as gemm_bf16.hip): 64 rows x 128 columns of fp32 per pass ----
  TC* Cb = (TC*)P.C + (long)batch * P.cbs;
  float* stg = (float*)smem_raw;
  const TC* Rb = (const TC*)P.residual;
  if (Rb) Rb += (long)batch * P.cbs;
  constexpr int C4 = BN / 4;
  constexpr int NVEC = 64 * C4;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if (i > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        stg[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * STG_LD + (wn * 2 + j) * 32 + (lane & 31)] = acc[i][j][r];
    __syncthreads();
    for (int v = tid; v < NVEC; v += NT) {
      const int row = v / C4, c4 = v % C4;
      const int m = m0 + ((row >> 5) * 2 + i) * 32 + (row & 31);
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
      if (P.row_scale) {
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
        Out4g<TC>::load(Rb + offr, rr);
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] += rr[e];
      }
      Out4g<TC>::store(Cb + off, o);
    }
  }
}

template <typename TC, int BK, int NST, int WNW>
int launch(const du_gemm_args& a, hipStream_t st) {
  constexpr int BN = WNW * 64;
  constexpr int MAIN_BYTES = NST * (BM + BN) * BK * 2;  // 64 KB (128x128, BK 64, 2 stages) / 48 KB (BK 32, 3 stages) / 96 KB (128x256)
  constexpr int STG_BYTES = 64 * (BN + 4) * 4;
  constexpr int LDS_BYTES = MAIN_BYTES > STG_BYTES ? MAIN_BYTES : STG_BYTES;
  GemmParams P = make_params(a, DU_PLAIN_ROW, DU_PLAIN_ROW, BM, BN, BK);
  P.tiles_m = (a.M + BM - 1) / BM;
  static const int group_env = DU_GETENV("DU_GLDS_GROUP_M") ? atoi(DU_GETENV("DU_GLDS_GROUP_M")) : 8;   // 0 / 1: row-major tile order
  P.group_m = group_env;
  dim3 grid(P.tiles_m * P.tiles_n, a.batch < 1 ? 1 : a.batch);
  auto kfn = gemm_nt_glds_kernel<TC, BK, NST, WNW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(WNW * 128), LDS_BYTES, st, P);
  return du_check_launch();
}

}  // namespace

bool du_gemm_glds_serves(const du_gemm_args& a) {
  if (a.a_mode != DU_PLAIN_ROW || a.b_mode != DU_PLAIN_ROW || a.dtype != DU_BF16) return false;
  if (a.K % 64 || a.split_k > 1 || a.N < 96 || a.M < 64) return false;
  static const bool off = DU_GETENV("DU_GEMM_NO_GLDS") != nullptr;   // debugging / A-B aid
  return !off;
}

// returns DU_ERR_UNSUPPORTED when the shape / mode is not served by this kernel (caller falls back to gemm_bf16.hip)
int du_gemm_nt_glds(const du_gemm_args& a, hipStream_t st) {
  if (!du_gemm_glds_serves(a)) return DU_ERR_UNSUPPORTED;
  static const char* var = DU_GETENV("DU_GLDS_VARIANT");             // "3": force the 3-stage BK=32 ring, "2": force the 2-stage BK=64 kernel
  // measured (tools/gemm_bench.py): the 3-stage BK=32 ring wins for short contractions (K <= 512: +8..20 %, more workgroups per
  // CU and a deeper DMA queue), the 2-stage BK=64 kernel for K >= 1024 (fewer barriers per flop)
  const bool ring = var ? var[0] == '3' : a.K <= 512;
  if (ring) {
    if (a.out_dtype == DU_BF16) return launch<bf16_t, 32, 3, 2>(a, st);
    return launch<float, 32, 3, 2>(a, st);
  }
  static const char* wide_env = DU_GETENV("DU_GLDS_WIDE");            // "1": 128 x 256 tiles (8 waves) for N >= 256, "0": never
  const bool wide = wide_env ? (wide_env[0] == '1' && a.N >= 256) : false;
  if (wide) {
    if (a.out_dtype == DU_BF16) return launch<bf16_t, 64, 2, 4>(a, st);
    return launch<float, 64, 2, 4>(a, st);
  }
  if (a.out_dtype == DU_BF16) return launch<bf16_t, 64, 2, 2>(a, st);
  return launch<float, 64, 2, 2>(a, st);
}
