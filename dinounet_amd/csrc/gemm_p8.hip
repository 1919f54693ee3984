// bf16 "NT" GEMM for the large contraction-contiguous products (the frozen ViT's qkv / proj / fc1 / fc2, layers/attention.py:88-90,
// layers/ffn_layers.py:43-49): 256 x 256 x 64 tiles, 8 waves (2 x 4), v_mfma_f32_32x32x16_bf16, multi-phase LDS-DMA pipeline.
//
//   C[m][n] = epilogue( alpha * sum_k A[m][k] * B[n][k] )
//
// Why a second NT kernel: the 128 x 128 two-barrier kernel (gemm_glds.hip) moves 32 KB L2->LDS per 2.1 MFLOP and drains its DMA
// queue at every K step (vmcnt(0) + barrier); it tops out at ~650-700 TF/s on the ViT-L shapes.  Here
//   * a 256 x 256 tile halves the operand bytes per flop (64 KB per 8.4 MFLOP),
//   * the K step is cut into 4 phases, one 64 x 32 "quadrant" pair of the wave's 128 x 64 output each (8 MFMAs); every phase
//     prefetches the NEXT phase's fragments into a second register set and issues ONE 16 KB half-tile of LDS-DMA
//     (buffer_load_dwordx4 ... lds), so loads, LDS reads and MFMAs of different phases overlap,
//   * the DMA queue is never drained in the main loop: each phase ends with a COUNTED s_waitcnt vmcnt(12) (six half-tiles stay
//     in flight, ~1.5 K steps of latency cover) + lgkmcnt(0) + one raw s_barrier.
//
// LDS map (128 KB): buffer b = K-tile parity, 64 KB each = { A-half0, A-half1, B-half0, B-half1 }, a half = 128 tile rows x 64 k
// (16 KB, 128-B rows).  A-half i holds tile rows [128 i, 128 i + 128); wave (wm, wn) owns rows wm*64 + [0,64) of EACH A half and
// rows wn*32 + [0,32) of EACH B half, so quadrant (i, j) of every wave reads only A-half i and B-half j: once all waves have read a
// half (one barrier later) the DMA for tile t+2 may overwrite it.  The 16-byte chunks of a row are XOR-swizzled by ((row >> 1) & 7)
// on the SOURCE address (the DMA destination is lane-linear) and un-swizzled in the ds_read_b128 fragment address.
//
// Phase program of K-tile t (p = t & 1; fragment sets A0f/A1f and Bx/By, By <-> Bx swap roles every tile):
//   q0: MFMA (A0f, B0)  | ds_read  B1(t)   -> free B set | DMA B0(t+2) -> B-half0[p]   (last read in q3(t-1))
//   q1: MFMA (A0f, B1)  | ds_read  A1(t)   -> A1f        | DMA B1(t+2) -> B-half1[p]   (last read in q0(t))
//   q2: MFMA (A1f, B1)  | ds_read  A0(t+1) -> A0f        | DMA A1(t+2) -> A-half1[p]   (last read in q1(t))
//   q3: MFMA (A1f, B0)  | ds_read  B0(t+1) -> free B set | DMA A0(t+3) -> A-half0[1-p] (last read in q2(t))
// A half-tile is therefore issued seven phases before its first read.  Issue order: A0(t) B0(t) B1(t) A1(t) A0(t+1) ...; at the end
// of phase q the half-tile read in phase q+1 is the 7th youngest of the 13 outstanding => vmcnt(2 * 6).  The last two K-tile pairs
// use exact smaller counts (nothing is issued past the end of K).
//
// Rows past M / N read as zeros through the buffer descriptor's bounds check (their products are never stored).  K % 128 == 0, K >= 256.
#include "common.h"
#include "gemm_params.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int HALF_B = 128 * PBK * 2;          // 16 KB
constexpr int BUF_B = 4 * HALF_B;              // 64 KB
constexpr int STG_LD = PBN + 4;                // fp32 staging row (epilogue)

template <typename TC> struct Out4p;
template <> struct Out4p<float> {
  static __device__ __forceinline__ void load(const float* p, float* v) { float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Out4p<bf16_t> {
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

template <int N> struct IC { static constexpr int value = N; };

__device__ __forceinline__ void wait_vm_halves(int h) {
  // outstanding LDS-DMA instructions allowed = 2 per half-tile
  switch (h) {
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <typename TC, int SCHED>
__global__ __launch_bounds__(512) void gemm_nt_p8_kernel(GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;     // bijective XCD remap: an XCD owns a run of tiles
  }
  int tm, tn;
  if (P.group_m > 1) {      // bands of group_m tile rows walked column by column: an XCD's ~32 resident tiles share few A / B panels
    const int band = P.group_m * P.tiles_n;
    const int g = tile / band, l = tile - g * band;
    const int first = g * P.group_m;
    const int gsz = min(P.tiles_m - first, P.group_m);
    tn = l / gsz; tm = first + (l - tn * gsz);
  } else {
    tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n;
  }
  const int m0 = tm * PBM, n0 = tn * PBN;
  const int batch = blockIdx.y;

  // ---- buffer descriptors based at the tile's first row; rows past the matrix fail the bounds check and load zeros ----
  const bf16_t* Ab = (const bf16_t*)P.a.p + (long)batch * P.a.bstride + (long)m0 * P.a.ld;
  const bf16_t* Bb = (const bf16_t*)P.b.p + (long)batch * P.b.bstride + (long)n0 * P.b.ld;
  long abytes = ((long)(P.M - m0) * P.a.ld - (P.a.ld - P.K)) * 2, bbytes = ((long)(P.N - n0) * P.b.ld - (P.b.ld - P.K)) * 2;
  if (abytes > 0x7fffffffL) abytes = 0x7fffffffL;
  if (bbytes > 0x7fffffffL) bbytes = 0x7fffffffL;
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)abytes, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bbytes, 0x00020000);

  // staging: piece (round r, wave w) of a half = its rows (r*8 + w)*8 .. +8 (1 KB, lane l -> row + l/8, physical chunk l%8)
  unsigned va[2][2], vb[2][2];
  {
    const int sw = ((wave & 1) << 2) | (lane >> 4);          // ((row >> 1) & 7) of this lane's row
    const int lc = (lane & 7) ^ sw;                          // logical k chunk stored at this lane's physical chunk
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int row = h * 128 + (r * 8 + wave) * 8 + (lane >> 3);
        va[h][r] = (unsigned)row * (unsigned)(P.a.ld * 2) + lc * 16;
        vb[h][r] = (unsigned)row * (unsigned)(P.b.ld * 2) + lc * 16;
      }
  }
  // which: 0 = A-half0, 1 = A-half1, 2 = B-half0, 3 = B-half1 of K-tile kt, into buffer buf
  auto stage = [&](auto which_c, auto buf_c, int kt) {
    constexpr int which = decltype(which_c)::value, buf = decltype(buf_c)::value;
    constexpr int h = which & 1;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      unsigned char* dst = smem + buf * BUF_B + which * HALF_B + (r * 8 + wave) * 1024;
      if constexpr (which < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, va[h][r], kt * (PBK * 2), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)dst, 16, vb[h][r], kt * (PBK * 2), 0, 0);
    }
  };

  // fragment addresses: row (lane & 31) of a 32-row block, logical chunk kk*2 + (lane >> 5), physical = logical ^ ((row >> 1) & 7)
  int L[4];
  {
    const int x = (lane >> 5) ^ ((lane >> 1) & 7);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) L[kk] = (lane & 31) * 128 + ((x ^ (2 * kk)) << 4);
  }
  const int aoff = wm * 64 * 128, boff = wn * 32 * 128;

  bf16x8 Af[2][2][4];     // [set = A half][row block][kk]
  bf16x8 Bf[2][4];        // [set][kk]
  f32x16 acc[2][2][2];    // [i][j][row block]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][b][r] = 0.f;

  auto readA = [&](auto set_c, auto buf_c) {      // A-half `set` of buffer `buf` -> Af[set]
    constexpr int set = decltype(set_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
        Af[set][b][kk] = *(const bf16x8*)(smem + buf * BUF_B + set * HALF_B + aoff + b * 4096 + L[kk]);
  };
  auto readB = [&](auto set_c, auto half_c, auto buf_c) {    // B-half `half` of buffer `buf` -> Bf[set]
    constexpr int set = decltype(set_c)::value, half = decltype(half_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
      Bf[set][kk] = *(const bf16x8*)(smem + buf * BUF_B + (2 + half) * HALF_B + boff + L[kk]);
  };
  auto mma = [&](auto i_c, auto j_c, auto bset_c) {
    constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value, bset = decltype(bset_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int b = 0; b < 2; b++)
        acc[i][j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[i][b][kk], Bf[bset][kk], acc[i][j][b], 0, 0, 0);
  };
  // pin the issue order of a phase.  The compiler orders every ds_read of the phase before its LDS-DMA issues (it must assume they
  // alias), so the reads ride behind the first four MFMAs and the two DMA issues behind the next two.
  auto pin = [&](auto nrd_c, bool) {
    constexpr int nrd = decltype(nrd_c)::value;
    if constexpr (SCHED == 1) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
        if (m < 4) __builtin_amdgcn_sched_group_barrier(0x100, nrd / 4, 0);       // 1 or 2 DS reads
        if (m == 4 || m == 5) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 VMEM (LDS-DMA)
      }
    }
  };

  const int nk = P.K / PBK;

  // one K-tile: PAR = t & 1 (compile time), TAIL = runtime guards + exact vmcnt for the last four tiles
  auto ktile = [&](auto par_c, auto tail_c, int t) {
    constexpr int p = decltype(par_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr int b0set = p, b1set = 1 - p;    // fragment set holding B-half0 / B-half1 of this tile
    auto finish = [&](int q) {
      __builtin_amdgcn_sched_barrier(0);      // phase boundary: register-only MFMAs must not drift across it either
      if constexpr (TAIL) {
        int h = 4 * (nk - t - 1) - q;
        h = h < 0 ? 0 : (h > 6 ? 6 : h);
        wait_vm_halves(h);
      } else {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // q0
    readB(IC<b1set>{}, IC<1>{}, IC<p>{});
    if (!TAIL || t + 2 < nk) stage(IC<2>{}, IC<p>{}, t + 2);
    mma(IC<0>{}, IC<0>{}, IC<b0set>{});
    pin(IC<4>{}, true);
    finish(0);
    // q1
    readA(IC<1>{}, IC<p>{});
    if (!TAIL || t + 2 < nk) stage(IC<3>{}, IC<p>{}, t + 2);
    mma(IC<0>{}, IC<1>{}, IC<b1set>{});
    pin(IC<8>{}, true);
    finish(1);
    // q2
    if (!TAIL || t + 1 < nk) readA(IC<0>{}, IC<1 - p>{});
    if (!TAIL || t + 2 < nk) stage(IC<1>{}, IC<p>{}, t + 2);
    mma(IC<1>{}, IC<1>{}, IC<b1set>{});
    pin(IC<8>{}, true);
    finish(2);
    // q3  (the next tile's B-half0 goes to the set that held this tile's B-half1)
    if (!TAIL || t + 1 < nk) readB(IC<b1set>{}, IC<0>{}, IC<1 - p>{});
    if (!TAIL || t + 3 < nk) stage(IC<0>{}, IC<1 - p>{}, t + 3);
    mma(IC<1>{}, IC<0>{}, IC<b0set>{});
    pin(IC<4>{}, true);
    finish(3);
  };

  // ---- prologue: K-tiles 0 and 1 entirely, then A0(2) once A-half0[0] has been read ----
  stage(IC<0>{}, IC<0>{}, 0); stage(IC<2>{}, IC<0>{}, 0); stage(IC<3>{}, IC<0>{}, 0); stage(IC<1>{}, IC<0>{}, 0);
  stage(IC<0>{}, IC<1>{}, 1); stage(IC<2>{}, IC<1>{}, 1); stage(IC<3>{}, IC<1>{}, 1); stage(IC<1>{}, IC<1>{}, 1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // A0(0), B0(0) and B1(0) (read in q0 of tile 0: no earlier phase waits for it) landed
  __builtin_amdgcn_s_barrier();
  readA(IC<0>{}, IC<0>{});
  readB(IC<0>{}, IC<0>{}, IC<0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stage(IC<0>{}, IC<0>{}, 2);

  int t = 0;
  for (; t + 6 <= nk; t += 2) {
    ktile(IC<0>{}, IC<false>{}, t);
    ktile(IC<1>{}, IC<false>{}, t + 1);
  }
  for (; t < nk; t += 2) {
    ktile(IC<0>{}, IC<true>{}, t);
    ktile(IC<1>{}, IC<true>{}, t + 1);
  }

  // ---- LDS-staged epilogue: 4 passes of 64 rows x 256 columns of fp32 (pass = quadrant row i, row block b) ----
  TC* Cb = (TC*)P.C + (long)batch * P.cbs;
  float* stg = (float*)smem;
  const TC* Rb = (const TC*)P.residual;
  if (Rb) Rb += (long)batch * P.cbs;
  constexpr int C4 = PBN / 4;
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    const int i = pass >> 1, b = pass & 1;
    if (pass > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        stg[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * STG_LD + j * 128 + wn * 32 + (lane & 31)] = acc[i][j][b][r];
    __syncthreads();
#pragma unroll 2
    for (int v = tid; v < 64 * C4; v += 512) {
      const int row = v / C4, c4 = v % C4;
      const int m = m0 + i * 128 + (row >> 5) * 64 + b * 32 + (row & 31);
      const int n = n0 + c4 * 4;
      if (m >= P.M || n >= P.N) continue;
      float4 tt = *(const float4*)(stg + row * STG_LD + c4 * 4);
      float o[4] = {tt.x * P.alpha, tt.y * P.alpha, tt.z * P.alpha, tt.w * P.alpha};
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
      const long off = (long)m * P.ldc + n;
      if (Rb) {
        float rr[4];
        Out4p<TC>::load(Rb + (long)m * P.ldr + n, rr);
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] += rr[e];
      }
      Out4p<TC>::store(Cb + off, o);
    }
  }
}

int g_p8_mode = -1;      // -1: heuristic, 0: off, 1: force where legal
int g_p8_sched = 1;
int g_p8_group = 4;

template <typename TC, int SCHED>
int launch_p8(const du_gemm_args& a, hipStream_t st) {
  constexpr int LDS_BYTES = 2 * BUF_B;       // 128 KB (the epilogue staging needs 64 * 260 * 4 = 65 KB of it)
  GemmParams P = make_params(a, DU_PLAIN_ROW, DU_PLAIN_ROW, PBM, PBN, PBK);
  P.tiles_m = (a.M + PBM - 1) / PBM;
  P.group_m = g_p8_group;
  dim3 grid(P.tiles_m * P.tiles_n, a.batch < 1 ? 1 : a.batch);
  auto kfn = gemm_nt_p8_kernel<TC, SCHED>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, grid, dim3(512), LDS_BYTES, st, P);
  return du_check_launch();
}

}  // namespace

// debugging / A-B knobs (within-process variant switching for tools/gemm_p8_bench.py); not part of the hot-path contract
extern "C" int du_set_option(int key, int value) {
  switch (key) {
    case 0: g_p8_mode = value; return DU_OK;
    case 1: g_p8_sched = value; return DU_OK;
    case 2: g_p8_group = value; return DU_OK;
    default: return DU_ERR_BAD_ARG;
  }
}

// true when the 256 x 256 multi-phase kernel can run this product at all
static bool p8_legal(const du_gemm_args& a) {
  if (a.a_mode != DU_PLAIN_ROW || a.b_mode != DU_PLAIN_ROW || a.dtype != DU_BF16) return false;
  if (a.store_mode != DU_STORE_PLAIN || a.split_k > 1) return false;
  if (a.K % 128 || a.K < 256 || a.M < 256 || a.N < 128 || a.N % 4) return false;
  if (a.lda % 8 || a.ldb % 8 || (((uintptr_t)a.A) & 15) || (((uintptr_t)a.B) & 15)) return false;
  if (a.a_batch_stride % 8 || a.b_batch_stride % 8) return false;
  if ((long)a.lda * 2 * 256 > 0x7fffffffL || (long)a.ldb * 2 * 256 > 0x7fffffffL) return false;
  return true;
}

// true when du_gemm routes this product to the 256 x 256 multi-phase kernel (legal, and the heuristic or the debug knob says so)
bool du_gemm_p8_wants(const du_gemm_args& a) {
  if (g_p8_mode == 0 || !p8_legal(a)) return false;
  if (g_p8_mode > 0) return true;
  // enough 256 x 256 tiles to occupy the 256 CUs, long enough contraction to amortise the 9-half-tile prologue
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256) * (a.batch < 1 ? 1 : a.batch);
  return tiles >= 192 && a.K >= 512;
}

// returns DU_ERR_UNSUPPORTED when this kernel cannot serve the product; the caller then uses gemm_glds.hip
int du_gemm_nt_p8(const du_gemm_args& a, hipStream_t st) {
  if (!p8_legal(a)) return DU_ERR_UNSUPPORTED;
  if (a.out_dtype == DU_BF16) return g_p8_sched ? launch_p8<bf16_t, 1>(a, st) : launch_p8<bf16_t, 0>(a, st);
  return g_p8_sched ? launch_p8<float, 1>(a, st) : launch_p8<float, 0>(a, st);
}
