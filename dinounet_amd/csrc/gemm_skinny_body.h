// The ragged-tail product of a tall NT GEMM (<= 64 rows x N x K, see gemm_skinny.hip) as a device function: one workgroup = 32 output columns
// x the WHOLE contraction, NW waves each taking every NW-th 64-wide K chunk, fragments straight from global memory, wave tiles parked in
// LDS slots [NW][64][33], then the du_gemm epilogue.  Used by gemm_skinny_fused_kernel (its own launch) and, for the ViT's 40-row tails, by
// EXTRA workgroups appended to the grid of the multi-phase tile kernels (gemm_p8.hip): the tail then costs no launch of its own.
#pragma once
#include "common.h"

namespace {

struct SkinnyEpi {
  void* C; long ldc; const void* residual; long ldr;
  const float* bias; const float* gamma; const float* row_scale;
  float alpha; int act, rs_rows, out_bf16;
  int row0;        // first output row's index for row_scale (the tail sits behind the head's rows)
  int qkv_H;       // > 0: DU_STORE_QKV_HEADS -- C is the base of the three head-major planes (ldc elements apart), row row0 + m = (b, token) of
  int qkv_N, qkv_Npad;   //   qkv_N tokens per sample, column n = (which, head, d): element [which][b][head][token][d] (bf16, no residual)
  // K-sliced units (gemm_p8.hip tails with a long contraction): this workgroup holds slice `slice` of `slices`; its 64 x 32 fp32 partial goes to
  // slab + slice * 2048 floats (sc1 stores), a ticket on *cnt names the last arriver, which adds the slabs IN SLICE ORDER (bit-reproducible)
  // and runs the epilogue; *cnt is zero between launches.  slices <= 1: the unit owns its columns (no slab, no counter).
  int slices, slice; float* slab; int* cnt;
};

constexpr int SK_BN = 32;          // output columns per workgroup
constexpr int SK_CHUNK = 64;       // contraction elements per wave step (4 MFMAs)

template <int NW, int CB = 2, bool SL = false, bool STG = false>
__device__ __forceinline__ void skinny_fused_body(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, int M, int N,
                                                  int K, const SkinnyEpi& P, float* sk_red, int blk) {
  constexpr int SLOT = 64 * (SK_BN + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blk * SK_BN;
  const int nrb = (M + 31) >> 5;
  const int kh = (lane >> 5) * 32;
  const int n = n0 + (lane & 31);
  const bf16_t* bp = B + (long)min(n, N - 1) * ldb + kh;
  const bf16_t* ap0 = A + (long)min(lane & 31, M - 1) * lda + kh;
  const bf16_t* ap1 = A + (long)min(32 + (lane & 31), M - 1) * lda + kh;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // epilogue operands of this thread's first output item are requested BEFORE the contraction (round 6: the unit is a chain of dependent
  // memory latencies -- fragments, then bias / LayerScale / residual -- at the end of a launch whose other workgroups are done; each link
  // removed is ~1.5-2 us off the product: 5-7 us per K = 1024 product and 18 us for fc2's K = 4096 tail before, profiles/r06_gemm_kspair_v1.txt)
  constexpr int NT = NW * 64, IPR = SK_BN / 4;
  const int total = M * IPR;
  struct EpiOps { float4 b, g, r; float rs; };
  auto fetch = [&](int i, EpiOps& E) {
    const int m = i / IPR, nn = n0 + (i % IPR) * 4;
    E.b = make_float4(0.f, 0.f, 0.f, 0.f); E.g = make_float4(1.f, 1.f, 1.f, 1.f); E.r = E.b; E.rs = 1.f;
    if (i >= total || nn >= N) return;
    if (P.bias) E.b = *(const float4*)(P.bias + nn);
    if (P.gamma) E.g = *(const float4*)(P.gamma + nn);
    if (P.row_scale) E.rs = P.row_scale[(P.row0 + m) / P.rs_rows];
    if (P.residual && P.qkv_H <= 0) {
      if (P.out_bf16) {
        const bf16x4 t = *(const bf16x4*)((const bf16_t*)P.residual + (long)m * P.ldr + nn);
        E.r = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
      } else {
        E.r = *(const float4*)((const float*)P.residual + (long)m * P.ldr + nn);
      }
    }
  };
  EpiOps E0;
  const bool sliced = SL && P.slices > 1;      // (SL = false: the slice code is not compiled in -- it costs the persistent kernels registers)
  if (!sliced) fetch(tid, E0);
  if constexpr (STG) {
    // Operands through LDS in whole lines (round 6).  The fragment-shaped loads below fetch 32 rows x 16 bytes per instruction: 32 cache lines
    // for 1 KB, four instructions per line -- the address path walks 4 x more lines than the data has (K = 1024: ~6100 line visits per unit,
    // ~2.6 us of the unit's ~6).  Here a K-step is 512 contraction elements of the 32 W rows + 64 A rows = 96 rows x 1 KB: one wave reads one
    // whole row per instruction (8 lines), 12 instructions per lane, all in flight; the rows are parked in LDS (row pitch 1 KB + 16 B: the
    // 32 rows of a fragment read start 4 banks apart) and every wave takes the fragments of its 64-wide chunk from there.  The next
    // K-step's loads are in flight under the MFMAs.  The staging area is the reduce slots' memory (they are written after the last step).
    static_assert(NW == 8, "one K-step = 8 chunks, one per wave");
    constexpr int KST = 512, PITCH = KST * 2 + 16, NPC = 96 * 64 / (NW * 64);
    unsigned char* stg = (unsigned char*)sk_red;
    const int nst = (K + KST - 1) / KST;
    bf16x8 cur[NPC];
    auto request = [&](int st) {
#pragma unroll
      for (int it = 0; it < NPC; it++) {
        const int pc = it * (NW * 64) + tid, row = pc >> 6, c16 = pc & 63;
        const int kk = st * KST + c16 * 8;
        const bf16_t* src = row < 32 ? B + (long)min(n0 + row, N - 1) * ldb : A + (long)min(row - 32, M - 1) * lda;
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = (bf16_t)0.f;
        cur[it] = (kk < K && (row < 64 || nrb > 1)) ? *(const bf16x8*)(src + kk) : z;
      }
    };
    request(0);
    for (int st = 0; st < nst; st++) {
      if (st > 0) __syncthreads();               // the previous step's fragment reads
#pragma unroll
      for (int it = 0; it < NPC; it++) {
        const int pc = it * (NW * 64) + tid;
        *(bf16x8*)(stg + (pc >> 6) * PITCH + (pc & 63) * 16) = cur[it];
      }
      if (st + 1 < nst) request(st + 1);
      __syncthreads();
      if (st * KST + wave * SK_CHUNK < K) {
        const unsigned char* fr = stg + (lane & 31) * PITCH + wave * 128 + (lane >> 5) * 64;
        bf16x8 fb[4], fa0[4], fa1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = *(const bf16x8*)(fr + j * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) fa0[j] = *(const bf16x8*)(fr + 32 * PITCH + j * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[j], fb[j], acc0, 0, 0, 0);
        if (nrb > 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) fa1[j] = *(const bf16x8*)(fr + 64 * PITCH + j * 16);
#pragma unroll
          for (int j = 0; j < 4; j++) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[j], fb[j], acc1, 0, 0, 0);
        }
      }
    }
    __syncthreads();                             // the slots below overwrite the staging area
  } else {
  // workgroup j starts its walk over the K chunks at a different chunk (and wraps): with long rows (K = 4096: 8 KB pitch) every load of
  // the launch otherwise hits the same few memory channels at the same time (32 rows at ONE column offset per load, all workgroups in step)
  const int nchunk = K / SK_CHUNK;
  const int rot = (int)((blk * 5u) % (unsigned)nchunk);
  // the fragments of up to CB chunks are requested together (K = 1024 on 8 waves: both of a wave's chunks in ONE round trip; CB = 4 spills
  // inside the persistent kernels)
  for (int base = wave; base < nchunk; base += NW * CB) {
    bf16x8 fb[CB][4], fa0[CB][4], fa1[CB][4];
#pragma unroll
    for (int u = 0; u < CB; u++) {
      const int ci = base + u * NW;
      if (ci < nchunk) {
        int cc = ci + rot; if (cc >= nchunk) cc -= nchunk;
        const int k = cc * SK_CHUNK;
#pragma unroll
        for (int j = 0; j < 4; j++) fb[u][j] = *(const bf16x8*)(bp + k + j * 8);
#pragma unroll
        for (int j = 0; j < 4; j++) fa0[u][j] = *(const bf16x8*)(ap0 + k + j * 8);
        if (nrb > 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) fa1[u][j] = *(const bf16x8*)(ap1 + k + j * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CB; u++) {
      if (base + u * NW < nchunk) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[u][j], fb[u][j], acc0, 0, 0, 0);
        if (nrb > 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[u][j], fb[u][j], acc1, 0, 0, 0);
        }
      }
    }
  }
  }
  {   // D layout: column lane & 31, row (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* slot = sk_red + wave * SLOT;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      slot[row * (SK_BN + 1) + (lane & 31)] = acc0[r];
      if (nrb > 1) slot[(32 + row) * (SK_BN + 1) + (lane & 31)] = acc1[r];
    }
  }
  __syncthreads();
  if (sliced) {
    // publish this slice's partial (item i -> 16 bytes at slab[slice][i]), then the ticket; the recipe of the in-launch split-K reduce:
    // sc1 (write-through) stores, every wave drains vmcnt, barrier, ONE relaxed agent-scope fetch_add; the reader uses sc1 loads
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.slab, 0, P.slices * 64 * SK_BN * 4, 0x00020000);
    for (int i = tid; i < total; i += NT) {
      const int m = i / IPR, c = (i % IPR) * 4;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const float* q = sk_red + w * SLOT + m * (SK_BN + 1) + c;
        o[0] += q[0]; o[1] += q[1]; o[2] += q[2]; o[3] += q[3];
      }
      typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_u32x4, o), rs, (P.slice * (64 * IPR) + i) * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile int* bc = (volatile int*)sk_red;
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(P.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == P.slices - 1) __hip_atomic_store(P.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bc[0] = t;
    }
    __syncthreads();
    if (bc[0] != P.slices - 1) return;
    fetch(tid, E0);
  }
  for (int i = tid; i < total; i += NT) {
    if (i != tid) fetch(i, E0);
    const int m = i / IPR, c = (i % IPR) * 4;
    const int nn = n0 + c;
    if (nn >= N) continue;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (sliced) {
      typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
      const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.slab, 0, P.slices * 64 * SK_BN * 4, 0x00020000);
      for (int sl = 0; sl < P.slices; sl++) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (sl * (64 * IPR) + i) * 16, 0, 16));
        o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
      }
    } else {
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const float* q = sk_red + w * SLOT + m * (SK_BN + 1) + c;
        o[0] += q[0]; o[1] += q[1]; o[2] += q[2]; o[3] += q[3];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] *= P.alpha;
    if (P.bias) { o[0] += E0.b.x; o[1] += E0.b.y; o[2] += E0.b.z; o[3] += E0.b.w; }
    if (P.act != DU_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = apply_act(o[e], P.act);
    }
    if (P.gamma) { o[0] *= E0.g.x; o[1] *= E0.g.y; o[2] *= E0.g.z; o[3] *= E0.g.w; }
    if (P.row_scale) {
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] *= E0.rs;
    }
    if (P.qkv_H > 0) {
      const int gm = P.row0 + m, b = gm / P.qkv_N, tk = gm - b * P.qkv_N;
      const int hd = P.qkv_H * 64, which = nn / hd, rem = nn - which * hd;
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (bf16_t)o[e];
      *(uint2*)((bf16_t*)P.C + (long)which * P.ldc + (((long)b * P.qkv_H + (rem >> 6)) * P.qkv_Npad + tk) * 64 + (rem & 63)) = __builtin_bit_cast(uint2, t);
    } else if (P.out_bf16) {
      if (P.residual) { o[0] += E0.r.x; o[1] += E0.r.y; o[2] += E0.r.z; o[3] += E0.r.w; }
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] = (bf16_t)o[e];
      *(uint2*)((bf16_t*)P.C + (long)m * P.ldc + nn) = __builtin_bit_cast(uint2, t);
    } else {
      if (P.residual) { o[0] += E0.r.x; o[1] += E0.r.y; o[2] += E0.r.z; o[3] += E0.r.w; }
      *(float4*)((float*)P.C + (long)m * P.ldc + nn) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

}  // namespace
