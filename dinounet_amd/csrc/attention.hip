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
#include <utility>
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
//
// Workgroup = 4 waves x 32 queries sharing the K / V tiles (64 keys) of one (batch, head), two LDS buffers, one barrier per tile.
// What the counters said about the previous version (rocprofv3 --pmc, ViT-L shape): the vector ALU was busy 49 % of the time and the
// matrix pipe 24 % -- 14 VALU instructions per MFMA -- and the register-staged K / V prefetch did not overlap anything (the staging
// registers were re-used as S accumulators, so every tile began by waiting for its own global loads).  Hence:
//   * K / V tiles travel HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass, no address
//     arithmetic (wave-uniform soffset per tile), rows past N read as zeros through the descriptor's bounds check.  The images are
//     unpadded; bank conflicts are avoided by XOR-swizzling the 16-byte chunk index on the SOURCE address (K: fragment reads of 32
//     rows x one chunk; V: the 4-row x 64-byte groups of ds_read_b64_tr_b16) and un-swizzling in the read address;
//   * the S^T accumulators start at -m (running row maximum) instead of 0, so exp2 needs no subtraction; the row sum uses packed adds;
//   * K fragments of both 32-key blocks are fetched up front and the two accumulation chains alternate; V fragments of PV step i+1
//     are fetched while step i multiplies; the row maximum crosses the wave halves with v_permlane32_swap (no LDS round trip).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float other_half_max(float x) {
  // max of the two wave halves' values, in every lane.  v_permlane32_swap exchanges the upper half of its first operand with the lower half
  // of its second: fed (x, copy of x), the first register then holds the lower half's value in every lane, the second the upper half's.
  // In inline asm on purpose: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) folds fmaxf(r[0], r[1]) to r[0] (also with an
  // opaque copy as second operand) -- the running maximum of rounds 2 / 3 silently came from the keys of the LOWER half-lanes only.  Found
  // in round 4 by the spiked-key test (a large score on an upper-half key overflowed the probabilities to inf / NaN); on unspiked data the
  // two halves' maxima are close, so nothing showed.  The s_nop covers the VALU-write -> v_permlane read hazard (2 wait states).
  float a = x, b;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  return fmaxf(a, b);
}

template <int DH, bool ABL>
__global__ __launch_bounds__(256, DH == 64 ? (ABL ? 3 : 4) : 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int N,
                                                       int Npad, int dbg_arg, unsigned long long* __restrict__ probe) {
  // ABL: timing ablations (tools/attn_ablate.py, du_set_option key 4): pieces of the tile program switched off by dbg bits; the product
  // kernel is the ABL = false instantiation, where every `dbg &` test folds away
  const int dbg = ABL ? dbg_arg : 0;
  constexpr int KT = 64;                     // keys per tile
  constexpr int ROWB = DH * 2;               // bytes per K / V row (128 / 256): unpadded LDS images
  constexpr int NKK = DH / 16;               // k-steps of the S^T product
  constexpr int NDB = DH / 32;               // 32-wide dv blocks of O^T
  constexpr int TILE_B = KT * ROWB;          // 8 / 16 KB per operand tile
  constexpr int BUF_B = 2 * TILE_B;          // K image then V image
  constexpr int RPP = 1024 / ROWB;           // rows per 1-KB DMA piece (8 / 4)
  constexpr int CPR = ROWB / 16;             // 16-byte chunks per row (8 / 16)
  constexpr int PIECES = TILE_B / 1024;      // DMA pieces per operand tile (8 / 16), dealt to the 4 waves: piece = wave + 4 i
  // K / V ring.  Two buffers (one tile of prefetch).  A three-buffer ring (two tiles in flight, 3 workgroups / CU) was measured in round 3:
  // 58.5 us against 56.0 us at the dinounet_l shape -- the tile period is not memory latency (tools/attn_ablate.py: the K / V tiles sit in
  // L2 / Infinity Cache; an EMPTY tile loop still costs half the kernel), and the third buffer costs a workgroup of occupancy.
  constexpr int NBUF = 2;
  constexpr int AHEAD = NBUF - 1;            // tiles requested beyond the one being multiplied
  constexpr int DMA_PER_TILE = 2 * (PIECES / 4);   // LDS-DMA instructions per wave and tile (K + V)
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * BUF_B];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  typedef __attribute__((address_space(3))) void lds_void;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, ql = lane & 31;
  // XCD-aware order: workgroup b lands on XCD b % 8, so with the plain (q tile, head) grid the query tiles of one head sat on
  // different XCDs and every XCD pulled every head's K / V through its own L2 (FETCH_SIZE 5.5x the algorithmic bytes).  All query
  // tiles of a head share an XCD: heads are dealt round-robin to the XCDs, the tiles of a head are consecutive local indices.
  int bh = blockIdx.y, qt = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    bh = (idx / (int)gridDim.x) * 8 + xcd;
    qt = idx % (int)gridDim.x;
  }
  const int q0 = (qt * 4 + wave) * 32;
  const bool active = q0 < N;        // wave-uniform: the last query tile of N = 1029 keeps only one wave busy
  // ABL bit 64: wave 0 of workgroup (1, 0) stamps s_memtime at the segment boundaries of every tile and leaves the per-segment cycle sums
  // in probe[0..7] (du_debug_attn_probe); each stamp drains the LDS / scalar queues, so the tile runs ~10 % slower than unprobed
  const bool probing = ABL && (dbg_arg & 64) && probe && blockIdx.x == 1 && blockIdx.y == 0 && wave == 0;
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  auto stamp = [&](int j) {
    if constexpr (ABL) {
      if (probing) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (j >= 0) seg[j] += t - t_prev;
        t_prev = t;
      }
    }
  };
  const bf16_t* Qb = Q + (long)bh * Npad * DH;

  // ---- K / V descriptors (N rows: later rows of the Npad-row arrays are never read) and this lane's DMA source offsets ----
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto make_srd = [&](const bf16_t* base) -> u32x4 {
    const unsigned long long a = (unsigned long long)(const void*)base;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);    // stride 0
    d[2] = __builtin_amdgcn_readfirstlane((unsigned)(N * ROWB));             // num_records (bytes): rows >= N read as zeros
    d[3] = 0x00020000u;
    return d;
  };
  const u32x4 rk = make_srd(K + (long)bh * Npad * DH), rv = make_srd(V + (long)bh * Npad * DH);
  unsigned vk_off, vv_off;
  {
    const int rip = lane / CPR, pc = lane % CPR;          // row inside the piece, physical chunk
    // K: chunk ^= (row >> 1) & 7 (128-B rows: 2 rows per 256-B bank row) / row & 15 (256-B rows); row = (wave + 4 i) * RPP + rip
    const int swk = DH == 64 ? ((((wave & 1) << 2) | (rip >> 1)) & 7) : ((wave * 4 + rip) & 15);
    // V: the 64-byte quarter index ^= (row >> 1) & 1 (128-B rows) / row & 3 (256-B rows)
    const int swv = DH == 64 ? (((rip >> 1) & 1) << 2) : ((rip & 3) << 2);
    vk_off = (unsigned)(rip * ROWB + ((pc ^ swk) << 4));
    vv_off = (unsigned)(rip * ROWB + ((pc ^ swv) << 4));
  }
  // LDS-DMA in inline asm: the compiler must not see these LDS writes -- it orders every ds_read_b64_tr_b16 behind all LDS-DMA it knows
  // of with a vmcnt(0) (the V reads of tile t would wait for the prefetch of tile t+1).  Ordering is ours: one s_waitcnt vmcnt(0) +
  // barrier per tile, below.  M0 (the LDS destination) is saved and restored inside the statement.
  auto dma16 = [&](const u32x4& srd, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
  };
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  auto dma_tile = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < PIECES / 4; i++) {
      const int piece = wave + 4 * i;
      const unsigned soff = (unsigned)((kt * KT + piece * RPP) * ROWB);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * BUF_B + piece * 1024);
      dma16(rk, vk_off, soff, dst);
      dma16(rv, vv_off, soff, dst + TILE_B);
    }
  };

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
  float m_run = 0.f;                 // running row maximum; the first tile always sets it (a start value of -1e30 would swamp the
                                     // scores in the -m accumulator init)

  float l0 = 0.f, l1 = 0.f;          // two partial row sums (packed adds)

  // K fragment address of row (kb * 32 + ql), logical chunk 2 kk + half: row * ROWB + ((chunk ^ swz(row)) << 4)
  const int ksw = DH == 64 ? ((ql >> 1) & 7) : (ql & 15);
  const int krow = ql * ROWB;
  // V^T fragment (ds_read_b64_tr_b16) of PV step (kb, st), dv block d: 4 key rows x 64 bytes per 16-lane group; the 64-byte quarter d
  // of a row sits at quarter d ^ f, f = (row >> 1) & 1 (128-B rows) / row & 3 (256-B rows), a per-lane constant
  const int g = lane >> 4, p16 = lane & 15;
  const int vf_sel = DH == 64 ? ((p16 >> 3) & 1) : ((p16 >> 2) & 3);
  const int vlane = (4 * (g >> 1) + (p16 >> 2)) * ROWB + 32 * (g & 1) + 8 * (p16 & 3);
  auto vfrag = [&](const unsigned char* Vs, int kb, int st, int d) -> bf16x8 {
    const unsigned char* vp = Vs + vlane + (kb * 32 + 16 * st) * ROWB + ((d ^ vf_sel) << 6);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)vp);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(vp + 8 * ROWB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  const int ntiles = (N + KT - 1) / KT;
  // counted wait: everything but the youngest `tiles` tiles' DMA has landed (vector memory returns in order)
  auto wait_all_but = [&](int tiles) {
    if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (DMA_PER_TILE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
  static_assert(AHEAD <= 2 && (DMA_PER_TILE == 4 || DMA_PER_TILE == 8), "wait_all_but() knows one tile of 4 or 8 pieces in flight");
  dma_tile(0, 0);
  if (AHEAD > 1 && ntiles > 1) dma_tile(1, 1);
  wait_all_but(AHEAD > 1 && ntiles > 1 ? 1 : 0);
  // The compiler cannot see the wait above (nor the one that ends every tile), so its own wait-count bookkeeping still has the four Q
  // fragment loads "in flight" at the loop head and guarded the S^T MFMAs of EVERY tile with s_waitcnt vmcnt(3) ... vmcnt(0) -- which the
  // hardware applies to what is really outstanding there: the LDS-DMA of the NEXT tile, issued a few instructions earlier.  Every tile
  // waited for its successor's loads to land before its 7th MFMA (the prefetch overlapped nothing; ~2 us per tile and wave, found in the
  // ISA in round 3).  Re-defining the fragments here makes the compiler settle its vmcnt debt before the loop.
#pragma unroll
  for (int kk = 0; kk < NKK; kk++) asm volatile("" : "+v"(qf[kk]));
  __syncthreads();
  int cur = 0, fill = AHEAD % NBUF;                      // ring slots: the tile being multiplied, the one requested now
  for (int kt = 0; kt < ntiles; kt++) {
    const bool more = kt + 1 < ntiles;
    stamp(-1);
    if (kt + AHEAD < ntiles && !(dbg & 32)) dma_tile(fill, kt + AHEAD);
    stamp(0);                                            // 0: DMA issue // into the slot read last in tile kt - 1 (everyone is past that tile's barrier)
    const unsigned char* Ks = smem + cur * BUF_B;
    const unsigned char* Vs = Ks + TILE_B;
    if (active) {
      // ---- S^T - m = K Q^T - m for the two 32-key blocks: all K fragments first, then the two accumulation chains alternating ----
      f32x16 s[2];
      auto kfrag = [&](int kb, int kk) -> bf16x8 {
        return *(const bf16x8*)(Ks + kb * 32 * ROWB + krow + (((kk * 2 + half) ^ ksw) << 4));
      };
      f32x16 cinit;                                      // -m_run: the C operand of the first S^T MFMAs
#pragma unroll
      for (int r = 0; r < 16; r++) cinit[r] = -m_run;
      bf16x8 kf[2][2];                                   // two k-steps of both key blocks in flight
#pragma unroll
      for (int kb = 0; kb < 2; kb++) { kf[kb][0] = kfrag(kb, 0); kf[kb][1] = kfrag(kb, 1); }
      if (dbg & 8) { s[0] = cinit; s[1] = cinit; } else {
      if (!(dbg & 128)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < NKK; kk++) {
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][kk & 1], qf[kk], kk == 0 ? cinit : s[kb], 0, 0, 0);
          if (kk + 2 < NKK) kf[kb][kk & 1] = kfrag(kb, kk + 2);
        }
      }
      if (!(dbg & 128)) __builtin_amdgcn_s_setprio(0);
      }
      if (probing) { asm volatile("" :: "v"(s[0][0]), "v"(s[1][0])); }
      stamp(1);                                          // 1: K fragment reads + S^T MFMAs (results landed)
      // first V fragments: in flight under the softmax
      bf16x8 vf[2][NDB];
#pragma unroll
      for (int d = 0; d < NDB; d++) vf[0][d] = vfrag(Vs, 0, 0, d);
      // ---- mask the tail keys (last tile only), online softmax in base 2 (q was pre-scaled by Dh^-1/2 * log2 e) ----
      if (!more) {
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int key = kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= N) s[kb][r] = -3e30f;
          }
      }
      float mx = s[0][0];
      if (!(dbg & 2)) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
      mx = other_half_max(mx);                           // tile maximum relative to the running maximum
      }
      // deferred rescale: keep the old running max while the tile's max exceeds it by <= 8 (P <= 2^8, exact in the fp32
      // accumulators, bf16 P keeps its relative precision); the branch is wave-uniform
      if (!(dbg & 2) && (kt == 0 || !__all(mx <= 8.0f))) {
        const float dm = kt == 0 ? mx : fmaxf(mx, 0.f);  // new max - old max (the first tile adopts its maximum whatever its sign)
        const float alpha = __builtin_amdgcn_exp2f(-dm);
        m_run += dm;
        l0 *= alpha; l1 *= alpha;
#pragma unroll
        for (int d = 0; d < NDB; d++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc_o[d][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) s[kb][r] -= dm;
      }
      stamp(2);                                          // 2: mask, running maximum, (rare) rescale
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float p0 = s[kb][r], p1 = s[kb][r + 1];
          if (!(dbg & 1)) {
            p0 = __builtin_amdgcn_exp2f(p0); p1 = __builtin_amdgcn_exp2f(p1);
            if constexpr (ABL) asm volatile("" : "+v"(p0), "+v"(p1));     // keeps the ablation's branch a branch (no speculated exp + select)
          }
          s[kb][r] = p0; s[kb][r + 1] = p1;
          l0 += p0; l1 += p1;
        }

      if (probing) { asm volatile("" :: "v"(s[0][15]), "v"(s[1][15]), "v"(l0), "v"(l1)); }
      stamp(3);                                          // 3: exp2 + row sums
      // ---- O^T += V^T P^T: B fragment of step (kb, st) = accumulator registers 8st..8st+7 of s[kb] (keys kb*32 + 16st + 4half +
      //      {0..3, 8..11}); the V fragments of the next step are fetched while this step's MFMAs run ----
#pragma unroll
      for (int step = 0; step < 4; step++) {
        const int kb = step >> 1, st = step & 1;
        if (step < 3) {
#pragma unroll
          for (int d = 0; d < NDB; d++) vf[(step + 1) & 1][d] = vfrag(Vs, (step + 1) >> 1, (step + 1) & 1, d);
        }
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; e++) pf[e] = (bf16_t)s[kb][8 * st + e];
        if (!(dbg & 4)) {
        if (!(dbg & 128)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int d = 0; d < NDB; d++) acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[step & 1][d], pf, acc_o[d], 0, 0, 0);
        if (!(dbg & 128)) __builtin_amdgcn_s_setprio(0);
        }
      }
    }
    // this wave's share of the NEXT tile has landed (the barrier = everyone's); the tile after it may stay in flight
    if (probing) { asm volatile("" :: "v"(acc_o[0][0]), "v"(acc_o[NDB - 1][15])); }
    stamp(4);                                            // 4: P conversion, V fragment reads, PV MFMAs (results landed)
    if (!(dbg & 16)) {
    wait_all_but(AHEAD > 1 && kt + 2 < ntiles ? 1 : 0);
    __syncthreads();
    }
    stamp(5);                                            // 5: end-of-tile DMA wait + barrier
    cur = cur + 1 == NBUF ? 0 : cur + 1;
    fill = fill + 1 == NBUF ? 0 : fill + 1;
  }

  if constexpr (ABL) {
    if (probing && lane == 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) probe[j] = seg[j];
    }
  }
  // ---- finalize: O[q][dv] = O^T[dv][q] / l ----
  const float l_part = l0 + l1;
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

// ------------------------------------------------------------------------------------------------------
// flash attention forward, round 4: 64 queries per wave (two 32-query blocks sharing every K / V fragment), two waves per SIMD
//
// Why: the 4-waves-per-SIMD kernel above spends ~340 instruction issues around 16 MFMAs per 64-key tile and a SIMD issues
// about one instruction per 4-5 cycles whatever it is; its ceiling at d_head 64 is 0.38 of the matrix pipe.  What this kernel removes:
//   * the running row maximum.  The scores of the FIRST tile a wave multiplies set the reference m of every query (a proper row maximum,
//     so the largest probability of a row is >= 1 for good: the row sum cannot underflow); every later tile is exponentiated against that
//     m without looking at its own maximum -- P = 2^(s - m) may exceed 1 by any factor fp32 holds (P enters the PV product as bf16, the row
//     sum and O accumulate in fp32: precision is relative, only the RANGE matters).  A tile whose lane-partial row sum exceeds 2^60 (or is
//     not finite) takes a cold path that multiplies its scores again straight from global memory, adopts the true maximum and rescales O
//     and l (tests force it with spiked keys and with the threshold turned down to zero).  32 v_max + a lane exchange + a vote per 32
//     scores and the max -> exp dependency are gone: per 32-query block and tile 32 v_exp + 32 v_add + 16 v_cvt_pk stay around 16 MFMAs;
//   * half of the K / V fragment reads, DMA issues, waits, barriers and loop overhead per MFMA: both query blocks share them;
//   * masks and -m bookkeeping in the loop: the ragged tile (keys >= N) is multiplied FIRST (softmax does not care about key order), in the
//     prologue that computes the masked maximum anyway; in the loop m rides in as the C operand of the first S^T MFMA of a chain.
// (The one-wave-per-SIMD / 512-register form of this kernel was compiled first: hipcc parks MFMA operands in the accumulator half of the
// register file and copies them back and forth -- 600 v_accvgpr_read / _write per tile -- so the kernel stays inside 256 VGPRs and two
// workgroups share a CU; the partner wave of a SIMD fills the matrix pipe while this one exponentiates.)
// K / V tiles arrive by LDS-DMA into a three-slot ring (two tiles of flight), one barrier per tile.
// NQB = 1 (d_head 128: O alone is 128 registers for two blocks): 32 queries per wave, the same program.
// ------------------------------------------------------------------------------------------------------
template <int DH, int NQB, bool PROBE = false, int OCC = 2, int ABL = 0, int PRIO = 0>
__global__ __launch_bounds__(256, OCC) void attn_fwd_w64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int N,
                                                                int Npad, float thresh, unsigned long long* __restrict__ probe) {
  constexpr int KT = 64;                     // keys per tile
  constexpr int QW = 32 * NQB;               // queries per wave
  constexpr int ROWB = DH * 2;               // bytes per K / V row
  constexpr int NKK = DH / 16;               // k-steps of the S^T product
  constexpr int NDB = DH / 32;               // 32-wide dv blocks of O^T
  constexpr int TILE_B = KT * ROWB;          // 8 / 16 KB per operand tile
  constexpr int SLOT_B = 2 * TILE_B;         // K image then V image
  constexpr int RPP = 1024 / ROWB;           // rows per 1-KB DMA piece
  constexpr int CPR = ROWB / 16;             // 16-byte chunks per row
  constexpr int PIECES = TILE_B / 1024;      // DMA pieces per operand tile, dealt to the 4 waves
  constexpr int DPO = PIECES / 4;            // LDS-DMA instructions per wave and operand tile
  constexpr int NSLOT = DH == 64 ? (OCC > 3 ? 2 : 3) : 2;    // 48 / 64 KB per workgroup, two workgroups per CU (OCC 4: 32 KB each)
  constexpr int AHEAD = NSLOT - 1;           // tiles requested beyond the one being multiplied
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * SLOT_B];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  unsigned long long t_entry = 0;
  if constexpr (PROBE) t_entry = __builtin_amdgcn_s_memtime();
  if constexpr (PRIO == 1 || PRIO == 2) {
    // static, DISTINCT priorities for the waves that share a SIMD (they belong to different workgroups: wave slot = HW_ID[3:0])
    const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));
    const unsigned pr = PRIO == 1 ? (slot & 3) : 3 - (slot & 3);
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, ql = lane & 31;
  int bh = blockIdx.y, qt = blockIdx.x;
  if ((gridDim.y & 7) == 0) {                // all query tiles of a head on one XCD (its K / V stay in that L2)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    const int gx = gridDim.x, hpx = gridDim.y >> 3;          // query tiles per head, heads per XCD
    if (gx > 1 && N % (4 * QW) != 0 && N % (4 * QW) <= 2 * QW) {
      // the LAST query tile of a head is nearly empty (N = 1029: 5 queries, one wave busy for the whole key walk): dispatched behind the
      // full tiles it ran alone at the end (+ 16 us at the dinounet_l shape); dispatched FIRST it shares a CU with a full workgroup and the
      // slot it frees takes a late full one
      if (idx < hpx) { bh = idx * 8 + xcd; qt = gx - 1; }
      else { const int j = idx - hpx; bh = (j / (gx - 1)) * 8 + xcd; qt = j % (gx - 1); }
    } else {
      bh = (idx / gx) * 8 + xcd;
      qt = idx % gx;
    }
  }
  // (Splitting the keys of a nearly empty last query tile over its four waves -- every wave its own reference maximum, partial (m, l, O)
  //  combined through LDS -- was built and measured in round 4: the second instantiation of the tile walk cost the common one its last
  //  registers (spilled Q fragments, + 8 us on the dinounet_l shape).  Dispatching those workgroups first, above, is what stayed.)
  const int q0 = (qt * 4 + wave) * QW;
  const bool active = q0 < N;                // wave-uniform
  const bf16_t* Qb = Q + (long)bh * Npad * DH;
  const bf16_t* Kb = K + (long)bh * Npad * DH;

  auto make_srd = [&](const bf16_t* base) __attribute__((always_inline)) -> u32x4 {
    const unsigned long long a = (unsigned long long)(const void*)base;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane((unsigned)(N * ROWB));             // rows >= N read as zeros
    d[3] = 0x00020000u;
    return d;
  };
  const u32x4 rk = make_srd(Kb), rv = make_srd(V + (long)bh * Npad * DH);
  unsigned vk_off, vv_off;                   // the LDS images and their source-side swizzles are those of attn_fwd_kernel
  {
    const int rip = lane / CPR, pc = lane % CPR;
    const int swk = DH == 64 ? ((((wave & 1) << 2) | (rip >> 1)) & 7) : ((wave * 4 + rip) & 15);
    const int swv = DH == 64 ? (((rip >> 1) & 1) << 2) : ((rip & 3) << 2);
    vk_off = (unsigned)(rip * ROWB + ((pc ^ swk) << 4));
    vv_off = (unsigned)(rip * ROWB + ((pc ^ swv) << 4));
  }
  auto dma16 = [&](const u32x4& srd, unsigned voff, unsigned soff, unsigned lds_addr) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
  };
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int ntiles = (N + KT - 1) / KT;
  // order of multiplication: step 0 = the ragged last tile, step i = tile i - 1; steps past the end name a tile beyond N (the DMA returns
  // zeros through the descriptor's bounds check and nobody reads them: every step issues the same number of pieces, so the counted waits
  // hold for any N)
  auto tile_of = [&](int i) __attribute__((always_inline)) -> int { return i == 0 ? ntiles - 1 : (i < ntiles ? i - 1 : ntiles); };
  auto dma_step = [&](int i, int slot) __attribute__((always_inline)) {
    const int tile = tile_of(i);
#pragma unroll
    for (int j = 0; j < DPO; j++) {
      const int piece = wave + 4 * j;
      const unsigned soff = (unsigned)((tile * KT + piece * RPP) * ROWB);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * SLOT_B + piece * 1024);
      dma16(rk, vk_off, soff, dst);
      dma16(rv, vv_off, soff, dst + TILE_B);
    }
  };

  // Q fragments (B operand): lane (q, half) holds Q[q][kk*16 + half*8 .. +8]
  bf16x8 qf[NQB][NKK];
#pragma unroll
  for (int qb = 0; qb < NQB; qb++) {
    int qr = q0 + qb * 32 + ql; if (qr > N - 1) qr = N - 1;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) qf[qb][kk] = *(const bf16x8*)(Qb + (long)qr * DH + kk * 16 + half * 8);
  }
#pragma unroll
  for (int a = 0; a < AHEAD; a++) dma_step(a, a);
#pragma unroll
  for (int qb = 0; qb < NQB; qb++)
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) asm volatile("" : "+v"(qf[qb][kk]));      // the compiler settles its own vmcnt debt (the Q loads) here

  f32x16 acc_o[NQB][NDB], s[NQB][2], cinit[NQB];
  bf16x8 pf[NQB][4];                         // P^T of the current tile: B fragments of the four PV steps, per query block
  float m[NQB], l[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; qb++) {
    l[qb] = 0.f; m[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) cinit[qb][r] = 0.f;
#pragma unroll
    for (int d = 0; d < NDB; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc_o[qb][d][r] = 0.f;
  }

  // fragment addresses
  const int ksw = DH == 64 ? ((ql >> 1) & 7) : (ql & 15);
  const int krow = ql * ROWB;
  const int g = lane >> 4, p16 = lane & 15;
  const int vf_sel = DH == 64 ? ((p16 >> 3) & 1) : ((p16 >> 2) & 3);
  const int vlane = (4 * (g >> 1) + (p16 >> 2)) * ROWB + 32 * (g & 1) + 8 * (p16 & 3);
  auto kfrag = [&](const unsigned char* Ks, int kb, int kk) __attribute__((always_inline)) -> bf16x8 {
    return *(const bf16x8*)(Ks + kb * 32 * ROWB + krow + (((kk * 2 + half) ^ ksw) << 4));
  };
  auto vfrag = [&](const unsigned char* Vs, int step, int d) __attribute__((always_inline)) -> bf16x8 {
    const unsigned char* vp = Vs + vlane + (16 * step) * ROWB + ((d ^ vf_sel) << 6);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)vp);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(vp + 8 * ROWB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  // S^T - m of every query block against the K image at Ks: each K fragment is read once and multiplied into all blocks
  // ABL (tools/attn_ablate.py: timing only, results are garbage): 1 no exp2, 2 no S^T MFMAs, 4 no PV MFMAs, 8 no K / V fragment reads,
  // 16 no DMA and no wait for it, 32 no barrier, 64 no row sums, 128 no bf16 packing
  auto qk = [&](const unsigned char* Ks, const f32x16 (&cin)[NQB]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < NKK; kk++)
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        bf16x8 a;
        if constexpr (ABL & 8) { a = qf[0][(kk + kb) % NKK]; asm volatile("" : "+v"(a)); } else a = kfrag(Ks, kb, kk);
#pragma unroll
        for (int qb = 0; qb < NQB; qb++) {
          if constexpr (ABL & 2) { if (kk == 0) s[qb][kb] = cin[qb]; asm volatile("" : "+v"(s[qb][kb]) : "v"(a)); }
          else s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[qb][kk], kk == 0 ? cin[qb] : s[qb][kb], 0, 0, 0);
        }
      }
  };
  auto pv = [&](const unsigned char* Vs) __attribute__((always_inline)) {
#pragma unroll
    for (int step = 0; step < 4; step++)
#pragma unroll
      for (int d = 0; d < NDB; d++) {
        bf16x8 a;
        if constexpr (ABL & 8) { a = qf[0][(step + d) % NKK]; asm volatile("" : "+v"(a)); } else a = vfrag(Vs, step, d);
#pragma unroll
        for (int qb = 0; qb < NQB; qb++) {
          if constexpr (ABL & 4) asm volatile("" : "+v"(acc_o[qb][d]) : "v"(a), "v"(pf[qb][step]));
          else acc_o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[qb][step], acc_o[qb][d], 0, 0, 0);
        }
      }
  };
  auto pv_qb = [&](int qb, const unsigned char* Vs) __attribute__((always_inline)) {
#pragma unroll
    for (int step = 0; step < 4; step++)
#pragma unroll
      for (int d = 0; d < NDB; d++) {
        bf16x8 a;
        if constexpr (ABL & 8) { a = qf[0][(step + d) % NKK]; asm volatile("" : "+v"(a)); } else a = vfrag(Vs, step, d);
        if constexpr (ABL & 4) asm volatile("" : "+v"(acc_o[qb][d]) : "v"(a), "v"(pf[qb][step]));
        else acc_o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[qb][step], acc_o[qb][d], 0, 0, 0);
      }
  };
  // P = 2^S (S already carries -m), packed for the PV steps; returns this lane's partial row sum of the tile
  auto expo = [&](int qb) __attribute__((always_inline)) -> float {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float p0, p1;
        if constexpr (ABL & 1) { p0 = s[qb][kb][r]; p1 = s[qb][kb][r + 1]; }
        else { p0 = __builtin_amdgcn_exp2f(s[qb][kb][r]); p1 = __builtin_amdgcn_exp2f(s[qb][kb][r + 1]); }
        if constexpr (!(ABL & 64)) { t0 += p0; t1 += p1; }
        if constexpr (ABL & 128) {
          if ((r & 3) == 0) { unsigned u = __builtin_bit_cast(unsigned, p0); asm volatile("" : "+v"(u) : "v"(p1));
                              bf16x2 t2 = __builtin_bit_cast(bf16x2, u); pf[qb][kb * 2 + (r >> 3)][r & 7] = t2[0]; pf[qb][kb * 2 + (r >> 3)][(r & 7) + 1] = t2[1]; }
          else { unsigned u = __builtin_bit_cast(unsigned, p1); asm volatile("" : "+v"(u) : "v"(p0));
                 bf16x2 t2 = __builtin_bit_cast(bf16x2, u); pf[qb][kb * 2 + (r >> 3)][r & 7] = t2[0]; pf[qb][kb * 2 + (r >> 3)][(r & 7) + 1] = t2[1]; }
        } else {
          pf[qb][kb * 2 + (r >> 3)][r & 7] = (bf16_t)p0;
          pf[qb][kb * 2 + (r >> 3)][(r & 7) + 1] = (bf16_t)p1;
        }
      }
    return t0 + t1;
  };
  // the cold path: tile `tile` of query block qb overflowed against the reference m: scores again (K fragments straight from global
  // memory), true maximum, rescale of what the block has accumulated
  auto rescue = [&](int qb, int tile) __attribute__((always_inline)) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    f32x16 rs[2];
#pragma unroll
    for (int kk = 0; kk < NKK; kk++)
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 a = *(const bf16x8*)(Kb + (long)(tile * KT + kb * 32 + ql) * DH + kk * 16 + half * 8);
        rs[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[qb][kk], kk == 0 ? z : rs[kb], 0, 0, 0);
      }
    float mx = rs[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, rs[kb][r]);
    mx = other_half_max(mx);
    const float mn = fmaxf(m[qb], mx);
    const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
    m[qb] = mn;
    l[qb] *= alpha;
#pragma unroll
    for (int d = 0; d < NDB; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc_o[qb][d][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; r++) cinit[qb][r] = -mn;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) s[qb][kb][r] = rs[kb][r] - mn;
  };

  // the first step of a wave: scores against a zero reference, (step 0: keys >= N masked,) row maximum -> m, P, PV
  auto first_step = [&](const unsigned char* Ks, bool ragged) __attribute__((always_inline)) {
    const int t0 = ntiles - 1;
    qk(Ks, cinit);                           // (cinit is still zero: the C operand costs no extra registers)
#pragma unroll
    for (int qb = 0; qb < NQB; qb++) {
      float mx = -3e38f;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = t0 * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (ragged && key >= N) s[qb][kb][r] = -3e38f;
          mx = fmaxf(mx, s[qb][kb][r]);
        }
      mx = other_half_max(mx);
      m[qb] = mx;
#pragma unroll
      for (int r = 0; r < 16; r++) cinit[qb][r] = -mx;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) s[qb][kb][r] -= mx;
      l[qb] = expo(qb);
    }
    pv(Ks + TILE_B);
  };
  // PROBE (tools/attn_ablate.py, du_set_option(4, 64)): wave 0 of workgroup (1, 0) stamps s_memtime at the segment boundaries of every step
  // and leaves the per-segment cycle sums in probe[0..7]; each stamp drains the wave's queues, so the probed wave runs slower than unprobed
  const bool probing = PROBE && probe && blockIdx.x == 1 && blockIdx.y == 0 && wave == 0;
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  unsigned long long t_begin = 0;
  auto stamp = [&](int j) __attribute__((always_inline)) {
    if constexpr (PROBE) {
      if (probing) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (j >= 0) seg[j] += t - t_prev;
        t_prev = t;
      }
    }
  };
  // one step after a wave's first: full tile, no masks, no maxima
  auto later_step = [&](const unsigned char* Ks, int i) __attribute__((always_inline)) {
    qk(Ks, cinit);
    if constexpr (PROBE) { if (probing) { asm volatile("" :: "v"(s[0][0][0]), "v"(s[NQB - 1][1][15])); } }
    stamp(1);
    float lt[NQB];
    if constexpr (NQB == 2 && PRIO == 3) {
      // the second block's exponentials ride beside the first block's PV MFMAs (the first block's rode beside the second block's S^T chain)
      lt[0] = expo(0);
      if (__builtin_expect(!__all(lt[0] < thresh), 0)) { rescue(0, i - 1); lt[0] = expo(0); }
      l[0] += lt[0];
      pv_qb(0, Ks + TILE_B);
      lt[1] = expo(1);
      if (__builtin_expect(!__all(lt[1] < thresh), 0)) { rescue(1, i - 1); lt[1] = expo(1); }
      l[1] += lt[1];
      pv_qb(1, Ks + TILE_B);
    } else {
#pragma unroll
      for (int qb = 0; qb < NQB; qb++) lt[qb] = expo(qb);
#pragma unroll
      for (int qb = 0; qb < NQB; qb++) {
        if (__builtin_expect(!__all(lt[qb] < thresh), 0)) {
          rescue(qb, i - 1);
          lt[qb] = expo(qb);
        }
        l[qb] += lt[qb];
      }
      if constexpr (PROBE) { if (probing) { asm volatile("" :: "v"(pf[0][0]), "v"(pf[NQB - 1][3]), "v"(l[0])); } }
      stamp(2);
      pv(Ks + TILE_B);
    }
    if constexpr (PROBE) { if (probing) { asm volatile("" :: "v"(acc_o[0][0][0]), "v"(acc_o[NQB - 1][NDB - 1][15])); } }
    stamp(3);
  };
  {
    // ---- step 0: the ragged tile with its masked row maximum ----
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 1) * 2 * DPO) : "memory");      // step 0 landed (this wave's share; the barrier = everyone's)
    __builtin_amdgcn_s_barrier();
    dma_step(AHEAD, AHEAD % NSLOT);
    if (active) first_step(smem, true);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 1) * 2 * DPO) : "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (PROBE) t_begin = __builtin_amdgcn_s_memtime();
    // ---- steps 1 .. ntiles - 1 ----
    int cur = 1 % NSLOT, fill = (1 + AHEAD) % NSLOT;
    for (int i = 1; i < ntiles; i++) {
      stamp(-1);
      if constexpr (!(ABL & 16)) dma_step(i + AHEAD, fill);               // into the slot read in step i - 1: everyone is past that step's barrier
      stamp(0);
      if (active) later_step(smem + cur * SLOT_B, i);
      if constexpr (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 1) * 2 * DPO) : "memory");    // step i + 1 has landed; later requests stay in flight
      stamp(4);
      if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
      stamp(5);
      cur = cur + 1 == NSLOT ? 0 : cur + 1;
      fill = fill + 1 == NSLOT ? 0 : fill + 1;
    }
  }

  if constexpr (PROBE) {
    if (probing && lane == 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) probe[j] = seg[j];
    }
    // census: when and where every workgroup's loop ran (HW_ID: wave slot, SIMD, CU, SE; XCC_ID): entries of 4 words behind the 8 sums
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (probe && wave == 0 && lane == 0 && wg < 1024) {
      probe[8 + 8 * wg] = t_begin;
      probe[8 + 8 * wg + 1] = __builtin_amdgcn_s_memtime();
      probe[8 + 8 * wg + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_REG_HW_ID, 32 bits
      probe[8 + 8 * wg + 3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));         // HW_REG_XCC_ID, 4 bits
      probe[8 + 8 * wg + 4] = t_entry;
    }
  }
  // ---- finalize: O[q][dv] = O^T[dv][q] / l ----
  if (!active) return;
  const int b = bh / H, h = bh % H;
#pragma unroll
  for (int qb = 0; qb < NQB; qb++) {
    const float lsum = l[qb] + __shfl_xor(l[qb], 32, 64);
    const float inv = 1.f / lsum;
    const int q = q0 + qb * 32 + ql;
    if (q < N) {
      bf16_t* op = O + ((long)b * N + q) * ((long)H * DH) + (long)h * DH;
#pragma unroll
      for (int d = 0; d < NDB; d++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; e++) o4[e] = (bf16_t)(acc_o[qb][d][g4 * 4 + e] * inv);
          *(bf16x4*)(op + d * 32 + 8 * g4 + 4 * half) = o4;
        }
    }
  }
  if constexpr (PROBE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (probe && wave == 0 && lane == 0 && wg < 1024) probe[8 + 8 * wg + 5] = __builtin_amdgcn_s_memtime();
  }
}

// ------------------------------------------------------------------------------------------------------
// the same algorithm as attn_fwd_w64_kernel (d_head 64, 64 queries per wave, no running maximum), software-pipelined INSIDE the wave
//
// What round 4's measurements said (tools/scratch/mix_bench.hip, the ablation table of tools/attn_ablate.py):
//   * v_mfma_f32_32x32x16_bf16 followed by 2 v_exp + 2 v_add + 1 v_cvt_pk IN THE SAME WAVE runs at 37.8 cycles per MFMA (5 plain VALU ops:
//     33.5) -- the vector ALU work hides under the matrix pipe when it is interleaved at that grain;
//   * the same work in PHASES (a run of MFMAs, then a run of VALU) does not overlap across the waves of a SIMD: the ablation of the phased
//     kernel is additive (MFMAs alone 14.5 us, softmax VALU alone 11.5 us, together 29 us), whatever the occupancy (2, 3, 4 waves).
// So every MFMA group needs an INDEPENDENT VALU chain beside it.  The two query blocks A, B of a wave alternate (' = the previous step):
//     G1:  S_A = K Q_A - m_A  (8 MFMA)   beside   P_B'[keys 32..63] = 2^S_B'        check B'
//     G2:  O_B += V' P_B'     (8 MFMA)   beside   P_A[keys 0..31]   = 2^S_A
//     G3:  S_B = K Q_B - m_B  (8 MFMA)   beside   P_A[keys 32..63]  = 2^S_A         check A
//     G4:  O_A += V P_A       (8 MFMA)   beside   P_B[keys 0..31]   = 2^S_B
// S is single-buffered (a block's scores are consumed before its next S^T chain starts), B trails A by half a step.  K / V arrive by LDS-DMA
// into two three-slot rings (V of the previous step is still read in G2), one barrier per step.
// ------------------------------------------------------------------------------------------------------
template <int N_> struct AIC { static constexpr int value = N_; };
template <bool PROBE = false, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int N,
                                                                 int Npad, float thresh, unsigned long long* __restrict__ probe) {
  constexpr int DH = 64, KT = 64, ROWB = 128, NKK = 4, NDB = 2, TILE_B = KT * ROWB, RPP = 8, CPR = 8, DPO = 2, NSLOT = 3;
  constexpr int KSLOT = 2;                   // K of step i is read in step i only: two slots (V: also in G2 of step i + 1: three)
  constexpr int QOFF = (KSLOT + NSLOT) * TILE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[QOFF + 4 * TILE_B];      // K ring, V ring, Q image of every wave (64 rows each)
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, ql = lane & 31;
  int bh = blockIdx.y, qt = blockIdx.x;
  if ((gridDim.y & 7) == 0) {                // all query tiles of a head on one XCD (its K / V stay in that L2)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
    bh = (idx / (int)gridDim.x) * 8 + xcd;
    qt = idx % (int)gridDim.x;
  }
  const int q0 = (qt * 4 + wave) * 64;
  const bool active = q0 < N;                // wave-uniform
  const bf16_t* Qb = Q + (long)bh * Npad * DH;
  const bf16_t* Kb = K + (long)bh * Npad * DH;

  auto make_srd = [&](const bf16_t* base) __attribute__((always_inline)) -> u32x4 {
    const unsigned long long a = (unsigned long long)(const void*)base;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane((unsigned)(N * ROWB));             // rows >= N read as zeros
    d[3] = 0x00020000u;
    return d;
  };
  const u32x4 rk = make_srd(Kb), rv = make_srd(V + (long)bh * Npad * DH);
  unsigned vk_off, vv_off;                   // LDS images and source-side swizzles of attn_fwd_kernel
  {
    const int rip = lane / CPR, pc = lane % CPR;
    const int swk = (((wave & 1) << 2) | (rip >> 1)) & 7;
    const int swv = ((rip >> 1) & 1) << 2;
    vk_off = (unsigned)(rip * ROWB + ((pc ^ swk) << 4));
    vv_off = (unsigned)(rip * ROWB + ((pc ^ swv) << 4));
  }
  auto dma16 = [&](const u32x4& srd, unsigned voff, unsigned soff, unsigned lds_addr) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
  };
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int ntiles = (N + KT - 1) / KT;
  // step 0 = the ragged last tile, step i = tile i - 1; steps past the end name a tile beyond N (zeros through the bounds check, never read)
  auto tile_of = [&](int i) __attribute__((always_inline)) -> int { return i == 0 ? ntiles - 1 : (i < ntiles ? i - 1 : ntiles); };
  auto dma_op = [&](const u32x4& srd, unsigned voff, int slot_base, int step) __attribute__((always_inline)) {
    const int tile = tile_of(step);
#pragma unroll
    for (int j = 0; j < DPO; j++) {
      const int piece = wave + 4 * j;
      const unsigned soff = (unsigned)((tile * KT + piece * RPP) * ROWB);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot_base + piece * 1024);
      dma16(srd, voff, soff, dst);
    }
  };
  auto dma_k = [&](int step) __attribute__((always_inline)) { dma_op(rk, vk_off, (step % KSLOT) * TILE_B, step); };
  auto dma_v = [&](int step) __attribute__((always_inline)) { dma_op(rv, vv_off, (KSLOT + step % NSLOT) * TILE_B, step); };
  auto kslot = [&](int step) __attribute__((always_inline)) -> const unsigned char* { return smem + (step % KSLOT) * TILE_B; };
  auto vslot = [&](int step) __attribute__((always_inline)) -> const unsigned char* { return smem + (KSLOT + step % NSLOT) * TILE_B; };

  // this wave's 64 query rows: an LDS image like a K tile's (rows past N read as zeros; they are never stored), 8 pieces of 8 rows; the B
  // fragments of the S^T chains are read from it (in registers they cost 32 VGPRs the pipelined loop does not have)
  const unsigned char* Qs = smem + QOFF + wave * TILE_B;
  {
    const u32x4 rq = make_srd(Qb);
    const int rip = lane / CPR, pc = lane % CPR;
#pragma unroll
    for (int piece = 0; piece < 8; piece++) {
      const int swq = (((piece & 1) << 2) | (rip >> 1)) & 7;
      const unsigned voff = (unsigned)(rip * ROWB + ((pc ^ swq) << 4));
      const unsigned soff = (unsigned)((q0 + piece * RPP) * ROWB);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + QOFF + wave * TILE_B + piece * 1024);
      dma16(rq, voff, soff, dst);
    }
  }
  dma_k(0); dma_v(0); dma_k(1); dma_v(1);

  f32x16 acc_o[2][NDB], s[2][2], cinit[2];
  bf16x8 pf[2][4];
  float m[2], l[2];
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    l[qb] = 0.f; m[qb] = 0.f;
#pragma unroll
    for (int d = 0; d < NDB; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc_o[qb][d][r] = 0.f;
  }
  const int ksw = (ql >> 1) & 7;
  const int krow = ql * ROWB;
  const int g = lane >> 4, p16 = lane & 15;
  const int vf_sel = (p16 >> 3) & 1;
  const int vlane = (4 * (g >> 1) + (p16 >> 2)) * ROWB + 32 * (g & 1) + 8 * (p16 & 3);
  auto kfrag = [&](const unsigned char* Ks, int kb, int kk) __attribute__((always_inline)) -> bf16x8 {
    return *(const bf16x8*)(Ks + kb * 32 * ROWB + krow + (((kk * 2 + half) ^ ksw) << 4));
  };
  auto vfrag = [&](const unsigned char* Vs, int step, int d) __attribute__((always_inline)) -> bf16x8 {
    const unsigned char* vp = Vs + vlane + (16 * step) * ROWB + ((d ^ vf_sel) << 6);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)vp);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(vp + 8 * ROWB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto qfrag = [&](int qb, int kk) __attribute__((always_inline)) -> bf16x8 { return kfrag(Qs, qb, kk); };
  auto qk = [&](int qb, const unsigned char* Ks, const f32x16& cin) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
      const bf16x8 b = qfrag(qb, kk);
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
        s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(Ks, kb, kk), b, kk == 0 ? cin : s[qb][kb], 0, 0, 0);
    }
  };
  auto pv = [&](int qb, const unsigned char* Vs) __attribute__((always_inline)) {
#pragma unroll
    for (int step = 0; step < 4; step++)
#pragma unroll
      for (int d = 0; d < NDB; d++)
        acc_o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(Vs, step, d), pf[qb][step], acc_o[qb][d], 0, 0, 0);
  };
  // P = 2^S of one 32-key block, packed for its two PV steps; returns this lane's partial row sum
  auto expo = [&](int qb, int kb) __attribute__((always_inline)) -> float {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __builtin_amdgcn_exp2f(s[qb][kb][r]), p1 = __builtin_amdgcn_exp2f(s[qb][kb][r + 1]);
      t0 += p0; t1 += p1;
      pf[qb][kb * 2 + (r >> 3)][r & 7] = (bf16_t)p0;
      pf[qb][kb * 2 + (r >> 3)][(r & 7) + 1] = (bf16_t)p1;
    }
    return t0 + t1;
  };
  // ---- the pipelined step as 32 pinned slots: slot n = MFMA n + the two exponentials of pair n + the sums / packing of pair n - 1 + the
  //      LDS reads of the fragments MFMA n + 2 needs; nothing moves across a slot boundary (sched_barrier) ----
  //   MFMA n:  0..7  S_A chain (kk = n >> 1, kb = n & 1)      8..15 O_B += V' P_B' (step = (n - 8) >> 1, d = n & 1)
  //           16..23 S_B chain                                24..31 O_A += V P_A
  //   pair n: chain n >> 3 = 0: (B, keys 32..63) of the previous step, 1: (A, 0..31), 2: (A, 32..63), 3: (B, 0..31); scores 2 (n & 7), + 1
  bf16x8 kfr[3], qfr[2], vfr[3];            // rotating fragment registers (compile-time indices)
  float pp0 = 0.f, pp1 = 0.f;               // the pair whose sums / packing are pending
  float ta0 = 0.f, ta1 = 0.f, tb0 = 0.f, tb1 = 0.f;      // partial row sums of the step in progress (block A, block B)
  auto chain_qb = [](int c) constexpr { return (c == 1 || c == 2) ? 0 : 1; };
  auto chain_kb = [](int c) constexpr { return (c == 0 || c == 2) ? 1 : 0; };
  auto pair_exp = [&](auto nc) __attribute__((always_inline)) {
    constexpr int n = decltype(nc)::value, c = (n >> 3) & 3, jj = n & 7, qb = (c == 1 || c == 2) ? 0 : 1, kb = (c == 0 || c == 2) ? 1 : 0;
    if constexpr (ABL & 1) { pp0 = s[qb][kb][2 * jj]; pp1 = s[qb][kb][2 * jj + 1]; }
    else {
      pp0 = __builtin_amdgcn_exp2f(s[qb][kb][2 * jj]);
      pp1 = __builtin_amdgcn_exp2f(s[qb][kb][2 * jj + 1]);
    }
  };
  auto pair_finish = [&](auto nc) __attribute__((always_inline)) {        // sums and packing of pair n (its exponentials are in pp0 / pp1)
    constexpr int n = decltype(nc)::value, c = (n >> 3) & 3, jj = n & 7, qb = (c == 1 || c == 2) ? 0 : 1, kb = (c == 0 || c == 2) ? 1 : 0;
    // (the empty asm statements pin the sums and the packed pair to THIS slot: left alone, the SLP vectoriser pairs the adds of different
    //  slots into v_pk_add_f32 and the conversions sink to the end of the region -- runs of VALU that no MFMA covers)
    if constexpr (!(ABL & 64)) {
      if constexpr (qb == 0) { ta0 += pp0; ta1 += pp1; asm volatile("" : "+v"(ta0), "+v"(ta1)); }
      else { tb0 += pp0; tb1 += pp1; asm volatile("" : "+v"(tb0), "+v"(tb1)); }
    }
    bf16x2 t2;
    unsigned u;
    if constexpr (ABL & 128) { u = __builtin_bit_cast(unsigned, pp0) ^ __builtin_bit_cast(unsigned, pp1); }
    else {
      t2[0] = (bf16_t)pp0; t2[1] = (bf16_t)pp1;
      u = __builtin_bit_cast(unsigned, t2);
      asm volatile("" : "+v"(u));
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, pf[qb][kb * 2 + (jj >> 2)]);
    w[jj & 3] = u;
    pf[qb][kb * 2 + (jj >> 2)] = __builtin_bit_cast(bf16x8, w);
  };
  // fragment loads of MFMA n (issued two slots ahead): S^T chains: K (kb, kk) into kfr[n % 3] and, for even n, Q (qb, kk) into qfr[kk & 1];
  // PV: V (step, d) into vfr[n % 3]
  auto frag_load = [&](auto nc, const unsigned char* Ks, const unsigned char* Vp, const unsigned char* Vc) __attribute__((always_inline)) {
    constexpr int n = decltype(nc)::value & 31, grp = n >> 3, jj = n & 7;
    if constexpr (ABL & 8) { } else
    if constexpr (grp == 0 || grp == 2) {
      constexpr int kk = jj >> 1, kb = jj & 1, qb = grp == 0 ? 0 : 1;
      kfr[n % 3] = kfrag(Ks, kb, kk);
      if constexpr (kb == 0) qfr[kk & 1] = qfrag(qb, kk);
    } else {
      constexpr int step = jj >> 1, d = jj & 1;
      vfr[n % 3] = vfrag(grp == 1 ? Vp : Vc, step, d);
    }
  };
  auto mfma_slot = [&](auto nc) __attribute__((always_inline)) {
    constexpr int n = decltype(nc)::value, grp = n >> 3, jj = n & 7;
    if constexpr (grp == 0 || grp == 2) {
      constexpr int kk = jj >> 1, kb = jj & 1, qb = grp == 0 ? 0 : 1;
      s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[n % 3], qfr[kk & 1], kk == 0 ? cinit[qb] : s[qb][kb], 0, 0, 0);
    } else {
      constexpr int step = jj >> 1, d = jj & 1, qb = grp == 1 ? 1 : 0;
      acc_o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[n % 3], pf[qb][step], acc_o[qb][d], 0, 0, 0);
    }
  };
  // the cold path (see attn_fwd_w64_kernel): step `step` of block qb overflowed against the reference m
  auto rescue = [&](int qb, int step) __attribute__((always_inline)) -> float {
    const int tile = step - 1;
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    f32x16 rs[2];
#pragma unroll
    for (int kk = 0; kk < NKK; kk++)
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 a = *(const bf16x8*)(Kb + (long)(tile * KT + kb * 32 + ql) * DH + kk * 16 + half * 8);
        rs[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qfrag(qb, kk), kk == 0 ? z : rs[kb], 0, 0, 0);
      }
    float mx = rs[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, rs[kb][r]);
    mx = other_half_max(mx);
    const float mn = fmaxf(m[qb], mx);
    const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
    m[qb] = mn;
    l[qb] *= alpha;
#pragma unroll
    for (int d = 0; d < NDB; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc_o[qb][d][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; r++) cinit[qb][r] = -mn;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) s[qb][kb][r] = rs[kb][r] - mn;
    const float a0 = expo(qb, 0);
    return a0 + expo(qb, 1);
  };

  // ---- step 0: the ragged tile with its masked row maximum; block A complete, block B up to its first 32 keys ----
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * DPO) : "memory");      // Q, K of step 0
  __builtin_amdgcn_s_barrier();
  if (active) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    const int t0 = ntiles - 1;
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      qk(qb, kslot(0), z);
      float mx = -3e38f;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = t0 * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= N) s[qb][kb][r] = -3e38f;
          mx = fmaxf(mx, s[qb][kb][r]);
        }
      mx = other_half_max(mx);
      m[qb] = mx;
#pragma unroll
      for (int r = 0; r < 16; r++) cinit[qb][r] = -mx;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) s[qb][kb][r] -= mx;
    }
    const float a0 = expo(0, 0);
    l[0] = a0 + expo(0, 1);
    // block B's first 32 keys in the loop's own form: pairs 24 .. 30 finished, pair 31 pending
    [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
      ([&] { pair_exp(AIC<24 + J>{}); if constexpr (J < 7) pair_finish(AIC<24 + J>{}); }(), ...);
    }(std::make_integer_sequence<int, 8>{});
  }
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DPO) : "memory");      // V of step 0
  __builtin_amdgcn_s_barrier();
  if (active) pv(0, vslot(0));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // K, V of step 1
  __builtin_amdgcn_s_barrier();

  // ---- steps 1 .. ntiles - 1 ----
  for (int i = 1; i < ntiles; i++) {
    if constexpr (!(ABL & 16)) { dma_v(i + 1); dma_k(i + 1); }             // V slot of step i - 2 (last read in G2 of step i - 1), K slot of step i - 1
    if (active) {
      const unsigned char* Ks = kslot(i);
      const unsigned char* Vp = vslot(i - 1);
      const unsigned char* Vc = vslot(i);
      frag_load(AIC<0>{}, Ks, Vp, Vc);
      frag_load(AIC<1>{}, Ks, Vp, Vc);
      __builtin_amdgcn_sched_barrier(0);
      auto run = [&](auto n0c, auto n1c) __attribute__((always_inline)) {      // slots [n0, n1)
        constexpr int n0 = decltype(n0c)::value, n1 = decltype(n1c)::value;
        [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
          ([&] {
            constexpr int n = n0 + J;
            mfma_slot(AIC<n>{});
            if constexpr (n != 8 && n != 24) pair_finish(AIC<(n + 31) & 31>{});      // pair n - 1 (slot 0: the last pair of the previous step's G4)
            pair_exp(AIC<n>{});
            if constexpr (n + 2 < 32) frag_load(AIC<n + 2>{}, Ks, Vp, Vc);
            if constexpr (n == 7 || n == 23) pair_finish(AIC<n>{});      // a check follows: nothing pending across it
            __builtin_amdgcn_sched_barrier(0);
          }(), ...);
        }(std::make_integer_sequence<int, n1 - n0>{});
      };
      // slot 0 finishes pair 31 of the previous step; slots 8 and 24 must NOT finish pairs 7 / 23 again
      run(AIC<0>{}, AIC<8>{});
      {
        float lt_b = tb0 + tb1;
        if (i > 1 && __builtin_expect(!__all(lt_b < thresh), 0)) lt_b = rescue(1, i - 1);
        l[1] += lt_b; tb0 = 0.f; tb1 = 0.f;
      }
      run(AIC<8>{}, AIC<24>{});
      {
        float lt_a = ta0 + ta1;
        if (__builtin_expect(!__all(lt_a < thresh), 0)) lt_a = rescue(0, i);
        l[0] += lt_a; ta0 = 0.f; ta1 = 0.f;
      }
      run(AIC<24>{}, AIC<32>{});
    }
    if constexpr (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // V, K of step i + 1 (requested at the top of this step)
    if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
  }
  // ---- drain: block B's last step ----
  if (!active) return;
  {
    pair_finish(AIC<31>{});
    float lt_b = tb0 + tb1 + expo(1, 1);
    if (ntiles > 1 && __builtin_expect(!__all(lt_b < thresh), 0)) lt_b = rescue(1, ntiles - 1);
    l[1] += lt_b;
    pv(1, vslot(ntiles - 1));
  }
  // ---- finalize: O[q][dv] = O^T[dv][q] / l ----
  const int b = bh / H, h = bh % H;
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    const float lsum = l[qb] + __shfl_xor(l[qb], 32, 64);
    const float inv = 1.f / lsum;
    const int q = q0 + qb * 32 + ql;
    if (q < N) {
      bf16_t* op = O + ((long)b * N + q) * ((long)H * DH) + (long)h * DH;
#pragma unroll
      for (int d = 0; d < NDB; d++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; e++) o4[e] = (bf16_t)(acc_o[qb][d][g4 * 4 + e] * inv);
          *(bf16x4*)(op + d * 32 + 8 * g4 + 4 * half) = o4;
        }
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

// rows [m_begin, m_begin + m_count) of the (B * N, 3 * H * Dh) projection only (the skinny tail of a product whose tile part was stored
// through DU_STORE_QKV_ROPE)
template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_split_rows_kernel(const T* __restrict__ rows, T* __restrict__ q, T* __restrict__ k,
                                                                  T* __restrict__ v, const float* __restrict__ sin_t,
                                                                  const float* __restrict__ cos_t, int N, int Npad, int H, int Dh,
                                                                  int prefix, float qscale, long m_begin, long total) {
  constexpr int V = Elem<T>::VEC;
  const int dv = Dh / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int d0 = (int)(i % dv) * V;
    long t = i / dv;
    const int h = (int)(t % H); t /= H;
    const int which = (int)(t % 3); t /= 3;          // t = local row
    const long m = m_begin + t;
    const int b = (int)(m / N), n = (int)(m - (long)b * N);
    const T* src = rows + ((t * 3 + which) * H + h) * Dh;
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

// RoPE + q scale applied IN PLACE to head-major q / k planes that the qkv product stored unrotated (DU_STORE_QKV_HEADS, round 6): a thread
// owns dimensions d0 .. d0 + 7 AND their rotate-half partners d0 + 32 .. of one (which, b, h, token) row, so every element is read once
// and written once -- 67 MB for ViT-L at batch 8 against the 101 MB of qkv_rope_split_kernel (v is already where it belongs).  Rows whose
// global index b * N + n is >= m_limit belong to the product's ragged tail (du_qkv_rope_split_rows writes them rotated): skipped.
// Same arithmetic as qkv_rope_split_kernel, term by term (layers/attention.py:16-27,66-85).
template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_inplace_kernel(T* __restrict__ q, T* __restrict__ k, const float* __restrict__ sin_t,
                                                               const float* __restrict__ cos_t, int B, int N, int Npad, int H, int Dh,
                                                               int prefix, float qscale, long m_limit, long total) {
  constexpr int V = Elem<T>::VEC;
  const int half = Dh / 2, dv = half / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int d0 = (int)(i % dv) * V;
    long t = i / dv;
    const int n = (int)(t % N); t /= N;
    const int h = (int)(t % H); t /= H;
    const int b = (int)(t % B);
    const int which = (int)(t / B);
    if ((long)b * N + n >= m_limit) continue;
    if (which == 1 && n < prefix) continue;                      // k prefix rows: nothing to do
    T* row = (which == 0 ? q : k) + (((long)b * H + h) * Npad + n) * Dh;
    Vec16<T> x1 = as_vec<T>(*(const uint4*)(row + d0)), x2 = as_vec<T>(*(const uint4*)(row + d0 + half));
    float o1[V], o2[V];
#pragma unroll
    for (int j = 0; j < V; j++) { o1[j] = to_f32(x1.v[j]); o2[j] = to_f32(x2.v[j]); }
    if (n >= prefix) {
      const float* sp = sin_t + (long)(n - prefix) * Dh + d0;
      const float* cp = cos_t + (long)(n - prefix) * Dh + d0;
#pragma unroll
      for (int j = 0; j < V; j++) {
        const float a = o1[j], c = o2[j];
        o1[j] = a * cp[j] + -1.f * c * sp[j];
        o2[j] = c * cp[half + j] + 1.f * a * sp[half + j];
      }
    }
    if (which == 0) {
#pragma unroll
      for (int j = 0; j < V; j++) { o1[j] *= qscale; o2[j] *= qscale; }
    }
    Vec16<T> r1, r2;
#pragma unroll
    for (int j = 0; j < V; j++) { r1.v[j] = from_f32<T>(o1[j]); r2.v[j] = from_f32<T>(o2[j]); }
    *(uint4*)(row + d0) = as_u4(r1);
    *(uint4*)(row + d0 + half) = as_u4(r2);
  }
}

extern "C" int du_qkv_rope_inplace(int dtype, void* q, void* k, const float* sin_t, const float* cos_t, int B, int N, int Npad, int H, int Dh,
                                   int prefix, float qscale, int64_t m_limit, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!q || !k || !sin_t || !cos_t || B <= 0 || N <= 0 || Npad < N || H <= 0 || prefix < 0 || prefix > N) return DU_ERR_BAD_ARG;
  if (dtype != DU_BF16 || Dh % 16 || Dh <= 0) return DU_ERR_UNSUPPORTED;
  const long total = 2L * B * H * N * (Dh / 2 / 8);
  long g = (total + 255) / 256; if (g > 65535 * 4) g = 65535 * 4;
  hipLaunchKernelGGL(qkv_rope_inplace_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (bf16_t*)q, (bf16_t*)k, sin_t, cos_t, B, N, Npad, H,
                     Dh, prefix, qscale, (long)m_limit, total);
  return du_check_launch();
}

int g_attn_impl = 0;     // du_set_option key 6: 0 = attn_fwd_w64_kernel (one wave per SIMD, 64 queries per wave), 1 = attn_fwd_kernel (round 2/3)
int g_attn_var = 0;      // du_set_option key 8: experimental variants of the w64 kernel (tools only)
int g_attn_thresh_log2 = 60;   // du_set_option key 7: log2 of the row-sum threshold of the w64 kernel's cold rescale path (<= -1000: every tile takes it)
int g_attn_w = 0;        // du_set_option key 4: ablation bits of attn_fwd_kernel<64, true> (tools/attn_ablate.py); 0 = the product kernel

extern "C" int du_qkv_rope_split_rows(int dtype, const void* qkv_rows, void* q, void* k, void* v, const float* sin_t, const float* cos_t,
                                      int B, int N, int Npad, int H, int Dh, int prefix, float qscale, int64_t m_begin, int m_count,
                                      void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == DU_BF16 ? 8 : 4;
  if (!qkv_rows || !q || !k || !v || !sin_t || !cos_t || B <= 0 || N <= 0 || Npad < N || H <= 0 || Dh <= 0 || (Dh / 2) % vec || prefix < 0 ||
      prefix > N || m_begin < 0 || m_count <= 0 || m_begin + m_count > (int64_t)B * N)
    return DU_ERR_BAD_ARG;
  const long total = (long)m_count * 3 * H * (Dh / vec);
  long g = (total + 255) / 256; if (g > 65535) g = 65535;
  if (dtype == DU_BF16)
    hipLaunchKernelGGL(qkv_rope_split_rows_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)qkv_rows, (bf16_t*)q, (bf16_t*)k,
                       (bf16_t*)v, sin_t, cos_t, N, Npad, H, Dh, prefix, qscale, (long)m_begin, total);
  else if (dtype == DU_F32)
    hipLaunchKernelGGL(qkv_rope_split_rows_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const float*)qkv_rows, (float*)q, (float*)k,
                       (float*)v, sin_t, cos_t, N, Npad, H, Dh, prefix, qscale, (long)m_begin, total);
  else return DU_ERR_BAD_ARG;
  return du_check_launch();
}

// device scratch of the ablation kernel's cycle probe (allocated on first use, never in the product path)
static unsigned long long* g_attn_probe = nullptr;
static unsigned long long* attn_probe_buffer() {
  if (!g_attn_probe && hipMalloc((void**)&g_attn_probe, (8 + 8 * 1024) * sizeof(unsigned long long)) != hipSuccess) g_attn_probe = nullptr;
  return g_attn_probe;
}
// Debug aid (tools/attn_ablate.py): copy the 8 per-segment cycle sums the last probed launch (du_set_option(4, bits | 64)) left behind.
// the census the probed w64 kernel leaves behind: 8 words per workgroup (loop begin, loop end in s_memtime ticks; HW_ID; XCC_ID; kernel entry; after the
// output stores have been acknowledged; 2 spare), first n workgroups
extern "C" int du_debug_attn_census(uint64_t* host, int n) {
  if (!host || n < 1 || n > 1024) return DU_ERR_BAD_ARG;
  if (!g_attn_probe) return DU_ERR_UNSUPPORTED;
  return hipMemcpy(host, g_attn_probe + 8, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? DU_OK : DU_ERR_LAUNCH;
}
extern "C" int du_debug_attn_probe(uint64_t* host8) {
  if (!host8) return DU_ERR_BAD_ARG;
  if (!g_attn_probe) return DU_ERR_UNSUPPORTED;
  return hipMemcpy(host8, g_attn_probe, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? DU_OK : DU_ERR_LAUNCH;
}

// Debug aid: blocks per CU the runtime's occupancy query admits for the attention kernels (which: 0 = w64 d_head 64, 1 = w64 d_head 128,
// 2 = the round-3 kernel d_head 64)
extern "C" int du_debug_attn_occupancy(int which) {
  int n = -1;
  hipError_t e;
  if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_w64_kernel<64, 2>, 256, 0);
  else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<128, false>, 256, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<64, false>, 256, 0);
  return e == hipSuccess ? n : -1;
}

extern "C" int du_attention_fwd(const void* q, const void* k, const void* v, void* out, int B, int H, int N, int Npad, int Dh,
                                void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || N <= 0 || Npad < N) return DU_ERR_BAD_ARG;
  if ((long)N * Dh * 2 > 0x7fffffffL) return DU_ERR_UNSUPPORTED;
  if (g_attn_impl == 0 && (g_attn_w & ~64) == 0 && Dh == 64) {
    // 64 queries per wave, two workgroups per CU (attn_fwd_w64_kernel); threshold of the cold rescale path: 2^60 unless a test turned it
    // down.  d_head 128 stays on attn_fwd_kernel (O alone is 128 registers for 64 queries; at 32 queries per wave the two kernels tie).
    // g_attn_var (du_set_option key 8, tools only): 1000 + bits = timing ablations, 7 = second block's exponentials beside the first block's
    // PV, 9 = the slot-pipelined kernel
    const float thresh = g_attn_thresh_log2 <= -1000 ? 0.f : ldexpf(1.f, g_attn_thresh_log2);
    const dim3 g64((N + 255) / 256, B * H), b64(256);
#define DU_W64_ARGS (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, H, N, Npad, thresh
    if (g_attn_w & 64) hipLaunchKernelGGL((attn_fwd_w64_kernel<64, 2, true>), g64, b64, 0, st, DU_W64_ARGS, attn_probe_buffer());
#define DU_ABL2(bits) else if (g_attn_var == 1000 + (bits)) hipLaunchKernelGGL((attn_fwd_w64_kernel<64, 2, false, 2, (bits)>), g64, b64, 0, st, DU_W64_ARGS, nullptr);
    DU_ABL2(1) DU_ABL2(2) DU_ABL2(4) DU_ABL2(6) DU_ABL2(193) DU_ABL2(8) DU_ABL2(48) DU_ABL2(56) DU_ABL2(199) DU_ABL2(255) DU_ABL2(249)
    else if (g_attn_var == 7) hipLaunchKernelGGL((attn_fwd_w64_kernel<64, 2, false, 2, 0, 3>), g64, b64, 0, st, DU_W64_ARGS, nullptr);
    else if (g_attn_var == 9) hipLaunchKernelGGL((attn_fwd_pipe_kernel<false>), g64, b64, 0, st, DU_W64_ARGS, nullptr);
    else if (g_attn_var == 1) hipLaunchKernelGGL((attn_fwd_w64_kernel<64, 2>), g64, b64, 0, st, DU_W64_ARGS, nullptr);
    else hipLaunchKernelGGL((attn_fwd_w64_kernel<64, 1, false, 4>), dim3((N + 127) / 128, B * H), b64, 0, st, DU_W64_ARGS, nullptr);
    return du_check_launch();
  }
  dim3 grid((N + 127) / 128, B * H), block(256);
  if (Dh == 64)
    if (g_attn_w) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, H, N, Npad, g_attn_w, attn_probe_buffer());
    else hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, H, N, Npad, 0, nullptr);
  else if (Dh == 128)
    hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, H, N, Npad, 0, nullptr);
  else return DU_ERR_UNSUPPORTED;
  return du_check_launch();
}

extern "C" int du_softmax_rows_f32(float* x, int64_t rows, int cols, int64_t ld, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || rows <= 0 || cols <= 0 || ld < cols) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, (long)rows, cols, (long)ld);
  return du_check_launch();
}
