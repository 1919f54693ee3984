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
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0]: lower half's value in every lane, r[1]: upper half's
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
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
  if (!g_attn_probe && hipMalloc((void**)&g_attn_probe, 8 * sizeof(unsigned long long)) != hipSuccess) g_attn_probe = nullptr;
  return g_attn_probe;
}
// Debug aid (tools/attn_ablate.py): copy the 8 per-segment cycle sums the last probed launch (du_set_option(4, bits | 64)) left behind.
extern "C" int du_debug_attn_probe(uint64_t* host8) {
  if (!host8) return DU_ERR_BAD_ARG;
  if (!g_attn_probe) return DU_ERR_UNSUPPORTED;
  return hipMemcpy(host8, g_attn_probe, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? DU_OK : DU_ERR_LAUNCH;
}

extern "C" int du_attention_fwd(const void* q, const void* k, const void* v, void* out, int B, int H, int N, int Npad, int Dh,
                                void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || N <= 0 || Npad < N) return DU_ERR_BAD_ARG;
  if ((long)N * Dh * 2 > 0x7fffffffL) return DU_ERR_UNSUPPORTED;
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
