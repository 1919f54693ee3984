// Streaming 3x3 convolution (stride 1, pad 1) for the 32-channel decoder layers at 512^2 (dinounet_training.py:581-592 via
// StackedConvBlocks: conv -> InstanceNorm -> LeakyReLU) and -- with flipped, transposed weights -- their data gradients.
//
// The LDS-tiled kernel of conv_halo.hip is instruction-issue bound on these layers (~620 instructions per wave and
// 128-pixel tile around 18 MFMAs, two LDS fragment reads per MFMA, an fp32 staging round trip and two barriers per tile; 0.34 of HBM
// peak).  This kernel removes the workgroup from the picture: ONE WAVE owns a strip of 32 image columns and walks down its rows.
//
//   * product transposed: D[co][pixel] = W[co][k] * X[k][pixel].  The weights are the MFMA A operand and stay in REGISTERS for the
//     whole strip (9 taps x Cin/16 fragments = 72 VGPRs at Cin = 32): no weight reads from LDS at all.
//   * the B operand of input row t (32 pixels + 2 halo pixels, channel-contiguous as it lies in HBM) is read from LDS once per
//     horizontal shift dx and k-step and used by THREE MFMAs -- it is tap (0, dx) of output row t + 1, tap (1, dx) of row t and
//     tap (2, dx) of row t - 1 -- into three rolling accumulators: 6 ds_read_b128 per 18 MFMAs instead of 36.
//   * rows arrive by LDS-DMA (buffer_load_dwordx4 ... lds) into a wave-private ring of 6 rows, 5 rows ahead: no registers and no
//     ds_write on the way in, no barrier anywhere (a wave waits on its own vmcnt only).  The 16-byte channel vectors of a pixel are
//     XOR-swizzled on the SOURCE side ((pixel >> 2) & 3) so the fragment reads are conflict-free in the ds_read_b128 lane groups.
//     The two halo columns of six rows come with one extra DMA per six rows into a side buffer (lanes whose shifted pixel falls on
//     them carry a different base / row pitch).
//   * with the pixel on the lane axis the accumulator holds 16 output channels of ONE pixel per lane: bf16 pairs, four
//     v_permlane32_swap, two 16-byte stores per lane and row -- no staging tile.  Channel statistics (sum, sum of squares for the
//     following InstanceNorm, of the bf16-rounded outputs) accumulate per lane over the whole strip and are reduced across lanes once per wave.
//
// Roofline: HBM.  Algorithmic bytes per pixel = (Cin + Cout) * 2; 18 MFMAs (576 matrix-pipe cycles) per 4 KB at 32 -> 32.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int S_NR = 6;                     // ring rows
constexpr int S_AHEAD = 5;                  // rows in flight
constexpr int S_ROW = 2048;                 // ring row: 32 pixels x 64 B
constexpr int S_EDGE = S_NR * S_ROW;        // two edge buffers of 1 KB behind the ring (6 rows x 2 pixels x 64 B used of each)
constexpr int S_WAVE_LDS = S_EDGE + 2048;   // 14 KB per wave

struct StripParams {
  const bf16_t* x; long ldx;
  const bf16_t* w;                          // [Cout][9 * 32], (tap, ci) column order
  const float* bias;
  bf16_t* y; long ldy;
  float* stats_part;                        // [B * nseg * strips][Cout][2] or null
  int Cout, B, H, W, RS, nseg, strips;
  int dbg;                                  // timing ablations (DU_STRIP_DEBUG): 1 = no output stores, 2 = no input traffic (all DMA out of range)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N_> struct SIC { static constexpr int value = N_; };

__device__ __forceinline__ void dma16(const u32x4& srd, unsigned voff, unsigned lds_addr) {
  // M0 = LDS destination (declared clobbered: the compiler has no use for it between two of these in this kernel, so no save / restore)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(srd), "s"(lds_addr) : "memory", "m0");
}
// a's lanes 32-63 <-> b's lanes 0-31
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int N_> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// NW = 4: four strips per workgroup, 32 output channels.  NW = 8 (Cout = 64): waves w and w + 4 own the SAME strip and the two 32-channel
// halves of its output -- each with its own ring (the second copy of the input comes from L2), running side by side so the two 64-byte
// halves of every 128-byte output pixel reach L2 together (two workgroups writing half lines at different times: 185 us instead of 2 x 58).
template <bool STATS, bool BIAS, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv3x3_strip_kernel(StripParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  // workgroup -> (image, row segment, group of 4 strips)
  const int sg = P.strips >> 2;
  int bi = blockIdx.x;
  const int sx4 = bi % sg; bi /= sg;
  const int seg = bi % P.nseg;
  const int b = bi / P.nseg;
  const int strip = sx4 * 4 + (wave & 3);
  const int x0 = strip * 32, r0 = seg * P.RS;
  const int co0 = (wave >> 2) * 32;
  const unsigned pb = (unsigned)P.ldx * 2u;                       // pixel pitch of the source in bytes
  unsigned char* lds = smem_raw + wave * S_WAVE_LDS;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds);

  // ---- weights: A fragment of (tap, kk) = w[co0 + n][tap * 32 + kk * 16 + h * 8 ..]
  bf16x8 Wf[9][2];
  {
    const bf16_t* wr = P.w + (long)(co0 + n) * 288 + h * 8;
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) Wf[tap][kk] = *(const bf16x8*)(wr + tap * 32 + kk * 16);
  }
  // bias as the C input of the first MFMA of every output row: accumulator register r = channel (r & 3) + 8 (r >> 2) + 4 h
  f32x16 biasv;
#pragma unroll
  for (int r = 0; r < 16; r++) biasv[r] = BIAS ? P.bias[co0 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;

  // ---- source descriptor (one image) and the per-lane DMA offsets
  u32x4 srd;
  {
    const unsigned long long base = (unsigned long long)(P.x + (long)b * P.H * P.W * P.ldx);
    srd[0] = (unsigned)base; srd[1] = (unsigned)(base >> 32) & 0xffffu;
    srd[2] = (unsigned)P.H * (unsigned)P.W * pb; srd[3] = 0x00020000u;
  }
  // main pieces i = 0, 1: slot s = 64 i + lane -> pixel p = s >> 2 of the 34-pixel halo row (p = 0: the left halo column),
  // physical vector s & 3 holds channel vector (s & 3) ^ ((p >> 2) & 3)
  int dvo[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int s = 64 * i + lane, p = s >> 2, v = (s & 3) ^ ((p >> 2) & 3);
    dvo[i] = (p - 1) * (int)pb + v * 16;
  }
  const bool kill_left = x0 == 0 && lane < 4;
  // edge piece: lane -> (row of the group rg = lane >> 3, pixel 32 + e, vector q); pixels 32, 33 have swizzle 0
  const int rg = lane >> 3, e_ = (lane >> 2) & 1;
  const int evo = (rg * P.W + 31 + e_) * (int)pb + (lane & 3) * 16;
  const bool edge_dead = rg >= S_NR || (e_ == 1 && x0 + 32 == P.W);
  constexpr unsigned OOB = 0x40000000u;                            // host checks H * W * pitch < 2^30
  auto row_scalar = [&](int trel) -> unsigned {                    // byte offset of (input row r0 - 1 + trel, column x0) or out of range
    const int yy = r0 - 1 + trel;
    return (yy >= 0 && yy < P.H && trel < P.RS + 2 && !(P.dbg & 2)) ? (unsigned)(yy * P.W + x0) * pb : OOB;
  };
  auto dma_row = [&](int trel, int slot) {
    const unsigned rs = __builtin_amdgcn_readfirstlane(row_scalar(trel));     // (the loop's induction variables land in VGPRs otherwise)
    unsigned o0 = (unsigned)dvo[0] + rs, o1 = (unsigned)dvo[1] + rs;
    o0 = kill_left ? OOB : o0;
    dma16(srd, o0, lds_base + slot * S_ROW);
    dma16(srd, o1, lds_base + slot * S_ROW + 1024);
  };
  auto dma_edge = [&](int group) {                                 // halo columns of input rows 6 group .. 6 group + 5
    const int yg = r0 - 1 + group * S_NR;
    const int yl = yg + rg;
    unsigned o = (unsigned)evo + (unsigned)(yg * P.W + x0) * pb;
    o = (edge_dead || yl < 0 || yl >= P.H || (P.dbg & 2)) ? OOB : o;
    dma16(srd, o, __builtin_amdgcn_readfirstlane(lds_base + S_EDGE + (group & 1) * 1024));
  };

  // ---- fragment addresses: B fragment of (dx, kk) = channels kk * 16 + h * 8 .. of halo pixel p = n + dx
  // dx = 0: always in the ring (pitch S_ROW: immediate offsets).  dx = 1, 2: lanes with p >= 32 read the edge buffer (pitch 128).
  int fb[3][2], pitch[3];
#pragma unroll
  for (int dx = 0; dx < 3; dx++) {
    const int p = n + dx;
    pitch[dx] = p < 32 ? S_ROW : 128;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int vec = kk * 2 + h;
      fb[dx][kk] = p < 32 ? p * 64 + ((vec ^ ((p >> 2) & 3)) << 4) : S_EDGE + (p - 32) * 64 + (vec << 4);
    }
  }
  const int etog1 = (n + 1 >= 32) ? 1024 : 0, etog2 = (n + 2 >= 32) ? 1024 : 0;

  // ---- output addressing
  const auto yrs = __builtin_amdgcn_make_buffer_rsrc((void*)P.y, 0, (int)((unsigned)P.B * (unsigned)P.H * (unsigned)P.W * (unsigned)P.ldy * 2u), 0x00020000);
  const unsigned yvo = (unsigned)n * (unsigned)P.ldy * 2u + (unsigned)co0 * 2u + (unsigned)h * 32u;

  f32x2 s1[8], s2[8];                       // per-lane (sum, sum of squares) of accumulator registers (2c, 2c + 1)
#pragma unroll
  for (int c = 0; c < 8; c++) { s1[c] = f32x2{0.f, 0.f}; s2[c] = f32x2{0.f, 0.f}; }

  f32x16 a0, a1, a2;
#pragma unroll
  for (int r = 0; r < 16; r++) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }

  // ---- prologue: edge group 0 and rows 0 .. AHEAD - 1
  dma_edge(0);
#pragma unroll
  for (int t = 0; t < S_AHEAD; t++) dma_row(t, t);

  // one input row trel: J = trel % 6 = its ring slot (static); EPI: the step completes an output row (trel >= 2); VMW: the vector-memory
  // operations issued after the two DMA pieces of row trel = what may still be in flight when the row is needed.  Steady state: the 2
  // stores of the step that issued it + 4 further steps x (2 pieces + 2 stores) = 18; the first seven steps have fewer (rows 0-4 come
  // from the prologue, steps 0 and 1 store nothing): 8 8 8 10 12 14 16.  Edge pieces are not counted: at most one operation more than
  // needed is waited for.  (Out-of-range dummy stores to make the count uniform do NOT work: a store the descriptor's range check drops
  // retires at once, ahead of older loads, and the counted wait then passes with the row still in flight -- seen as NaN rows at the
  // head of the segments of every workgroup that started on a busy chip.)
  auto step = [&](auto jc, auto ec, auto wc, int trel, f32x16& accN, f32x16& accM, f32x16& accO) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    constexpr bool EPI = decltype(ec)::value != 0;
    vm_wait<decltype(wc)::value>();
    // refill the slot of the previous row (its fragments were consumed by the previous step's MFMAs)
    if constexpr (J == 1) dma_edge(trel / S_NR + 1);
    dma_row(trel + S_AHEAD, (J + S_AHEAD) % S_NR);

    // Schedule of a step (pinned with scheduling barriers: left alone the compiler hoists all reads, spills the statistics, and puts the
    // whole epilogue behind the last MFMA where nothing covers it).  Phase A: the six MFMAs that complete the oldest accumulator.
    // Phase B: the twelve MFMAs of the two younger accumulators, with the epilogue of the completed row (bf16 pairs, statistics,
    // half-wave swaps, stores) spread over their shadows -- a wave issues ~5 VALU operations under one 32-cycle MFMA for free.  The
    // fragments are read twice (once per phase): LDS reads are cheap here (12 per 18 MFMAs), registers are not.
    auto fr = [&](auto ic) __attribute__((always_inline)) -> bf16x8 {
      constexpr int dx = decltype(ic)::value >> 1, kk = decltype(ic)::value & 1;
      if constexpr (dx == 0) return *(const bf16x8*)(lds + fb[0][kk] + J * S_ROW);
      else return *(const bf16x8*)(lds + fb[dx][kk] + J * pitch[dx]);
    };
    auto mO = [&](auto ic, const bf16x8& f) __attribute__((always_inline)) {
      constexpr int dx = decltype(ic)::value >> 1, kk = decltype(ic)::value & 1;
      accO = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[6 + dx][kk], f, accO, 0, 0, 0);
    };
    auto mNM = [&](auto ic, const bf16x8& f) __attribute__((always_inline)) {
      constexpr int dx = decltype(ic)::value >> 1, kk = decltype(ic)::value & 1;
      if constexpr (dx == 0 && kk == 0) {
        if constexpr (BIAS) accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[0][0], f, biasv, 0, 0, 0);
        else {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; r++) z[r] = 0.f;
          accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[0][0], f, z, 0, 0, 0);
        }
      } else accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[dx][kk], f, accN, 0, 0, 0);
      accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[3 + dx][kk], f, accM, 0, 0, 0);
    };
    unsigned pk[8];
    // epilogue chunk c (0..7): registers 2c, 2c + 1 of the completed accumulator -> one bf16 pair + their statistics
    auto epi = [&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      if constexpr (EPI) {
        const bf16x2 t = {(bf16_t)accO[2 * c], (bf16_t)accO[2 * c + 1]};
        pk[c] = __builtin_bit_cast(unsigned, t);
        if constexpr (STATS) {
          // of the ROUNDED values: what the following norm layer reads (and what autocast's InstanceNorm of a bf16 tensor sees)
          const f32x2 v = {__builtin_bit_cast(float, pk[c] << 16), __builtin_bit_cast(float, pk[c] & 0xffff0000u)};
          s1[c] += v; s2[c] = __builtin_elementwise_fma(v, v, s2[c]);
          asm volatile("" : "+v"(s1[c]), "+v"(s2[c]));       // pinned here: their only use is after the loop, and the compiler sinks them behind the step
        }
      }
    };
    // lanes h = 0 hold channels {0-3, 8-11, 16-19, 24-27}, h = 1 {4-7, 12-15, 20-23, 28-31}: swap (pk[i], pk[4 + i]) across the halves ->
    // h = 0: channels 0-15, h = 1: 16-31, in register order 0 1 4 5 2 3 6 7
#define SB __builtin_amdgcn_sched_barrier(0)
    bf16x8 b0 = fr(SIC<0>{}), b1 = fr(SIC<1>{}), b2 = fr(SIC<2>{});
    SB;
    if constexpr (EPI) {
      mO(SIC<0>{}, b0); b0 = fr(SIC<3>{}); SB;
      mO(SIC<1>{}, b1); b1 = fr(SIC<4>{}); SB;
      mO(SIC<2>{}, b2); b2 = fr(SIC<5>{}); SB;
      mO(SIC<3>{}, b0); b0 = fr(SIC<0>{}); SB;
      mO(SIC<4>{}, b1); b1 = fr(SIC<1>{}); SB;
      mO(SIC<5>{}, b2); b2 = fr(SIC<2>{}); SB;
    }
    mNM(SIC<0>{}, b0); b0 = fr(SIC<3>{}); SB;
    mNM(SIC<1>{}, b1); b1 = fr(SIC<4>{}); epi(SIC<0>{}); epi(SIC<1>{}); SB;
    mNM(SIC<2>{}, b2); b2 = fr(SIC<5>{}); epi(SIC<2>{}); epi(SIC<3>{}); epi(SIC<4>{}); SB;
    mNM(SIC<3>{}, b0); epi(SIC<5>{}); epi(SIC<6>{}); epi(SIC<7>{}); SB;
    mNM(SIC<4>{}, b1);
    if constexpr (EPI) { swap_halves(pk[0], pk[4]); swap_halves(pk[1], pk[5]); swap_halves(pk[2], pk[6]); swap_halves(pk[3], pk[7]); }
    SB;
    mNM(SIC<5>{}, b2);
    // output row r0 + trel - 2 is complete
    if constexpr (EPI) {
      const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((long)b * P.H + (r0 + trel - 2)) * P.W + x0) * (unsigned)P.ldy * 2u);
      const u32x4 lo = {pk[0], pk[1], pk[4], pk[5]}, hi = {pk[2], pk[3], pk[6], pk[7]};
      if (!(P.dbg & 1)) {
        __builtin_amdgcn_raw_buffer_store_b128(lo, yrs, yvo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, yrs, yvo + 16u, so, 0);
      }
    }
    SB;
#undef SB
  };

  step(SIC<0>{}, SIC<0>{}, SIC<8>{}, 0, a0, a1, a2);
  step(SIC<1>{}, SIC<0>{}, SIC<8>{}, 1, a2, a0, a1);
  step(SIC<2>{}, SIC<1>{}, SIC<8>{}, 2, a1, a2, a0);
  step(SIC<3>{}, SIC<1>{}, SIC<10>{}, 3, a0, a1, a2);
  step(SIC<4>{}, SIC<1>{}, SIC<12>{}, 4, a2, a0, a1);
  step(SIC<5>{}, SIC<1>{}, SIC<14>{}, 5, a1, a2, a0);
  fb[1][0] ^= etog1; fb[1][1] ^= etog1; fb[2][0] ^= etog2; fb[2][1] ^= etog2;      // rows 6-11: edge columns in the other buffer
  step(SIC<0>{}, SIC<1>{}, SIC<16>{}, 6, a0, a1, a2);
  for (int trel = 7, left = P.RS - 5;; trel += S_NR, left -= S_NR) {       // RS >= 8: at least 3 rows are left
    step(SIC<1>{}, SIC<1>{}, SIC<18>{}, trel, a2, a0, a1); if (left <= 1) break;
    step(SIC<2>{}, SIC<1>{}, SIC<18>{}, trel + 1, a1, a2, a0); if (left <= 2) break;
    step(SIC<3>{}, SIC<1>{}, SIC<18>{}, trel + 2, a0, a1, a2); if (left <= 3) break;
    step(SIC<4>{}, SIC<1>{}, SIC<18>{}, trel + 3, a2, a0, a1); if (left <= 4) break;
    step(SIC<5>{}, SIC<1>{}, SIC<18>{}, trel + 4, a1, a2, a0); if (left <= 5) break;
    fb[1][0] ^= etog1; fb[1][1] ^= etog1; fb[2][0] ^= etog2; fb[2][1] ^= etog2;
    step(SIC<0>{}, SIC<1>{}, SIC<18>{}, trel + 5, a0, a1, a2); if (left <= 6) break;
  }

  if constexpr (STATS) {
    if (P.stats_part) {
      vm_wait<0>();                                  // every (dummy) DMA into this wave's LDS has landed: reuse it
      float* red = (float*)lds;                      // [64 lanes][33]
#pragma unroll
      for (int r = 0; r < 16; r++) { red[lane * 33 + r] = s1[r >> 1][r & 1]; red[lane * 33 + 16 + r] = s2[r >> 1][r & 1]; }
      // lane j: value vi = j & 31 of half j >> 5, summed over that half's 32 lanes
      const int vi = lane & 31;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 32; q++) t += red[(h * 32 + q) * 33 + vi];
      const int r = vi & 15, which = vi >> 4;
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long part = ((long)b * P.nseg + seg) * P.strips + strip;
      P.stats_part[(part * P.Cout + co) * 2 + which] = t;
    }
  }
}

}  // namespace

template <int NW>
static int strip_launch(const StripParams& P, hipStream_t st) {
  const dim3 grid(P.B * P.nseg * (P.strips / 4));
  const size_t lds = NW * S_WAVE_LDS;
  const bool stats = P.stats_part != nullptr, bias = P.bias != nullptr;
  auto go = [&](auto kfn) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DU_ERR_LAUNCH;
    hipLaunchKernelGGL(kfn, grid, dim3(NW * 64), lds, st, P);
    return du_check_launch();
  };
  if (stats) return bias ? go(conv3x3_strip_kernel<true, true, NW>) : go(conv3x3_strip_kernel<true, false, NW>);
  return bias ? go(conv3x3_strip_kernel<false, true, NW>) : go(conv3x3_strip_kernel<false, false, NW>);
}

// segmentation of an image for the strip kernel: rows per segment (0 = shape not served)
static int strip_rows(int B, int H, int W) {
  if (W % 128 || H % 8) return 0;
  // about two workgroups per CU: B * (W / 128) * (H / RS) >= 512 where the image allows it, segments of at least 8 rows
  int rs = H;
  while (rs >= 16 && rs % 2 == 0 && (long)B * (W / 128) * (H / rs) < 512) rs /= 2;
  return rs;
}

// number of partial-statistics rows du_conv3x3_halo writes for this shape (what the caller allocates: parts x Cout x 2 fp32)
extern "C" int du_conv3x3_halo_parts(int C1, int Cin, int Cout, int B, int H, int W) {
  static const bool off = getenv("DU_CONV_STRIP") && atoi(getenv("DU_CONV_STRIP")) == 0;
  const int rs = strip_rows(B, H, W);
  if (!off && rs && C1 == Cin && Cin == 32 && (Cout == 32 || Cout == 64)) return B * (H / rs) * (W / 32);
  if (H % 8 || W % 16) return 0;
  return B * (H / 8) * (W / 16);
}

// the strip kernel behind du_conv3x3_halo; DU_ERR_UNSUPPORTED = shape not served (the LDS-tiled kernel takes it)
extern "C" int du_conv3x3_strip(const void* x, int64_t ldx, int Cin, int Cout, int B, int H, int W, const void* w, const float* bias,
                                void* y, int64_t ldy, float* stats_part, void* stream) {
  static const bool off = getenv("DU_CONV_STRIP") && atoi(getenv("DU_CONV_STRIP")) == 0;
  if (off || Cin != 32 || !(Cout == 32 || Cout == 64)) return DU_ERR_UNSUPPORTED;
  const int rs = strip_rows(B, H, W);
  if (!rs) return DU_ERR_UNSUPPORTED;
  if ((long)H * W * ldx * 2 >= (1L << 30) - 65536 || (long)B * H * W * ldy * 2 >= (1L << 31)) return DU_ERR_UNSUPPORTED;
  StripParams P{};
  P.x = (const bf16_t*)x; P.ldx = ldx; P.w = (const bf16_t*)w; P.bias = bias; P.y = (bf16_t*)y; P.ldy = ldy; P.stats_part = stats_part;
  static const int dbg = getenv("DU_STRIP_DEBUG") ? atoi(getenv("DU_STRIP_DEBUG")) : 0;
  P.dbg = dbg;
  P.Cout = Cout; P.B = B; P.H = H; P.W = W; P.RS = rs; P.nseg = H / rs; P.strips = W / 32;
  hipStream_t st = (hipStream_t)stream;
  return Cout == 32 ? strip_launch<4>(P, st) : strip_launch<8>(P, st);
}
