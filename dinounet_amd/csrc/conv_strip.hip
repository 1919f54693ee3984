// Streaming 3x3 convolution (stride 1, pad 1) for the 32- and 64-channel decoder layers at 512^2 / 256^2 (dinounet_training.py:581-592
// via StackedConvBlocks: conv -> InstanceNorm -> LeakyReLU; :614 the channel concat in front of the first conv of a stage) and -- with
// flipped, transposed weights -- their data gradients.
//
// The LDS-tiled kernel of conv_halo.hip is instruction-issue bound on these layers (~620 instructions per wave and
// 128-pixel tile around 18 MFMAs, two LDS fragment reads per MFMA, an fp32 staging round trip and two barriers per tile; 0.34 of HBM
// peak on 32 -> 32).  This kernel takes the workgroup out of the picture: ONE WAVE owns a strip of 32 image columns and walks down
// its rows.
//
//   * product transposed: D[co][pixel] = W[co][k] * X[k][pixel].  The weights are the MFMA A operand and stay in REGISTERS for the
//     whole strip (9 taps x 2 fragments = 72 VGPRs per 32 input channels); with 64 input channels the second 32 channels' weights
//     are read from an LDS image, one conflict-free ds_read_b128 per MFMA.
//   * the B operand of input row t (32 pixels + 2 halo pixels, channel-contiguous as it lies in HBM) is read from LDS once per
//     horizontal shift dx and k-step and used by THREE MFMAs -- it is tap (0, dx) of output row t + 1, tap (1, dx) of row t and
//     tap (2, dx) of row t - 1 -- into three rolling accumulators.
//   * rows arrive by LDS-DMA (buffer_load_dwordx4 ... lds) into a ring of 6 rows, 5 rows ahead: no registers and no ds_write on the
//     way in.  A ring row is one 2 KB plane per 32 input channels (32 pixels x 64 B; a fused concat simply feeds the two planes
//     from its two tensors).  The 16-byte channel vectors of a pixel are XOR-swizzled on the SOURCE side ((pixel >> 2) & 3) so the
//     fragment reads are conflict-free in the ds_read_b128 lane groups.  The two halo columns of six rows come with one extra DMA
//     per six rows and plane into a side buffer (lanes whose shifted pixel falls on them carry a different base / row pitch).
//   * 32 output channels: a wave owns its ring, waits on its own vmcnt only, no barrier anywhere.  64 output channels: waves w and
//     w + 4 own the two 32-channel halves of the SAME strip and share its ring (each issues half of the DMA pieces; one s_barrier
//     per row publishes them) -- two private rings read everything twice, and the second read is no cheaper than the first.
//   * with the pixel on the lane axis the accumulator holds 16 output channels of ONE pixel per lane: bf16 pairs, four
//     v_permlane32_swap, two 16-byte stores per lane and row -- no staging tile.  Channel statistics (sum, sum of squares for the
//     following InstanceNorm, of the bf16-rounded outputs) accumulate per lane over the whole strip and are reduced across lanes once
//     per wave.  All of that runs in the shadow of the row's MFMAs.
//
// Roofline: HBM.  Algorithmic bytes per pixel = (Cin + Cout) * 2; 18 MFMAs (576 matrix-pipe cycles) per 4 KB at 32 -> 32.
#include <stdlib.h>
#include <utility>
#include "common.h"

namespace {

constexpr int S_NR = 6;                     // ring rows
constexpr int S_AHEAD = 5;                  // rows in flight
constexpr int S_PL = 2048;                  // one plane of a ring row: 32 pixels x 64 B

struct StripParams {
  const bf16_t* x; long ldx;                // plane 0: channels [0, 32) of x
  const bf16_t* x2; long ldx2;              // plane 1 (NP = 2): channels [0, 32) of x2 (fused concat) -- or null: channels [32, 64) of x
  const bf16_t* w;                          // [Cout][9 * Cin], (tap, ci) column order
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

// NP: 32-channel planes of the input (Cin = 32 NP).  NCO: 32-channel halves of the output, one wave each per strip (Cout = 32 NCO).
template <int NP, int NCO, bool STATS, bool BIAS>
__global__ __launch_bounds__(256 * NCO, (NP == 1 && NCO == 1) ? 2 : 1) void conv3x3_strip_kernel(StripParams P) {
  constexpr int ROWB = NP * S_PL;                       // ring row
  constexpr int S_EDGE = S_NR * ROWB;                   // two edge buffers behind the ring: [buffer][plane][6 rows x 2 pixels x 64 B (1 KB)]
  constexpr int EDGEB = NP * 1024;
  constexpr int RING = S_EDGE + 2 * EDGEB;              // per strip
  constexpr int WL_OFF = 4 * RING;                      // NP = 2: LDS image of the second plane's weights [NCO][18 fragments][64 lanes][16 B]
  constexpr int BL_OFF = WL_OFF + (NP == 2 ? NCO * 18 * 1024 : 0);      // NP = 2: bias as accumulator image [NCO][64 lanes][16 fp32]
  constexpr int NF = 6 * NP;                            // B fragments of an input row: (dx, plane, kk)
  constexpr int PW = NCO == 1 ? 2 * NP : NP;            // main DMA pieces this wave issues per row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int part = wave >> 2;                           // which 32-channel half of the output (NCO = 2)
  // workgroup -> (image, row segment, group of 4 strips)
  const int sg = P.strips >> 2;
  int bi = blockIdx.x;
  const int sx4 = bi % sg; bi /= sg;
  const int seg = bi % P.nseg;
  const int b = bi / P.nseg;
  const int strip = sx4 * 4 + (wave & 3);
  const int x0 = strip * 32, r0 = seg * P.RS;
  const int co0 = part * 32;
  constexpr int Cin = 32 * NP;
  unsigned char* lds = smem_raw + (wave & 3) * RING;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds);

  // ---- source descriptors (one image per plane) and the per-lane DMA offsets
  u32x4 srd[NP];
  unsigned pbp[NP];                                     // pixel pitch of the plane's source in bytes
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const bool second = p == 1 && P.x2 != nullptr;
    const bf16_t* src = second ? P.x2 : P.x + 32 * p;
    const long ld = second ? P.ldx2 : P.ldx;
    pbp[p] = (unsigned)ld * 2u;
    const unsigned long long base = (unsigned long long)(src + (long)b * P.H * P.W * ld);
    srd[p][0] = (unsigned)base; srd[p][1] = (unsigned)(base >> 32) & 0xffffu;
    srd[p][2] = (unsigned)P.H * (unsigned)P.W * pbp[p] - (p == 1 && !second ? 64u : 0u); srd[p][3] = 0x00020000u;
  }
  // main pieces (plane, i = 0, 1): LDS slot s = 64 i + lane of the plane -> pixel p = s >> 2 of the 34-pixel halo row (p = 0: the left halo
  // column), physical vector s & 3 holds channel vector (s & 3) ^ ((p >> 2) & 3).  NCO = 2: this wave issues the pieces i = part.
  constexpr unsigned OOB = 0x40000000u;                 // host checks H * W * pitch < 2^30
  const bool kill_left = x0 == 0 && lane < 4;           // piece i = 0
  // edge piece of a plane: lane -> (row of the group rg = lane >> 3, pixel 32 + e, vector q); pixels 32, 33 have swizzle 0
  const int rg = lane >> 3, e_ = (lane >> 2) & 1;
  const bool edge_dead = rg >= S_NR || (e_ == 1 && x0 + 32 == P.W);
  auto row_scalar = [&](int trel) -> int {              // pixel index of (input row r0 - 1 + trel, column x0) or -1
    const int yy = r0 - 1 + trel;
    return (yy >= 0 && yy < P.H && trel < P.RS + 2 && !(P.dbg & 2)) ? yy * P.W + x0 : -1;
  };
  auto dma_row = [&](int trel, int slot) {
    const int rp = __builtin_amdgcn_readfirstlane(row_scalar(trel));          // (the loop's induction variables land in VGPRs otherwise)
#pragma unroll
    for (int p = 0; p < NP; p++) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        if (NCO == 2 && i != part) continue;
        const int s = 64 * i + lane, px = s >> 2, v = (s & 3) ^ ((px >> 2) & 3);
        unsigned o = (unsigned)((rp + px - 1) * (int)pbp[p] + v * 16);
        o = (rp < 0 || (i == 0 && kill_left)) ? OOB : o;
        dma16(srd[p], o, __builtin_amdgcn_readfirstlane(lds_base + slot * ROWB + p * S_PL + i * 1024));
      }
    }
  };
  auto dma_edge = [&](int group) {                      // halo columns of input rows 6 group .. 6 group + 5
    const int yg = r0 - 1 + group * S_NR;
    const int yl = yg + rg;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      if (NCO == 2 && part != (NP == 2 ? p : 0)) continue;        // NCO = 2: plane p's edge by part p (one plane: by part 0)
      unsigned o = (unsigned)((yl * P.W + x0 + 31 + e_) * (int)pbp[p] + (lane & 3) * 16);
      o = (edge_dead || yl < 0 || yl >= P.H || (P.dbg & 2)) ? OOB : o;
      dma16(srd[p], o, __builtin_amdgcn_readfirstlane(lds_base + S_EDGE + (group & 1) * EDGEB + p * 1024));
    }
  };

  // ---- fragment addresses: B fragment (dx, plane, kk) = channels plane * 32 + kk * 16 + h * 8 .. of halo pixel p = n + dx
  // dx = 0: always in the ring (immediate offsets for plane and row).  dx = 1, 2: lanes with p >= 32 read the edge buffer (row pitch 128).
  int fb0[2], fbx[2][NP][2], pitch[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) fb0[kk] = n * 64 + (((kk * 2 + h) ^ ((n >> 2) & 3)) << 4);
#pragma unroll
  for (int dx = 1; dx < 3; dx++) {
    const int p = n + dx;
    pitch[dx - 1] = p < 32 ? ROWB : 128;
#pragma unroll
    for (int pl = 0; pl < NP; pl++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int vec = kk * 2 + h;
        fbx[dx - 1][pl][kk] = p < 32 ? pl * S_PL + p * 64 + ((vec ^ ((p >> 2) & 3)) << 4) : S_EDGE + pl * 1024 + (p - 32) * 64 + (vec << 4);
      }
  }
  const int etog1 = (n + 1 >= 32) ? EDGEB : 0, etog2 = (n + 2 >= 32) ? EDGEB : 0;

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

  // (the weights are fetched BEHIND the ring's first rows: their latency and the rows' overlap; the compiler's waits for these loads
  //  also cover the older DMA pieces, which is harmless)
  // ---- weights of plane 0: A fragment of (tap, kk) = w[co0 + n][tap * Cin + kk * 16 + h * 8 ..], in registers
  bf16x8 Wf[9][2];
  {
    const bf16_t* wr = P.w + (long)(co0 + n) * 9 * Cin + h * 8;
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) Wf[tap][kk] = *(const bf16x8*)(wr + tap * Cin + kk * 16);
  }
  // bias as the C input of the first MFMA of every output row: accumulator register r = channel (r & 3) + 8 (r >> 2) + 4 h.  NP = 1: in
  // registers; NP = 2 (registers are short): an LDS image read back just before that MFMA
  f32x16 biasv;
  if constexpr (NP == 1) {
#pragma unroll
    for (int r = 0; r < 16; r++) biasv[r] = BIAS ? P.bias[co0 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
  }
  const unsigned char* wl = smem_raw + WL_OFF + part * 18 * 1024 + lane * 16;          // + (tap * 2 + kk) * 1024
  const float* bl = (const float*)(smem_raw + BL_OFF + part * 4096 + lane * 64);
  if constexpr (NP == 2) {
    // fragment f = tap * 2 + kk of plane 1; the four waves of a part share the work
    const bf16_t* wr = P.w + (long)(co0 + n) * 9 * Cin + 32 + h * 8;
    for (int f = wave & 3; f < 18; f += 4)
      *(uint4*)(smem_raw + WL_OFF + part * 18 * 1024 + f * 1024 + lane * 16) = *(const uint4*)(wr + (f >> 1) * Cin + (f & 1) * 16);
    if constexpr (BIAS) {
      if ((wave & 3) == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) ((float*)(smem_raw + BL_OFF + part * 4096 + lane * 64))[r] = P.bias[co0 + (r & 3) + 8 * (r >> 2) + 4 * h];
      }
    }
  }

  if constexpr (NP == 2) __syncthreads();   // the weight / bias images


  // one input row trel: J = trel % 6 = its ring slot (static); EPI: the step completes an output row (trel >= 2); VMW: the vector-memory
  // operations this wave issued after its DMA pieces of row trel = what may still be in flight when the row is needed.  Steady state:
  // the 2 stores of the step that issued them + 4 further steps x (PW pieces + 2 stores); the first seven steps have fewer (rows 0-4 come
  // from the prologue, steps 0 and 1 store nothing): 4 PW + 2 clamp(trel - 2, 0, 5).  Edge pieces are not counted: at most one operation
  // more than needed is waited for.  (Out-of-range dummy stores to make the count uniform do NOT work: a store the descriptor's range
  // check drops retires at once, ahead of older loads, and the counted wait then passes with the row still in flight -- seen as NaN
  // rows at the head of the segments of every workgroup that started on a busy chip.)
  auto step = [&](auto jc, auto ec, auto tc, int trel, f32x16& accN, f32x16& accM, f32x16& accO) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    constexpr bool EPI = decltype(ec)::value != 0;
    constexpr int TR = decltype(tc)::value;             // min(trel, 7)
    vm_wait<4 * PW + 2 * (TR < 2 ? 0 : (TR > 7 ? 5 : TR - 2))>();
    if constexpr (NCO == 2) asm volatile("s_barrier" ::: "memory");      // the partner's pieces of the row; it has read row trel - 1
    // refill the slot of the previous row (its fragments were consumed by the previous step's MFMAs)
    if constexpr (J == 1) dma_edge(trel / S_NR + 1);
    dma_row(trel + S_AHEAD, (J + S_AHEAD) % S_NR);

    // Schedule of a step (pinned with scheduling barriers: left alone the compiler hoists all reads, spills the statistics, and puts the
    // whole epilogue behind the last MFMA where nothing covers it).  Phase A: the 6 NP MFMAs that complete the oldest accumulator.
    // Phase B: the 12 NP MFMAs of the two younger accumulators, with the epilogue of the completed row (bf16 pairs, statistics,
    // half-wave swaps, stores) spread over their shadows -- a wave issues ~5 VALU operations under one 32-cycle MFMA for free.  The
    // fragments are read twice (once per phase): LDS reads are cheap here, registers are not.
    // fragment i -> (dx, plane, kk)
    auto fr = [&](auto ic) __attribute__((always_inline)) -> bf16x8 {
      constexpr int i = decltype(ic)::value % NF;
      constexpr int dx = i / (2 * NP), pl = (i >> 1) % NP, kk = i & 1;
      if constexpr (dx == 0) return *(const bf16x8*)(lds + fb0[kk] + pl * S_PL + J * ROWB);
      else return *(const bf16x8*)(lds + fbx[dx - 1][pl][kk] + J * pitch[dx - 1]);
    };
    // A fragment of fragment i for the accumulator fed by kernel row dyr: registers (plane 0) or the LDS image (plane 1)
    auto wfrag = [&](auto ic, auto dc) __attribute__((always_inline)) -> bf16x8 {
      constexpr int i = decltype(ic)::value % NF, dyr = decltype(dc)::value;
      constexpr int dx = i / (2 * NP), pl = (i >> 1) % NP, kk = i & 1;
      if constexpr (pl == 0) return Wf[dyr * 3 + dx][kk];
      else return *(const bf16x8*)(wl + ((dyr * 3 + dx) * 2 + kk) * 1024);
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
#define SB __builtin_amdgcn_sched_barrier(0)
    bf16x8 bq[3], aq[2][2];                               // B fragments three deep; LDS weight fragments two slots ahead: [slot parity][N / M or O]
    bq[0] = fr(SIC<0>{}); bq[1] = fr(SIC<1>{}); bq[2] = fr(SIC<2>{});
    SB;
    if constexpr (EPI) {
      // ---- phase A
      [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
        ([&] {
          constexpr int i = I, pl = (i >> 1) % NP;
          if constexpr (pl == 0) accO = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(SIC<i>{}, SIC<2>{}), bq[i % 3], accO, 0, 0, 0);
          else accO = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[i & 1][0], bq[i % 3], accO, 0, 0, 0);
          bq[i % 3] = fr(SIC<i + 3>{});                   // wraps into phase B's first fragments
          // LDS weight fragments TWO slots ahead (a slot is one 32-cycle MFMA here, an LDS read takes ~130): the plane-0 slots of a dx
          // fetch the plane-1 weights of the same dx
          if constexpr (i + 2 < NF && pl == 0 && NP == 2) aq[i & 1][0] = wfrag(SIC<i + 2>{}, SIC<2>{});
          SB;
        }(), ...);
      }(std::make_integer_sequence<int, NF>{});
    }
    // ---- phase B
    f32x16 cin;
    if constexpr (NP == 2 && BIAS) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 t = *(const f32x4*)(bl + 4 * q);
        cin[4 * q] = t[0]; cin[4 * q + 1] = t[1]; cin[4 * q + 2] = t[2]; cin[4 * q + 3] = t[3];
      }
    } else if constexpr (BIAS) {
      cin = biasv;
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) cin[r] = 0.f;
    }
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
      ([&] {
        constexpr int i = I, pl = (i >> 1) % NP;
        if constexpr (pl == 0) {
          accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(SIC<i>{}, SIC<0>{}), bq[i % 3], i == 0 ? cin : accN, 0, 0, 0);
          accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(SIC<i>{}, SIC<1>{}), bq[i % 3], accM, 0, 0, 0);
        } else {
          accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[i & 1][0], bq[i % 3], accN, 0, 0, 0);
          accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[i & 1][1], bq[i % 3], accM, 0, 0, 0);
        }
        if constexpr (i + 3 < NF) bq[i % 3] = fr(SIC<i + 3>{});
        if constexpr (i + 2 < NF && pl == 0 && NP == 2) { aq[i & 1][0] = wfrag(SIC<i + 2>{}, SIC<0>{}); aq[i & 1][1] = wfrag(SIC<i + 2>{}, SIC<1>{}); }
        // the completed row's epilogue, spread over slots 1 .. NF - 2
        if constexpr (NF == 6) {
          if constexpr (i == 1) { epi(SIC<0>{}); epi(SIC<1>{}); }
          if constexpr (i == 2) { epi(SIC<2>{}); epi(SIC<3>{}); epi(SIC<4>{}); }
          if constexpr (i == 3) { epi(SIC<5>{}); epi(SIC<6>{}); epi(SIC<7>{}); }
        } else {
          if constexpr (i >= 1 && i <= 8) epi(SIC<(i >= 1 && i <= 8) ? i - 1 : 0>{});
        }
        // lanes h = 0 hold channels {0-3, 8-11, 16-19, 24-27}, h = 1 {4-7, 12-15, 20-23, 28-31}: swap (pk[i], pk[4 + i]) across the halves
        // -> h = 0: channels 0-15, h = 1: 16-31, in register order 0 1 4 5 2 3 6 7
        if constexpr (EPI && i == NF - 2) { swap_halves(pk[0], pk[4]); swap_halves(pk[1], pk[5]); swap_halves(pk[2], pk[6]); swap_halves(pk[3], pk[7]); }
        if constexpr (i < NF - 1) SB;
      }(), ...);
    }(std::make_integer_sequence<int, NF>{});
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
  auto flip_edges = [&]() {                             // next block of six rows: its edge columns sit in the other edge buffer
#pragma unroll
    for (int pl = 0; pl < NP; pl++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) { fbx[0][pl][kk] ^= etog1; fbx[1][pl][kk] ^= etog2; }
  };

  step(SIC<0>{}, SIC<0>{}, SIC<0>{}, 0, a0, a1, a2);
  step(SIC<1>{}, SIC<0>{}, SIC<1>{}, 1, a2, a0, a1);
  step(SIC<2>{}, SIC<1>{}, SIC<2>{}, 2, a1, a2, a0);
  step(SIC<3>{}, SIC<1>{}, SIC<3>{}, 3, a0, a1, a2);
  step(SIC<4>{}, SIC<1>{}, SIC<4>{}, 4, a2, a0, a1);
  step(SIC<5>{}, SIC<1>{}, SIC<5>{}, 5, a1, a2, a0);
  flip_edges();
  step(SIC<0>{}, SIC<1>{}, SIC<6>{}, 6, a0, a1, a2);
  for (int trel = 7, left = P.RS - 5;; trel += S_NR, left -= S_NR) {       // RS >= 8: at least 3 rows are left
    step(SIC<1>{}, SIC<1>{}, SIC<7>{}, trel, a2, a0, a1); if (left <= 1) break;
    step(SIC<2>{}, SIC<1>{}, SIC<7>{}, trel + 1, a1, a2, a0); if (left <= 2) break;
    step(SIC<3>{}, SIC<1>{}, SIC<7>{}, trel + 2, a0, a1, a2); if (left <= 3) break;
    step(SIC<4>{}, SIC<1>{}, SIC<7>{}, trel + 3, a2, a0, a1); if (left <= 4) break;
    step(SIC<5>{}, SIC<1>{}, SIC<7>{}, trel + 4, a1, a2, a0); if (left <= 5) break;
    flip_edges();
    step(SIC<0>{}, SIC<1>{}, SIC<7>{}, trel + 5, a0, a1, a2); if (left <= 6) break;
  }

  if constexpr (STATS) {
    if (P.stats_part) {
      vm_wait<0>();                                  // every (dummy) DMA into the ring has landed ...
      if constexpr (NCO == 2) __syncthreads();       // ... and the partner is done with it: reuse it
      // two passes (sums, then sums of squares) through this wave's share of its ring: [64 lanes][17]
      float* red = (float*)(smem_raw + (wave & 3) * RING + part * (RING / 2));
      static_assert(64 * 17 * 4 <= RING / 2, "reduction scratch");
      const int r = lane & 15;
      float tot[2];
#pragma unroll
      for (int which = 0; which < 2; which++) {
#pragma unroll
        for (int q = 0; q < 16; q++) red[lane * 17 + q] = which ? s2[q >> 1][q & 1] : s1[q >> 1][q & 1];
        // lanes with (lane & 31) < 16: accumulator register r of half h, summed over that half's 32 lanes
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 32; q++) t += red[(h * 32 + q) * 17 + r];
        tot[which] = t;
      }
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long prt = ((long)b * P.nseg + seg) * P.strips + strip;
      if ((lane & 31) < 16) *(float2*)(P.stats_part + (prt * P.Cout + co) * 2) = make_float2(tot[0], tot[1]);
    }
  }
}

template <int NP, int NCO>
int strip_launch(const StripParams& P, hipStream_t st) {
  const dim3 grid(P.B * P.nseg * (P.strips / 4));
  constexpr int RING = S_NR * NP * S_PL + 2 * NP * 1024;
  const size_t lds = 4 * RING + (NP == 2 ? NCO * (18 * 1024 + 4096) : 0);
  const bool stats = P.stats_part != nullptr, bias = P.bias != nullptr;
  auto go = [&](auto kfn) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)DU_ERR_LAUNCH;
    hipLaunchKernelGGL(kfn, grid, dim3(256 * NCO), lds, st, P);
    return du_check_launch();
  };
  if (stats) return bias ? go(conv3x3_strip_kernel<NP, NCO, true, true>) : go(conv3x3_strip_kernel<NP, NCO, true, false>);
  return bias ? go(conv3x3_strip_kernel<NP, NCO, false, true>) : go(conv3x3_strip_kernel<NP, NCO, false, false>);
}

// segmentation of an image for the strip kernel: rows per segment (0 = shape not served).  Workgroups = B * (W / 128) * (H / RS): two per
// CU where the image allows it (one per CU for the kernels that keep one workgroup resident), segments of at least 8 rows
int strip_rows(int B, int H, int W, int want) {
  if (W % 128 || H % 8) return 0;
  int rs = H;
  while (rs >= 16 && rs % 2 == 0 && (long)B * (W / 128) * (H / rs) < want) rs /= 2;
  return rs;
}
bool strip_off() {
  static const bool off = DU_GETENV("DU_CONV_STRIP") && atoi(DU_GETENV("DU_CONV_STRIP")) == 0;
  return off;
}
// which instantiation serves (C1, Cin, Cout): 0 = none, else NP * 10 + NCO
int strip_kind(int C1, int Cin, int Cout, bool concat) {
  if (strip_off() || !(Cout == 32 || Cout == 64)) return 0;
  if (Cin == 32 && !concat) return 10 + Cout / 32;
  if (Cin == 64 && (!concat || C1 == 32)) return 20 + Cout / 32;
  return 0;
}
int strip_want(int kind) { return kind == 11 ? 512 : 256; }

}  // namespace

// number of partial-statistics rows du_conv3x3_halo writes for this shape (what the caller allocates: parts x Cout x 2 fp32)
extern "C" int du_conv3x3_halo_parts(int C1, int Cin, int Cout, int B, int H, int W) {
  const int kind = strip_kind(C1, Cin, Cout, C1 != Cin);
  const int rs = kind ? strip_rows(B, H, W, strip_want(kind)) : 0;
  if (rs) return B * (H / rs) * (W / 32);
  if (H % 8 || W % 16) return 0;
  return B * (H / 8) * (W / 16);
}

// the strip kernel behind du_conv3x3_halo; DU_ERR_UNSUPPORTED = shape not served (the LDS-tiled kernel takes it)
extern "C" int du_conv3x3_strip(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int C1, int Cin, int Cout, int B, int H, int W,
                                const void* w, const float* bias, void* y, int64_t ldy, float* stats_part, void* stream) {
  if (!x2) C1 = Cin;
  const int kind = strip_kind(C1, Cin, Cout, x2 != nullptr);
  const int rs = kind ? strip_rows(B, H, W, strip_want(kind)) : 0;
  if (!rs) return DU_ERR_UNSUPPORTED;
  const long ldmax = ldx > ldx2 ? ldx : ldx2;
  if ((long)H * W * ldmax * 2 >= (1L << 30) - 65536 || (long)B * H * W * ldy * 2 >= (1L << 31)) return DU_ERR_UNSUPPORTED;
  StripParams P{};
  P.x = (const bf16_t*)x; P.ldx = ldx; P.x2 = (const bf16_t*)x2; P.ldx2 = ldx2; P.w = (const bf16_t*)w; P.bias = bias;
  P.y = (bf16_t*)y; P.ldy = ldy; P.stats_part = stats_part;
  static const int dbg = DU_GETENV("DU_STRIP_DEBUG") ? atoi(DU_GETENV("DU_STRIP_DEBUG")) : 0;
  P.dbg = dbg;
  P.Cout = Cout; P.B = B; P.H = H; P.W = W; P.RS = rs; P.nseg = H / rs; P.strips = W / 32;
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 11: return strip_launch<1, 1>(P, st);
    case 12: return strip_launch<1, 2>(P, st);
    case 21: return strip_launch<2, 1>(P, st);
    case 22: return strip_launch<2, 2>(P, st);
  }
  return DU_ERR_UNSUPPORTED;
}
