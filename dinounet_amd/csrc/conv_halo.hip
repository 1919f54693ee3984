// LDS-tiled direct 3x3 convolution (stride 1, pad 1) on NHWC bf16 for the U-Net decoder / FAPM / SPM-stem convolutions
// (dinounet_training.py:581-592 via StackedConvBlocks, dinov3_adapter.py:243-249) and -- with spatially flipped, transposed
// weights -- their data gradients.
//
// Why not the implicit-GEMM kernel (gemm_bf16.hip, IM2COL_ROW): it re-gathers every input pixel once per filter tap (9x the L1/TA
// traffic) and, with Cout = 32/64, has only 4-8 MFMAs of work per barrier.  Here a workgroup stages the (8+2) x (16+2) input halo
// of an 8 x 16 output tile ONCE per channel chunk and the 9 taps read their MFMA A fragments from it at shifted pixel addresses;
// the chunk's weights for all 9 taps sit in LDS too (for Cin <= 64 they are loaded once per workgroup: workgroups are persistent
// and walk tiles), so one staging step feeds 9 * CK/16 * Cout/32 MFMAs per wave.  The next halo (and weight chunk) is prefetched
// into registers while the current one is multiplied.  The fp32 tile goes through LDS so every output pixel is written as one
// contiguous Cout-channel row; the per-channel sums / sums of squares of the tile (InstanceNorm / BatchNorm statistics of the
// following norm layer) fall out of the same staged tile and are written as partials (no second pass over the output).
//
// Roofline: HBM for Cout <= 64 at 256^2 / 512^2 (algorithmic bytes = (Cin + Cout) * 2 per pixel), MFMA for the 128-channel layers.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16;              // output tile (128 pixels = 4 waves x 32)
constexpr int HW_ = (TH + 2) * (TW + 2);    // 180 halo pixels
constexpr int HXW = TW + 2;

struct HaloParams {
  const bf16_t* x; long ldx;                // channels [0, C1)
  const bf16_t* x2; long ldx2;              // channels [C1, Cin) (fused concat) or null
  int C1, Cin, Cout;
  int B, H, W;
  const bf16_t* w;                          // [Cout][9 * Cin], (tap, ci) column order
  const float* bias;                        // [Cout] or null
  bf16_t* y; long ldy;
  float* stats_part;                        // [ntiles][Cout][2] partial (sum, sum of squares) or null
  int tilesX, tilesY, ntiles;
};

template <int CK, int TN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(HaloParams P) {
  constexpr int COUT = TN * 32;
  constexpr int LD = CK + 8;                            // LDS pixel / weight-row pitch (bf16): 16-byte padded
  constexpr int CV = CK / 8;                            // 16-byte vectors per pixel per chunk
  constexpr int HV = (HW_ * CV + 255) / 256;            // halo vectors per thread
  constexpr int WV = (9 * COUT * CV + 255) / 256;       // weight vectors per thread
  // halo image: pixel pitch LD (144 / 80 bytes: the 16 pixels of a row land on 16 different 16-byte bank groups), ROW pitch padded to a
  // multiple of 256 bytes -- the A-fragment ds_read_b128 serves lanes {0-3, 12-15} of one image row together with lanes {20-27} of the
  // NEXT row in one LDS cycle; with the natural pitch (18 pixels) the second row started on another bank phase and collided with the
  // first (SQ_LDS_BANK_CONFLICT 22-35 % of the LDS cycles, profiles/r02_pmc_sq_cycles_eager.txt)
  constexpr int HROW = ((HXW * LD * 2 + 255) / 256 * 256) / 2;
  constexpr int W_EL = 9 * COUT * LD, H_EL = (TH + 2) * HROW;
  constexpr int STG_LD = COUT + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ws = (bf16_t*)smem_raw;                       // [9][COUT][LD]
  bf16_t* Hs = Ws + W_EL;                               // [180][LD]
  const int nch = P.Cin / CK;
  // fp32 staging tile [128][COUT + 4]: behind the halo when the weights stay resident (one chunk), else over the weights
  float* stg = (nch == 1) ? (float*)(smem_raw + (size_t)(W_EL + H_EL) * 2) : (float*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const int py = 2 * wave + ((lane & 31) >> 4), px = lane & 15;   // this lane's output pixel inside the tile (A-fragment row)

  // TWO halo chunks in flight (round 3): with one workgroup-tile of prefetch per wave the HBM-bound layers were latency x concurrency
  // bound (2 workgroups / CU x 11-23 KB in flight = 6-12 MB chip-wide at ~2.5 us loaded latency ~ 2.4-3 TB/s, what was measured).
  // The two register sets are used by two STATIC copies of the chunk step (step(hregA); step(hregB); ...): with a run-time set index the
  // compiler's wait-count pass cannot tell which set the oldest loads went to and drains the whole queue (s_waitcnt vmcnt(0)) before
  // every install.  The loads are raw buffer loads of ONE image (descriptor rebuilt per chunk from wave-uniform values): a pixel outside
  // the image -- or a chunk past the end of this workgroup's work -- gets an out-of-range offset and reads zeros, no branch per load.
  uint4 hregA[HV], hregB[HV], wreg[WV];
  // Addressing diet (round 3: the kernel is instruction-issue bound -- ~620 instructions per wave and tile around 18-72 MFMAs; a SIMD
  // issues about one instruction per 4-5 cycles whatever its kind, tools/attn_ablate.py).  Per halo slot of this thread only constants:
  // rel = (dy - 1) * W + (dx - 1) pixels from the tile origin, the byte offset of its channel vector, and three bit masks (slot does not
  // exist / sits in the halo's left column / right column).  Rows above and below the image fall outside the per-image buffer descriptor
  // by themselves (negative or >= H * W offsets); the left / right columns wrap into the neighbouring row, so they are killed through
  // the masks when the tile touches that border: bit 31 of the offset = out of range.
  int rel_pix[HV];
  const unsigned cv_bytes = (unsigned)(tid % CV) * 16u;   // 256 % CV == 0: the same channel vector in every slot
  unsigned masks = 0;                                   // bits 0-7 dead slots, 8-15 left column, 16-23 right column
#pragma unroll
  for (int i = 0; i < HV; i++) {
    const int v = tid + i * 256;
    const int pix = v / CV;
    rel_pix[i] = (pix / HXW - 1) * P.W + (pix % HXW - 1);
    if (v >= HW_ * CV) masks |= 1u << i;
    else if (pix % HXW == 0) masks |= 0x100u << i;
    else if (pix % HXW == HXW - 1) masks |= 0x10000u << i;
  }
  static_assert(HV <= 8 && 256 % CV == 0, "slot masks are 8 bits wide");
  struct TileAt { int tile, tx, ty, b; };               // a tile of this workgroup's sequence and its (x, y, image) coordinates
  int step_x, step_y, step_b;                           // gridDim.x tiles further on, in those coordinates
  {
    const int g = (int)gridDim.x, per = P.tilesX * P.tilesY;
    step_b = g / per; step_y = (g % per) / P.tilesX; step_x = g % P.tilesX;
  }
  auto tile_at = [&](int tile) {
    TileAt t; t.tile = tile; t.tx = tile % P.tilesX;
    const int q = tile / P.tilesX;
    t.ty = q % P.tilesY; t.b = q / P.tilesY;
    return t;
  };
  auto tile_next = [&](TileAt& t) {                     // + gridDim.x tiles without a division
    t.tile += (int)gridDim.x;
    t.tx += step_x; if (t.tx >= P.tilesX) { t.tx -= P.tilesX; t.ty++; }
    t.ty += step_y; if (t.ty >= P.tilesY) { t.ty -= P.tilesY; t.b++; }
    t.b += step_b;
  };
  auto halo_load = [&](uint4 (&hreg)[HV], const TileAt& t, int ch) {
    const bool live = t.tile < P.ntiles;
    const int tx0 = t.tx * TW, ty0 = t.ty * TH;
    const int c0 = ch * CK;
    const bf16_t* src = P.x; long ld = P.ldx; int cofs = c0;
    if (c0 >= P.C1) { src = P.x2; ld = P.ldx2; cofs = c0 - P.C1; }
    const unsigned pixb = (unsigned)ld * 2u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)(live ? t.b : 0) * P.H * P.W * ld), 0,
                                                      (int)((unsigned)P.H * (unsigned)P.W * pixb), 0x00020000);
    const unsigned base = (unsigned)(ty0 * P.W + tx0) * pixb + (unsigned)cofs * 2u;      // wave-uniform
    const unsigned sel = 0xffu | (tx0 == 0 ? 0xff00u : 0u) | (tx0 + TW == P.W ? 0xff0000u : 0u);       // wave-uniform
    unsigned kill = masks & sel;
    kill = (kill | (kill >> 8) | (kill >> 16) | (live ? 0u : 0xffu)) & 0xffu;
#pragma unroll
    for (int i = 0; i < HV; i++) {
      unsigned off = (unsigned)__mul24(rel_pix[i], (int)pixb) + cv_bytes + base;
      off |= (kill << (31 - i)) & 0x80000000u;
      hreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
  auto halo_store = [&](const uint4 (&hreg)[HV]) {
#pragma unroll
    for (int i = 0; i < HV; i++) {
      const int v = tid + i * 256;
      if (v < HW_ * CV) {
        const int pix = v / CV;
        *(uint4*)(Hs + (pix / HXW) * HROW + (pix % HXW) * LD + (v % CV) * 8) = hreg[i];
      }
    }
  };
  auto w_load = [&](int ch) {
#pragma unroll
    for (int i = 0; i < WV; i++) {
      const int v = tid + i * 256;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (v < 9 * COUT * CV) {
        const int cv = v % CV, row = v / CV;            // row = tap * COUT + co
        const int tap = row / COUT, co = row % COUT;
        r = *(const uint4*)(P.w + (long)co * 9 * P.Cin + (long)tap * P.Cin + ch * CK + cv * 8);
      }
      wreg[i] = r;
    }
  };
  auto w_store = [&]() {
#pragma unroll
    for (int i = 0; i < WV; i++) {
      const int v = tid + i * 256;
      if (v < 9 * COUT * CV) *(uint4*)(Ws + (v / CV) * LD + (v % CV) * 8) = wreg[i];
    }
  };

  if ((int)blockIdx.x >= P.ntiles) return;
  // the bias, read from memory ONCE (a load in the epilogue queues behind the halo prefetches: vector memory returns in order) and kept in
  // LDS; it starts the accumulators, so the epilogue adds nothing
  float* bias_s;
  {
    constexpr size_t MAIN_B = (size_t)(W_EL + H_EL) * 2, STG_B = (size_t)128 * STG_LD * 4 + (size_t)4 * COUT * 2 * 4;
    bias_s = (float*)(smem_raw + (nch == 1 ? MAIN_B + STG_B : (MAIN_B > STG_B ? MAIN_B : STG_B)));
    if (tid < COUT) bias_s[tid] = P.bias ? P.bias[tid] : 0.f;          // visible after the first install barrier
  }
  // chunk sequence of this (persistent) workgroup: (tile, 0..nch-1), (tile + grid, 0..nch-1), ...; `cur` = the tile being multiplied,
  // (`pre`, pre_ch) = the chunk the next halo request is for (two chunks ahead)
  TileAt cur = tile_at((int)blockIdx.x), pre = cur;
  int pre_ch = 0;
  auto pre_advance = [&]() { if (++pre_ch == nch) { pre_ch = 0; tile_next(pre); } };
  w_load(0);
  halo_load(hregA, pre, pre_ch); pre_advance();
  halo_load(hregB, pre, pre_ch); pre_advance();
  bool w_pending = true;
  int ch = 0;
  f32x16 acc[TN];
  // one chunk: install it from `hreg`, refill `hreg` with the chunk after the next one, multiply; the tile's epilogue after its last chunk.
  // Returns false when this workgroup has no further chunk.
  auto step = [&](uint4 (&hreg)[HV]) -> bool {
    {
      halo_store(hreg);
      if (w_pending) w_store();
      __syncthreads();
      if (ch == 0) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const float bv = bias_s[j * 32 + (lane & 31)];
#pragma unroll
          for (int r = 0; r < 16; r++) acc[j][r] = bv;
        }
      }
      {
        // next chunk: its weights (one chunk ahead: they come from L2), before the halo: the next install waits for the weights only
        const int c1 = ch + 1 == nch ? 0 : ch + 1;
        const bool next_live = ch + 1 < nch || cur.tile + (int)gridDim.x < P.ntiles;
        if (next_live && nch > 1) w_load(c1);
        w_pending = nch > 1;
        halo_load(hreg, pre, pre_ch); pre_advance();    // the chunk after it: its halo
      }
      // ---- 9 taps x CK/16 k-steps ----
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3, dx = tap % 3;
        const bf16_t* arow = Hs + (py + dy) * HROW + (px + dx) * LD + half * 8;
        const bf16_t* brow = Ws + (tap * COUT + (lane & 31)) * LD + half * 8;
#pragma unroll
        for (int kk = 0; kk < CK / 16; kk++) {
          const bf16x8 fa = *(const bf16x8*)(arow + kk * 16);
#pragma unroll
          for (int j = 0; j < TN; j++) {
            const bf16x8 fb = *(const bf16x8*)(brow + j * 32 * LD + kk * 16);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
          }
        }
      }
      __syncthreads();       // everyone is done with Hs / Ws before the next chunk (or the staging tile) overwrites them
    }
    if (ch != nch - 1) { ch++; return true; }
    // ---- epilogue: stage the 128 x COUT fp32 tile, add bias, row stores, per-channel statistics ----
    // accumulator register r of lane l: output pixel (wave*32 + (r&3) + 8*(r>>2) + 4*(l>>5)), channel j*32 + (l&31)
#pragma unroll
    for (int j = 0; j < TN; j++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        stg[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * STG_LD + j * 32 + (lane & 31)] = acc[j][r];
    }
    __syncthreads();
    {
      const int tile = cur.tile, tx0 = cur.tx * TW, ty0 = cur.ty * TH, b = cur.b;
      constexpr int C8 = COUT / 8;                     // 16-byte channel vectors per output pixel (4 / 8 / 16)
      // thread = (pixel, channel vector); 256 % C8 == 0, so a thread keeps the same channel vector for all its pixels and can
      // accumulate that vector's statistics (of the bf16-rounded values the norm layer will read) in registers on the way
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { s1[e] = 0.f; s2[e] = 0.f; }
      const int c8 = tid % C8;
      for (int v = tid; v < 128 * C8; v += 256) {
        const int p = v / C8;
        // tile pixel p: wave = p / 32, (p % 32) / 16 = row inside the wave's 2 rows, p % 16 = x
        const int oy = ty0 + 2 * (p >> 5) + ((p & 31) >> 4), ox = tx0 + (p & 15);
        const float4 a0 = *(const float4*)(stg + p * STG_LD + c8 * 8);
        const float4 a1 = *(const float4*)(stg + p * STG_LD + c8 * 8 + 4);
        const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        Vec16<bf16_t> o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          o.v[e] = (bf16_t)f[e];
          const float vq = (float)o.v[e];
          s1[e] += vq; s2[e] += vq * vq;
        }
        *(uint4*)(P.y + (((long)b * P.H + oy) * P.W + ox) * P.ldy + c8 * 8) = as_u4(o);
      }
      if (P.stats_part) {
        // lanes that share a channel vector sit C8 apart: inside the 16-lane rows they are combined by DPP row shifts (full-rate VALU,
        // out-of-row sources read 0), the 16 row results of the workgroup through LDS.  (The xor-shuffle tree this replaces was 48-64
        // ds_bpermute + as many adds and waits per tile -- a tenth of the tile program, §6.38 -- and the reason the 32-channel layers
        // took their statistics from a separate pass over the tensor.)
        if constexpr (C8 <= 8) {
#pragma unroll
          for (int e = 0; e < 8; e++) {
            if constexpr (C8 == 4) { s1[e] += DU_DPP_F32(s1[e], 0x114); s2[e] += DU_DPP_F32(s2[e], 0x114); }      // row_shr:4
            s1[e] += DU_DPP_F32(s1[e], 0x118); s2[e] += DU_DPP_F32(s2[e], 0x118);                                   // row_shr:8
          }
        }
        __syncthreads();                               // every thread has read its part of the staged tile: reuse it
        float* part = stg;                             // [16 = wave * 4 + row][COUT][2]
        if ((lane & 15) >= 16 - C8) {
          const int slot = wave * 4 + (lane >> 4);
#pragma unroll
          for (int e = 0; e < 8; e += 2)
            *(float4*)(part + ((slot * COUT) + c8 * 8 + e) * 2) = make_float4(s1[e], s2[e], s1[e + 1], s2[e + 1]);
        }
        __syncthreads();
        if (tid < COUT) {
          float t1 = 0.f, t2s = 0.f;
#pragma unroll
          for (int q2 = 0; q2 < 16; q2++) { t1 += part[(q2 * COUT + tid) * 2]; t2s += part[(q2 * COUT + tid) * 2 + 1]; }
          P.stats_part[((long)tile * COUT + tid) * 2] = t1;
          P.stats_part[((long)tile * COUT + tid) * 2 + 1] = t2s;
        }
      }
    }
    __syncthreads();
    ch = 0;
    tile_next(cur);
    return cur.tile < P.ntiles;
  };
  for (;;) {
    if (!step(hregA)) break;
    if (!step(hregB)) break;
  }
}

template <int CK, int TN>
int launch(const HaloParams& P, hipStream_t st) {
  constexpr int COUT = TN * 32, LD = CK + 8;
  constexpr int HROW = ((HXW * LD * 2 + 255) / 256 * 256) / 2;
  const int nch = P.Cin / CK;
  const size_t main_bytes = (size_t)(9 * COUT * LD + (TH + 2) * HROW) * 2;
  const size_t stg_bytes = (size_t)128 * (COUT + 4) * 4 + (size_t)4 * COUT * 2 * 4;   // fp32 tile + statistics scratch [4][COUT][2]
  const size_t lds = (nch == 1 ? main_bytes + stg_bytes : (main_bytes > stg_bytes ? main_bytes : stg_bytes)) + (size_t)COUT * 4;   // + bias
  if (lds > 160 * 1024) return DU_ERR_UNSUPPORTED;
  // the halo loads address one image through a 32-bit buffer descriptor whose out-of-range sentinel is offset 2^31
  if ((long)P.H * P.W * (P.ldx > P.ldx2 ? P.ldx : P.ldx2) * 2 >= (1L << 31)) return DU_ERR_UNSUPPORTED;
  auto kfn = conv3x3_halo_kernel<CK, TN>;
  if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DU_ERR_LAUNCH;
  const int per_cu = (int)((160 * 1024) / lds);
  int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
  if (grid > P.ntiles) grid = P.ntiles;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, P);
  return du_check_launch();
}

}  // namespace

// x (B,H,W,C1) [+ x2 (B,H,W,Cin-C1)] NHWC bf16 with pixel strides ldx/ldx2; w bf16 [Cout][9*Cin] in (tap, ci) column order;
// y (B,H,W,Cout) bf16, pixel stride ldy.  stats_part (nullable): (du_conv3x3_halo_parts(...), Cout, 2) fp32 partial sums, image-major.
// Returns DU_ERR_UNSUPPORTED for shapes this kernel does not serve (caller falls back to the implicit-GEMM path).
extern "C" int du_conv3x3_halo(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int C1, int Cin, int Cout, int B, int H, int W,
                               const void* w, const float* bias, void* y, int64_t ldy, float* stats_part, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DU_ERR_BAD_ARG;
  if (H % TH || W % TW || ldx % 8 || ldy % 8 || (x2 && (ldx2 % 8 || C1 % 8)) || Cin % 8) return DU_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)x2)) & 15) return DU_ERR_UNSUPPORTED;
  if (!x2) C1 = Cin;
  {                // streaming strip kernel (conv_strip.hip) where it serves the shape
    const int rc = du_conv3x3_strip(x, ldx, x2, ldx2, C1, Cin, Cout, B, H, W, w, bias, y, ldy, stats_part, stream);
    if (rc != DU_ERR_UNSUPPORTED) return rc;
    // The caller sized stats_part with du_conv3x3_halo_parts(), which knows the channel counts and the image but not the strides / byte
    // sizes the strip kernel also declines on (a per-image input >= 1 GiB, an output >= 2 GiB).  When the strip kernel was the planned
    // one, the tile kernel below would write B * (H / 8) * (W / 16) partial rows into a buffer of B * (H / rs) * (W / 32) and the
    // finalize step would run with the wrong count: decline instead (the caller retries without epilogue statistics).
    if (stats_part && du_conv3x3_halo_parts(C1, Cin, Cout, B, H, W) != B * (H / TH) * (W / TW)) return DU_ERR_UNSUPPORTED;
  }
  HaloParams P{};
  P.x = (const bf16_t*)x; P.ldx = ldx; P.x2 = (const bf16_t*)x2; P.ldx2 = ldx2; P.C1 = C1; P.Cin = Cin; P.Cout = Cout;
  P.B = B; P.H = H; P.W = W; P.w = (const bf16_t*)w; P.bias = bias; P.y = (bf16_t*)y; P.ldy = ldy; P.stats_part = stats_part;
  P.tilesX = W / TW; P.tilesY = H / TH; P.ntiles = B * P.tilesX * P.tilesY;
  // channel chunk: 64 when both sources split on 64-channel boundaries, else 32
  static const bool ck32 = DU_GETENV("DU_HALO_CK32") != nullptr;     // A-B aid: 32-channel chunks for the 64-output layers too (2 workgroups / CU)
  const bool c64 = Cin % 64 == 0 && C1 % 64 == 0;
  const bool c32 = Cin % 32 == 0 && C1 % 32 == 0;
  if (Cout == 32) { if (c64) return launch<64, 1>(P, st); if (c32) return launch<32, 1>(P, st); }
  if (Cout == 64) { if (c64 && !ck32) return launch<64, 2>(P, st); if (c32) return launch<32, 2>(P, st); }
  if (Cout == 128) { if (c32) return launch<32, 4>(P, st); }
  return DU_ERR_UNSUPPORTED;
}

// =====================================================================================================================
// Weight gradient of the same convolution:  dW[co][tap][ci] = sum_pix dY[pix][co] * X[pix + tap][ci]
//
// As an MFMA product the contraction runs over PIXELS, the slow index of both NHWC operands, so both fragments are fetched with the
// LDS transpose read (ds_read_b64_tr_b16) from images that are stored exactly as they lie in HBM: the (8+2) x (16+2) input halo
// [pixel][ci] (staged once per tile and chunk; the 9 taps read it at shifted pixel addresses) and the dY tile [pixel][co].
// A k-step = the 16 pixels of one tile row.  The 9 * (Cout/32) * (CK/32) accumulator tiles of a channel chunk are dealt round-robin
// to the 4 waves and stay in registers while the (persistent) workgroup walks its tiles; each workgroup then writes ONE fp32 partial
// of dW and a finalize kernel adds the partials (no atomics).  The implicit-GEMM path re-gathered X nine times and had 4 MFMAs per
// barrier (633 us for the 512^2 64->32 layer).
// =====================================================================================================================
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_v4;

__host__ __device__ constexpr int tr_pitch(int n) { return (n * 2) % 256 == 64 || (n * 2) % 256 == 192 ? n : n + 32; }

struct WgradParams {
  const bf16_t* x; long ldx; const bf16_t* x2; long ldx2; int C1, Cin, Cout;
  const bf16_t* dy; long lddy;
  int B, H, W;
  float* part;              // [gridDim.x][Cout * 9 * Cin (+ Cout with with_db)] fp32
  int tilesX, tilesY, ntiles;
  int with_db;              // also emit the bias gradient: per-workgroup column sums of dY behind the slab's weight part
};

template <int CK, int MT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_halo_kernel(WgradParams P) {
  constexpr int COUT = MT * 32;
  constexpr int NB = CK / 32;                           // 32-wide ci blocks per chunk
  constexpr int NTL = 9 * MT * NB;                      // accumulator tiles per chunk
  constexpr int TPW = (NTL + 3) / 4;                    // per wave
  constexpr int PX = tr_pitch(CK), PD = tr_pitch(COUT);
  constexpr int CV = CK / 8, DV = COUT / 8;
  constexpr int HV = (HW_ * CV + 255) / 256, YV = (128 * DV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Xs = (bf16_t*)smem_raw;                       // [180][PX]
  bf16_t* Ys = Xs + HW_ * PX;                           // [128][PD]   (tile pixel = ty * 16 + tx)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, p16 = lane & 15;
  const int txk = 8 * (g >> 1) + (p16 >> 2);            // pixel column this lane addresses in a transpose read (second read: + 4)
  const int oc = 16 * (g & 1) + 4 * (p16 & 3);          // outer-index offset inside a 32-wide block
  const int nch = P.Cin / CK;

  uint4 hreg[HV], yreg[YV];
  auto tile_origin = [&](int tile, int& b, int& ty0, int& tx0) {
    tx0 = (tile % P.tilesX) * TW;
    const int t2 = tile / P.tilesX;
    ty0 = (t2 % P.tilesY) * TH; b = t2 / P.tilesY;
  };
  auto loads = [&](int tile, int ch) {
    int b, ty0, tx0;
    tile_origin(tile, b, ty0, tx0);
    const int c0 = ch * CK;
    const bf16_t* src = P.x; long ld = P.ldx; int cofs = c0;
    if (c0 >= P.C1) { src = P.x2; ld = P.ldx2; cofs = c0 - P.C1; }
#pragma unroll
    for (int i = 0; i < HV; i++) {
      const int v = tid + i * 256;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (v < HW_ * CV) {
        const int pix = v / CV, cv = v % CV;
        const int gy = ty0 + pix / HXW - 1, gx = tx0 + pix % HXW - 1;
        if (gy >= 0 && gy < P.H && gx >= 0 && gx < P.W)
          r = *(const uint4*)(src + (((long)b * P.H + gy) * P.W + gx) * ld + cofs + cv * 8);
      }
      hreg[i] = r;
    }
#pragma unroll
    for (int i = 0; i < YV; i++) {
      const int v = tid + i * 256;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (v < 128 * DV) {
        const int pix = v / DV, dv = v % DV;
        r = *(const uint4*)(P.dy + (((long)b * P.H + ty0 + (pix >> 4)) * P.W + tx0 + (pix & 15)) * P.lddy + dv * 8);
      }
      yreg[i] = r;
    }
  };
  // bias gradient (with_db): 256 % DV == 0, so a thread's dY vectors always cover the same 8 channels (tid % DV): it keeps their sums over
  // all its tiles of chunk 0 in registers; reduced over the threads through LDS after the tile loop
  float dbs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto stores = [&](bool sum_dy) {
#pragma unroll
    for (int i = 0; i < HV; i++) {
      const int v = tid + i * 256;
      if (v < HW_ * CV) *(uint4*)(Xs + (v / CV) * PX + (v % CV) * 8) = hreg[i];
    }
#pragma unroll
    for (int i = 0; i < YV; i++) {
      const int v = tid + i * 256;
      if (v < 128 * DV) {
        *(uint4*)(Ys + (v / DV) * PD + (v % DV) * 8) = yreg[i];
        if (sum_dy) {
          const bf16x8 t = __builtin_bit_cast(bf16x8, yreg[i]);
#pragma unroll
          for (int e = 0; e < 8; e++) dbs[e] += (float)t[e];
        }
      }
    }
  };
  auto trfrag = [&](const bf16_t* q, int pitch) -> bf16x8 {      // q: this lane's address of the first 4-pixel read
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)q);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(q + 4 * pitch));
    s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
  };

  if ((int)blockIdx.x >= P.ntiles) return;
  for (int ch = 0; ch < nch; ch++) {
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    // this wave's accumulator tiles: t = wave + 4 i  ->  (tap, mt, nb)
    int xoff[TPW], yoff[TPW];
#pragma unroll
    for (int i = 0; i < TPW; i++) {
      const int t = wave + 4 * i;
      const int nb = t % NB, mt = (t / NB) % MT, tap = t / (NB * MT);
      const int dy_ = tap / 3, dx_ = tap % 3;
      xoff[i] = (dy_ * HXW + txk + dx_) * PX + nb * 32 + oc;      // + ty * HXW * PX per k-step
      yoff[i] = txk * PD + mt * 32 + oc;                          // + ty * 16 * PD per k-step
    }
    int tile = blockIdx.x;
    loads(tile, ch);
    while (tile < P.ntiles) {
      stores(P.with_db && ch == 0);
      __syncthreads();
      const int ntile = tile + gridDim.x;
      if (ntile < P.ntiles) loads(ntile, ch);
#pragma unroll
      for (int ty = 0; ty < TH; ty++) {
#pragma unroll
        for (int i = 0; i < TPW; i++) {
          if (wave + 4 * i < NTL) {
            const bf16x8 fa = trfrag(Ys + ty * 16 * PD + yoff[i], PD);
            const bf16x8 fb = trfrag(Xs + ty * HXW * PX + xoff[i], PX);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      tile = ntile;
    }
    const long slab = (long)COUT * 9 * P.Cin + (P.with_db ? COUT : 0);
    if (P.with_db && ch == 0) {        // every LDS read of the tile loop is behind its last barrier: Xs is free as scratch
      float* red = (float*)smem_raw;   // [256][8]
#pragma unroll
      for (int e = 0; e < 8; e++) red[tid * 8 + e] = dbs[e];
      __syncthreads();
      if (tid < COUT) {                // channel tid = group (tid >> 3) element (tid & 7); contributors: threads with t % DV == tid >> 3
        float sacc = 0.f;
        for (int t = tid >> 3; t < 256; t += DV) sacc += red[t * 8 + (tid & 7)];
        P.part[(long)blockIdx.x * slab + (long)COUT * 9 * P.Cin + tid] = sacc;
      }
      __syncthreads();
    }
    // flush this chunk's tiles: part[block][co][tap * Cin + ch * CK + ci]
    float* dst = P.part + (long)blockIdx.x * slab;
#pragma unroll
    for (int i = 0; i < TPW; i++) {
      const int t = wave + 4 * i;
      if (t < NTL) {
        const int nb = t % NB, mt = (t / NB) % MT, tap = t / (NB * MT);
        const int ci = ch * CK + nb * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          dst[((long)co * 9 + tap) * P.Cin + ci] = acc[i][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same product with every fragment read ONCE per tile and two tiles in flight.
//
// What bounded the kernel above (profiles/r04: 1.3-2.9 TB/s, the matrix pipe 17 % busy): (1) a wave's accumulator tiles were dealt
// round-robin over (tap, mt, nb), so every MFMA fetched both of its fragments -- 4 transpose reads per MFMA, 2x what the LDS pipe
// delivers per matrix-pipe slot; (2) one tile (20-39 KB per CU) of loads in flight, consumed one iteration later: 5 MB on the chip
// against the ~16 MB that 8 TB/s x 2 us of loaded latency ask for; (3) two barriers per tile.
// Here a wave owns ALL 9 TAPS of one (mt, nb) block pair (9 accumulators = 144 registers) over RW of the tile's 8 rows:
//   * the dY fragment of a k-step (one tile row) is read once and feeds 9 MFMAs;
//   * the X fragment of halo row r and column shift dx serves tap (dy, dx) at k-step r - dy: a rolling window of 3 rows x 3 shifts
//     stays in registers, 3 new fragments per k-step -- 8 transpose reads per 9 MFMAs instead of 36;
//   * two LDS stages and two register sets: the loads of tile i + 3 are issued while tile i is multiplied (two tiles = 40-78 KB per
//     CU in flight), a tile's registers -> LDS copy goes to the stage that is not being read, ONE barrier per tile;
//   * (mt, nb) pairs fewer than 4 (32 output channels, or one 32-channel input block): the waves split the tile's rows RG ways and
//     their accumulators are added through LDS once, after the tile loop (tap by tap, two alternating 16 KB buffers).
// Slab layout, bias-gradient sums and the finalize pass are those of the kernel above.
// ---------------------------------------------------------------------------------------------------------------------
template <int CK, int MT>
constexpr size_t wgrad_rows_lds() { return (size_t)2 * (HW_ * tr_pitch(CK) + 128 * tr_pitch(MT * 32)) * 2; }

// one workgroup per CU with two register sets of prefetch, except the smallest form (32 -> 32 channels: 39 KB, 246 registers): two per CU
template <int CK, int MT>
constexpr bool wgrad_rows_two() { return !(MT == 1 && wgrad_rows_lds<CK, MT>() <= 80 * 1024); }

template <int CK, int MT>
__global__ __launch_bounds__(256, (wgrad_rows_two<CK, MT>() ? 1 : 2)) void conv3x3_wgrad_rows_kernel(WgradParams P) {
  constexpr int COUT = MT * 32;
  constexpr int NB = CK / 32;
  static_assert(MT * NB == 1 || MT * NB == 2 || MT * NB == 4, "4 waves = (mt, nb) pairs x row groups");
  constexpr int RG = 4 / (MT * NB), RW = TH / RG;       // row groups, tile rows per wave
  constexpr int PX = tr_pitch(CK), PD = tr_pitch(COUT);
  constexpr int CV = CK / 8, DV = COUT / 8;
  constexpr int HV = (HW_ * CV + 255) / 256, YV = (128 * DV + 255) / 256;
  constexpr int STAGE = HW_ * PX + 128 * PD;            // bf16 elements: [180][PX] halo + [128][PD] dY tile
  constexpr bool TWO = wgrad_rows_two<CK, MT>();        // one workgroup per CU: two register sets of prefetch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* S = (bf16_t*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, p16 = lane & 15;
  const int txk = 8 * (g >> 1) + (p16 >> 2);            // pixel column this lane addresses in a transpose read (second read: + 4)
  const int oc = 16 * (g & 1) + 4 * (p16 & 3);          // outer-index offset inside a 32-wide block
  const int nb = wave % NB, mt = (wave / NB) % MT, rg = wave / (NB * MT);
  const int r0 = rg * RW;
  const int xoffb = txk * PX + nb * 32 + oc, yoffb = txk * PD + mt * 32 + oc;
  const int nch = P.Cin / CK;
  const int G = gridDim.x;

  // per-thread constants of the staging copies (the same for every tile: without them the compiler rebuilt -- and kept live -- the
  // divisions of every one of the 5 load and 3 store sites): LDS offsets, packed halo coordinates, pixel offsets relative to the tile
  int hlds[HV], ylds[YV], hpix[HV], ypix[YV];
  unsigned hpk[HV];
#pragma unroll
  for (int i = 0; i < HV; i++) {
    const int v = tid + i * 256, pix = v / CV, cv = v % CV, py = pix / HXW, px = pix % HXW;
    hlds[i] = pix * PX + cv * 8;
    hpix[i] = (py - 1) * P.W + (px - 1);                 // x ld + cv * 8 + chunk offset: per chunk below
    hpk[i] = v < HW_ * CV ? (unsigned)py | ((unsigned)px << 8) : 0xffffu;
  }
#pragma unroll
  for (int i = 0; i < YV; i++) {
    const int v = tid + i * 256, pix = v / DV, dv = v % DV;
    ylds[i] = pix * PD + dv * 8;
    ypix[i] = ((pix >> 4) * P.W + (pix & 15)) * (int)P.lddy + dv * 8;
  }
  static_assert((128 * DV) % 256 == 0, "whole dY vectors per thread");
  // buffer descriptors: a halo pixel outside the image is requested at an offset beyond num_records and comes back as zeros (no branch)
  const unsigned pixels = (unsigned)P.B * (unsigned)P.H * (unsigned)P.W;
  const auto yrs = __builtin_amdgcn_make_buffer_rsrc((void*)P.dy, 0, (int)(pixels * (unsigned)P.lddy * 2u), 0x00020000);
  float dbs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // bias gradient: see the kernel above
  auto stores = [&](const uint4 (&hreg)[HV], const uint4 (&yreg)[YV], int stage, bool sum_dy) {
    bf16_t* Xs = S + stage * STAGE;
    bf16_t* Ys = Xs + HW_ * PX;
#pragma unroll
    for (int i = 0; i < HV; i++)
      if ((i + 1) * 256 <= HW_ * CV || tid + i * 256 < HW_ * CV) *(uint4*)(Xs + hlds[i]) = hreg[i];
#pragma unroll
    for (int i = 0; i < YV; i++) {
      *(uint4*)(Ys + ylds[i]) = yreg[i];
      if (sum_dy) {
        const bf16x8 t = __builtin_bit_cast(bf16x8, yreg[i]);
#pragma unroll
        for (int e = 0; e < 8; e++) dbs[e] += (float)t[e];
      }
    }
  };
  auto trfrag = [&](const bf16_t* q, int pitch) -> bf16x8 {      // q: this lane's address of the first 4-pixel read
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)q);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(q + 4 * pitch));
    s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
  };

  if ((int)blockIdx.x >= P.ntiles) return;
  for (int ch = 0; ch < nch; ch++) {
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    auto compute = [&](int stage) {
      const bf16_t* Xs = S + stage * STAGE;
      const bf16_t* Ys = Xs + HW_ * PX;
      const bf16_t* xq = Xs + r0 * HXW * PX + xoffb;
      const bf16_t* yq = Ys + r0 * 16 * PD + yoffb;
      bf16x8 xw[3][3];
#pragma unroll
      for (int d = 0; d < 2; d++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) xw[d][dx] = trfrag(xq + (d * HXW + dx) * PX, PX);
      bf16x8 fa = trfrag(yq, PD);
#pragma unroll
      for (int t = 0; t < RW; t++) {
        // this step's reads: halo row t + 2 (first used by the 7th MFMA below) and the NEXT step's dY fragment; the step boundary is
        // pinned -- left alone the scheduler hoists the reads of all 8 steps to the top (128 more live registers: it spilled)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) xw[(t + 2) % 3][dx] = trfrag(xq + ((t + 2) * HXW + dx) * PX, PX);
        bf16x8 fan = fa;
        if (t + 1 < RW) fan = trfrag(yq + (t + 1) * 16 * PD, PD);
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++)
            acc[d * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xw[(t + d) % 3][dx], acc[d * 3 + dx], 0, 0, 0);
        fa = fan;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const bool sdy = P.with_db && ch == 0;
    const int c0 = ch * CK;
    const bool second = c0 >= P.C1;
    const int ldc = (int)(second ? P.ldx2 : P.ldx), cofs = second ? c0 - P.C1 : c0;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? P.x2 : P.x), 0, (int)(pixels * (unsigned)ldc * 2u), 0x00020000);
    int hrel[HV];
#pragma unroll
    for (int i = 0; i < HV; i++) hrel[i] = hpix[i] * ldc + (tid + i * 256) % CV * 8 + cofs;
    auto loads = [&](uint4 (&hreg)[HV], uint4 (&yreg)[YV], int tile) {
      // a tile beyond the last one: every lane asks beyond num_records (zeros, no traffic; the set is never copied to LDS).  No branch
      // around the loads: with a path that issues none the compiler's counted waits in front of the OTHER set's copy collapse to
      // vmcnt(0), i.e. one tile in flight again
      const bool live = tile < P.ntiles;
      const int tx0 = (tile % P.tilesX) * TW, t2 = tile / P.tilesX;
      const int ty0 = (t2 % P.tilesY) * TH, b = t2 / P.tilesY;
      const int org = (b * P.H + ty0) * P.W + tx0;        // pixel index of the tile's first output pixel
#pragma unroll
      for (int i = 0; i < HV; i++) {
        const unsigned gy = (unsigned)(ty0 - 1) + (hpk[i] & 0xffu), gx = (unsigned)(tx0 - 1) + (hpk[i] >> 8);
        const unsigned off = live && gy < (unsigned)P.H && gx < (unsigned)P.W ? (unsigned)(org * ldc + hrel[i]) * 2u : 0x80000000u;
        hreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < YV; i++)
        yreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yrs, live ? (org * (int)P.lddy + ypix[i]) * 2 : (int)0x80000000u, 0, 0));
    };
    uint4 hA[HV], yA[YV];
    // workgroup i runs on XCD i % 8: give each XCD G / 8 CONSECUTIVE tiles of every round (two tile rows of a 256-wide image), so the
    // one-pixel halo ring a tile shares with its neighbours is found in that XCD's L2 instead of being fetched once per XCD
    int tile = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3), st = 0;
    if constexpr (TWO) {
      uint4 hB[HV], yB[YV];
      loads(hA, yA, tile);
      loads(hB, yB, tile + G);
      stores(hA, yA, 0, sdy);
      loads(hA, yA, tile + 2 * G);
      __syncthreads();
      for (;;) {
        // stage st holds `tile`; set B holds tile + G, set A tile + 2 G (both in flight)
        if (tile + G < P.ntiles) stores(hB, yB, st ^ 1, sdy);
        loads(hB, yB, tile + 3 * G);
        compute(st);
        __syncthreads();
        tile += G; st ^= 1;
        if (tile >= P.ntiles) break;
        // set A holds tile + G, set B tile + 2 G
        if (tile + G < P.ntiles) stores(hA, yA, st ^ 1, sdy);
        loads(hA, yA, tile + 3 * G);
        compute(st);
        __syncthreads();
        tile += G; st ^= 1;
        if (tile >= P.ntiles) break;
      }
    } else {                           // two workgroups per CU: one register set each (256 registers per wave), the CU still has two tiles in flight
      loads(hA, yA, tile);
      stores(hA, yA, 0, sdy);
      loads(hA, yA, tile + G);
      __syncthreads();
      for (;;) {
        if (tile + G < P.ntiles) stores(hA, yA, st ^ 1, sdy);
        loads(hA, yA, tile + 2 * G);
        compute(st);
        __syncthreads();
        tile += G; st ^= 1;
        if (tile >= P.ntiles) break;
      }
    }
    // (every LDS read of the tile loop is behind its last barrier: the stages are free as scratch)
    const long slab = (long)COUT * 9 * P.Cin + (P.with_db ? COUT : 0);
    if (sdy) {
      float* red = (float*)smem_raw;   // [256][8]
#pragma unroll
      for (int e = 0; e < 8; e++) red[tid * 8 + e] = dbs[e];
      __syncthreads();
      if (tid < COUT) {                // channel tid = group (tid >> 3) element (tid & 7); contributors: threads with t % DV == tid >> 3
        float sacc = 0.f;
        for (int t = tid >> 3; t < 256; t += DV) sacc += red[t * 8 + (tid & 7)];
        P.part[(long)blockIdx.x * slab + (long)COUT * 9 * P.Cin + tid] = sacc;
      }
      __syncthreads();
    }
    if constexpr (RG > 1) {            // add the row groups of each (mt, nb) pair into its rg = 0 wave
      float4* red = (float4*)smem_raw; // [2][4 waves][4][64 lanes] float4 = 2 x 16 KB
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        float4* buf = red + (tap & 1) * 1024;
        if (rg > 0) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            buf[(wave * 4 + j) * 64 + lane] = make_float4(acc[tap][4 * j], acc[tap][4 * j + 1], acc[tap][4 * j + 2], acc[tap][4 * j + 3]);
        }
        __syncthreads();               // (the buffer of tap - 1 is read before this barrier, that of tap + 1 written after it)
        if (rg == 0) {
#pragma unroll
          for (int o = 1; o < RG; o++) {
            const int w2 = wave + o * NB * MT;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float4 v = buf[(w2 * 4 + j) * 64 + lane];
              acc[tap][4 * j] += v.x; acc[tap][4 * j + 1] += v.y; acc[tap][4 * j + 2] += v.z; acc[tap][4 * j + 3] += v.w;
            }
          }
        }
      }
      __syncthreads();
    }
    // flush this chunk's tiles: part[block][co][tap * Cin + ch * CK + ci]
    if (rg == 0) {
      float* dst = P.part + (long)blockIdx.x * slab;
      // this lane's part of the 144 store addresses, made opaque here: left visible, the compiler builds all of them at kernel entry
      // (they do not depend on the tile loop) and parks them in scratch -- 140 spills in, 137 reloads out
      int voff = ((mt * 32 + 4 * (lane >> 5)) * 9) * P.Cin + ch * CK + nb * 32 + (lane & 31);
      asm volatile("" : "+v"(voff));
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
#pragma unroll
        for (int r = 0; r < 16; r++) dst[voff + (((r & 3) + 8 * (r >> 2)) * 9 + tap) * P.Cin] = acc[tap][r];
      }
    }
  }
}

template <int CK, int MT>
int launch_wgrad_rows(WgradParams& P, int max_blocks, hipStream_t st) {
  constexpr size_t lds = wgrad_rows_lds<CK, MT>();
  auto kfn = conv3x3_wgrad_rows_kernel<CK, MT>;
  static bool attr_set = false;          // once per instantiation (as launch_pp / rk_launch)
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DU_ERR_LAUNCH;
    attr_set = true;
  }
  int grid = max_blocks < P.ntiles ? max_blocks : P.ntiles;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, P);
  return du_check_launch();
}

template <int CK, int MT>
int launch_wgrad(WgradParams& P, int max_blocks, hipStream_t st) {
  constexpr int COUT = MT * 32;
  const size_t lds = (size_t)(HW_ * tr_pitch(CK) + 128 * tr_pitch(COUT)) * 2;
  auto kfn = conv3x3_wgrad_halo_kernel<CK, MT>;
  if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DU_ERR_LAUNCH;
  int grid = max_blocks < P.ntiles ? max_blocks : P.ntiles;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, P);
  return du_check_launch();
}

}  // namespace

int g_wgrad_rows = 1;     // du_set_option key 13: 1 = conv3x3_wgrad_rows_kernel for 32 / 64 output channels, 128 on the grouped launch's in-place gather
                          // (default); 2 = 128 output channels on the rows kernel too (round 6: 106 / 200 us for the two layers against 400 us of
                          // grouped launch, but x1.0001 in the step -- profiles/r06_ab_wgrad128_side_v1.txt -- so it stays opt-in); 0 = the round-3 kernel

// number of workgroups (= partial dW slabs) the weight-gradient kernel uses for this shape; 0 = shape not served
extern "C" int du_conv3x3_wgrad_halo_blocks(int C1, int Cin, int Cout, int B, int H, int W) {
  if (H % TH || W % TW || B <= 0) return 0;
  const int ntiles = B * (H / TH) * (W / TW);
  if (Cout == 128 && g_wgrad_rows >= 2 && Cin % 32 == 0 && C1 % 32 == 0 && Cin <= 256 &&
      (long)B * H * W * (Cin > 128 ? Cin : 128) * 2 < 0x7fffffffL) {
    // round 6: the 128-output layers of the first decoder stage (dinounet_training.py:581-592) on the rows kernel, one 32-output block per
    // wave over 32-channel input chunks; the partial slabs are 0.6-1.2 MB each: 128 workgroups (75-150 MB for the finalize pass to read)
    return ntiles < 128 ? ntiles : 128;
  }
  if (!(Cout == 32 || Cout == 64) || Cin % 32 || C1 % 32) return 0;   // (round-3 kernel: 9 accumulator tiles per wave spill at 128 outputs)
  if ((long)Cout * 9 * Cin > 80L * 1024) return 0;                 // larger filters: MFMA-bound anyway, partial slabs too big
  // persistent workgroups = partial dW slabs.  36-50 KB of LDS and 4 waves each: one per CU leaves every SIMD with a single wave and
  // nothing to switch to while it waits for its LDS writes / barrier / transpose reads (1.4 TB/s measured); two per CU double the
  // slab traffic of the finalize (<= 2 x 75 MB) and hide that latency.  DU_HALO_WGRAD_BLOCKS overrides (A-B aid).
  // measured (bench A-B, round 3): 512^2 64->32 257 -> 168 us, 32->32 138 -> 93 us with 512 workgroups; the 64-output layers at 256^2
  // (147-295 KB slabs) lose 10 % to the doubled finalize traffic: two per CU only while a slab stays under 80 KB
  static const int cap_env = DU_GETENV("DU_HALO_WGRAD_BLOCKS") ? atoi(DU_GETENV("DU_HALO_WGRAD_BLOCKS")) : 0;
  int cap = cap_env > 0 ? cap_env : ((long)Cout * 9 * Cin * 4 <= 80L * 1024 ? 512 : 256);
  // round-5 kernel: two LDS stages; all forms but 32 -> 32 channels run one workgroup per CU (72-118 KB), so a second slab per CU buys nothing
  // (keyed on the kernel that will run: past the 32-bit offset limit -- judged here on the dense tensors, ld = channels -- the round-3
  //  kernel takes the shape and keeps the workgroup count it was tuned for)
  const long pix = (long)B * H * W;
  const bool small = pix * (Cin > C1 ? Cin : C1) * 2 < 0x7fffffffL && pix * Cout * 2 < 0x7fffffffL;
  if (g_wgrad_rows && small && cap_env <= 0 && (Cout == 64 || (Cin % 64 == 0 && C1 % 64 == 0))) cap = 256;
  return ntiles < cap ? ntiles : cap;
}

// x / x2 as in du_conv3x3_halo, dy (B,H,W,Cout) bf16; part: du_conv3x3_wgrad_halo_blocks(...) x Cout x 9*Cin fp32 scratch;
// dw (Cout, 9*Cin) fp32 in (tap, ci) column order is OVERWRITTEN (sum of the partial slabs, via du_strip_finalize).
extern "C" int du_conv3x3_wgrad_halo(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int C1, int Cin, int Cout, int B, int H,
                                     int W, const void* dy, int64_t lddy, float* part, float* dw, int with_db, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || !dy || !part || !dw) return DU_ERR_BAD_ARG;
  if (!x2) C1 = Cin;
  const int blocks = du_conv3x3_wgrad_halo_blocks(C1, Cin, Cout, B, H, W);
  if (blocks <= 0 || ldx % 8 || lddy % 8 || (x2 && ldx2 % 8)) return DU_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)x2)) & 15) return DU_ERR_UNSUPPORTED;
  WgradParams P{};
  P.x = (const bf16_t*)x; P.ldx = ldx; P.x2 = (const bf16_t*)x2; P.ldx2 = ldx2; P.C1 = C1; P.Cin = Cin; P.Cout = Cout;
  P.dy = (const bf16_t*)dy; P.lddy = lddy; P.B = B; P.H = H; P.W = W; P.part = part;
  P.tilesX = W / TW; P.tilesY = H / TH; P.ntiles = B * P.tilesX * P.tilesY;
  P.with_db = with_db ? 1 : 0;
  const bool c64 = Cin % 64 == 0 && C1 % 64 == 0;
  int rc = DU_ERR_UNSUPPORTED;
  // the round-5 kernel addresses the three tensors through 32-bit buffer offsets
  const long pix = (long)B * H * W;
  const bool small = pix * ldx * 2 < 0x7fffffffL && pix * lddy * 2 < 0x7fffffffL && (!x2 || pix * ldx2 * 2 < 0x7fffffffL);
  if (Cout == 128) {
    if (!(g_wgrad_rows >= 2 && small)) return DU_ERR_UNSUPPORTED;
    rc = launch_wgrad_rows<32, 4>(P, blocks, st);
  } else if (g_wgrad_rows && small) {
    if (Cout == 32) rc = c64 ? launch_wgrad_rows<64, 1>(P, blocks, st) : launch_wgrad_rows<32, 1>(P, blocks, st);
    else if (Cout == 64) rc = c64 ? launch_wgrad_rows<64, 2>(P, blocks, st) : launch_wgrad_rows<32, 2>(P, blocks, st);
  } else if (Cout == 32) rc = c64 ? launch_wgrad<64, 1>(P, blocks, st) : launch_wgrad<32, 1>(P, blocks, st);
  else if (Cout == 64) rc = launch_wgrad<32, 2>(P, blocks, st);        // 32-channel chunks: 5 accumulator tiles per wave, no spills
  if (rc != DU_OK) return rc;
  // finalize works on (C, 2) pairs: C = elements / 2; with_db the Cout bias-gradient sums ride behind the weight gradient
  return du_strip_finalize(part, dw, 1, blocks, (Cout * 9 * Cin + (with_db ? Cout : 0)) / 2, stream);
}
