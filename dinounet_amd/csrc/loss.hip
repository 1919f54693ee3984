// Fused soft-Dice + cross-entropy loss of the reference trainer on fp32 NCHW logits:
//   DC_and_CE_loss (dinounet/training/loss/compound_losses.py:8-56) = RobustCrossEntropyLoss (mean over pixels)
//   + MemoryEfficientSoftDiceLoss(batch_dice=True, do_bg=False, smooth=1e-5) (dice.py:58-119):
//       dc_c = (2 I_c + s) / clip(G_c + P_c + s, 1e-8),   I_c = sum p_c [t = c],  P_c = sum p_c,  G_c = sum [t = c]   (c >= 1)
//       loss = CE - mean_c dc_c
// One pass over the logits produces the softmax-dependent sums (stock torch needs ~25 launches and several full-size temporaries and
// its multi-block reductions misbehave under hipGraph replay); the backward pass recomputes the softmax and writes d loss / d logits.
// HBM-bound: forward reads K*4 + 8 bytes per pixel, backward reads the same and writes K*4.
#include "common.h"

namespace {

constexpr int MAXK = 16;

// sums layout: [0] = sum of -log p_target, then for c = 1..K-1: [1 + 3(c-1) + {0,1,2}] = (I_c, P_c, G_c)
template <int K>
__global__ __launch_bounds__(256) void dice_ce_partial_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                              float* __restrict__ part, int B, long HW) {
  constexpr int NS = 1 + 3 * (K - 1);
  float acc[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) acc[i] = 0.f;
  const long npix = (long)B * HW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / HW, r = p - b * HW;
    const float* lp = logits + b * K * HW + r;
    float v[K];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; k++) { v[k] = lp[(long)k * HW]; mx = fmaxf(mx, v[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) { v[k] = __expf(v[k] - mx); se += v[k]; }
    const float inv = 1.f / se;
    const int t = (int)target[p];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const float pk = v[k] * inv;
      if (k == t) acc[0] -= __logf(fmaxf(pk, 1e-38f));
      if (k >= 1) {
        acc[1 + 3 * (k - 1) + 1] += pk;
        if (k == t) { acc[1 + 3 * (k - 1)] += pk; acc[1 + 3 * (k - 1) + 2] += 1.f; }
      }
    }
  }
  __shared__ float red[4][NS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NS) part[(long)blockIdx.x * NS + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void dice_ce_sum_kernel(const float* __restrict__ part, float* __restrict__ sums, int blocks, int NS) {
  __shared__ float red[256];
  for (int i = 0; i < NS; i++) {
    float a = 0.f;
    for (int b = threadIdx.x; b < blocks; b += 256) a += part[(long)b * NS + i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) sums[i] = red[0];
    __syncthreads();
  }
}

// loss and the per-class backward coefficients: d loss / d p_c(pixel) = coef[2(c-1)] * [t = c] + coef[2(c-1) + 1]
__global__ void dice_ce_coef_kernel(const float* __restrict__ sums, float* __restrict__ loss, float* __restrict__ coef, int K,
                                    float inv_npix, float smooth, float grad_mult) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float dc_mean = 0.f;
  const float invc = 1.f / (float)(K - 1);
  for (int c = 1; c < K; c++) {
    const float I = sums[1 + 3 * (c - 1)], P = sums[1 + 3 * (c - 1) + 1], G = sums[1 + 3 * (c - 1) + 2];
    const float num = 2.f * I + smooth;
    const float raw = G + P + smooth;
    const float den = fmaxf(raw, 1e-8f);
    dc_mean += num / den * invc;
    coef[2 * (c - 1)] = -2.f * invc / den * grad_mult;
    coef[2 * (c - 1) + 1] = (raw > 1e-8f ? num / (den * den) * invc : 0.f) * grad_mult;
  }
  loss[0] = sums[0] * inv_npix - dc_mean;
}

template <int K>
__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                          const float* __restrict__ coef, const float* __restrict__ gout,
                                                          float* __restrict__ dlogits, int B, long HW, float inv_npix) {
  float ca[K], cb[K];
  ca[0] = 0.f; cb[0] = 0.f;
#pragma unroll
  for (int c = 1; c < K; c++) { ca[c] = coef[2 * (c - 1)]; cb[c] = coef[2 * (c - 1) + 1]; }
  const float go = gout ? gout[0] : 1.f;
  const long npix = (long)B * HW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / HW, r = p - b * HW;
    const float* lp = logits + b * K * HW + r;
    float v[K];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; k++) { v[k] = lp[(long)k * HW]; mx = fmaxf(mx, v[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) { v[k] = __expf(v[k] - mx); se += v[k]; }
    const float inv = 1.f / se;
    const int t = (int)target[p];
    float g[K], dot = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
      v[k] *= inv;
      g[k] = (k == t ? ca[k] : 0.f) + cb[k];
      dot += g[k] * v[k];
    }
    float* dp = dlogits + b * K * HW + r;
#pragma unroll
    for (int k = 0; k < K; k++) dp[(long)k * HW] = go * ((v[k] - (k == t ? 1.f : 0.f)) * inv_npix + v[k] * (g[k] - dot));
  }
}

int loss_grid(long npix) { long g = (npix + 255) / 256 / 4; if (g < 1) g = 1; if (g > 2048) g = 2048; return (int)g; }

}  // namespace

extern "C" int64_t du_dice_ce_ws_elems(int B, int K, int64_t HW) {
  if (B <= 0 || K < 2 || K > MAXK || HW <= 0) return 0;
  return (int64_t)loss_grid((long)B * HW) * (1 + 3 * (K - 1));
}

#define LOSS_K_SWITCH(K, CALL) \
  switch (K) { case 2: { CALL(2); break; } case 3: { CALL(3); break; } case 4: { CALL(4); break; } case 5: { CALL(5); break; } \
               case 6: { CALL(6); break; } case 7: { CALL(7); break; } case 8: { CALL(8); break; } default: return DU_ERR_UNSUPPORTED; }

// sums (1 + 3(K-1)) fp32 <- per-pixel softmax sums of this rank's batch
extern "C" int du_dice_ce_sums(const float* logits, const int64_t* target, float* sums, int B, int K, int64_t HW, float* ws,
                               int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!logits || !target || !sums || !ws || B <= 0 || K < 2 || HW <= 0) return DU_ERR_BAD_ARG;
  const int grid = loss_grid((long)B * HW);
  const int NS = 1 + 3 * (K - 1);
  if (ws_elems < (int64_t)grid * NS) return DU_ERR_BAD_ARG;
#define CALL(KK) hipLaunchKernelGGL(dice_ce_partial_kernel<KK>, dim3(grid), dim3(256), 0, st, logits, target, ws, B, (long)HW)
  LOSS_K_SWITCH(K, CALL)
#undef CALL
  hipLaunchKernelGGL(dice_ce_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, sums, grid, NS);
  return du_check_launch();
}

// loss (1) and coef (2(K-1)) from (possibly all-reduced) sums; npix = pixels of THIS rank (the CE term is a local mean);
// grad_mult = world size when the dice sums were all-reduced (backward of the all-gather sums the identical coefficient over ranks)
extern "C" int du_dice_ce_finish(const float* sums, float* loss, float* coef, int K, int64_t npix, float smooth, float grad_mult,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!sums || !loss || !coef || K < 2 || K > MAXK || npix <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(dice_ce_coef_kernel, dim3(1), dim3(64), 0, st, sums, loss, coef, K, 1.f / (float)npix, smooth, grad_mult);
  return du_check_launch();
}

// dlogits = grad_out[0] * d loss / d logits   (grad_out: device scalar, NULL = 1)
extern "C" int du_dice_ce_bwd(const float* logits, const int64_t* target, const float* coef, const float* grad_out, float* dlogits,
                              int B, int K, int64_t HW, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!logits || !target || !coef || !dlogits || B <= 0 || K < 2 || HW <= 0) return DU_ERR_BAD_ARG;
  const long npix = (long)B * HW;
  long g = (npix + 255) / 256; if (g > 8192) g = 8192;
#define CALL(KK) hipLaunchKernelGGL(dice_ce_bwd_kernel<KK>, dim3((unsigned)g), dim3(256), 0, st, logits, target, coef, grad_out, dlogits, B, (long)HW, 1.f / (float)npix)
  LOSS_K_SWITCH(K, CALL)
#undef CALL
  return du_check_launch();
}
