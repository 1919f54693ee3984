// Fused gradient clipping + Nesterov SGD over all trainable tensors (gfx950) -- SURVEY.md 8(f) rank 1, second half.
//
// Replaces, per train step, torch.nn.utils.clip_grad_norm_(params, 12) (nnUNetTrainer.py:922) followed by torch.optim.SGD.step()
// (nnUNetTrainer.py:486,924: momentum 0.99, nesterov, weight decay 3e-5): ~20 multi-tensor launches of 15-25 us each.  Three launches:
//   (1) sqnorm_kernel      one workgroup per 4096-element chunk of some gradient -> one fp32 partial sum of squares (no atomics)
//   (2) clip_coef_kernel   one workgroup: total_norm = sqrt(sum), coef = min(1, max_norm / (total_norm + 1e-6))   (clip_grad_norm_ semantics)
//   (3) sgd_kernel         per chunk: g *= coef (written back, like clip_grad_norm_);  d = g + wd * p;  buf = mu * buf + d;
//                          d = nesterov ? d + mu * buf : buf;  p -= lr * d          (torch/optim/sgd.py _single_tensor_sgd, dampening 0;
//                          a zero-initialised buffer reproduces torch's first-step "buf = d")
// Tensors are described by a device table (4 x int64 per tensor: param, grad, momentum buffer, numel) and a workgroup prefix, like
// du_pack_weights.  Hyper-parameters live in DEVICE memory (hyper[0..4] = lr, momentum, weight_decay, max_norm, nesterov) so that a
// captured hipGraph follows the learning-rate schedule without re-capture.
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 4096;

__device__ __forceinline__ int row_of_block(const int64_t* __restrict__ bprefix, int n, long bid) {
  int lo = 0, hi = n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bprefix[mid] <= bid) lo = mid; else hi = mid; }
  return lo;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ bprefix, int n,
                                                     float* __restrict__ partials) {
  __shared__ float red[4];
  const int row = row_of_block(bprefix, n, blockIdx.x);
  const float* g = (const float*)table[row * 4 + 1];
  const long numel = table[row * 4 + 3];
  const long e0 = ((long)blockIdx.x - bprefix[row]) * OPT_CHUNK;
  const long e1 = min(numel, e0 + OPT_CHUNK);
  float acc = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {
    for (long e = e0 + threadIdx.x * 4; e < e1; e += 1024) {
      if (e + 4 <= e1) { const float4 v = *(const float4*)(g + e); acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
      else for (long t = e; t < e1; t++) acc += g[t] * g[t];
    }
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += 256) acc += g[e] * g[e];
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// out[0] = total_norm, out[1] = clip coefficient
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int nblocks, const float* __restrict__ hyper,
                                                        float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partials[i];
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    const float c = hyper[3] / (norm + 1e-6f);
    out[0] = norm;
    out[1] = c < 1.f ? c : 1.f;
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ bprefix, int n,
                                                  const float* __restrict__ hyper, const float* __restrict__ normcoef) {
  const int row = row_of_block(bprefix, n, blockIdx.x);
  float* p = (float*)table[row * 4 + 0];
  float* g = (float*)table[row * 4 + 1];
  float* m = (float*)table[row * 4 + 2];
  const long numel = table[row * 4 + 3];
  const long e0 = ((long)blockIdx.x - bprefix[row]) * OPT_CHUNK;
  const long e1 = min(numel, e0 + OPT_CHUNK);
  const float lr = hyper[0], mu = hyper[1], wd = hyper[2], coef = normcoef[1];
  const bool nesterov = hyper[4] != 0.f;
  auto upd = [&](float& pv, float& gv, float& mv) {
    gv *= coef;
    float d = gv + wd * pv;
    mv = mu * mv + d;
    d = nesterov ? d + mu * mv : mv;
    pv -= lr * d;
  };
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m)) & 15) == 0;
  if (al) {
    for (long e = e0 + threadIdx.x * 4; e < e1; e += 1024) {
      if (e + 4 <= e1) {
        float4 pv = *(float4*)(p + e), gv = *(float4*)(g + e), mv = *(float4*)(m + e);
        upd(pv.x, gv.x, mv.x); upd(pv.y, gv.y, mv.y); upd(pv.z, gv.z, mv.z); upd(pv.w, gv.w, mv.w);
        *(float4*)(p + e) = pv; *(float4*)(g + e) = gv; *(float4*)(m + e) = mv;
      } else {
        for (long t = e; t < e1; t++) upd(p[t], g[t], m[t]);
      }
    }
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += 256) upd(p[e], g[e], m[e]);
  }
}

}  // namespace

extern "C" int64_t du_clip_sgd_ws_elems(int nblocks) { return nblocks > 0 ? (int64_t)nblocks + 2 : 0; }

extern "C" int du_clip_sgd(const int64_t* table, const int64_t* bprefix, int n_tensors, int nblocks, const float* hyper, float* ws,
                           int64_t ws_elems, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!table || !bprefix || !hyper || !ws || n_tensors <= 0 || nblocks <= 0 || ws_elems < (int64_t)nblocks + 2) return DU_ERR_BAD_ARG;
  float* normcoef = ws + nblocks;          // [total_norm, clip coefficient] behind the partials
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nblocks), dim3(256), 0, st, table, bprefix, n_tensors, ws);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nblocks, hyper, normcoef);
  hipLaunchKernelGGL(sgd_kernel, dim3(nblocks), dim3(256), 0, st, table, bprefix, n_tensors, hyper, (const float*)normcoef);
  return du_check_launch();
}
