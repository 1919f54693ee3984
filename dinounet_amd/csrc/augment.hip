// GPU-side training augmentation for 2D slices: the transforms of nnUNetTrainer.get_training_transforms
// (dinounet/training/nnUNetTrainer/nnUNetTrainer.py:684-776) as HIP kernels over NCHW fp32 batches already resident in HBM, so the
// host-side batchgenerators worker pool (:601-650) is not on the critical path of a 200+ slices/s step.
//
// The reference delegates every transform to `batchgenerators` (requirements.txt; not vendored in the reference tree): parity is
// UNPINNED against that package; each kernel restates the package's published algorithm and is tested against a numpy / scipy
// restatement (oracle/augment_oracle.py).  One documented deviation: image resampling uses the Keys bicubic kernel (a = -0.5) where
// batchgenerators calls scipy.ndimage.map_coordinates(order=3) (cubic B-spline with prefilter); the two agree to O(h^3) on smooth data.
//
//   du_aug_spatial      SpatialTransform (rotation, isotropic scale, centre crop) + MirrorTransform folded into one resampling pass;
//                       labels follow batchgenerators' order-1 rule (per label: bilinear interpolation of its one-hot map >= 0.5 with
//                       cval -1 outside the image, ascending labels overwrite; pixels no label claims are 0, which is also what
//                       RemoveLabelTransform(-1, 0) leaves)
//   du_aug_plane_stats  per (sample, channel) plane: mean, std (population), min, max
//   du_aug_noise_mult   GaussianNoiseTransform (x + N(0, sigma^2), sigma ~ U(0, 0.1) drawn by the host) then BrightnessMultiplicativeTransform (x * m)
//   du_aug_contrast     ContrastAugmentationTransform, preserve_range: clip((x - mean) * f + mean, min, max)
//   du_aug_gamma        GammaTransform body: [negate] ((x - min) / (range + 1e-7))^gamma * range + min   (retain_stats: du_aug_affine)
//   du_aug_affine       x * a + b per plane (retain_stats rescale, final negate)
//   du_aug_blur         GaussianBlurTransform: separable Gaussian, radius int(4 sigma + 0.5), reflect borders (scipy gaussian_filter)
//   du_aug_lowres       SimulateLowResolutionTransform: nearest-neighbour down to round(size * zoom), cubic back up
// Per-plane parameters are device arrays (one entry per (sample, channel)); a plane whose parameter says "off" is copied unchanged.
#include "common.h"

namespace {

__device__ __forceinline__ float keys(float t) {          // Keys cubic convolution kernel, a = -0.5
  t = fabsf(t);
  if (t <= 1.f) return (1.5f * t - 2.5f) * t * t + 1.f;
  if (t < 2.f) return ((-0.5f * t + 2.5f) * t - 4.f) * t + 2.f;
  return 0.f;
}

// bicubic sample of plane p (H x W) at (y, x); taps outside the image contribute `cval`
__device__ __forceinline__ float sample_cubic(const float* __restrict__ p, int H, int W, float y, float x, float cval) {
  const int y0 = (int)floorf(y), x0 = (int)floorf(x);
  float acc = 0.f;
#pragma unroll
  for (int j = -1; j <= 2; j++) {
    const int yy = y0 + j;
    const float wy = keys(y - (float)yy);
#pragma unroll
    for (int i = -1; i <= 2; i++) {
      const int xx = x0 + i;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? p[(long)yy * W + xx] : cval;
      acc += wy * keys(x - (float)xx) * v;
    }
  }
  return acc;
}

// params per sample: [m00, m01, m10, m11, flip_y, flip_x]: input coordinate = centre_in + M * (output coordinate - centre_out)
__global__ __launch_bounds__(256) void aug_spatial_kernel(const float* __restrict__ din, const float* __restrict__ sin_, const float* __restrict__ prm,
                                                          float* __restrict__ dout, float* __restrict__ sout, int B, int C, int Hi, int Wi,
                                                          int Ho, int Wo) {
  const long total = (long)B * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    long t = i / Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float* q = prm + b * 6;
    const int ys = q[4] != 0.f ? Ho - 1 - yo : yo, xs = q[5] != 0.f ? Wo - 1 - xo : xo;      // MirrorTransform flips the output arrays
    const float cy = (float)ys - 0.5f * (Ho - 1), cx = (float)xs - 0.5f * (Wo - 1);
    const float y = q[0] * cy + q[1] * cx + (0.5f * Hi - 0.5f);
    const float x = q[2] * cy + q[3] * cx + (0.5f * Wi - 0.5f);
    for (int c = 0; c < C; c++)
      dout[(((long)b * C + c) * Ho + yo) * Wo + xo] = sample_cubic(din + ((long)b * C + c) * Hi * Wi, Hi, Wi, y, x, 0.f);
    if (sin_) {
      const float* sp = sin_ + (long)b * Hi * Wi;
      const int y0 = (int)floorf(y), x0 = (int)floorf(x);
      const float fy = y - (float)y0, fx = x - (float)x0;
      float lab[4], w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
        lab[k] = (yy >= 0 && yy < Hi && xx >= 0 && xx < Wi) ? sp[(long)yy * Wi + xx] : -1.f;
      }
      float best = 0.f;        // result starts at 0; labels are visited in ascending order and overwrite
      float wout = 0.f;        // a neighbour outside the image reads cval = -1 in EVERY one-hot map: it counts against each label
#pragma unroll
      for (int j = 0; j < 4; j++) wout += lab[j] < 0.f ? w[j] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (lab[k] < 0.f) continue;
        float s = -wout;
#pragma unroll
        for (int j = 0; j < 4; j++) s += (lab[j] == lab[k]) ? w[j] : 0.f;
        if (s >= 0.5f && lab[k] > best) best = lab[k];
      }
      sout[((long)b * Ho + yo) * Wo + xo] = best;
    }
  }
}

// one workgroup per plane: out[p] = (mean, population std, min, max)
__global__ __launch_bounds__(1024) void aug_plane_stats_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
  __shared__ float red[4][16];
  const float* p = x + (long)blockIdx.x * n;
  float s = 0.f, s2 = 0.f, mn = 3.4e38f, mx = -3.4e38f;
  for (long i = threadIdx.x * 4L; i < n; i += 4096) {
    const float4 v = *(const float4*)(p + i);
    s += v.x + v.y + v.z + v.w;
    s2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  s = wave_sum(s); s2 = wave_sum(s2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s; red[1][w] = s2; red[2][w] = mn; red[3][w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 3.4e38f, d = -3.4e38f;
    for (int k = 0; k < 16; k++) { a += red[0][k]; b += red[1][k]; c = fminf(c, red[2][k]); d = fmaxf(d, red[3][k]); }
    const float mean = a / (float)n;
    float var = b / (float)n - mean * mean;
    out[blockIdx.x * 4 + 0] = mean;
    out[blockIdx.x * 4 + 1] = sqrtf(fmaxf(var, 0.f));
    out[blockIdx.x * 4 + 2] = c;
    out[blockIdx.x * 4 + 3] = d;
  }
}

__device__ __forceinline__ unsigned hash32(unsigned x) {     // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// standard normal from the counter (plane, element, seed): Box-Muller on two hashed uniforms
__device__ __forceinline__ float normal_at(unsigned seed, unsigned plane, unsigned idx) {
  const unsigned h1 = hash32(idx * 2654435761U ^ hash32(plane + 0x9e3779b9U * seed));
  const unsigned h2 = hash32(h1 ^ 0x85ebca6bU);
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

__global__ __launch_bounds__(256) void aug_noise_mult_kernel(float* __restrict__ x, const float* __restrict__ sigma, const float* __restrict__ mult,
                                                             long n, unsigned seed) {
  const int p = blockIdx.y;
  const float sg = sigma[p], m = mult[p];
  if (sg == 0.f && m == 1.f) return;
  float* q = x + (long)p * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = q[i];
    if (sg != 0.f) v += sg * normal_at(seed, (unsigned)p, (unsigned)i);
    q[i] = v * m;
  }
}

// stats (P, 4) of the CURRENT contents; factor[p] == 1 leaves the plane alone
__global__ __launch_bounds__(256) void aug_contrast_kernel(float* __restrict__ x, const float* __restrict__ factor, const float* __restrict__ st, long n) {
  const int p = blockIdx.y;
  const float f = factor[p];
  if (f == 1.f) return;
  const float mean = st[p * 4], mn = st[p * 4 + 2], mx = st[p * 4 + 3];
  float* q = x + (long)p * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    q[i] = fminf(fmaxf((q[i] - mean) * f + mean, mn), mx);
}

// gamma[p] <= 0: plane untouched.  invert[p] != 0: the transform acts on -x (stats must be those of -x: min' = -max, max' = -min)
__global__ __launch_bounds__(256) void aug_gamma_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ invert,
                                                        const float* __restrict__ st, long n) {
  const int p = blockIdx.y;
  const float g = gamma[p];
  if (g <= 0.f) return;
  const bool inv = invert[p] != 0.f;
  const float mn = inv ? -st[p * 4 + 3] : st[p * 4 + 2], mx = inv ? -st[p * 4 + 2] : st[p * 4 + 3];
  const float rng = mx - mn;
  float* q = x + (long)p * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = inv ? -q[i] : q[i];
    q[i] = __powf(fmaxf((v - mn) / (rng + 1e-7f), 0.f), g) * rng + mn;
  }
}

__global__ __launch_bounds__(256) void aug_affine_kernel(float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b, long n) {
  const int p = blockIdx.y;
  const float aa = a[p], bb = b[p];
  if (aa == 1.f && bb == 0.f) return;
  float* q = x + (long)p * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) q[i] = q[i] * aa + bb;
}

__device__ __forceinline__ int reflect(int i, int n) {      // scipy "reflect": (d c b a | a b c d | d c b a)
  while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
  return i;
}
// one separable pass along `axis` (0: rows / y, 1: columns / x); sigma[p] <= 0 copies the plane
__global__ __launch_bounds__(256) void aug_blur_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ sigma, int H, int W,
                                                       int axis) {
  const int p = blockIdx.y;
  const float sg = sigma[p];
  const long n = (long)H * W;
  const float* q = x + (long)p * n;
  float* o = y + (long)p * n;
  const int r = sg > 0.f ? (int)(4.0f * sg + 0.5f) : 0;
  float wsum = 0.f;
  for (int k = -r; k <= r; k++) wsum += __expf(-0.5f * (float)(k * k) / (sg * sg + 1e-30f));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (r == 0) { o[i] = q[i]; continue; }
    const int yy = (int)(i / W), xx = (int)(i % W);
    float acc = 0.f;
    for (int k = -r; k <= r; k++) {
      const float wk = __expf(-0.5f * (float)(k * k) / (sg * sg));
      const int ys = axis == 0 ? reflect(yy + k, H) : yy, xs = axis == 1 ? reflect(xx + k, W) : xx;
      acc += wk * q[(long)ys * W + xs];
    }
    o[i] = acc / wsum;
  }
}

// zoom[p] >= 1 or <= 0: copy.  Low-resolution grid (Hl, Wl) = round(size * zoom); its samples are the nearest input pixels
// (skimage.transform.resize order 0: source index = floor((i + 0.5) * H / Hl)); the output is the cubic interpolation of that grid
// evaluated at ((y + 0.5) * Hl / H - 0.5), edge samples replicated (resize mode "edge")
__global__ __launch_bounds__(256) void aug_lowres_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ zoom, int H, int W) {
  const int p = blockIdx.y;
  const float z = zoom[p];
  const long n = (long)H * W;
  const float* q = x + (long)p * n;
  float* o = y + (long)p * n;
  const bool off = !(z > 0.f && z < 1.f);
  const int Hl = max((int)rintf(H * z), 1), Wl = max((int)rintf(W * z), 1);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (off) { o[i] = q[i]; continue; }
    const int yy = (int)(i / W), xx = (int)(i % W);
    const float fy = ((float)yy + 0.5f) * (float)Hl / (float)H - 0.5f, fx = ((float)xx + 0.5f) * (float)Wl / (float)W - 0.5f;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    float acc = 0.f;
#pragma unroll
    for (int j = -1; j <= 2; j++) {
      const int yl = min(max(y0 + j, 0), Hl - 1);
      const int ys = min((int)(((float)yl + 0.5f) * (float)H / (float)Hl), H - 1);
      const float wy = keys(fy - (float)(y0 + j));
#pragma unroll
      for (int k = -1; k <= 2; k++) {
        const int xl = min(max(x0 + k, 0), Wl - 1);
        const int xs = min((int)(((float)xl + 0.5f) * (float)W / (float)Wl), W - 1);
        acc += wy * keys(fx - (float)(x0 + k)) * q[(long)ys * W + xs];
      }
    }
    o[i] = acc;
  }
}

inline dim3 plane_grid(long n, int planes) {
  long g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  return dim3((unsigned)g, (unsigned)planes);
}

}  // namespace

extern "C" int du_aug_spatial(const float* data, const float* seg, const float* params, float* data_out, float* seg_out, int B, int C, int Hi,
                              int Wi, int Ho, int Wo, void* stream) {
  if (!data || !params || !data_out || (seg && !seg_out) || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return DU_ERR_BAD_ARG;
  const long total = (long)B * Ho * Wo;
  long g = (total + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(aug_spatial_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, data, seg, params, data_out, seg_out, B, C, Hi, Wi, Ho, Wo);
  return du_check_launch();
}

extern "C" int du_aug_plane_stats(const float* x, float* stats, int planes, int64_t n, void* stream) {
  if (!x || !stats || planes <= 0 || n <= 0 || n % 4) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_plane_stats_kernel, dim3(planes), dim3(1024), 0, (hipStream_t)stream, x, stats, (long)n);
  return du_check_launch();
}

extern "C" int du_aug_noise_mult(float* x, const float* sigma, const float* mult, int planes, int64_t n, int seed, void* stream) {
  if (!x || !sigma || !mult || planes <= 0 || n <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_noise_mult_kernel, plane_grid(n, planes), dim3(256), 0, (hipStream_t)stream, x, sigma, mult, (long)n, (unsigned)seed);
  return du_check_launch();
}

extern "C" int du_aug_contrast(float* x, const float* factor, const float* stats, int planes, int64_t n, void* stream) {
  if (!x || !factor || !stats || planes <= 0 || n <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_contrast_kernel, plane_grid(n, planes), dim3(256), 0, (hipStream_t)stream, x, factor, stats, (long)n);
  return du_check_launch();
}

extern "C" int du_aug_gamma(float* x, const float* gamma, const float* invert, const float* stats, int planes, int64_t n, void* stream) {
  if (!x || !gamma || !invert || !stats || planes <= 0 || n <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_gamma_kernel, plane_grid(n, planes), dim3(256), 0, (hipStream_t)stream, x, gamma, invert, stats, (long)n);
  return du_check_launch();
}

extern "C" int du_aug_affine(float* x, const float* a, const float* b, int planes, int64_t n, void* stream) {
  if (!x || !a || !b || planes <= 0 || n <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_affine_kernel, plane_grid(n, planes), dim3(256), 0, (hipStream_t)stream, x, a, b, (long)n);
  return du_check_launch();
}

extern "C" int du_aug_blur(const float* x, float* y, const float* sigma, int planes, int H, int W, int axis, void* stream) {
  if (!x || !y || x == y || !sigma || planes <= 0 || H <= 0 || W <= 0 || (axis != 0 && axis != 1)) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_blur_kernel, plane_grid((long)H * W, planes), dim3(256), 0, (hipStream_t)stream, x, y, sigma, H, W, axis);
  return du_check_launch();
}

extern "C" int du_aug_lowres(const float* x, float* y, const float* zoom, int planes, int H, int W, void* stream) {
  if (!x || !y || x == y || !zoom || planes <= 0 || H <= 0 || W <= 0) return DU_ERR_BAD_ARG;
  hipLaunchKernelGGL(aug_lowres_kernel, plane_grid((long)H * W, planes), dim3(256), 0, (hipStream_t)stream, x, y, zoom, H, W);
  return du_check_launch();
}
