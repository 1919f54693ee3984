// Do back-to-back MFMAs of one wave block the VALU issue of ANOTHER wave on the same SIMD?  Workgroups [0, 256) run role X, [256, 512) role Y
// (b and b + 256 share a CU: tools/scratch/census.py).  Roles: 0 idle (exit at once), 1 back-to-back MFMAs (4 accumulators), 2 v_add stream,
// 3 v_exp stream, 4 MFMA + 5 VALU interleaved, 5 MFMAs with an s_nop 7 between them (one per ~64 cycles)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void role(int r, int iters, float* out, unsigned long long* cyc) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  f32x16 acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  bf16x8 fa, fb; for (int i = 0; i < 8; i++) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (r == 1) for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
  }
  if (r == 2) for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(a[(i + 1) & 15]));
  }
  if (r == 3) for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 15]));
  }
  if (r == 4) for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %4, %0, %1" : "+v"(a[(2 * j) & 15]), "+v"(a[(2 * j + 1) & 15]), "+v"(a[(2 * j + 8) & 15]), "+v"(a[(2 * j + 9) & 15]), "=v"(a[(2 * j + 4) & 15]));
    }
  }
  if (r == 6) for (int it = 0; it < iters; it++) {      // exps of this group, adds + cvt on the PREVIOUS group's exps
#pragma unroll
    for (int j = 0; j < 16; j++) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %2, %2, %5\n\tv_exp_f32 %1, %1\n\tv_add_f32 %3, %3, %6\n\tv_cvt_pk_bf16_f32 %4, %5, %6" : "+v"(a[(2 * j) & 7]), "+v"(a[(2 * j + 1) & 7]), "+v"(a[8]), "+v"(a[9]), "=v"(a[10 + (j & 3)]) : "v"(a[(2 * j + 6) & 7]), "v"(a[(2 * j + 7) & 7]));
    }
  }
  if (r == 7) for (int it = 0; it < iters; it++) {      // the same with plain VALU ops only (no transcendentals): 5 v_add per MFMA
#pragma unroll
    for (int j = 0; j < 16; j++) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
      asm volatile("v_add_f32 %0, %0, %5\n\tv_add_f32 %2, %2, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %5, %6" : "+v"(a[(2 * j) & 7]), "+v"(a[(2 * j + 1) & 7]), "+v"(a[8]), "+v"(a[9]), "=v"(a[10 + (j & 3)]) : "v"(a[(2 * j + 6) & 7]), "v"(a[(2 * j + 7) & 7]));
    }
  }
  if (r == 8) for (int it = 0; it < iters; it++) {      // 2 exps per MFMA only
#pragma unroll
    for (int j = 0; j < 16; j++) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(a[(2 * j) & 7]), "+v"(a[(2 * j + 1) & 7]));
    }
  }
  if (r == 9) for (int it = 0; it < iters; it++) {      // 3 plain VALU per MFMA
#pragma unroll
    for (int j = 0; j < 16; j++) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
      asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %3, %4" : "+v"(a[8]), "+v"(a[9]), "=v"(a[10 + (j & 3)]) : "v"(a[(2 * j + 6) & 7]), "v"(a[(2 * j + 7) & 7]));
    }
  }
  if (r == 5) for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) { acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0); asm volatile("s_nop 7"); }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; i++) s += a[i]; for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k(int rx, int ry, int ix, int iy, float* out, unsigned long long* cyc) {
  if (blockIdx.x < 256) role(rx, ix, out, cyc); else role(ry, iy, out, cyc);
}
// all workgroups the same role; per-workgroup [begin, end] ticks so the co-resident span can be read
__global__ __launch_bounds__(256) void k2(int r, int iters, float* out, unsigned long long* cyc) { role(r, iters, out, cyc); }
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const char* nm[] = {"idle", "MFMA back-to-back", "v_add stream", "v_exp stream", "MFMA + 5 VALU interleaved", "MFMA every ~64 cycles", "MFMA + 2exp 2add cvt (piped)", "MFMA + 5 v_add", "MFMA + 2 exp", "MFMA + 3 v_add"};
  const int work[] = {0, 16, 64, 64, 16, 16, 16, 16, 16, 16};     // units (MFMAs or VALU ops) per iteration
  int pairs[][2] = {{6, 0}};
  for (auto& p : pairs) {
    const int itx = 4000, ity = p[1] ? 4000 : 0;
    hipMemset(cyc, 0, 512 * 8);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, p[0], p[1], itx, ity, out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[512]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0, my = 0; for (int i = 0; i < 256; i++) { mx += h[i]; my += h[256 + i]; } mx /= 256; my /= 256;
    printf("X = %-26s Y = %-26s : X %6.1f cycles per %s", nm[p[0]], nm[p[1]], mx / ((double)itx * work[p[0]]), p[0] == 2 || p[0] == 3 ? "VALU op" : "MFMA");
    if (p[1]) printf("   Y %6.1f cycles per %s", my / ((double)ity * work[p[1]]), p[1] == 2 || p[1] == 3 ? "VALU op" : "MFMA");
    printf("\n");
  }
  for (int r : {1, 6, 7, 8, 9}) for (int wgs : {1, 2, 3, 4}) {
    const int iters = 4000;
    hipLaunchKernelGGL(k2, dim3(256 * wgs), dim3(256), 0, 0, r, iters, out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h[0]) * 256 * wgs, hipMemcpyDeviceToHost);
    double mx = 0; for (int i = 0; i < 256 * wgs; i++) mx = h[i] > mx ? h[i] : mx;      // the slowest (youngest) wave = the span of the co-resident set
    printf("%-30s %d waves / SIMD: span %.1f cycles per MFMA group and wave -> SIMD-level %.1f cycles per MFMA\n", nm[r], wgs, mx / (iters * 16.0), mx / (iters * 16.0) / wgs);
  }
  return 0;
}
