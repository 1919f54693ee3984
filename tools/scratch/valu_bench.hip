// micro-benchmark: cycles per instruction of a few VALU ops on one SIMD, by waves per SIMD (1, 2, 4), with and without MFMAs beside them
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP 64
template <int OP, bool MF>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  f32x16 acc; for (int i = 0; i < 16; i++) acc[i] = 0.f;
  bf16x8 fa, fb; for (int i = 0; i < 8; i++) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 16; r++) {
      if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        if (OP == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(a[(i + 2) & 15])); }
        if (OP == 1) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[i + 1])); }
        if (OP == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(f32x2*)&a[i]) : "v"(*(f32x2*)&a[(i + 2) & 15])); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(f32x2*)&a[i]) : "v"(*(f32x2*)&a[(i + 4) & 15])); }
        if (OP == 3) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15])); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[i + 1]) : "v"(a[(i + 3) & 15]), "v"(a[(i + 2) & 15])); }
        if (OP == 4) { asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15])); asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i + 1]) : "v"(a[(i + 3) & 15]), "v"(a[(i + 2) & 15])); }
        if (OP == 5) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15])); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i + 1]) : "v"(a[(i + 2) & 15])); }
        if (OP == 6) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15])); asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i + 1]) : "v"(a[(i + 3) & 15]), "v"(a[(i + 2) & 15])); }
        if (OP == 7) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(f32x2*)&a[i]) : "v"(*(f32x2*)&a[(i + 2) & 15])); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(f32x2*)&a[i]) : "v"(*(f32x2*)&a[(i + 4) & 15])); }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; i++) s += a[i]; for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, bool MF> void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 200;
  for (int wgs = 1; wgs <= 4; wgs *= 2) {          // workgroups of 4 waves per CU -> waves per SIMD
    hipLaunchKernelGGL((k<OP, MF>), dim3(256 * wgs), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h[0]) * 256 * wgs, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256 * wgs; i++) m += h[i]; m /= 256 * wgs;
    const double n = (double)iters * REP;            // VALU instructions per wave (+ iters * REP / 16 MFMAs when MF)
    printf("%-34s %d wave/SIMD: %7.2f cycles per VALU instruction per wave; SIMD-level %6.2f cycles per VALU instruction%s\n", name, wgs, m / n, m / n / wgs,
           MF ? " (1 MFMA per 16 VALU: 32 cycles of matrix pipe each)" : "");
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  run<0, false>("v_add_f32", out, cyc); run<5, false>("v_fma_f32", out, cyc); run<1, false>("v_exp_f32", out, cyc); run<2, false>("v_pk_add_f32", out, cyc);
  run<7, false>("v_pk_mul/fma_f32", out, cyc); run<3, false>("v_cvt_pk_bf16_f32", out, cyc); run<4, false>("v_dot2c_f32_bf16", out, cyc); run<6, false>("v_max3_f32", out, cyc);
  run<0, true>("v_add_f32 + MFMA", out, cyc); run<1, true>("v_exp_f32 + MFMA", out, cyc); run<2, true>("v_pk_add_f32 + MFMA", out, cyc); run<4, true>("v_dot2c + MFMA", out, cyc);
  return 0;
}
