// One asm block per loop body: 16 MFMAs (4 accumulators round robin) with N fillers after each, in exactly this order.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define M(i) "v_mfma_f32_32x32x16_bf16 %" #i ", %4, %5, %" #i "\n\t"
#define A(d, s) "v_add_f32 %" #d ", %" #d ", %" #s "\n\t"
#define E(d) "v_exp_f32 %" #d ", %" #d "\n\t"
#define C(d, s, t) "v_cvt_pk_bf16_f32 %" #d ", %" #s ", %" #t "\n\t"
// fillers use operands 6..21 (16 VGPRs)
#define F0
#define F3a A(6, 14) A(7, 15) A(8, 16)
#define F3b A(9, 17) A(10, 18) A(11, 19)
#define F5a A(6, 14) A(7, 15) A(8, 16) A(9, 17) A(10, 18)
#define F5b A(11, 19) A(12, 20) A(13, 21) A(6, 15) A(7, 16)
#define X5a E(6) E(7) A(14, 10) A(15, 11) C(20, 10, 11)
#define X5b E(8) E(9) A(16, 12) A(17, 13) C(21, 12, 13)
#define X5c E(10) E(11) A(14, 6) A(15, 7) C(20, 6, 7)
#define X5d E(12) E(13) A(16, 8) A(17, 9) C(21, 8, 9)
#define X2a E(6) E(7)
#define X2b E(8) E(9)
#define BODY(fa, fb, fc, fd) M(0) fa M(1) fb M(2) fc M(3) fd M(0) fa M(1) fb M(2) fc M(3) fd M(0) fa M(1) fb M(2) fc M(3) fd M(0) fa M(1) fb M(2) fc M(3) fd
template <int V, bool AG>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-4f + i * 1e-3f;
  f32x16 acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  bf16x8 fa, fb; for (int i = 0; i < 8; i++) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#define OPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(fa), "v"(fb), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15])
#define OPSA : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(fa), "v"(fb), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15])
    if constexpr (AG) {
      if constexpr (V == 0) asm volatile(BODY(F0, F0, F0, F0) OPSA);
      if constexpr (V == 1) asm volatile(BODY(F3a, F3b, F3a, F3b) OPSA);
      if constexpr (V == 2) asm volatile(BODY(F5a, F5b, F5a, F5b) OPSA);
      if constexpr (V == 3) asm volatile(BODY(X5a, X5b, X5c, X5d) OPSA);
      if constexpr (V == 4) asm volatile(BODY(X2a, X2b, X2a, X2b) OPSA);
    } else {
      if constexpr (V == 0) asm volatile(BODY(F0, F0, F0, F0) OPS);
      if constexpr (V == 1) asm volatile(BODY(F3a, F3b, F3a, F3b) OPS);
      if constexpr (V == 2) asm volatile(BODY(F5a, F5b, F5a, F5b) OPS);
      if constexpr (V == 3) asm volatile(BODY(X5a, X5b, X5c, X5d) OPS);
      if constexpr (V == 4) asm volatile(BODY(X2a, X2b, X2a, X2b) OPS);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; i++) s += a[i]; for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V, bool AG> void run(const char* name, float* out, unsigned long long* cyc) {
  for (int wgs = 1; wgs <= 3; wgs++) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<V, AG>), dim3(256 * wgs), dim3(256), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h[0]) * 256 * wgs, hipMemcpyDeviceToHost);
    double mx = 0; for (int i = 0; i < 256 * wgs; i++) mx = h[i] > mx ? h[i] : mx;
    printf("%-40s %s %d waves / SIMD: %.1f cycles per MFMA per wave -> SIMD-level %.1f cycles per MFMA\n", name, AG ? "acc in AGPR" : "acc in VGPR", wgs, mx / (iters * 16.0), mx / (iters * 16.0) / wgs);
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  run<0, false>("MFMA only", out, cyc); run<1, false>("MFMA + 3 v_add", out, cyc); run<2, false>("MFMA + 5 v_add", out, cyc); run<3, false>("MFMA + 2 exp 2 add 1 cvt", out, cyc); run<4, false>("MFMA + 2 exp", out, cyc);
  run<0, true>("MFMA only", out, cyc); run<2, true>("MFMA + 5 v_add", out, cyc); run<3, true>("MFMA + 2 exp 2 add 1 cvt", out, cyc);
  return 0;
}
