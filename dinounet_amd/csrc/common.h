// Common device helpers for the gfx950 kernels (wave64, MFMA).  gfx950 only: no portability macros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dinounet_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DU_WAVE 64

// Debugging / A-B knobs read from the environment exist only in builds made with -DDU_DEBUG_KNOBS (DINOUNET_DEBUG_KNOBS=1 python -m
// dinounet_amd._build: the measurement tools' build).  The release library has none: every default below is what ships and no test or
// deployment can change kernel selection through the environment (tests/test_cpu_oracle_and_boundary.py checks the
// binary).  Run-time switching that the product itself needs goes through du_set_option.
#include <stdlib.h>
#ifdef DU_DEBUG_KNOBS
#define DU_GETENV(name) getenv(name)
#else
#define DU_GETENV(name) ((const char*)nullptr)
#endif

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int VEC = 4; static constexpr int DT = DU_F32; };
template <> struct Elem<bf16_t> { static constexpr int VEC = 8; static constexpr int DT = DU_BF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// 16-byte vector <-> element arrays
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};
template <typename T> __device__ __forceinline__ Vec16<T> as_vec(uint4 u) { return __builtin_bit_cast(Vec16<T>, u); }
template <typename T> __device__ __forceinline__ uint4 as_u4(Vec16<T> v) { return __builtin_bit_cast(uint4, v); }

// Wave-wide reductions, result in every lane.  Inside the 16-lane rows by DPP (v_*_dpp: one full-rate VALU op per step, no LDS), across
// the four rows by scalar reads -- the xor-shuffle tree (six dependent ds_bpermute round trips through the LDS pipe, ~100 cycles each)
// paced every kernel that reduces once per row or per step: the MSDA grad_value kernel ran 3x faster with its step mask reduced this way.
#define DU_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
  v += DU_DPP_F32(v, 0xB1);       // quad_perm [1,0,3,2]
  v += DU_DPP_F32(v, 0x4E);       // quad_perm [2,3,0,1]
  v += DU_DPP_F32(v, 0x141);      // row_half_mirror
  v += DU_DPP_F32(v, 0x140);      // row_mirror: every lane holds its row's sum
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, DU_DPP_F32(v, 0xB1));
  v = fmaxf(v, DU_DPP_F32(v, 0x4E));
  v = fmaxf(v, DU_DPP_F32(v, 0x141));
  v = fmaxf(v, DU_DPP_F32(v, 0x140));
  const int b = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// acc + the sum of the 8 bf16 of an MFMA operand fragment, in fp32: four v_dot2c_f32_bf16 against (1, 1) instead of eight converts + eight
// adds (the bias-gradient side sums of the weight-gradient kernels run in the MFMA loop's VALU shadow: instruction count is what they cost)
__device__ __forceinline__ float frag_sum8(const bf16x8& f, float acc) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const bf16x2_t pr = {f[2 * e], f[2 * e + 1]};
    acc = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, acc, false);
  }
  return acc;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case DU_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case DU_ACT_RELU: return v > 0.f ? v : 0.f;
    case DU_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}
// d act(z) / dz evaluated at pre-activation z
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case DU_ACT_GELU: {
      const float c = 0.70710678118654752440f, ip = 0.39894228040143267794f;  // 1/sqrt(2), 1/sqrt(2 pi)
      return 0.5f * (1.0f + erff(z * c)) + z * ip * __expf(-0.5f * z * z);
    }
    case DU_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case DU_ACT_LEAKY: return z > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}

static inline int du_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DU_OK : DU_ERR_LAUNCH;
}
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
