// Shared device helpers and host-side error plumbing for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lx2v_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LX2V_DEVICE __device__ __forceinline__

// One 16-byte chunk = 8 bf16 values.
LX2V_DEVICE void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}

LX2V_DEVICE uint32_t pack2(float lo, float hi) {
  bf16x2 t;
  t[0] = (__bf16)lo;  // round-to-nearest-even, NaN preserved (v_cvt_pk_bf16_f32)
  t[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, t);
}

LX2V_DEVICE u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

LX2V_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

LX2V_DEVICE float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) written as x * sigmoid(2u); evaluated as written
  // (no re-association under -ffast-math), so that every kernel with this epilogue rounds alike
#pragma clang fp reassociate(off)
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x / (1.0f + __expf(-2.0f * u));
}

// ---- host side -------------------------------------------------------------------------------
void lx2v_set_error(const char* fmt, ...);

// Kernel-selection switches (A/B and test knobs).  Read from the environment once, on first use;
// lx2v_set_option() overrides them afterwards.  Launchers read this struct, never the environment.
struct Lx2vOptions {
  int attn_kernel;      // LX2V_ATTN_KERNEL=x16|w4|w8 -> 16 (default), 4, 8
  int attn_qb;          // LX2V_ATTN_QB=4|5 fixes the query blocks per wave of the x16 kernel; 0 = by cost
  int attn_tail;        // LX2V_ATTN_TAIL=0: no second launch for a sparse last round
  int attn_round_sync;  // LX2V_ATTN_ROUND_SYNC=0: no round alignment (a locality hint: same results)
  int attn_variant;     // LX2V_ATTN_VARIANT=0: 8-wave kernel without pipelined K-fragment reads
  int attn_fp8_kernel;  // LX2V_ATTN_FP8_KERNEL=w4|w8 -> 4 (default: one wave per SIMD), 8
  int attn_fp8_p;       // LX2V_ATTN_FP8_P=linear -> 1: the one-wave-per-SIMD fp8 kernel takes P = 2^x with a linear mantissa, without exps (faster, see attn_fp8_w4.hip); 0 = exact exps (default)
  int gemm_kernel;      // LX2V_GEMM_KERNEL=simple|pingpong -> 1 | 2; 0 = each GEMM's default (bf16: persistent one-wave-per-SIMD kernel, fp8: simple)
};
const Lx2vOptions& lx2v_options();

#define LX2V_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      lx2v_set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)

#define LX2V_CHECK_LAUNCH(what)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      lx2v_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return LX2V_EHIP;                                                           \
    }                                                                             \
  } while (0)

static inline bool lx2v_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
