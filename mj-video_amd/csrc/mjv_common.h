// Shared device helpers for the gfx950 kernels of libmjv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "mjv.h"

typedef uint16_t u16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MJV_DEV __device__ __forceinline__

// bf16 <-> f32.  The plain cast lowers to v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving).
MJV_DEV float bf2f(u16 u) { return __uint_as_float(((unsigned)u) << 16); }
MJV_DEV u16 f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(u16, b);
}
// round an fp32 value through bf16 (a torch op boundary in the reference's bf16 path): ONE v_cvt_pk_bf16_f32 with the value
// in the HIGH half and zero in the low half - the packed word is the rounded value's fp32 bit pattern, no shift / mask
typedef __attribute__((ext_vector_type(2))) float mjv_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 mjv_bf16x2;
MJV_DEV float rbf(float f) {
  return __builtin_bit_cast(float, __builtin_convertvector(mjv_f32x2{0.f, f}, mjv_bf16x2));
}
MJV_DEV unsigned pack2bf(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// GELU of a bf16 value whose magnitude lies past the table (|x| >= 128), as torch's CPU bf16 kernel returns it (x * 0.5 * (1 + erf)
// in fp32, rounded): x for finite x > 0 until 2 x overflows fp32 (bf16 patterns 0x7f00..0x7f7f -> +inf), -0 for finite x < 0, NaN
// for +-inf (inf * 0), a NaN operand quieted with its payload kept.  u = the value's fp32 bits, mag = (u >> 16) & 0x7fff.
MJV_DEV unsigned gelu_beyond_table(unsigned u, unsigned mag) {
  const bool neg = (int)u < 0;
  const unsigned finite = neg ? 0x80000000u : (mag >= 0x7f00u ? 0x7f800000u : u);
  const unsigned nonfinite = (u & 0x7fffffffu) == 0x7f800000u ? (neg ? 0x7fc00000u : 0xffc00000u) : (u | 0x00400000u);
  return mag >= 0x7f80u ? nonfinite : finite;
}

MJV_DEV void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
MJV_DEV u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
  return v;
}

MJV_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
MJV_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side -------------------------------------------------------------------------------
void mjv_set_error(const char* fmt, ...);
int mjv_check_launch(const char* what);
int mjv_device_cus();   // multiProcessorCount of the current device (cached per device)

// MXFP8 operands (gemm_fp8.hip); mjv_gemm_bf16 hands descriptors that name them over
int mjv_gemm_mxfp8_dispatch(const mjv_gemm_desc* d, void* stream);

// profiler hooks (capi.cpp)
struct MjvProfScope {
  MjvProfScope(const char* tag, hipStream_t s, double flops, double bytes);
  ~MjvProfScope();
  int slot;
  hipStream_t stream;
};

#define MJV_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      mjv_set_error(__VA_ARGS__);         \
      return MJV_E_ARG;                   \
    }                                     \
  } while (0)
