// MXFP8 (OCP e4m3 elements, e8m0 scale per 32-element block) device helpers shared by the quantiser, the norm kernels and
// the fp8 GEMM's epilogue.  Format and scale layout: include/mjv.h "MXFP8 operand format"; pinned by oracle/ref_fp8.py.
#pragma once
#include "mjv_common.h"

typedef __attribute__((ext_vector_type(2))) short mx8_s16x2;
typedef __attribute__((ext_vector_type(8))) int mx8_i32x8;

// largest bf16 magnitude (15-bit pattern: ordering of the patterns = ordering of the magnitudes) of 8 packed values
MJV_DEV unsigned mx8_amax8(const u32x4& v) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = v[i] & 0x7fff7fffu;
    m = max(m, max(a & 0xffffu, a >> 16));
  }
  return m;
}
// e8m0 byte of a block whose largest magnitude has the bf16 pattern u (E:8 | m:7): E - 8, one more when 1.m > 1.75 (then
// amax / 2^(E-8) would exceed 448 = 1.75 * 2^8), at least 1 (the scale stays a NORMAL fp32 for the conversion instruction)
MJV_DEV unsigned mx8_scale_byte(unsigned u) {
  const int t = (int)((u + 0x1fu) >> 7) - 8;
  return (unsigned)(t < 1 ? 1 : t);
}
MJV_DEV float mx8_scale_f32(unsigned b) { return __uint_as_float(b << 23); }   // 2^(b - 127), b >= 1
// 8 packed bf16 -> 8 e4m3 bytes, each x / scale rounded to nearest even (v_cvt_scalef32_pk_fp8_bf16 divides by its scale
// operand: tools/micro/mfma_scale_probe.hip, profiles/r04_a_mfma_scale_probe_v1.txt)
MJV_DEV u32x2 mx8_cvt8(const u32x4& v, float scale) {
  // (the elements go through named scalars: __builtin_bit_cast applied to the subscript expression v[i] itself made hipcc
  // (ROCm 7.2) convert v[0] four times)
  const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
  mx8_s16x2 lo = {0, 0}, hi = {0, 0};
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(mjv_bf16x2, e0), scale, false);
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(mjv_bf16x2, e1), scale, true);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(mjv_bf16x2, e2), scale, false);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(mjv_bf16x2, e3), scale, true);
  return u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
// byte offset of the scale of (row, 32-element block starting at column col) inside a scale buffer with `groups` 64-row groups
MJV_DEV long mx8_scale_offset(long row, int col, long groups) {
  const long kt = col >> 7;
  const int kb = (col >> 5) & 3;
  return ((kt * groups + (row >> 6)) << 8) + ((row & 15) << 4) + (kb << 2) + ((row >> 4) & 3);
}
// One lane holds 8 consecutive columns of a row (16 bytes of bf16), the four lanes 4q .. 4q+3 of a quad one 32-element block:
// quantises the block, returns the lane's 8 bytes and the block's scale byte (same in the four lanes)
MJV_DEV u32x2 mx8_quantize_quad(const u32x4& v, unsigned& scale_byte) {
  unsigned m = mx8_amax8(v);
  // quad exchange by DPP (quad_perm [1,0,3,2] then [2,3,0,1]): no LDS round trip
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));
  scale_byte = mx8_scale_byte(m);
  return mx8_cvt8(v, mx8_scale_f32(scale_byte));
}
