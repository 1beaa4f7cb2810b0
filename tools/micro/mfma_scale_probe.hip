// Probe of the gfx950 block-scaled fp8 MFMA and the fp8 conversions, run once on the GPU box before the MX-fp8 GEMM was
// written (tools only; hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_scale_probe.hip -o tools/micro/mfma_scale_probe).
//  1. operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: hypothesis H1 = lane l holds row / column
//     l & 15 and the 32 CONSECUTIVE k of block l >> 4 (byte j of its 8 dwords = k 32 (l >> 4) + j), its scale operand is the
//     e8m0 of that (row, block); checked with integer data and random per-(row, block) scales against a host sum;
//     op_sel picks the byte of the scale register.
//  2. v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_f32 / v_cvt_scalef32_pk_fp8_bf16 against a host round-to-nearest-even e4m3fn
//     conversion over every bf16 bit pattern (which way the scale goes, what happens beyond 448, NaN).
//  3. issue rate: cycles per MFMA of the scaled K = 128 form, the plain fp8 K = 32 form and the bf16 K = 32 form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(2))) short v2s;
typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- host e4m3fn
static float e4m3_to_f(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static uint8_t f_to_e4m3_sat(float f) {   // RNE, saturating to +-448, NaN -> 0x7f
  if (f != f) return 0x7f;
  const uint8_t s = signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 464.0f) return s | 0x7e;   // (448 + 480) / 2 = 464 rounds to even = 480 -> overflow; saturate
  // search nearest (256 candidates is fine for a probe)
  int best = 0; float bd = 1e30f;
  for (int c = 0; c < 0x7f; ++c) {
    const float v = e4m3_to_f((uint8_t)c);
    const float d = fabsf(v - a);
    if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
  }
  return s | (uint8_t)best;
}

// ---- 1. layout
template <int OPSEL>
__global__ void mfma_once(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, OPSEL, sa[threadIdx.x], OPSEL,
                                                         sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}

// ---- 2. conversions: in = fp32 values; out[0] = cvt_pk_fp8_f32 byte, out[1] = scalef32 (scale 4.0) byte, out[2] = scalef32 bf16 byte
__global__ void cvt_probe(const float* in, int n, uint8_t* o0, uint8_t* o1, uint8_t* o2, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  const int r = __builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false);
  o0[i] = (uint8_t)(r & 0xff);
  const v2s r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32((v2s){0, 0}, x, 0.f, scale, false);
  o1[i] = (uint8_t)(r1[0] & 0xff);
  const v2bf bb = {(__bf16)x, (__bf16)0.f};
  const v2s r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16((v2s){0, 0}, bb, scale, false);
  o2[i] = (uint8_t)(r2[0] & 0xff);
}

// ---- 3. rates
template <int KIND>
__global__ void rate(unsigned long long* out, const v8i* src, int iters) {
  v8i a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else if constexpr (KIND == 1) {
        const long la = ((long)a[1] << 32) | (unsigned)a[0], lb = ((long)b[1] << 32) | (unsigned)b[0];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, acc[i], 0, 0, 0);
      } else {
        v8bf ab = __builtin_bit_cast(v8bf, __builtin_shufflevector(a, a, 0, 1, 2, 3));
        v8bf bb = __builtin_bit_cast(v8bf, __builtin_shufflevector(b, b, 0, 1, 2, 3));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 12345.678f) out[0] = 0;
}

int main() {
  // ------------------------------------------------------------------ 1
  std::vector<float> A(16 * 128), B(128 * 16);
  std::vector<uint8_t> A8(16 * 128), B8(128 * 16);
  std::vector<int> SA(16 * 4), SB(16 * 4);
  srand(7);
  for (int r = 0; r < 16; ++r)
    for (int k = 0; k < 128; ++k) {
      const int va = rand() % 17 - 8, vb = rand() % 17 - 8;
      A[r * 128 + k] = (float)va; A8[r * 128 + k] = f_to_e4m3_sat((float)va);
      B[k * 16 + r] = (float)vb; B8[k * 16 + r] = f_to_e4m3_sat((float)vb);
    }
  for (int i = 0; i < 64; ++i) { SA[i] = 127 + rand() % 7 - 3; SB[i] = 127 + rand() % 7 - 3; }
  std::vector<double> ref(256, 0.0);
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n)
      for (int k = 0; k < 128; ++k)
        ref[m * 16 + n] += (double)A[m * 128 + k] * ldexp(1.0, SA[m * 4 + k / 32] - 127) * (double)B[k * 16 + n] * ldexp(1.0, SB[n * 4 + k / 32] - 127);
  v8i ha[64], hb[64];
  int hsa[64], hsb[64];
  v8i *da, *db; int *dsa, *dsb; v4f* dc;
  CK(hipMalloc(&da, sizeof(ha))); CK(hipMalloc(&db, sizeof(hb))); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16));
  for (int hyp = 0; hyp < 2; ++hyp) {
    for (int opsel = 0; opsel < 4; ++opsel) {
      for (int l = 0; l < 64; ++l) {
        uint8_t ba[32], bb[32];
        const int rc = l & 15, g = l >> 4;
        for (int j = 0; j < 32; ++j) {
          const int k = hyp == 0 ? 32 * g + j : 8 * g + (j & 7) + 32 * (j >> 3);
          ba[j] = A8[rc * 128 + k];
          bb[j] = B8[k * 16 + rc];
        }
        memcpy(&ha[l], ba, 32); memcpy(&hb[l], bb, 32);
        hsa[l] = (SA[rc * 4 + g] << (8 * opsel)) | (opsel ? 0x55 : 0);
        hsb[l] = (SB[rc * 4 + g] << (8 * opsel)) | (opsel ? 0x33 : 0);
      }
      CK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
      CK(hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice));
      switch (opsel) {
        case 0: hipLaunchKernelGGL(mfma_once<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); break;
        case 1: hipLaunchKernelGGL(mfma_once<1>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); break;
        case 2: hipLaunchKernelGGL(mfma_once<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); break;
        default: hipLaunchKernelGGL(mfma_once<3>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); break;
      }
      CK(hipDeviceSynchronize());
      v4f hc[64];
      CK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
      // D layout: with the first operand = "A" (rows m), lane l: col n = l & 15, rows m = 4 (l >> 4) + r
      int bad_std = 0, bad_t = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int n = l & 15, m = 4 * (l >> 4) + r;
          if (fabs(hc[l][r] - ref[m * 16 + n]) > 1e-3 * (1 + fabs(ref[m * 16 + n]))) ++bad_std;
          if (fabs(hc[l][r] - ref[n * 16 + m]) > 1e-3 * (1 + fabs(ref[n * 16 + m]))) ++bad_t;
        }
      printf("layout: hypothesis %s, op_sel %d: mismatches with D[row = 4 (l >> 4) + r][col = l & 15]: %d / 256 (transposed: %d)\n",
             hyp == 0 ? "H1 (32 consecutive k per lane)" : "H2 (4 x 8 interleaved)", opsel, bad_std, bad_t);
    }
  }
  // scale semantics when the A-side scale differs across the 4 lanes groups only (sanity of e8m0: 127 = 1.0)
  // ------------------------------------------------------------------ 2
  {
    const int n = 65536;
    std::vector<float> in(n);
    for (int i = 0; i < n; ++i) { uint32_t u = (uint32_t)i << 16; memcpy(&in[i], &u, 4); }
    float* din; uint8_t *d0, *d1, *d2;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&d0, n)); CK(hipMalloc(&d1, n)); CK(hipMalloc(&d2, n));
    CK(hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice));
    for (float scale : {1.0f, 4.0f, 0.25f}) {
      hipLaunchKernelGGL(cvt_probe, dim3(n / 256), dim3(256), 0, 0, din, n, d0, d1, d2, scale);
      CK(hipDeviceSynchronize());
      std::vector<uint8_t> o0(n), o1(n), o2(n);
      CK(hipMemcpy(o0.data(), d0, n, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), d1, n, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), d2, n, hipMemcpyDeviceToHost));
      int bad0 = 0, bad1_div = 0, bad1_mul = 0, bad2_div = 0, bad2_mul = 0, shown = 0;
      for (int i = 0; i < n; ++i) {
        const float x = in[i];
        if (x != x) {
          if (scale == 1.0f && shown < 3) { printf("  NaN input %08x -> cvt_pk %02x scalef32 %02x bf16 %02x\n", (unsigned)i << 16, o0[i], o1[i], o2[i]); ++shown; }
          continue;
        }
        const uint8_t e0 = f_to_e4m3_sat(x), ediv = f_to_e4m3_sat(x / scale), emul = f_to_e4m3_sat(x * scale);
        // +-0: compare values
        auto same = [](uint8_t a, uint8_t b) { return a == b || ((a & 0x7f) == 0 && (b & 0x7f) == 0); };
        if (!same(o0[i], e0)) { if (bad0 < 4 && scale == 1.0f) printf("  cvt_pk_fp8_f32(%g) = %02x (host saturating RNE %02x)\n", x, o0[i], e0); ++bad0; }
        if (!same(o1[i], ediv)) ++bad1_div;
        if (!same(o1[i], emul)) ++bad1_mul;
        if (!same(o2[i], ediv)) ++bad2_div;
        if (!same(o2[i], emul)) ++bad2_mul;
      }
      printf("cvt: scale %g: cvt_pk_fp8_f32 vs host saturating RNE: %d differ; scalef32_pk_fp8_f32 differs from cvt(x / s): %d, from cvt(x * s): %d; "
             "scalef32_pk_fp8_bf16: %d / %d\n", scale, bad0, bad1_div, bad1_mul, bad2_div, bad2_mul);
    }
    // what a value beyond the format's range becomes
    for (float x : {448.f, 464.f, 465.f, 480.f, 1000.f, 1e30f, INFINITY, -1000.f}) {
      CK(hipMemcpy(din, &x, 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(1), 0, 0, din, 1, d0, d1, d2, 1.0f);
      CK(hipDeviceSynchronize());
      uint8_t a, b, c;
      CK(hipMemcpy(&a, d0, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, d1, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&c, d2, 1, hipMemcpyDeviceToHost));
      printf("cvt: x = %g -> cvt_pk %02x (%g)  scalef32_f32 %02x  scalef32_bf16 %02x\n", x, a, e4m3_to_f(a), b, c);
    }
  }
  // ------------------------------------------------------------------ 3
  {
    v8i hsrc[128];
    for (int i = 0; i < 128; ++i)
      for (int j = 0; j < 8; ++j) hsrc[i][j] = (rand() & 0x77777777) | 0x30303030;   // random finite e4m3 / bf16-ish patterns
    v8i* dsrc; unsigned long long* dout;
    CK(hipMalloc(&dsrc, sizeof(hsrc))); CK(hipMalloc(&dout, 8 * 1024));
    CK(hipMemcpy(dsrc, hsrc, sizeof(hsrc), hipMemcpyHostToDevice));
    const int iters = 2000;
    for (int waves = 1; waves <= 2; ++waves) {
      for (int kind = 0; kind < 3; ++kind) {
        // 256 or 512 threads = 1 or 2 waves per SIMD, 256 workgroups (every CU busy)
        for (int rep = 0; rep < 2; ++rep) {
          if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256 * waves), 0, 0, dout, dsrc, iters);
          else if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256 * waves), 0, 0, dout, dsrc, iters);
          else hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256 * waves), 0, 0, dout, dsrc, iters);
          CK(hipDeviceSynchronize());
        }
        unsigned long long h[256];
        CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        const char* nm[] = {"scale_f32_16x16x128_f8f6f4 (e4m3)", "f32_16x16x32_fp8_fp8", "f32_16x16x32_bf16"};
        const double flop[] = {2.0 * 16 * 16 * 128, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32};
        const double cyc = s / 256 / (iters * 8.0);
        printf("rate: %-36s %d wave(s)/SIMD: %.2f cycles per MFMA per wave -> %.0f flop/cycle/CU\n", nm[kind], waves, cyc,
               flop[kind] / cyc * 4 * waves);
      }
    }
  }
  return 0;
}
