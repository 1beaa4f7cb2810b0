// Micro-benchmark: can one wave's VALU / transcendental stream issue under another wave's MFMAs on the same SIMD?
// 512-thread workgroups, one per CU: waves 0-3 run an MFMA loop, waves 4-7 a VALU loop (mode bit 0 / bit 1 enable them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int mode) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mf = wave < 4;
  if (mf && !(mode & 1)) return;
  if (!mf && !(mode & 2)) return;
  if (mf) {
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a = {}, b = {};
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (KIND == 0) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);         // 16 v_fma per iteration
        if (KIND == 1) x[j] = __builtin_amdgcn_exp2f(x[j]) ;                // 16 v_exp per iteration
        if (KIND == 2) { x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f); x[j] = __builtin_fmaf(x[j], 0.9999f, 0.25f);
                         x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f); x[j] = __builtin_fmaf(x[j], 0.9999f, 0.25f); } // 64 fma
      }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <int KIND>
void run(const char* name, float* out, int iters) {
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s mode=%d (%s)  %.3f ms\n", name, mode, mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both", ms);
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;   // 16 MFMA (512 cyc) per iteration on the matrix waves
  run<0>("fma16", out, iters);
  run<1>("exp16", out, iters);
  run<2>("fma64", out, iters);
  return 0;
}
