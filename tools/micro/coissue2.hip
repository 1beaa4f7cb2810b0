// Micro-benchmark 2: VALU / transcendental instructions interleaved with MFMAs in ONE wave's stream (one wave per SIMD),
// and the same streams run by two co-resident waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// NF fma + NE exp after every MFMA, all independent of the MFMA results
template <int NF, int NE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int nwaves) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  f32x16 acc0 = {}, acc1 = {};
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = x[i] * 0.01f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) x[(j + f) & 7] = __builtin_fmaf(x[(j + f) & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int e = 0; e < NE; ++e) y[(j + e) & 7] = __builtin_amdgcn_exp2f(y[(j + e) & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += x[i] + y[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NF, int NE>
void run(float* out, int iters) {
  for (int nw = 4; nw <= 8; nw += 4) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NF, NE>), dim3(256), dim3(512), 0, 0, out, iters, nw);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, NE>), dim3(256), dim3(512), 0, 0, out, iters, nw);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per MFMA slot: ms / (iters * 8 * waves_per_simd) in ns
    const double ns = ms * 1e6 / ((double)iters * 8 * (nw / 4));
    printf("fma/MFMA=%d exp/MFMA=%d waves/SIMD=%d  %.3f ms  %.1f ns per MFMA slot (32 cyc = %.1f ns at 2.4 GHz)\n", NF, NE, nw / 4, ms, ns, 32 / 2.4);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  run<0, 0>(out, iters);
  run<2, 0>(out, iters);
  run<4, 0>(out, iters);
  run<6, 0>(out, iters);
  run<8, 0>(out, iters);
  run<12, 0>(out, iters);
  run<0, 1>(out, iters);
  run<0, 2>(out, iters);
  run<0, 4>(out, iters);
  run<4, 1>(out, iters);
  run<4, 2>(out, iters);
  return 0;
}
