// Micro-benchmark of the gfx950 issue model the attention kernel is designed against (MI355X_MICROARCH.md, "vector-instruction
// ISSUE cost"): how many cycles does ONE wave's stream {1 v_mfma_f32_32x32x16_bf16 + NF independent vector fillers} take per
// MFMA, for several filler kinds, at 1 / 2 / 4 waves per SIMD; and do the un-interleaved phases {8 MFMA}{NV VALU} of
// DIFFERENT waves of one SIMD overlap.  Cycles are s_memtime deltas of wave 0 of every workgroup (median printed), so the
// answer does not depend on the clock the chip holds.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue_model.hip -o tools/micro/issue_model && tools/micro/issue_model
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { F_FMA = 0, F_EXP = 1, F_CVT = 2, F_PKFMA = 3, F_MAX3 = 4, F_PKADD = 5, F_MIX = 6, F_AND = 7 };

template <int KIND>
__device__ __forceinline__ void filler(float& x, float& y, f32x2& p, unsigned& u) {
  if constexpr (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == F_EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(y));
  if constexpr (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
  if constexpr (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
  if constexpr (KIND == F_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  if constexpr (KIND == F_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u) : "v"(u));
}

// interleaved stream: per MFMA, NF fillers on NF distinct register sets (independent of one another and of the MFMA)
template <int KIND, int NF>
__global__ __launch_bounds__(1024) void k_inter(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc = {};
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float x[8], y[8]; f32x2 p[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = 0.999f; p[i] = f32x2{x[i], 1.f}; u[i] = threadIdx.x + i; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if constexpr (KIND == F_MIX) {   // the softmax mix per 16 scores of a 32x32x16 gap at D = 64: see main()
          if (f % 13 < 2) filler<F_EXP>(x[f % 8], y[f % 8], p[f % 8], u[f % 8]);
          else if (f % 13 < 4) filler<F_CVT>(x[f % 8], y[f % 8], p[f % 8], u[f % 8]);
          else filler<F_FMA>(x[f % 8], y[f % 8], p[f % 8], u[f % 8]);
        } else {
          filler<KIND>(x[f % 8], y[f % 8], p[f % 8], u[f % 8]);
        }
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1] + __uint_as_float(u[i] & 0x3f800000u);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// phased stream: {8 MFMA}{NV fma fillers}, every wave of the SIMD runs the same program (as the round-1 attention does)
template <int NV>
__global__ __launch_bounds__(1024) void k_phase(float* out, unsigned long long* cyc, int iters, int stagger) {
  f32x16 acc = {};
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float x[8], y = 0.999f;
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (stagger && (wave & 4)) {   // second half of the waves starts in the VALU phase
#pragma unroll
    for (int f = 0; f < NV; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f % 8]) : "v"(y));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
    for (int f = 0; f < NV; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f % 8]) : "v"(y));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;

template <typename L>
static void measure(const char* name, int threads, int iters, int per_iter_mfma, L launch) {
  hipMemset(g_cyc, 0, 256 * 16 * 8);
  launch(threads);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch(threads);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(256 * 16);
  hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> w;
  for (int b = 0; b < 256; ++b) for (int i = 0; i < threads / 64; ++i) w.push_back(c[b * 16 + i]);
  std::sort(w.begin(), w.end());
  const double med = (double)w[w.size() / 2];
  const double per_mfma = med / ((double)iters * per_iter_mfma);
  // throughput view: all waves of a SIMD together issue (threads / 256) x the MFMAs in `med` cycles
  printf("%-28s waves/SIMD=%d  cycles per wave per MFMA %7.1f   SIMD cycles per MFMA %6.1f   (%.3f ms, clock %.2f GHz)\n", name,
         threads / 256, per_mfma, per_mfma / (threads / 256), ms, med / (ms * 1e6));
}

template <int KIND, int NF>
static void run_inter(const char* name, int iters) {
  char buf[64];
  for (int threads : {256, 512, 1024}) {
    snprintf(buf, sizeof buf, "%s x%d / MFMA", name, NF);
    measure(buf, threads, iters, 8, [&](int t) { hipLaunchKernelGGL((k_inter<KIND, NF>), dim3(256), dim3(t), 0, 0, g_out, g_cyc, iters); });
  }
}
template <int NV>
static void run_phase(int iters) {
  char buf[64];
  for (int stagger = 0; stagger < 2; ++stagger)
    for (int threads : {256, 512, 1024}) {
      snprintf(buf, sizeof buf, "phased 8 MFMA | %d fma%s", NV, stagger ? " stag" : "");
      measure(buf, threads, iters, 8, [&](int t) { hipLaunchKernelGGL((k_phase<NV>), dim3(256), dim3(t), 0, 0, g_out, g_cyc, iters, stagger); });
    }
}

int main(int argc, char** argv) {
  hipMalloc(&g_out, 256 * 1024 * 4);
  hipMalloc(&g_cyc, 256 * 16 * 8);
  const int iters = 4000;
  if (argc > 1) {   // counter calibration subset (run under rocprofv3 --pmc): known MFMA / VALU busy fractions
    run_inter<F_FMA, 0>("none", iters);
    run_inter<F_FMA, 6>("v_fma_f32", iters);
    run_inter<F_FMA, 16>("v_fma_f32", iters);
    run_phase<64>(iters);
    return 0;
  }
  run_inter<F_FMA, 0>("none", iters);
  run_inter<F_FMA, 4>("v_fma_f32", iters);
  run_inter<F_FMA, 6>("v_fma_f32", iters);
  run_inter<F_FMA, 8>("v_fma_f32", iters);
  run_inter<F_FMA, 12>("v_fma_f32", iters);
  run_inter<F_FMA, 16>("v_fma_f32", iters);
  run_inter<F_EXP, 2>("v_exp_f32", iters);
  run_inter<F_EXP, 4>("v_exp_f32", iters);
  run_inter<F_EXP, 8>("v_exp_f32", iters);
  run_inter<F_CVT, 8>("v_cvt_pk_bf16_f32", iters);
  run_inter<F_PKFMA, 4>("v_pk_fma_f32", iters);
  run_inter<F_PKFMA, 8>("v_pk_fma_f32", iters);
  run_inter<F_PKADD, 8>("v_pk_add_f32", iters);
  run_inter<F_MAX3, 8>("v_max3_f32", iters);
  run_inter<F_AND, 8>("v_and_b32", iters);
  run_inter<F_MIX, 13>("mix(2exp,2cvt,9fma)", iters);
  run_inter<F_MIX, 26>("mix(4exp,4cvt,18fma)", iters);
  run_phase<32>(iters);
  run_phase<64>(iters);
  run_phase<128>(iters);
  return 0;
}
