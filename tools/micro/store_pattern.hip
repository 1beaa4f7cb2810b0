// Micro-benchmark: how fast does a 256 x 256 bf16 output tile per workgroup reach memory when the 8 waves store it straight from
// the MFMA fragment layout, against the LDS-staged row layout the GEMM epilogue uses?   (DESIGN §7 "what comes next")
//  pattern 0: rows      - one instruction = 2 rows x 512 B (32 lanes x 16 B per row): what pass B of the epilogue does
//  pattern 1: 64-B runs - lane (l15, l4) owns 16 consecutive columns of row l15: one instruction = 16 rows x 64 B
//                         (the fragment layout if the weight rows are dealt to the MFMA A rows as n = 16 l4 + 4 j + e)
//  pattern 2: 32-B runs - lane owns 4 consecutive columns per fragment: one instruction = 16 rows x 32 B, 8 B per lane (round 1)
// mode bit 0: also read a residual tile in the same pattern and add it; bit 1: streaming (nt) stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int P, int MODE>
__global__ __launch_bounds__(512) void k(unsigned short* out, const unsigned short* res, long ld, int tiles_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  unsigned short* base = out + (long)tm * 256 * ld + tn * 256;
  const unsigned short* rbase = res + (long)tm * 256 * ld + tn * 256;
  const unsigned v0 = tid * 2654435761u;
  if constexpr (P == 0) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const long off = (long)(it * 16 + tid / 32) * ld + (tid % 32) * 8;
      u32x4 v = {v0 + it, v0 ^ it, v0, v0 * 3};
      if constexpr (MODE & 1) { const u32x4 r = *(const u32x4*)(rbase + off); v += r; }
      if constexpr (MODE & 2) __builtin_nontemporal_store(v, (u32x4*)(base + off));
      else *(u32x4*)(base + off) = v;
    }
  } else if constexpr (P == 1) {
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long off = (long)(wr * 128 + i * 16 + l15) * ld + wc * 64 + l4 * 16 + h * 8;
        u32x4 v = {v0 + i, v0 ^ h, v0, v0 * 3};
        if constexpr (MODE & 1) { const u32x4 r = *(const u32x4*)(rbase + off); v += r; }
        if constexpr (MODE & 2) __builtin_nontemporal_store(v, (u32x4*)(base + off));
        else *(u32x4*)(base + off) = v;
      }
  } else {
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long off = (long)(wr * 128 + i * 16 + l15) * ld + wc * 64 + j * 16 + l4 * 4;
        u32x2 v = {v0 + i, v0 ^ j};
        if constexpr (MODE & 1) { const u32x2 r = *(const u32x2*)(rbase + off); v += r; }
        if constexpr (MODE & 2) __builtin_nontemporal_store(v, (u32x2*)(base + off));
        else *(u32x2*)(base + off) = v;
      }
  }
}

template <int P, int MODE>
void run(unsigned short* out, const unsigned short* res, int M, int N) {
  const int tiles = (M / 256) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<P, MODE>), dim3(tiles), dim3(512), 0, 0, out, res, (long)N, N / 256);
  hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<P, MODE>), dim3(tiles), dim3(512), 0, 0, out, res, (long)N, N / 256);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double bytes = (double)M * N * 2 * ((MODE & 1) ? 2 : 1);
  printf("M=%d N=%d pattern %d %s%s: %.3f ms  %.0f GB/s\n", M, N, P, (MODE & 1) ? "read+write" : "write", (MODE & 2) ? " nt" : "", ms, bytes / ms / 1e6);
}

int main() {
  const int M = 65536;
  for (int N : {1024, 4096}) {
    unsigned short *out, *res;
    hipMalloc(&out, (size_t)M * N * 2); hipMalloc(&res, (size_t)M * N * 2);
    hipMemset(res, 1, (size_t)M * N * 2);
    run<0, 0>(out, res, M, N); run<1, 0>(out, res, M, N); run<2, 0>(out, res, M, N);
    run<0, 2>(out, res, M, N); run<1, 2>(out, res, M, N); run<2, 2>(out, res, M, N);
    run<0, 1>(out, res, M, N); run<1, 1>(out, res, M, N); run<2, 1>(out, res, M, N);
    run<0, 3>(out, res, M, N); run<1, 3>(out, res, M, N); run<2, 3>(out, res, M, N);
    hipFree(out); hipFree(res);
  }
  return 0;
}
