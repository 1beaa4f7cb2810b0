// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), fp32 accumulate.
//
// Structure (cdna_hip_programming.md §5): 128x128 output tile per 256-thread workgroup (4 waves, 2x2,
// 64x64 per wave as 4x4 MFMA 16x16x32 tiles), BK = 64, both operand tiles staged global -> LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier per K tile.  The LDS image is
// lane-linear (a DMA requirement), so the bank-conflict swizzle is applied on the per-lane SOURCE address
// (16-byte chunk c of row r is stored at chunk c ^ (r & 7)) and undone on the ds_read_b128 side.
//
// Operand roles are swapped w.r.t. the textbook: the WEIGHT tile is the MFMA A operand and the
// ACTIVATION tile the B operand, so each lane ends up with 4 consecutive output COLUMNS of one output
// row (D row = n, D col = m) and the epilogue stores 8 bytes per lane instead of 4 scattered bf16.
//
// Epilogues reproduce the reference's bf16 op boundaries (one rounding per torch op).
#include "mjv_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int LDS_BYTES = 4 * TILE_BYTES;

struct GemmArgs {
  const u16* A; long lda;
  const u16* W; long ldw;
  u16* C; long ldc;
  int M, N, K;
  const u16* bias;
  const u16* scale;
  const u16* res; long ldr;
  int res_mod, res_off;
  int out_group, out_pad;
  const int* out_rows;
  int tiles_m, tiles_n;
};

MJV_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
MJV_DEV float silu(float x) { return x / (1.0f + __expf(-x)); }

// stage one 128 x 64 bf16 tile: 16 DMA pieces of 8 rows x 128 B; wave w issues pieces 4w .. 4w+3
MJV_DEV void stage_tile(const u16* __restrict__ src, long ld, int row0, int max_row, int k0, char* lds_tile,
                        int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    int gr = row0 + r;
    gr = gr < max_row ? gr : max_row;
    const u16* g = src + (long)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a
  // contiguous range of the (m-major, n-minor) tile list: its tiles then share A panels through its L2.
  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
  const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile(p.A, p.lda, m0, p.M - 1, 0, smem, wave, lane);
  stage_tile(p.W, p.ldw, n0, p.N - 1, 0, smem + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
    if (t + 1 < nk) {
      stage_tile(p.A, p.lda, m0, p.M - 1, (t + 1) * BK, nxt, wave, lane);
      stage_tile(p.W, p.ldw, n0, p.N - 1, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* As = cur;
    const char* Ws = cur + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[4], af[4];
      const int g = kk * 4 + l4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + l15;
        wf[j] = *(const bf16x8*)(Ws + r * 128 + ((g ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + l15;
        af[i] = *(const bf16x8*)(As + r * 128 + ((g ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane holds, for tile (i, j), rows n = 4*l4 .. 4*l4+3 (registers) of column m = l15
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= p.M) continue;
    long orow = m;
    if (p.out_rows) {
      orow = p.out_rows[m];
    } else if (p.out_group > 0) {
      const int gq = m / p.out_group;
      orow = (long)gq * (p.out_group + p.out_pad) + p.out_pad + (m - gq * p.out_group);
    }
    if constexpr (EPI == MJV_EPI_SILU_MUL) {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = n0 + wn * 64 + j * 16 + l4 * 4;  // row of the interleaved weight (gate block)
        if (n >= p.N) continue;
        const int oc = (n0 + wn * 64) / 2 + (j / 2) * 16 + l4 * 4;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gte = rbf(acc[i][j][r]);
          const float up = rbf(acc[i][j + 1][r]);
          o[r] = rbf(silu(gte)) * up;
        }
        u32x2 v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        *(u32x2*)(p.C + orow * p.ldc + oc) = v;
      }
    } else {
      long rrow = m;
      if (p.res_mod > 0) rrow = p.res_off + (m % p.res_mod);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + l4 * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if (p.bias) {
          const u32x2 bb = *(const u32x2*)(p.bias + n);
          v[0] += __uint_as_float(bb[0] << 16);
          v[1] += __uint_as_float(bb[0] & 0xffff0000u);
          v[2] += __uint_as_float(bb[1] << 16);
          v[3] += __uint_as_float(bb[1] & 0xffff0000u);
        }
        if constexpr (EPI == MJV_EPI_BIAS_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(rbf(v[r]));
        } else if constexpr (EPI == MJV_EPI_BIAS_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if constexpr (EPI == MJV_EPI_SCALE_RES) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
          if (p.scale) {
            const u32x2 ss = *(const u32x2*)(p.scale + n);
            v[0] = rbf(v[0] * __uint_as_float(ss[0] << 16));
            v[1] = rbf(v[1] * __uint_as_float(ss[0] & 0xffff0000u));
            v[2] = rbf(v[2] * __uint_as_float(ss[1] << 16));
            v[3] = rbf(v[3] * __uint_as_float(ss[1] & 0xffff0000u));
          }
          const u32x2 rr = *(const u32x2*)(p.res + rrow * p.ldr + n);
          v[0] += __uint_as_float(rr[0] << 16);
          v[1] += __uint_as_float(rr[0] & 0xffff0000u);
          v[2] += __uint_as_float(rr[1] << 16);
          v[3] += __uint_as_float(rr[1] & 0xffff0000u);
        }
        u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)(p.C + orow * p.ldc + n) = o;
      }
    }
  }
}

template <int EPI>
int launch(const GemmArgs& a, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  const int grid = a.tiles_m * a.tiles_n;
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(grid), dim3(256), LDS_BYTES, s, a);
  return mjv_check_launch("gemm_bf16");
}

}  // namespace

extern "C" int mjv_gemm_bf16(const mjv_gemm_desc* d, void* stream) {
  MJV_REQUIRE(d && d->A && d->W && d->C, "gemm: null pointer");
  MJV_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  MJV_REQUIRE(d->K % 64 == 0, "gemm: K=%d must be a multiple of 64", d->K);
  MJV_REQUIRE(d->N % 8 == 0, "gemm: N=%d must be a multiple of 8", d->N);
  MJV_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0 && d->ldc % 4 == 0, "gemm: leading dims must be multiples of 8");
  MJV_REQUIRE(d->lda >= d->K && d->ldw >= d->K, "gemm: lda/ldw smaller than K");
  MJV_REQUIRE(((uintptr_t)d->A | (uintptr_t)d->W) % 16 == 0 && (uintptr_t)d->C % 8 == 0, "gemm: misaligned pointer");
  if (d->epilogue == MJV_EPI_SCALE_RES) {
    MJV_REQUIRE(d->res != nullptr && d->ldr % 4 == 0, "gemm: SCALE_RES needs a residual");
  }
  if (d->epilogue == MJV_EPI_SILU_MUL) {
    MJV_REQUIRE(d->N % 32 == 0 && d->bias == nullptr, "gemm: SILU_MUL needs N %% 32 == 0 and no bias");
  }
  GemmArgs a;
  a.A = d->A; a.lda = d->lda; a.W = d->W; a.ldw = d->ldw; a.C = d->C; a.ldc = d->ldc;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.bias = d->bias; a.scale = d->scale; a.res = d->res; a.ldr = d->ldr;
  a.res_mod = d->res_mod; a.res_off = d->res_off; a.out_group = d->out_group; a.out_pad = d->out_pad;
  a.out_rows = d->out_rows;
  a.tiles_m = (d->M + BM - 1) / BM;
  a.tiles_n = (d->N + BN - 1) / BN;
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * d->M * (double)d->N * d->K;
  const double bytes = 2.0 * ((double)d->M * d->K + (double)d->N * d->K + (double)d->M * d->N);
  switch (d->epilogue) {
    case MJV_EPI_BIAS: { MjvProfScope ps("gemm_bias", s, flops, bytes); return launch<MJV_EPI_BIAS>(a, s); }
    case MJV_EPI_BIAS_GELU: { MjvProfScope ps("gemm_bias_gelu", s, flops, bytes); return launch<MJV_EPI_BIAS_GELU>(a, s); }
    case MJV_EPI_BIAS_RELU: { MjvProfScope ps("gemm_bias_relu", s, flops, bytes); return launch<MJV_EPI_BIAS_RELU>(a, s); }
    case MJV_EPI_SCALE_RES: { MjvProfScope ps("gemm_scale_res", s, flops, bytes); return launch<MJV_EPI_SCALE_RES>(a, s); }
    case MJV_EPI_SILU_MUL: { MjvProfScope ps("gemm_silu_mul", s, flops, bytes); return launch<MJV_EPI_SILU_MUL>(a, s); }
    default: mjv_set_error("gemm: unknown epilogue %d", d->epilogue); return MJV_E_ARG;
  }
}
