// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] @ W[N,K]^T), fp32 accumulate.
//
// Two kernels share the operand layout, the LDS image and the epilogues:
//
//  * gemm256_kernel - the production shape.  256x256x64 tile, 512 threads (8 waves as 2(M) x 4(N), 128x64 per
//    wave, 128 accumulator VGPRs), 128 KiB LDS = 2 K-tiles x 4 half-tiles (128 rows x 64 k) of 16 KiB.  Each
//    K-tile runs as 2 phases {ds_read the register sub-tiles | issue TWO half-tiles of LDS-DMA one / two K-tiles ahead |
//    barrier | 32 MFMA | barrier}; DMA stays in flight across barriers behind a COUNTED s_waitcnt vmcnt(4) once per
//    K-tile (never 0 in the steady state), raw s_barrier only.  The two M-groups of waves run one barrier apart,
//    so on every SIMD one wave is in its MFMA segment while its partner reads LDS / issues DMA
//    (cdna_hip_programming.md §5 "8-phase template", MI355X_MICROARCH.md "Two waves per SIMD").
//  * gemm128_kernel - 128x128x64 tile, 256 threads, double-buffered, one barrier per K-tile; used for small
//    problems (tiny configs, the M = batch gating layers) where a 256^2 tile would be mostly padding, and for the
//    tail rows peeled off an under-filled last round of the 256^2 kernel.  Under-filled long-K launches of it split
//    K over more workgroups (SPLIT) and finish in splitk_finish_kernel.
//
// What bounds them (measured): a CU fills LDS from L2 at about 66-73 GB/s and from the Infinity Cache at about 33;
// the 256^2 K-tile needs 64 KiB per 8.4 MFLOP, so with the 81 % L2 hit rate of an 8 x 4 tile patch per XCD the main
// loop runs at 1.1-1.5 us per K-tile against 0.87 us of MFMA time: the large shapes sit at 1 300-1 480 TFLOP/s.
//
// Operand staging is 16-byte LDS-DMA (global_load_lds_dwordx4).  The LDS image must be lane-linear, so the
// bank-conflict swizzle is applied on the per-lane SOURCE address (16-B chunk c of row r is stored at chunk
// c ^ (r & 7)) and undone on the ds_read_b128 side; reads are conflict-free.
//
// Operand roles are swapped w.r.t. the textbook: the WEIGHT tile is the MFMA A operand and the ACTIVATION tile
// the B operand, so each lane ends up with 4 consecutive output COLUMNS of one output row (D row = n, D col = m)
// and the epilogue moves 8 bytes per lane per access.
//
// Epilogues reproduce the reference's bf16 op boundaries (one rounding per torch op).
#include "mjv_common.h"
#include "gelu_table.h"
#include <atomic>

namespace {

// exact-erf GELU on bf16 as a finite function (tools/gen_gelu_table.py): bit-identical to torch's CPU bf16 GELU
__device__ const u16 g_gelu_table[MJV_GELU_TABLE_LEN] = MJV_GELU_TABLE_INIT;

// x given as an fp32 value that is exactly representable in bf16; tab = table in LDS or global memory.
// Branch-free on purpose (selects only): a divergent version costs ~3x more than the table read it guards.
MJV_DEV float gelu_lut(float xf, const u16* tab) {
  const unsigned u = __float_as_uint(xf);
  const unsigned mag = (u >> 16) & 0x7fffu;
  const unsigned rel = mag - MJV_GELU_LO;                       // wraps to a huge value below the table
  const unsigned sgn = (unsigned)((int)u >> 31);                // all ones for x < 0
  const bool in_tab = rel < (unsigned)MJV_GELU_R;
  const unsigned idx = in_tab ? rel + (sgn & MJV_GELU_NEG_OFF) : 0u;
  const unsigned t = tab[idx];
  const unsigned big = gelu_beyond_table(u, mag);               // beyond the table (mjv_common.h)
  const unsigned small = __float_as_uint(0.5f * xf);            // below the table: x / 2
  const unsigned other = mag < MJV_GELU_LO ? small : big;
  return __uint_as_float(in_tab ? (t << 16) : other);
}

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
  int gm;        // group-M of the tile order
  int nt_store;  // streaming stores for the output (host heuristic)
  int m_base;  // absolute index of row 0 of A (tail launches): output / residual rows are computed from m_base + m
  float* ws;   // split-K: fp32 partial tiles, [tile][slice][fragment][thread] float4 (the accumulators as they sit in registers)
  int split;   // K slices per tile (1 = no split)
  // MJV_EPI_ROPE_QKV
  const u16 *rope_cos, *rope_sin;
  const int* rope_pos;
  u16 *rope_q, *rope_k;
  long rope_ldq, rope_ldk;
  int rope_group;
  // norm folded into the consumer (mjv.h "row_scale"): lin = row_scale[m] * acc - row_shift[m] * col_shift[n] + bias
  const float *row_scale, *row_shift, *col_shift, *bias_f32;
};

// x * rcp(1 + e^-x): v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division; the result is rounded to bf16 next, and
// e^-x is a 1-2 ulp v_exp_f32 already, so the quotient's last fp32 bit carries no information the reference shares
MJV_DEV float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
MJV_DEV unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// 16-byte output store, streaming (nt) on request.  The nt form is inline assembly: written as `if (nt)
// __builtin_nontemporal_store(...) else plain store`, hipcc (ROCm 7.2) merges the two arms into ONE plain store and the hint is
// gone (the kernels of rounds 1-3 contained no nt store at all - found in round 4 by reading the ISA).  Counted in vmcnt like
// any store; nothing ever waits for it.
MJV_DEV void store16(u16* dst, const u32x4& val, int nt) {
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(val) : "memory");
  else *(u32x4*)dst = val;
}

MJV_DEV long out_row_of(const GemmArgs& p, int m) {
  if (p.out_rows) return p.out_rows[m];
  if (p.out_group > 0) {
    const int gq = m / p.out_group;
    return (long)gq * (p.out_group + p.out_pad) + p.out_pad + (m - gq * p.out_group);
  }
  return m;
}

// SiLU-mul: gate fragment (weight rows n..n+3 of a w1 block) and the matching up fragment (w3 block)
MJV_DEV void store_silu(const GemmArgs& p, const f32x4& g, const f32x4& u, long orow, int oc, float rs = 1.f) {
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = rbf(silu(rbf(g[r] * rs))) * rbf(u[r] * rs);
  const u32x2 v = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
  *(u32x2*)(p.C + orow * p.ldc + oc) = v;
}

// Epilogue of the small kernels for an NI x NJ block of 16x16 accumulator fragments of one thread (fragment (i, j): output
// row mrel[i] of the launch, columns ncol[j] .. ncol[j] + 3).  One 16x16 fragment: the lane holds output row m, columns n .. n+3.  Every
// global load of the block - output-row map, bias, LayerScale, residual - is issued before the first use, from clamped
// (always valid) addresses: fragment by fragment behind row / column conditions (round 1), each load got a branch and an
// s_waitcnt vmcnt(0) of its own, three to four dependent L2 round trips per fragment, sixteen fragments per thread.
template <int EPI, int NI, int NJ>
MJV_DEV void store_frags(const GemmArgs& p, const f32x4* acc /* [NI][NJ] */, const int (&mrel)[NI], const int (&ncol)[NJ]) {
  static_assert(EPI != MJV_EPI_SILU_MUL, "SiLU pairs go through store_silu");
  int mc[NI], nc[NJ];
  long orow[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) mc[i] = p.m_base + (mrel[i] < p.M ? mrel[i] : p.M - 1);
#pragma unroll
  for (int j = 0; j < NJ; ++j) nc[j] = ncol[j] < p.N ? ncol[j] : p.N - 4;
  if (p.out_rows) {
    int t[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) t[i] = p.out_rows[mc[i]];
#pragma unroll
    for (int i = 0; i < NI; ++i) orow[i] = t[i];
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) orow[i] = out_row_of(p, mc[i]);
  }
  u32x2 bb[NJ], ss[NJ], rr[NI][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) bb[j] = ss[j] = u32x2{0u, 0u};
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) bb[j] = *(const u32x2*)(p.bias + nc[j]);
  }
  // norm folded into this GEMM: lin = row_scale[m] acc - row_shift[m] col_shift[n] + bias_f32[n] (uniform branches)
  float rsc[NI], rsh[NI];
  f32x4 csh[NJ], bf[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) { rsc[i] = 1.f; rsh[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) csh[j] = bf[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.row_scale) {
#pragma unroll
    for (int i = 0; i < NI; ++i) rsc[i] = p.row_scale[mc[i]];
    if (p.row_shift) {
#pragma unroll
      for (int i = 0; i < NI; ++i) rsh[i] = p.row_shift[mc[i]];
#pragma unroll
      for (int j = 0; j < NJ; ++j) { csh[j] = *(const f32x4*)(p.col_shift + nc[j]); bf[j] = *(const f32x4*)(p.bias_f32 + nc[j]); }
    }
  }
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
    if (p.scale) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) ss[j] = *(const u32x2*)(p.scale + nc[j]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const long rrow = p.res_mod > 0 ? (long)(p.res_off + (mc[i] % p.res_mod)) : (long)mc[i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) rr[i][j] = *(const u32x2*)(p.res + rrow * p.ldr + nc[j]);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (mrel[i] >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (ncol[j] >= p.N) continue;
      const f32x4 a = acc[i * NJ + j];
      float v[4] = {a[0], a[1], a[2], a[3]};
      if (p.row_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], rsc[i], fmaf(-rsh[i], csh[j][r], bf[j][r]));
      }
      if (p.bias) {
        v[0] += __uint_as_float(bb[j][0] << 16);
        v[1] += __uint_as_float(bb[j][0] & 0xffff0000u);
        v[2] += __uint_as_float(bb[j][1] << 16);
        v[3] += __uint_as_float(bb[j][1] & 0xffff0000u);
      }
      if constexpr (EPI == MJV_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_lut(rbf(v[r]), g_gelu_table);
      } else if constexpr (EPI == MJV_EPI_BIAS_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if constexpr (EPI == MJV_EPI_SCALE_RES) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
        if (p.scale) {
          v[0] = rbf(v[0] * __uint_as_float(ss[j][0] << 16));
          v[1] = rbf(v[1] * __uint_as_float(ss[j][0] & 0xffff0000u));
          v[2] = rbf(v[2] * __uint_as_float(ss[j][1] << 16));
          v[3] = rbf(v[3] * __uint_as_float(ss[j][1] & 0xffff0000u));
        }
        v[0] += __uint_as_float(rr[i][j][0] << 16);
        v[1] += __uint_as_float(rr[i][j][0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[i][j][1] << 16);
        v[3] += __uint_as_float(rr[i][j][1] & 0xffff0000u);
      }
      const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *(u32x2*)(p.C + orow[i] * p.ldc + ncol[j]) = o;
    }
  }
}

MJV_DEV void tile_of_vblock(const GemmArgs& p, int nwg, int b, int& tm, int& tn);
MJV_DEV void tile_of_block(const GemmArgs& p, int& tm, int& tn) { tile_of_vblock(p, gridDim.x, blockIdx.x, tm, tn); }
MJV_DEV void tile_of_vblock(const GemmArgs& p, const int nwg, const int b, int& tm, int& tn) {
  // XCD-aware tile order.  Blocks b and b+8 share an XCD (round-robin dispatch), so every XCD gets a contiguous range
  // of a tile list that is itself ordered in groups of GM m-tiles x all n-tiles, n-major inside a group: the ~32
  // tiles an XCD runs concurrently then form a GM x (32/GM) patch that shares GM activation panels and 32/GM weight
  // panels through that XCD's 4 MiB L2 (row-major order shares 1 + 32 panels: the whole weight matrix was re-fetched
  // over the fabric once per m-tile row - rocprofv3 FETCH_SIZE 4.7 GB per w1|w3 launch for 139 MB of operands).
  const int GM = p.gm;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
  const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
  const int per_group = GM * p.tiles_n;
  const int grp = t / per_group, in_grp = t - grp * per_group;
  const int first_m = grp * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  tn = in_grp / gsz;
  tm = first_m + (in_grp - tn * gsz);
}

// ============================================================================================ 128 x 128
namespace t128 {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;
constexpr int LDS_BYTES = 4 * TILE_BYTES;

MJV_DEV void stage_tile(const u16* __restrict__ src, long ld, int row0, int max_row, int k0, char* lds_tile, int wave, int lane) {
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

// SPLIT: the launch has split slices per tile (workgroup b = tile b / split, slice b % split); a slice covers a
// contiguous share of the K-tiles and leaves its accumulators, as they sit in registers, in the fp32 workspace;
// splitk_finish_kernel sums the slices in slice order (deterministic) and applies the epilogue.
template <int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  int tm, tn;
  const int tile_id = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x;
  const int slice = SPLIT ? (int)blockIdx.x % p.split : 0;
  tile_of_vblock(p, SPLIT ? (int)gridDim.x / p.split : (int)gridDim.x, tile_id, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = p.K / BK;
  const int kt0 = SPLIT ? (int)((long)nk_all * slice / p.split) : 0;          // this slice's K-tiles [kt0, kt1)
  const int nk = (SPLIT ? (int)((long)nk_all * (slice + 1) / p.split) : nk_all) - kt0;
  stage_tile(p.A, p.lda, m0, p.M - 1, kt0 * BK, smem, wave, lane);
  stage_tile(p.W, p.ldw, n0, p.N - 1, kt0 * BK, smem + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
    if (t + 1 < nk) {
      stage_tile(p.A, p.lda, m0, p.M - 1, (kt0 + t + 1) * BK, nxt, wave, lane);
      stage_tile(p.W, p.ldw, n0, p.N - 1, (kt0 + t + 1) * BK, nxt + TILE_BYTES, wave, lane);
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

  if constexpr (SPLIT) {
    f32x4* img = (f32x4*)p.ws + ((long)tile_id * p.split + slice) * (16 * 256) + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) img[(i * 4 + j) * 256] = acc[i][j];
    return;
  }

  if constexpr (EPI == MJV_EPI_SILU_MUL) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mrel = m0 + wm * 64 + i * 16 + l15;
      if (mrel >= p.M) continue;
      const long orow = out_row_of(p, p.m_base + mrel);
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = n0 + wn * 64 + j * 16 + l4 * 4;
        if (n >= p.N) continue;
        store_silu(p, acc[i][j], acc[i][j + 1], orow, (n0 + wn * 64) / 2 + (j / 2) * 16 + l4 * 4,
                   p.row_scale ? p.row_scale[p.m_base + mrel] : 1.f);
      }
    }
  } else {
    int mrel[4], ncol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mrel[i] = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) ncol[j] = n0 + wn * 64 + j * 16 + l4 * 4;
    store_frags<EPI, 4, 4>(p, &acc[0][0], mrel, ncol);
  }
}

// Second half of a split-K launch: one thread per (tile, thread of the GEMM workgroup); sums the slices' register images
// in slice order and runs the same per-fragment epilogue code as the unsplit kernel.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_finish_kernel(GemmArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  int tm, tn;
  tile_of_vblock(p, gridDim.x, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const f32x4* img = (const f32x4*)p.ws + (long)blockIdx.x * p.split * (16 * 256) + tid;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = img[(i * 4 + j) * 256];
  for (int sl = 1; sl < p.split; ++sl) {
    const f32x4* im2 = img + (long)sl * (16 * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += im2[(i * 4 + j) * 256];
  }
  if constexpr (EPI == MJV_EPI_SILU_MUL) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mrel = m0 + wm * 64 + i * 16 + l15;
      if (mrel >= p.M) continue;
      const long orow = out_row_of(p, p.m_base + mrel);
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = n0 + wn * 64 + j * 16 + l4 * 4;
        if (n >= p.N) continue;
        store_silu(p, acc[i][j], acc[i][j + 1], orow, (n0 + wn * 64) / 2 + (j / 2) * 16 + l4 * 4,
                   p.row_scale ? p.row_scale[p.m_base + mrel] : 1.f);
      }
    }
  } else {
    int mrel[4], ncol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mrel[i] = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) ncol[j] = n0 + wn * 64 + j * 16 + l4 * 4;
    store_frags<EPI, 4, 4>(p, &acc[0][0], mrel, ncol);
  }
}
}  // namespace t128

// ============================================================================================ 64 x 32 (skinny M)
// Small-M problems: the rows peeled off an under-filled last round of the 256^2 kernel (M = 64 in the vision tower, 80
// for w1|w3), the batch-sized gating layers.  Such a launch is pure latency - a chain of K / 64 dependent LDS fills on
// the few CUs it occupies, each fill bounded by one CU's L2 -> LDS rate - so the tile is cut small in N (32 weight rows:
// 12 KiB per K-tile instead of 32) to put the chain on many CUs and shorten every link, and two K-tiles of LDS-DMA stay
// in flight (three buffers, one barrier per K-tile, counted vmcnt).  Same operand layout, swizzle, fragment maps and
// epilogue code as the 128^2 kernel; 4 waves, wave w owns activation rows 16 w .. 16 w + 15 of the tile and all 32
// weight rows (so a SiLU gate / up pair sits in one lane).
namespace t64 {
constexpr int BM = 64, BK = 64;
constexpr int A_BYTES = BM * BK * 2;
constexpr int NBUF = 3;
template <int NJ>
struct Cfg {   // NJ 16-row weight fragments per tile: BN = 32 (skinny M <= 128) or 128 (tails of ~1000 rows)
  static constexpr int BN = 16 * NJ;
  static constexpr int W_BYTES = BN * BK * 2;
  static constexpr int TILE_BYTES = A_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = NBUF * TILE_BYTES;
  static constexpr int W_PIECES = BN / 8 / 4;        // 1-KiB DMA pieces of the weight tile per wave
  static constexpr int DMA_PER_TILE = 2 + W_PIECES;  // LDS-DMA instructions a wave issues per K-tile
};

template <int NJ>
MJV_DEV void stage(const GemmArgs& p, int m0, int n0, int k0, char* buf, int wave, int lane) {
  using C = Cfg<NJ>;
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // activation rows: 8 pieces of 8 rows, two per wave
    const int piece = wave * 2 + i;
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    int gr = m0 + r;
    gr = gr < p.M - 1 ? gr : p.M - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + (long)gr * p.lda + k0 + c * 8),
                                     (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < C::W_PIECES; ++i) {   // weight rows
    const int piece = wave * C::W_PIECES + i;
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    int gr = n0 + r;
    gr = gr < p.N - 1 ? gr : p.N - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.W + (long)gr * p.ldw + k0 + c * 8),
                                     (__attribute__((address_space(3))) void*)(buf + A_BYTES + piece * 1024), 16, 0, 0);
  }
}

template <int EPI, int NJ>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
  using C = Cfg<NJ>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tm = (int)blockIdx.x % p.tiles_m, tn = (int)blockIdx.x / p.tiles_m;
  const int m0 = tm * BM, n0 = tn * C::BN;
  const int nk = p.K / BK;
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage<NJ>(p, m0, n0, 0, smem, wave, lane);
  if (nk > 1) stage<NJ>(p, m0, n0, BK, smem + C::TILE_BYTES, wave, lane);
  for (int t = 0; t < nk; ++t) {
    // all but the youngest K-tile's LDS-DMA of this wave must have landed
    if (t + 1 < nk) {
      if constexpr (C::DMA_PER_TILE == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // K-tile t visible to every wave; buffer (t + 2) % 3 (read in iteration t - 1) is free
    if (t + 2 < nk) stage<NJ>(p, m0, n0, (t + 2) * BK, smem + ((t + 2) % NBUF) * C::TILE_BYTES, wave, lane);
    const char* As = smem + (t % NBUF) * C::TILE_BYTES;
    const char* Ws = As + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int g = kk * 4 + l4;
      const int ra = wave * 16 + l15;
      const bf16x8 af = *(const bf16x8*)(As + ra * 128 + ((g ^ (ra & 7)) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int rw = j * 16 + l15;
        const bf16x8 wf = *(const bf16x8*)(Ws + rw * 128 + ((g ^ (rw & 7)) << 4));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[j], 0, 0, 0);
      }
    }
  }
  const int mrel = m0 + wave * 16 + l15;
  if constexpr (EPI == MJV_EPI_SILU_MUL) {
    if (mrel >= p.M) return;
    const long orow = out_row_of(p, p.m_base + mrel);
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
      const int n = n0 + j * 16 + l4 * 4;
      if (n < p.N) store_silu(p, acc[j], acc[j + 1], orow, n0 / 2 + (j / 2) * 16 + l4 * 4, p.row_scale ? p.row_scale[p.m_base + mrel] : 1.f);
    }
  } else {
    const int mr[1] = {mrel};
    int ncol[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ncol[j] = n0 + j * 16 + l4 * 4;
    store_frags<EPI, 1, NJ>(p, &acc[0], mr, ncol);
  }
}
}  // namespace t64

// ============================================================================================ 256 x 256
namespace t256 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;   // 16 KiB: 128 rows x 64 k
constexpr int PIPE_BYTES = 8 * HALF_BYTES;  // 2 K-tiles x {W0, W1, A0, A1}
constexpr int EPI_PITCH = 256 * 2 + 16;     // bf16 output tile staged for coalesced stores: 528-B rows (conflict-free)
constexpr int EPI_TILE_BYTES = 256 * EPI_PITCH;
constexpr int GELU_BYTES = MJV_GELU_TABLE_LEN * 2;
constexpr int LDS_BYTES = EPI_TILE_BYTES + GELU_BYTES;  // 150 528 B >= PIPE_BYTES: one workgroup per CU either way (the table's
                                                        // place in the persistent kernel's map; the one-tile GELU kernel has its own, below)
static_assert(LDS_BYTES >= PIPE_BYTES && LDS_BYTES <= 160 * 1024 && EPI_TILE_BYTES % 16 == 0, "LDS budget");
// FUSE kernels (a norm folded into this GEMM, mjv.h "row_scale"): the tile's 256 row_scale / row_shift / col_shift / bias_f32
// floats, 1 KiB each, past the GELU table - fetched by LDS-DMA in the prologue (older than every K-tile DMA the counted waits
// count), read in pass A: no register lives through the main loop for them
constexpr int ST_OFF = LDS_BYTES;
constexpr int LDS_BYTES_FUSE = LDS_BYTES + 4096;
static_assert(ST_OFF % 16 == 0 && LDS_BYTES_FUSE <= 160 * 1024, "LDS budget (fused norm)");

// GELU kernels (round 4): the table's two sign halves sit 64 KiB apart - x > 0 at LDS byte 0, x < 0 at byte 65536 - so that
// the gather address is the rounded value's own bit pattern: (fp32 bits >> 15) - 2 LO = 65536 sign + 2 (magnitude - LO), two
// vector instructions where the contiguous table took five (magnitude, -LO, sign mask, & R, +) and a per-element range
// compare.  A 16-KiB pipeline slot is given up at each of the two addresses (the eight half-tiles use slots 1-3 and 5-9 of the
// ten the 160 KiB hold), the folded-norm vectors go behind the first table, and the staged output tile - which no longer has
// 132 contiguous KiB - is split into rows 0..100 below the second table and rows 101..255 above it.
constexpr int G_NEG = 65536;
constexpr int G_ST = 7680;                       // FUSE vectors (4 KiB), behind the 7 680-byte table half
constexpr int G_A0 = G_ST + 4096, G_AROWS = (G_NEG - G_A0) / EPI_PITCH;     // rows 0 .. G_AROWS - 1
constexpr int G_B0 = G_NEG + 7680;                                            // rows G_AROWS .. 255
constexpr int LDS_BYTES_G = 160 * 1024;
static_assert(MJV_GELU_NEG_OFF * 2 <= G_ST && G_A0 + G_AROWS * EPI_PITCH <= G_NEG && G_NEG + MJV_GELU_NEG_OFF * 2 <= G_B0 &&
              G_B0 + (256 - G_AROWS) * EPI_PITCH <= LDS_BYTES_G && G_A0 % 16 == 0 && G_B0 % 16 == 0, "GELU LDS map");
template <bool GL>
MJV_DEV int half_off(int idx) {   // LDS byte offset of pipeline half-tile idx = (K-tile & 1) * 4 + {W0, W1, A0, A1}
  return GL ? (idx + 1 + (idx >= 3 ? 1 : 0)) * HALF_BYTES : idx * HALF_BYTES;
}
template <bool GL>
MJV_DEV int erow_off(int ml) {    // LDS byte offset of row ml of the staged output tile
  return GL ? ml * EPI_PITCH + (ml < G_AROWS ? G_A0 : G_B0 - G_AROWS * EPI_PITCH) : ml * EPI_PITCH;
}
// general form of the table GELU on the split table (gelu_lut's logic; tab = LDS base)
MJV_DEV float gelu_lut_split(float xf, const char* lds) {
  const unsigned u = __float_as_uint(xf);
  const unsigned mag = (u >> 16) & 0x7fffu;
  const unsigned rel = mag - MJV_GELU_LO;
  const unsigned sgn = (unsigned)((int)u >> 31);
  const bool in_tab = rel < (unsigned)MJV_GELU_R;
  const unsigned t = *(const u16*)(lds + (sgn & G_NEG) + (in_tab ? rel * 2 : 0u));
  const unsigned big = gelu_beyond_table(u, mag);               // beyond the table (mjv_common.h)
  const unsigned small = __float_as_uint(0.5f * xf);
  const unsigned other = mag < MJV_GELU_LO ? small : big;
  return __uint_as_float(in_tab ? (t << 16) : other);
}

// per-lane global source pointers of the two 1-KiB DMA pieces a wave issues for each of the four half-tiles
// (W rows 0-127, W rows 128-255, A rows 0-127, A rows 128-255) at k = 0: the row clamp and the chunk swizzle are loop
// invariant, so a stage is two {pointer + k offset, global_load_lds} pairs - the DMA issue sits in the load segment that
// has to hide under the partner wave's 16-MFMA segment
struct StagePtrs {
  const u16* src[4][2];
};

MJV_DEV void init_stage_ptrs(StagePtrs& sp, const GemmArgs& p, int m0, int n0, int wave, int lane) {
#pragma unroll
  for (int which = 0; which < 4; ++which) {
    const bool is_w = which < 2;
    const u16* base = is_w ? p.W : p.A;
    const long ld = is_w ? p.ldw : p.lda;
    const int row0 = (is_w ? n0 : m0) + (which & 1) * 128;
    const int max_row = (is_w ? p.N : p.M) - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);
      int gr = row0 + r;
      gr = gr < max_row ? gr : max_row;
      sp.src[which][i] = base + (long)gr * ld + c * 8;
    }
  }
}

// stage half-tile WHICH of K-tile t (no-op past the last K-tile)
template <int WHICH, bool GL = false>
MJV_DEV void stage_half(const StagePtrs& sp, int t, int nk, char* smem, int wave) {
  if (t >= nk) return;
  char* dst = smem + half_off<GL>((t & 1) * 4 + WHICH) + wave * 2048;
  const int k0 = t * BK;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp.src[WHICH][i] + k0),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
}

#define MJV_BARRIER()                      \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

// VAR 0: production.  VAR 3 / 4: timing experiments (no epilogue / no global stores) used to attribute the epilogue's
// cost; selected with mjv_gemm_set_tile(1000 + VAR), never by the automatic path.
// Tried and rejected on this structure (A/B in one process, tools/gemm_bench.py; code in git history): delaying every
// other first-round workgroup by half a tile time so that the CUs' epilogue store bursts stop colliding (round 2:
// -1 ... -20 % on every model shape - the delay is never recovered, so the lockstep burst is not what the epilogue waits
// for); issuing the
// LDS-DMA in the middle of the MFMA segment instead of the load segment (-5 %); a persistent one-workgroup-per-CU form
// that prefetches the next tile's first K-tile under the epilogue (neutral: s_waitcnt vmcnt is in-order, the first DMA
// wait also waits for the epilogue's stores); a role-split persistent form where 4 waves issue all DMA and the other 4
// all global stores (-5...-35 %: spills + half-width pass B); touching the residual tile's 1024 cache lines (one dword each)
// three K-tiles before the end of the main loop so that pass B finds it in L2 (round 2: neutral on proj / fc2 / wo / w2).
// FUSE: 0 = plain; 1 = lin = row_scale[m] * acc (+ bias): an RMSNorm folded into this GEMM (gain in W, rstd here);
// 2 = lin = row_scale[m] * acc - row_shift[m] * col_shift[n] + bias_f32[n]: a LayerNorm folded into it
MJV_DEV void stage_fused_vectors(const GemmArgs& p, int fuse, int m0, int n0, char* smem, int wave, int lane, int st_off) {
  if (wave >= (fuse == 2 ? 4 : 1)) return;
  const float* src = wave == 0 ? p.row_scale + p.m_base + m0 : wave == 1 ? p.row_shift + p.m_base + m0 : wave == 2 ? p.col_shift + n0 : p.bias_f32 + n0;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                   (__attribute__((address_space(3))) void*)(smem + st_off + wave * 1024), 16, 0, 0);
}

template <int EPI, int VAR, int FUSE = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;   // wave owns output rows m: wr*128.., columns n: wc*64..
  const int l15 = lane & 15, l4 = lane >> 4;
  int tm, tn;
  // VAR 7: split-K slice of a tile (workgroup b = tile b / split, slice b % split), as in the 128-tile kernel: the
  // accumulators go to the fp32 workspace as they sit in registers, splitk_finish256_kernel sums them in slice order
  constexpr bool SPLIT = VAR == 7;
  constexpr bool GL = EPI == MJV_EPI_BIAS_GELU;      // the GELU kernels' LDS map (tables 64 KiB apart, see G_NEG)
  constexpr int ST = GL ? G_ST : ST_OFF;             // where the folded-norm vectors sit
  const int tile_id = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x;
  const int slice = SPLIT ? (int)blockIdx.x % p.split : 0;
  tile_of_vblock(p, SPLIT ? (int)gridDim.x / p.split : (int)gridDim.x, tile_id, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk_all = p.K / BK;
  const int kt0 = SPLIT ? (int)((long)nk_all * slice / p.split) : 0;          // this slice's K-tiles [kt0, kt0 + nk)
  const int nk = (SPLIT ? (int)((long)nk_all * (slice + 1) / p.split) : nk_all) - kt0;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane LDS byte offsets inside a half-tile (row pitch 128 B, chunk swizzle ^ (row & 7)); the row's low 3
  // bits are l15 & 7 for every fragment, so one XOR term serves all of them
  const int sw = l15 & 7;
  int a_off[2], w_off[2];  // [kk]
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_off[kk] = l15 * 128 + (((kk * 4 + l4) ^ sw) << 4);
    w_off[kk] = ((wc & 1) * 64 + l15) * 128 + (((kk * 4 + l4) ^ sw) << 4);
  }
  const int a_half = 2 + wr;   // this wave's activation half-tile index within a K-tile
  const int w_half = wc >> 1;  // this wave's weight half-tile index

  // VAR 6 (diagnostic build, tools/gemm_bench.py): s_memtime stamps of wave 0 -> p.ws as 8 x uint64 per workgroup
  // {start, prologue done, main loop done, pass A done, end}: the shares of a tile's life, never its length
  auto stamp = [&]() -> unsigned long long {
    if constexpr (VAR == 6) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      return t;
    } else {
      return 0ull;
    }
  };
  // (same build: the 100 MHz constant-rate counter beside the shader-clock one around the main loop -> the clock the chip
  // holds inside it = d[2] / d[6] x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6)
  auto stamp_rt = [&]() -> unsigned long long {
    if constexpr (VAR == 6) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      return t;
    } else {
      return 0ull;
    }
  };
  const unsigned long long t_start = stamp();
  // bias of this lane's 4 x 4 output columns: requested before the first DMA (oldest in the in-order vmcnt queue, so the
  // counted waits below mean what they meant), used in pass A - loaded there, each of the four loads was followed by its
  // own s_waitcnt vmcnt(0): four dependent L2 round trips, 2 k of pass A's 2.6 k cycles
  u32x2 braw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    braw[j] = u32x2{0u, 0u};
    const int nl = wc * 64 + j * 16 + l4 * 4;
    if (EPI != MJV_EPI_SILU_MUL && p.bias && n0 + nl < p.N) braw[j] = *(const u32x2*)(p.bias + n0 + nl);
  }
  // (LayerScale vector of this thread's pass-B columns: same treatment)
  constexpr int OUT_COLS_E = (EPI == MJV_EPI_SILU_MUL) ? 128 : 256;
  u32x4 scraw = {0u, 0u, 0u, 0u};
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
    const int ne = n0 + (tid % (OUT_COLS_E / 8)) * 8;
    if (p.scale && ne < p.N) scraw = *(const u32x4*)(p.scale + ne);
  }
  StagePtrs sp;
  init_stage_ptrs(sp, p, m0, n0, wave, lane);
  if constexpr (EPI == MJV_EPI_BIAS_GELU) {
    // the GELU table (15 KB) goes to its two LDS windows - pipeline slots 0 and 4, which the K-tiles never use - by LDS-DMA, issued
    // before the first K-tile (older in the vmcnt queue than everything the counted waits below count): copied at the head of the
    // epilogue it cost a global round trip and a barrier per tile
    // the two sign halves (MJV_GELU_NEG_OFF entries = 480 16-byte chunks each) to LDS bytes 0.. and 65536..: one
    // instruction per half and wave, older in the vmcnt queue than every K-tile DMA the counted waits count
    static_assert(MJV_GELU_NEG_OFF % 8 == 0 && MJV_GELU_NEG_OFF / 8 <= 8 * 64, "eight 64-lane instructions cover a table half");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = wave * 64 + lane;
      if (chunk < MJV_GELU_NEG_OFF / 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const u32x4*)(g_gelu_table + i * MJV_GELU_NEG_OFF) + chunk),
                                         (__attribute__((address_space(3))) void*)(smem + i * G_NEG + wave * 1024), 16, 0, 0);
    }
  }
  if constexpr (SPLIT) {
#pragma unroll
    for (int which = 0; which < 4; ++which)
#pragma unroll
      for (int i = 0; i < 2; ++i) sp.src[which][i] += kt0 * BK;
  }
  if constexpr (FUSE != 0) stage_fused_vectors(p, FUSE, m0, n0, smem, wave, lane, ST);
  // ---- prologue: K-tile 0 completely, W halves of K-tile 1
  stage_half<0, GL>(sp, 0, nk, smem, wave);
  stage_half<1, GL>(sp, 0, nk, smem, wave);
  stage_half<2, GL>(sp, 0, nk, smem, wave);
  stage_half<3, GL>(sp, 0, nk, smem, wave);
  stage_half<0, GL>(sp, 1, nk, smem, wave);
  stage_half<1, GL>(sp, 1, nk, smem, wave);
  // the 128 accumulator registers are zeroed HERE, under the first K-tile's flight time: left alone, the compiler sinks the
  // v_movs (256 of them: one set per side of the loop-entry branch) below the wait and the barriers, where they are 1-2 k
  // exposed cycles per tile
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
  if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MJV_BARRIER();
  if (wr == 1) MJV_BARRIER();  // stagger the second M-group by one barrier
  const unsigned long long t_pro = stamp();
  const unsigned long long r_pro = stamp_rt();

  bf16x8 af[4][2], wf[2][2][2];  // af[i][kk]; wf[ns][j][kk]

#define MJV_LOAD_A(MS)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
      af[i][kk] = *(const bf16x8*)(abase + ((MS) * 64 + i * 16) * 128 + a_off[kk]);
#define MJV_LOAD_W(NS)                                                                          \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
      wf[NS][j][kk] = *(const bf16x8*)(wbase + ((NS) * 32 + j * 16) * 128 + w_off[kk]);
#define MJV_MFMA_K(MS, NS, KK)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)             \
      acc[(MS) * 4 + i][(NS) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
          wf[NS][j][KK], af[i][KK], acc[(MS) * 4 + i][(NS) * 2 + j], 0, 0, 0);
#define MJV_MFMA(MS, NS)                               \
  __builtin_amdgcn_s_setprio(1);                       \
  MJV_MFMA_K(MS, NS, 0)                                \
  MJV_MFMA_K(MS, NS, 1)                                \
  __builtin_amdgcn_s_setprio(0);

  // Two phases per K-tile, four barriers (round 1 ran four phases of one 64 x 32 quadrant each, eight barriers: the same
  // reads, DMA and MFMAs cut into segments twice as long run +2 ... +7 % on every model shape - fewer barrier round trips,
  // and a 16-read segment has a 32-MFMA segment of the partner wave to hide under instead of 8 under 16).
  //   phase I : read A rows 0-63 and all 64 W columns of K-tile t | DMA the A halves of K-tile t+1 | barrier | 32 MFMA | barrier
  //   phase II: read A rows 64-127 | DMA the W halves of K-tile t+2 | retire K-tile t+1 (vmcnt) | barrier | 32 MFMA | barrier
  // Barrier intervals: M-group 0 reads in I_4t (phase I) and I_4t+2 (phase II); group 1 runs one barrier later.  The W halves
  // of K-tile t are dead after group 1's phase-I reads (I_4t+1), their buffer is refilled from I_4t+2 on; the A halves are
  // dead after group 1's phase-II reads (I_4t+3), refilled from I_4t+4 on.  K-tile t+1 is retired by every wave before the
  // barrier that ends its phase-II read segment (group 1: end of I_4t+3), one barrier before group 0 first reads it (I_4t+4).
  // (One phase per K-tile - all reads, 64 MFMAs, two barriers - fits in 250 VGPRs but needs the DMA issue and its retiring
  // wait placed per M-group to stay race-free, and then runs 10-20 % slower than this.  Giving the A halves one more barrier
  // interval between issue and retire - group 0 retiring after its phase-II MFMAs, group 1 issuing at the head of its
  // phase-II MFMA segment - measured 0 ... -4 %: the retiring wait is not what the loop waits for.)
  for (int t = 0; t < nk; ++t) {
    const char* abase = smem + half_off<GL>((t & 1) * 4 + a_half);
    const char* wbase = smem + half_off<GL>((t & 1) * 4 + w_half);
    MJV_LOAD_W(0)
    MJV_LOAD_W(1)
    MJV_LOAD_A(0)
    stage_half<2, GL>(sp, t + 1, nk, smem, wave);   // A rows 0-127 of K-tile t+1
    stage_half<3, GL>(sp, t + 1, nk, smem, wave);   // A rows 128-255 of K-tile t+1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MJV_BARRIER();
    MJV_MFMA(0, 0)
    MJV_MFMA(0, 1)
    MJV_BARRIER();
    MJV_LOAD_A(1)
    stage_half<0, GL>(sp, t + 2, nk, smem, wave);   // W rows 0-127 of K-tile t+2
    stage_half<1, GL>(sp, t + 2, nk, smem, wave);   // W rows 128-255 of K-tile t+2
    if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // everything but the 2 half-tiles of t+2 issued last
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MJV_BARRIER();
    MJV_MFMA(1, 1)
    MJV_MFMA(1, 0)
    MJV_BARRIER();
  }
  if (wr == 0) MJV_BARRIER();  // matches the stagger barrier of the second M-group
  // every load of this wave has landed (the last K-tile's vmcnt(0) above is inline assembly, which the compiler's wait-count
  // pass does not read): said once more in a form it does read, so that the first use of the bias registers in pass A does
  // not get a conservative vmcnt(0) of its own - that one would also wait for the residual rows requested just below, whose
  // flight time pass A is there to cover
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
  const unsigned long long t_main = stamp();
  const unsigned long long r_main = stamp_rt();
#undef MJV_LOAD_A
#undef MJV_LOAD_W
#undef MJV_MFMA
#undef MJV_MFMA_K

  if constexpr (SPLIT) {
    f32x4* img = (f32x4*)p.ws + ((long)tile_id * p.split + slice) * (32 * 512) + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) img[(i * 4 + j) * 512] = acc[i][j];
    return;
  }
  if constexpr (VAR == 3) {  // timing experiment: main loop only (accumulators kept alive, nothing stored)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
    return;
  }

  // ---- epilogue, two passes through LDS (the pipeline buffers are dead: every DMA was retired by the last
  // vmcnt(0) and the last ds_read is 5 barriers back):
  //  pass A (fragment layout: lane = output row m, 4 consecutive columns n): + bias, first bf16 rounding, then the
  //         per-element activation (GELU table / ReLU / SiLU*up) -> bf16 tile in LDS;
  //  pass B (row layout: 32 lanes x 16 B = one 512-B output row): LayerScale / residual, coalesced 16-B global
  //         loads and stores (an 8-B-per-lane fragment store touches 16 rows x 32 B per instruction and ran the
  //         K = 1024 GEMMs at half speed).
  char* etile = smem;
  // (the GELU table halves were copied to LDS bytes 0.. and 65536.. - around the pipeline buffers - by the prologue's first DMA instructions)
  constexpr int OUT_COLS = (EPI == MJV_EPI_SILU_MUL) ? 128 : 256;
  // pass B geometry: 32 (16 for SiLU) lanes x 16 B = one output row per pass
  constexpr int LANES_PER_ROW = OUT_COLS / 8;          // 16-B chunks per output row
  constexpr int ROWS_PER_PASS = 512 / LANES_PER_ROW;
  constexpr int PASSES = 256 / ROWS_PER_PASS;
  const int c8 = (tid % LANES_PER_ROW) * 8;
  const int nout0 = (EPI == MJV_EPI_SILU_MUL) ? n0 / 2 : n0;
  const int nlim = (EPI == MJV_EPI_SILU_MUL) ? p.N / 2 : p.N;
  const int n = nout0 + c8;
  const int ml0 = tid / LANES_PER_ROW;
  // residual rows of this thread's 16 output chunks: issued BEFORE pass A - the operand fragment registers of the main
  // loop (64 per lane) are dead, and s_memtime stamps put 13-19 k of a SCALE_RES tile's 15-22 k pass-B cycles on this
  // 128 KB-per-CU read arriving at the per-CU share of HBM bandwidth; pass A (2.7 k cycles) and the barrier now run under it
  u32x4 rsv[(EPI == MJV_EPI_SCALE_RES) ? PASSES : 1];
  float sc[8];
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
    unpack8(scraw, sc);
    if (p.res_mod > 0) {   // uniform: residual rows repeat with a period (position embeddings)
#pragma unroll
      for (int it = 0; it < PASSES; ++it) {
        const int ml = it * ROWS_PER_PASS + ml0;
        rsv[it] = u32x4{0u, 0u, 0u, 0u};
        if (m0 + ml < p.M && n < nlim) {
          const int m = p.m_base + m0 + ml;
          rsv[it] = *(const u32x4*)(p.res + (long)(p.res_off + (m % p.res_mod)) * p.ldr + n);
        }
      }
    } else {               // residual row == output row: one base pointer, constant row step
      const u16* rp = p.res + (long)(p.m_base + m0 + ml0) * p.ldr + n;
      const long rstep = (long)ROWS_PER_PASS * p.ldr;
#pragma unroll
      for (int it = 0; it < PASSES; ++it) {
        rsv[it] = u32x4{0u, 0u, 0u, 0u};
        // (nt LOADS for these once-read rows were measured in round 4 - inline asm + a hand-placed wait: proj +2.7 %, fc2 -0.7 %,
        // wo -5 %, w2 +0.7 %: not kept)
        if (m0 + it * ROWS_PER_PASS + ml0 < p.M && n < nlim) rsv[it] = *(const u32x4*)(rp + it * rstep);
      }
    }
  }
  // FUSE: this lane's eight rows' row_scale (and row_shift) out of the LDS copy; the Linear's value before its rounding is
  // lin(i, j, r) = acc * rs[i] + bq[r], bq = bias - row_shift[i] * col_shift (one more fma per element for a LayerNorm)
  float rs8[FUSE ? 8 : 1], rh8[FUSE == 2 ? 8 : 1];
  if constexpr (FUSE != 0) {
    const float* st = (const float*)(smem + ST);
#pragma unroll
    for (int i = 0; i < 8; ++i) rs8[i] = st[wr * 128 + i * 16 + l15];
    if constexpr (FUSE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rh8[i] = st[256 + wr * 128 + i * 16 + l15];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nl = wc * 64 + j * 16 + l4 * 4;   // column inside the 256-wide weight tile
    const u32x2 bb = braw[j];
    float b4[4] = {__uint_as_float(bb[0] << 16), __uint_as_float(bb[0] & 0xffff0000u),
                   __uint_as_float(bb[1] << 16), __uint_as_float(bb[1] & 0xffff0000u)};
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FUSE == 2) {
      const f32x4 bq = *(const f32x4*)(smem + ST + 3072 + nl * 4);
      c4 = *(const f32x4*)(smem + ST + 2048 + nl * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) b4[r] = bq[r];
    }
    auto lin = [&](int i, int jj, int r) __attribute__((always_inline)) -> float {
      if constexpr (FUSE == 0) return acc[i][jj][r] + b4[r];
      else if constexpr (FUSE == 1) return fmaf(acc[i][jj][r], rs8[i], b4[r]);
      else return fmaf(acc[i][jj][r], rs8[i], fmaf(-rh8[i], c4[r], b4[r]));
    };
    if (EPI == MJV_EPI_SILU_MUL && (j & 1)) continue;
    if constexpr (EPI == MJV_EPI_BIAS_GELU) {
      // GELU by table, a whole column group (8 fragments = 32 elements per lane) at a time: the indices of all 32 first, ONE
      // wave-wide range test, then 32 LDS gathers in flight together.  Fragment by fragment (round 2, first form) every
      // quad of gathers was followed by its own s_waitcnt lgkmcnt(0) and its own wave vote: 32 exposed LDS latencies per lane.
      // (round 4) the gather address is the rounded value's own bit pattern: (bits >> 15) - 2 LO = 65536 sign + 2 (|x| bits - LO).
      // Fast path = every element of every lane of the wave inside the table (|x| in [2^-23, 128): the table runs far past 5.56,
      // where GELU becomes x / -0, because a trained ViT has some |x| > 5.56 in nearly every 2048-element vote), decided from the
      // running min / max of |x| (the magnitudes' order is the patterns' order); otherwise the general form (x / 2 below the
      // table, x or -0 above).
      unsigned ubs[8][4];
      float amax = 0.f, amin = __uint_as_float(0x7f000000u);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xr = rbf(lin(i, j, r));
          ubs[i][r] = __float_as_uint(xr);
          // IEEE-754-2019 maximum / minimum (v_maximum3_f32 / v_minimum3_f32 on gfx950): a NaN operand makes the result NaN, so
          // a NaN pre-activation fails both range tests below and the vote takes the general form, which propagates it
          // (fmaxf / fminf DROP a NaN operand: ADVICE r4)
          amax = __builtin_elementwise_maximum(amax, fabsf(xr));
          amin = __builtin_elementwise_minimum(amin, fabsf(xr));
        }
      const bool all_in = amin >= __uint_as_float((unsigned)MJV_GELU_LO << 16) && amax < __uint_as_float((unsigned)MJV_GELU_HI << 16);
      if (__all(all_in)) {
        unsigned t[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) t[i][r] = *(const u16*)(smem + ((ubs[i][r] >> 15) - 2u * MJV_GELU_LO));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ml = wr * 128 + i * 16 + l15;
          // table entries are bf16 bit patterns: two of them side by side are the packed pair as it is
          const u32x2 o = {t[i][0] | (t[i][1] << 16), t[i][2] | (t[i][3] << 16)};
          *(u32x2*)(smem + erow_off<GL>(ml) + nl * 2) = o;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ml = wr * 128 + i * 16 + l15;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_lut_split(__uint_as_float(ubs[i][r]), smem);
          const u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
          *(u32x2*)(smem + erow_off<GL>(ml) + nl * 2) = o;
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ml = wr * 128 + i * 16 + l15;
      float v[4];
      int col;
      if constexpr (EPI == MJV_EPI_SILU_MUL) {
        const float rs = FUSE ? rs8[i] : 1.f;   // (no bias on this epilogue: the folded RMSNorm is the row factor alone)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(silu(rbf(FUSE ? acc[i][j][r] * rs : acc[i][j][r]))) * rbf(FUSE ? acc[i][j + 1][r] * rs : acc[i][j + 1][r]);
        col = wc * 32 + (j >> 1) * 16 + l4 * 4;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = lin(i, j, r);
        if constexpr (EPI == MJV_EPI_BIAS_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        col = nl;
      }
      const u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};   // one v_cvt_pk_bf16_f32 per pair
      *(u32x2*)(etile + erow_off<GL>(ml) + col * 2) = o;
    }
  }
  const unsigned long long t_passA = stamp();
  {
    if constexpr (EPI == MJV_EPI_ROPE_QKV) {
      // wqkv: the 256-column tile holds two whole 128-wide heads, so a row's rotate_half partner (column d +- 64 of the
      // same head) is in the staged tile.  q / k heads: out = bf16(bf16(x cos) + bf16(rot sin)) to the de-interleaved q / k
      // buffers; v heads: copied to their columns of C.  Same three roundings, same operation order as rope_split_kernel.
      // every row's position first (the accumulators are dead: their registers hold these loads across the barrier), then
      // the cos / sin rows of HALF the passes at a time, all in flight together - a position -> table-row -> use chain per
      // pass would put two dependent global-memory latencies on each of the 16 passes of the tile
      const int gs = (p.rope_group + 2) * 128;        // columns per kv group
      const int blk_col = c8 & 128;                    // which of the tile's two heads this thread works on
      const int ncol = n0 + blk_col;                   // first column of that head
      const int grp = ncol / gs, within = (ncol - grp * gs) >> 7;   // kv group, head slot inside it (uniform per half-row)
      const int dcol = c8 & 127;                       // column inside the head
      const bool is_v = within == p.rope_group + 1, is_k = within == p.rope_group;
      const bool col_ok = n < nlim;
      // unconditional loads from clamped rows: behind a per-row condition every one of the 16 loads got its own branch and
      // its own s_waitcnt vmcnt(0) - sixteen dependent L2 round trips before the barrier
      int posv[PASSES];
      const int* pos_base = p.rope_pos + (long)p.m_base;
#pragma unroll
      for (int it = 0; it < PASSES; ++it) {
        const int mr = m0 + it * ROWS_PER_PASS + ml0;
        posv[it] = pos_base[mr < p.M ? mr : p.M - 1];
      }
      __syncthreads();
      const float sgn = dcol < 64 ? -1.f : 1.f;        // rotate_half: (-x2, x1)
      constexpr int HALF = PASSES / 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 cs_raw[HALF], sn_raw[HALF];
        if (!is_v) {
#pragma unroll
          for (int i = 0; i < HALF; ++i) {
            cs_raw[i] = *(const u32x4*)(p.rope_cos + (long)posv[h * HALF + i] * 128 + dcol);
            sn_raw[i] = *(const u32x4*)(p.rope_sin + (long)posv[h * HALF + i] * 128 + dcol);
          }
        }
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const int it = h * HALF + i;
          const int ml = it * ROWS_PER_PASS + ml0;
          if (m0 + ml >= p.M || !col_ok) continue;
          const long m = (long)p.m_base + m0 + ml;
          const u32x4 val = *(const u32x4*)(etile + ml * EPI_PITCH + c8 * 2);
          if (is_v) {
            *(u32x4*)(p.C + m * p.ldc + n) = val;
            continue;
          }
          const u32x4 par = *(const u32x4*)(etile + ml * EPI_PITCH + (c8 ^ 64) * 2);
          float x[8], y[8], cs[8], sn[8], o[8];
          unpack8(val, x);
          unpack8(par, y);
          unpack8(cs_raw[i], cs);
          unpack8(sn_raw[i], sn);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rbf(x[e] * cs[e]) + rbf((sgn * y[e]) * sn[e]);
          u16* dst = is_k ? p.rope_k + m * p.rope_ldk + (long)grp * 128 + dcol
                          : p.rope_q + m * p.rope_ldq + ((long)grp * p.rope_group + within) * 128 + dcol;
          *(u32x4*)dst = pack8(o);
        }
      }
      return;
    }
    __syncthreads();
    // plain row mapping (output row == residual row == m) is the common case: keep the integer divisions of the
    // row maps (CLS slot / pos-emb period / <IMG_CONTEXT> scatter) out of it
    const bool plain = !p.out_rows && p.out_group <= 0 && p.res_mod <= 0;
    u16* crow = p.C + (long)(p.m_base + m0 + ml0) * p.ldc + n;
    const long cstep = (long)ROWS_PER_PASS * p.ldc;
    // all of this thread's rows out of the staged tile first (the accumulator registers are free): one row at a time the
    // loop was {ds_read, s_waitcnt lgkmcnt(0), store} sixteen times, a full LDS latency per row
    u32x4 vals[PASSES];
#pragma unroll
    for (int it = 0; it < PASSES; ++it) vals[it] = *(const u32x4*)(etile + erow_off<GL>(it * ROWS_PER_PASS + ml0) + c8 * 2);

#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int ml = it * ROWS_PER_PASS + ml0;
      if (m0 + ml >= p.M || n >= nlim) continue;
      u32x4 val = vals[it];
      if constexpr (EPI == MJV_EPI_SCALE_RES) {
        float v[8], rs[8];
        unpack8(val, v);
        if (p.scale) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] * sc[e]);
        }
        unpack8(rsv[it], rs);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rs[e];
        val = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
      }
      u16* dst = plain ? crow + it * cstep : p.C + out_row_of(p, p.m_base + m0 + ml) * p.ldc + n;
      if constexpr (VAR == 4) {
        asm volatile("" ::"v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(dst));
      } else {
        // streaming (nt) stores for the short-K GEMMs with large outputs: their store bursts otherwise evict the
        // operand panels the XCD's CUs share from its 4 MiB L2 (measured +5-7 % on the ViT K = 1024 GEMMs, neutral
        // to slightly negative on the LLM shapes, which therefore keep plain stores)
        store16(dst, val, p.nt_store);
      }
    }
  }
  if constexpr (VAR == 6) {
    const unsigned long long t_end = stamp();
    if (p.ws && tid == 0) {
      unsigned long long* d = (unsigned long long*)p.ws + (long)blockIdx.x * 8;
      d[0] = t_start; d[1] = t_pro - t_start; d[2] = t_main - t_pro; d[3] = t_passA - t_main; d[4] = t_end - t_passA;
      d[5] = t_end - t_start; d[6] = r_main - r_pro;
    }
  }
}

// Second half of a 256-tile split-K launch: 8 workgroups per tile (workgroup = 32 output rows of each of the tile's 8
// waves' row blocks, i.e. fragment row i of every thread of the GEMM workgroup) so that the fp32 slices - a few tens of MB -
// are read by a few hundred workgroups, not by one per tile; per fragment, sums the slices' register images in slice order
// (deterministic) and runs the per-fragment epilogue of the small kernels.
template <int EPI>
__global__ __launch_bounds__(512) void splitk_finish256_kernel(GemmArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tile_id = (int)blockIdx.x >> 3, i = (int)blockIdx.x & 7;
  int tm, tn;
  tile_of_vblock(p, (int)gridDim.x >> 3, tile_id, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const f32x4* img = (const f32x4*)p.ws + (long)tile_id * p.split * (32 * 512) + tid;
  f32x4 fr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fr[j] = img[(i * 4 + j) * 512];
  for (int sl = 1; sl < p.split; ++sl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fr[j] += img[(long)sl * (32 * 512) + (i * 4 + j) * 512];
  }
  const int mrel = m0 + wr * 128 + i * 16 + l15;
  if constexpr (EPI == MJV_EPI_SILU_MUL) {
    if (mrel >= p.M) return;
    const long orow = out_row_of(p, p.m_base + mrel);
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const int n = n0 + wc * 64 + j * 16 + l4 * 4;
      if (n >= p.N) continue;
      store_silu(p, fr[j], fr[j + 1], orow, n0 / 2 + wc * 32 + (j / 2) * 16 + l4 * 4, p.row_scale ? p.row_scale[p.m_base + mrel] : 1.f);
    }
  } else {
    const int mr[1] = {mrel};
    int ncol[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ncol[j] = n0 + wc * 64 + j * 16 + l4 * 4;
    store_frags<EPI, 1, 4>(p, &fr[0], mr, ncol);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// gemm256p_kernel (round 3): the same 256 x 256 tile, main loop and epilogue arithmetic as gemm256_kernel, PERSISTENT - one
// workgroup per CU walks the tiles b, b + grid, b + 2 grid ... of the XCD-aware order - so that a tile's fixed costs stop
// being paid in series:
//  * the next tile's first K-tile (64 KiB: all four half-tiles) is requested right after the main loop, into pipeline set 0,
//    and lands under the epilogue; the next main loop starts from a resident K-tile instead of a 3.5-5 k cycle prologue
//    (workgroup start, 12 DMA issues, an L2 / fabric round trip with every CU doing the same at once);
//  * the epilogue no longer owns all of LDS: it runs in two row halves of 128 x 256 through a 66 KiB window over pipeline set 1
//    (dead after the main loop; set 0 is where the prefetch lands), pass A by the four waves that own the rows, pass B by all;
//  * the stores of tile i drain under the main loop of tile i + 1 (nothing waits for them until the first counted wait there),
//    the GELU table is loaded once per workgroup, kernel-argument / launch costs once per CU.
// Requires an even number of K-tiles >= 4 (the last K-tile then sits in set 1); other problems use gemm256_kernel.
constexpr int EPI_HALF_BYTES = 128 * EPI_PITCH;                       // 66 KiB window for one 128-row half of the output tile
constexpr int EPI_WIN_OFF = 4 * HALF_BYTES;                           // = pipeline set 1
constexpr int GELU_OFF_P = EPI_WIN_OFF + EPI_HALF_BYTES;              // the table sits past the window, untouched by the loops
static_assert(GELU_OFF_P + GELU_BYTES <= LDS_BYTES, "persistent layout fits the kernel's LDS allocation");

// FUSE (see gemm256_kernel): the tile's row / column vectors go to one of TWO 4-KiB LDS areas (tile parity): those of the NEXT
// tile are requested just before its prefetched K-tile 0 - older in the queue than everything the counted waits leave in flight -
// while the current tile's pass A still reads its own.
constexpr int LDS_BYTES_FUSE_P = LDS_BYTES + 8192;
static_assert(LDS_BYTES_FUSE_P <= 160 * 1024, "LDS budget (fused norm, persistent)");

template <int EPI, int FUSE = 0>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int n_tiles = p.tiles_m * p.tiles_n;
  const int nk = p.K / BK;

  const int sw = l15 & 7;
  int a_off[2], w_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_off[kk] = l15 * 128 + (((kk * 4 + l4) ^ sw) << 4);
    w_off[kk] = ((wc & 1) * 64 + l15) * 128 + (((kk * 4 + l4) ^ sw) << 4);
  }
  const int a_half = 2 + wr;
  const int w_half = wc >> 1;

  if constexpr (EPI == MJV_EPI_BIAS_GELU) {   // once per workgroup; older than every DMA the counted waits count
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = (i * 8 + wave) * 64 + lane;
      if (chunk < MJV_GELU_TABLE_LEN / 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const u32x4*)g_gelu_table + chunk),
                                         (__attribute__((address_space(3))) void*)(smem + GELU_OFF_P + (i * 8 + wave) * 1024), 16, 0, 0);
    }
  }

  constexpr int OUT_COLS = (EPI == MJV_EPI_SILU_MUL) ? 128 : 256;
  constexpr int LANES_PER_ROW = OUT_COLS / 8;
  constexpr int ROWS_PER_PASS = 512 / LANES_PER_ROW;
  constexpr int HPASSES = 128 / ROWS_PER_PASS;          // pass-B passes per row half
  const int c8 = (tid % LANES_PER_ROW) * 8;
  const int ml0 = tid / LANES_PER_ROW;
  const int nlim = (EPI == MJV_EPI_SILU_MUL) ? p.N / 2 : p.N;
  char* const ewin = smem + EPI_WIN_OFF;
  const u16* gtab = (const u16*)(smem + GELU_OFF_P);

  bool prefetched = false;
  int stores_behind = 0;   // store instructions this wave issued after the prefetch of the tile about to start (a lower bound)
  int st_par = 0;          // which of the two vector areas the current tile uses (FUSE)

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int tm, tn;
    tile_of_vblock(p, n_tiles, tile, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int st_off = ST_OFF + st_par * 4096;
    if constexpr (FUSE != 0) {
      if (!prefetched) stage_fused_vectors(p, FUSE, m0, n0, smem, wave, lane, st_off);   // (first tile of this workgroup)
    }
    StagePtrs sp;   // (recomputed per tile: cheaper than 16 registers carried through the epilogue)
    init_stage_ptrs(sp, p, m0, n0, wave, lane);
    const int nout0 = (EPI == MJV_EPI_SILU_MUL) ? n0 / 2 : n0;
    const int n = nout0 + c8;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x2 braw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      braw[j] = u32x2{0u, 0u};
      const int nl = wc * 64 + j * 16 + l4 * 4;
      if (EPI != MJV_EPI_SILU_MUL && p.bias && n0 + nl < p.N) braw[j] = *(const u32x2*)(p.bias + n0 + nl);
    }
    u32x4 scraw = {0u, 0u, 0u, 0u};
    if constexpr (EPI == MJV_EPI_SCALE_RES) {
      const int ne = n0 + (tid % (OUT_COLS / 8)) * 8;
      if (p.scale && ne < p.N) scraw = *(const u32x4*)(p.scale + ne);
    }

    // ---- prologue: K-tile 0 (already requested under the previous tile's epilogue, or requested here), W halves of K-tile 1
    if (!prefetched) {
      stage_half<0>(sp, 0, nk, smem, wave);
      stage_half<1>(sp, 0, nk, smem, wave);
      stage_half<2>(sp, 0, nk, smem, wave);
      stage_half<3>(sp, 0, nk, smem, wave);
    }
    stage_half<0>(sp, 1, nk, smem, wave);
    stage_half<1>(sp, 1, nk, smem, wave);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
    // K-tile 0 must have landed.  vmcnt counts loads, stores and DMA together, in issue order: behind the prefetched K-tile 0
    // this wave has issued the previous tile's pass-B stores (2 * HPASSES of them when that tile was interior - every store
    // instruction executed) and the 4 DMA instructions just above; allowing exactly that many to stay outstanding waits for
    // K-tile 0 WITHOUT waiting for the stores to drain (a smaller count is always safe, a larger one never is)
    if (stores_behind == 2 * HPASSES) {
      if constexpr (HPASSES == 8) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    MJV_BARRIER();
    if (wr == 1) MJV_BARRIER();  // stagger the second M-group by one barrier

    bf16x8 af[4][2], wf[2][2][2];
#define MJV_LOAD_A(MS)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
      af[i][kk] = *(const bf16x8*)(abase + ((MS) * 64 + i * 16) * 128 + a_off[kk]);
#define MJV_LOAD_W(NS)                                                                          \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
      wf[NS][j][kk] = *(const bf16x8*)(wbase + ((NS) * 32 + j * 16) * 128 + w_off[kk]);
#define MJV_MFMA_K(MS, NS, KK)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)             \
      acc[(MS) * 4 + i][(NS) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
          wf[NS][j][KK], af[i][KK], acc[(MS) * 4 + i][(NS) * 2 + j], 0, 0, 0);
#define MJV_MFMA(MS, NS)                               \
  __builtin_amdgcn_s_setprio(1);                       \
  MJV_MFMA_K(MS, NS, 0)                                \
  MJV_MFMA_K(MS, NS, 1)                                \
  __builtin_amdgcn_s_setprio(0);

#ifdef MJV_BENCH   // bench build, variant 1008: shader-clock and 100 MHz stamps around this tile's main loop (wave 0)
    unsigned long long bt0 = 0, br0 = 0;
    if (p.ws) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(bt0), "=s"(br0)::"memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    for (int t = 0; t < nk; ++t) {   // (the two-phase loop of gemm256_kernel, unchanged)
      const char* abase = smem + ((t & 1) * 4 + a_half) * HALF_BYTES;
      const char* wbase = smem + ((t & 1) * 4 + w_half) * HALF_BYTES;
      MJV_LOAD_W(0)
      MJV_LOAD_W(1)
      MJV_LOAD_A(0)
      stage_half<2>(sp, t + 1, nk, smem, wave);
      stage_half<3>(sp, t + 1, nk, smem, wave);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MJV_BARRIER();
      MJV_MFMA(0, 0)
      MJV_MFMA(0, 1)
      MJV_BARRIER();
      MJV_LOAD_A(1)
      stage_half<0>(sp, t + 2, nk, smem, wave);
      stage_half<1>(sp, t + 2, nk, smem, wave);
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MJV_BARRIER();
      MJV_MFMA(1, 1)
      MJV_MFMA(1, 0)
      MJV_BARRIER();
    }
    if (wr == 0) MJV_BARRIER();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) in the form the compiler's wait-count pass reads (see gemm256_kernel)
#ifdef MJV_BENCH
    if (p.ws) {
      unsigned long long bt1, br1;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(bt1), "=s"(br1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (tid == 0) {
        unsigned long long* d = (unsigned long long*)p.ws + (long)blockIdx.x * 8;
        d[2] += bt1 - bt0; d[6] += br1 - br0; d[7] += 1;
      }
    }
#endif
#undef MJV_LOAD_A
#undef MJV_LOAD_W
#undef MJV_MFMA
#undef MJV_MFMA_K

    // ---- residual rows of this thread's 2 x HPASSES output chunks (issued before the prefetch: their waits must not wait for it)
    const bool plain = !p.out_rows && p.out_group <= 0 && p.res_mod <= 0;
    // (one row half at a time: 32 registers instead of 64; the second half's rows are requested when the first half's pass B
    // has consumed its own - by then the prefetch, older in the queue, has landed)
    u32x4 rsv[(EPI == MJV_EPI_SCALE_RES) ? 2 * HPASSES : 1];
    float sc[8];
    // window row w of half hf <-> tile row (w < 64 ? 0 : 128) + 64 hf + (w & 63): both M-groups of waves take part in every half
    auto tile_row = [&](int hf, int w) { return ((w >> 6) << 7) + hf * 64 + (w & 63); };
    if constexpr (EPI == MJV_EPI_SCALE_RES) {
      unpack8(scraw, sc);
#pragma unroll
      for (int it = 0; it < 2 * HPASSES; ++it) {
        const int ml = tile_row(it / HPASSES, (it % HPASSES) * ROWS_PER_PASS + ml0);
        rsv[it] = u32x4{0u, 0u, 0u, 0u};
        if (m0 + ml < p.M && n < nlim) {
          const int m = p.m_base + m0 + ml;
          const long rrow = p.res_mod > 0 ? (long)(p.res_off + (m % p.res_mod)) : (long)m;
          rsv[it] = *(const u32x4*)(p.res + rrow * p.ldr + n);
        }
      }
    }

    // ---- the next tile's first K-tile: requested now (set 0 is dead: its last reader was K-tile nk - 2), lands under the epilogue
    const int next = tile + (int)gridDim.x;
    prefetched = false;
    // (interior tile: every lane of every pass-B store instruction is inside the problem, so all 2 * HPASSES execute)
    stores_behind = (m0 + BM <= p.M && n0 + BN <= p.N) ? 2 * HPASSES : 0;
    if (next < n_tiles) {
      int tm2, tn2;
      tile_of_vblock(p, n_tiles, next, tm2, tn2);
      StagePtrs sp_next;
      init_stage_ptrs(sp_next, p, tm2 * BM, tn2 * BN, wave, lane);
      if constexpr (FUSE != 0) stage_fused_vectors(p, FUSE, tm2 * BM, tn2 * BN, smem, wave, lane, ST_OFF + (st_par ^ 1) * 4096);
      stage_half<0>(sp_next, 0, nk, smem, wave);
      stage_half<1>(sp_next, 0, nk, smem, wave);
      stage_half<2>(sp_next, 0, nk, smem, wave);
      stage_half<3>(sp_next, 0, nk, smem, wave);
      prefetched = true;
    }

    // ---- epilogue in two row halves through the window over set 1
    float rs8[FUSE ? 8 : 1], rh8[FUSE == 2 ? 8 : 1];
    if constexpr (FUSE != 0) {
      const float* st = (const float*)(smem + st_off);
#pragma unroll
      for (int i = 0; i < 8; ++i) rs8[i] = st[wr * 128 + i * 16 + l15];
      if constexpr (FUSE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rh8[i] = st[256 + wr * 128 + i * 16 + l15];
      }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      {   // pass A by all eight waves: each wave's fragments 4 hf .. 4 hf + 3 (its rows 64 hf .. 64 hf + 63) -> window rows 64 wr ..
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = wc * 64 + j * 16 + l4 * 4;
          const u32x2 bb = braw[j];
          float b4[4] = {__uint_as_float(bb[0] << 16), __uint_as_float(bb[0] & 0xffff0000u),
                         __uint_as_float(bb[1] << 16), __uint_as_float(bb[1] & 0xffff0000u)};
          f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (FUSE == 2) {
            const f32x4 bq = *(const f32x4*)(smem + st_off + 3072 + nl * 4);
            c4 = *(const f32x4*)(smem + st_off + 2048 + nl * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = bq[r];
          }
          auto lin = [&](int i, int jj, int r) __attribute__((always_inline)) -> float {   // i = fragment 0 .. 7 of the wave
            if constexpr (FUSE == 0) return acc[i][jj][r] + b4[r];
            else if constexpr (FUSE == 1) return fmaf(acc[i][jj][r], rs8[i], b4[r]);
            else return fmaf(acc[i][jj][r], rs8[i], fmaf(-rh8[i], c4[r], b4[r]));
          };
          if (EPI == MJV_EPI_SILU_MUL && (j & 1)) continue;
          if constexpr (EPI == MJV_EPI_BIAS_GELU) {
            unsigned ubs[4][4], idx[4][4];
            bool all_in = true;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                ubs[i][r] = __float_as_uint(rbf(lin(4 * hf + i, j, r)));
                const unsigned rel = ((ubs[i][r] >> 16) & 0x7fffu) - MJV_GELU_LO;
                all_in = all_in && (rel < (unsigned)MJV_GELU_R);
                idx[i][r] = rel + (ubs[i][r] >> 31) * (unsigned)MJV_GELU_NEG_OFF;
              }
            if (__all(all_in)) {
              unsigned t[4][4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) t[i][r] = gtab[idx[i][r]];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const u32x2 o = {t[i][0] | (t[i][1] << 16), t[i][2] | (t[i][3] << 16)};
                *(u32x2*)(ewin + (wr * 64 + i * 16 + l15) * EPI_PITCH + nl * 2) = o;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_lut(__uint_as_float(ubs[i][r]), gtab);
                const u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                *(u32x2*)(ewin + (wr * 64 + i * 16 + l15) * EPI_PITCH + nl * 2) = o;
              }
            }
            continue;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v[4];
            int col;
            if constexpr (EPI == MJV_EPI_SILU_MUL) {
              const float rs = FUSE ? rs8[4 * hf + i] : 1.f;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                v[r] = rbf(silu(rbf(FUSE ? acc[4 * hf + i][j][r] * rs : acc[4 * hf + i][j][r]))) *
                       rbf(FUSE ? acc[4 * hf + i][j + 1][r] * rs : acc[4 * hf + i][j + 1][r]);
              col = wc * 32 + (j >> 1) * 16 + l4 * 4;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = lin(4 * hf + i, j, r);
              if constexpr (EPI == MJV_EPI_BIAS_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              col = nl;
            }
            const u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
            *(u32x2*)(ewin + (wr * 64 + i * 16 + l15) * EPI_PITCH + col * 2) = o;
          }
        }
      }
      __syncthreads();
      // pass B by all eight waves: window rows it * ROWS_PER_PASS + ml0
      u32x4 vals[HPASSES];
#pragma unroll
      for (int it = 0; it < HPASSES; ++it) vals[it] = *(const u32x4*)(ewin + (it * ROWS_PER_PASS + ml0) * EPI_PITCH + c8 * 2);
#pragma unroll
      for (int it = 0; it < HPASSES; ++it) {
        const int ml = tile_row(hf, it * ROWS_PER_PASS + ml0);
        if (m0 + ml >= p.M || n >= nlim) continue;
        u32x4 val = vals[it];
        if constexpr (EPI == MJV_EPI_SCALE_RES) {
          float v[8], rs[8];
          unpack8(val, v);
          if (p.scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] * sc[e]);
          }
          unpack8(rsv[hf * HPASSES + it], rs);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rs[e];
          val = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        }
        u16* dst = p.C + (plain ? (long)(p.m_base + m0 + ml) : out_row_of(p, p.m_base + m0 + ml)) * p.ldc + n;
        store16(dst, val, p.nt_store);
      }
      __syncthreads();   // the window is rewritten by the next half / refilled by the next tile's K-tile 1
    }
    st_par ^= 1;
  }
}

}  // namespace t256

thread_local int g_num_cus = 256;   // CUs of the device of the call in flight (mjv_device_cus), set in mjv_gemm_bf16

// Measurement knobs.  The product library (default build) has none: the values below are constants, the kernel variants that
// skip work (wrong results by construction) are not compiled, and what a caller may choose - the tile kernel, for parity
// tests - travels in the call's descriptor (mjv_gemm_desc.tile).  -DMJV_BENCH (make bench -> libmjv_hip_bench.so, loaded by
// tools/gemm_bench.py and friends) turns them into process-wide settings behind mjv_bench_gemm_set().
// ============================================================================================ 128 x 256, two workgroups per CU
// Round 6: the structural alternative to "one 256 x 256 workgroup owns the CU" for the SHORT-K Linears of the
// vision tower (qkv, fc1, proj: K = 1024 = 16 K-tiles of the 256^2 kernel, where the un-overlapped epilogue is a quarter to a
// third of a tile's life).  A 128 (activation rows) x 256 (weight rows) tile with HALF the accumulators' footprint per
// workgroup: 256 threads = 4 waves as 2 (M) x 2 (N), each wave a 64 x 128 sub-tile = the same 128 accumulator registers per lane,
// K-steps of 32 (one 16x16x32 MFMA deep) so that three pipeline stages are 72 KiB and TWO workgroups share a CU: one
// workgroup's epilogue (LDS round trip, table gathers, global stores) runs beside the other's main loop - the hardware
// dispatches a new workgroup whenever one retires, so the two drift apart by themselves.  Price: 1.5 x the L2 -> LDS fill per flop
// of the 256^2 tile.  LDS image per stage: A rows then W rows, 64-byte rows (32 k), 16-byte chunk c of row r at chunk
// c ^ ((-(r >> 2)) & 3) - the 16 lanes of a ds_read_b128 group then differ in (r & 3, chunk'), conflict-free by enumeration.
// Epilogues: bias / bias + GELU (a 6-KiB sub-range of the committed table, |x| in [2^-9, 8): below it GELU = x / 2, above it x or
// -0 for every bf16 input - checked exhaustively) / LayerScale + residual; plain output rows only.
namespace t2 {
constexpr int BM = 128, BN = 256, BK = 32, NST = 3;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES, PIPE = NST * STAGE;
constexpr int EP = BN * 2 + 16;                                       // staged output row pitch (528 B: conflict-free 8-B fragment writes)
static_assert(BM * EP <= PIPE, "the staged output tile reuses the pipeline buffers");
constexpr int GT_LO = 0x3B00, GT_HI = 0x4100, GT_R = GT_HI - GT_LO;   // |x| bit patterns the LDS copy of the table covers
constexpr int GT_BYTES = 2 * GT_R * 2;                                // both signs
static_assert(GT_LO >= MJV_GELU_LO && GT_HI <= MJV_GELU_HI && (GT_R * 2) % 1024 == 0 && ((GT_LO - MJV_GELU_LO) * 2) % 16 == 0, "sub-table");
constexpr int LDS_BYTES = PIPE + GT_BYTES;
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

MJV_DEV float gelu_sub(float xf, const u16* tab) {
  const unsigned u = __float_as_uint(xf);
  const unsigned mag = (u >> 16) & 0x7fffu, rel = mag - GT_LO;
  const bool in_tab = rel < (unsigned)GT_R;
  const unsigned t = tab[(u >> 31) * GT_R + (in_tab ? rel : 0u)];
  const unsigned other = mag < GT_LO ? __float_as_uint(0.5f * xf) : gelu_beyond_table(u, mag);
  return __uint_as_float(in_tab ? (t << 16) : other);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  int tm, tn;
  tile_of_block(p, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K / BK;

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias of this lane's 8 x 4 output columns, requested first (oldest entries of the in-order vmcnt queue)
  u32x2 braw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    braw[j] = u32x2{0u, 0u};
    const int n = n0 + wn * 128 + j * 16 + l4 * 4;
    if (p.bias && n < p.N) braw[j] = *(const u32x2*)(p.bias + n);
  }
  // this wave's six 1-KiB DMA pieces of a stage: pieces w, w + 4, ... of {A rows 0-127 (8 pieces), W rows 0-255 (16)}; a piece =
  // 16 rows x 64 B, lane -> (row = lane >> 2, LDS chunk = lane & 3), source chunk = LDS chunk ^ swizzle(row)
  const u16* src[6];
  int dst[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int piece = wave + 4 * q;
    const bool is_w = piece >= 8;
    const int r = (is_w ? piece - 8 : piece) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((-(r >> 2)) & 3);
    int gr = (is_w ? n0 : m0) + r;
    const int mx = (is_w ? p.N : p.M) - 1;
    gr = gr < mx ? gr : mx;
    src[q] = (is_w ? p.W + (long)gr * p.ldw : p.A + (long)gr * p.lda) + c * 8;
    dst[q] = (is_w ? A_BYTES + (piece - 8) * 1024 : piece * 1024);
  }
  auto stage = [&](int t) __attribute__((always_inline)) {
    char* base = smem + (t % NST) * STAGE;
#pragma unroll
    for (int q = 0; q < 6; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + t * BK),
                                       (__attribute__((address_space(3))) void*)(base + dst[q]), 16, 0, 0);
  };
  if constexpr (EPI == MJV_EPI_BIAS_GELU) {   // the table's sub-range behind the pipeline: 6 pieces of 1 KiB, older than every K-step DMA
#pragma unroll
    for (int q = 0; q < 6; ++q)
      if ((q & 3) == wave) {
        const int sgn = q / 3, part = q % 3;
        const u16* g = g_gelu_table + sgn * MJV_GELU_NEG_OFF + (GT_LO - MJV_GELU_LO) + part * 512 + lane * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + PIPE + q * 1024), 16, 0, 0);
      }
  }
  stage(0);
  if (nk > 1) stage(1);

  // per-lane fragment offsets inside a stage: rows of 64 B, chunk l4 ^ swizzle(row); (row >> 2) & 3 == (l15 >> 2) for every fragment
  const int fsw = (l4 ^ ((-(l15 >> 2)) & 3)) << 4;
  const int a_off = (wm * 64 + l15) * 64 + fsw;
  const int w_off = A_BYTES + (wn * 128 + l15) * 64 + fsw;

  for (int t = 0; t < nk; ++t) {
    // stage t has landed (this wave's pieces: everything but the 6 of stage t + 1), every wave is past its reads of stage t - 1
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MJV_BARRIER();
    if (t + 2 < nk) stage(t + 2);
    const char* sb = smem + (t % NST) * STAGE;
    bf16x8 af[4], wf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = *(const bf16x8*)(sb + w_off + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sb + a_off + i * 16 * 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), in a form the compiler's wait-count pass reads (bias registers)
  MJV_BARRIER();                        // every wave is done with the pipeline buffers: the staged tile takes their place

  // ---- pass A (fragment layout: lane = output row, 4 consecutive columns): + bias, first rounding, activation -> bf16 tile in LDS
  const u16* gtab = (const u16*)(smem + PIPE);
  float sc4[8][4];
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wn * 128 + j * 16 + l4 * 4;
      u32x2 sr = {0x3f803f80u, 0x3f803f80u};
      if (p.scale && n < p.N) sr = *(const u32x2*)(p.scale + n);
      sc4[j][0] = __uint_as_float(sr[0] << 16); sc4[j][1] = __uint_as_float(sr[0] & 0xffff0000u);
      sc4[j][2] = __uint_as_float(sr[1] << 16); sc4[j][3] = __uint_as_float(sr[1] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float b4[4] = {__uint_as_float(braw[j][0] << 16), __uint_as_float(braw[j][0] & 0xffff0000u),
                         __uint_as_float(braw[j][1] << 16), __uint_as_float(braw[j][1] & 0xffff0000u)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][j][r] + b4[r]);
      if constexpr (EPI == MJV_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_sub(v[r], gtab);
      } else if constexpr (EPI == MJV_EPI_SCALE_RES) {
        if (p.scale) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rbf(v[r] * sc4[j][r]);
        }
      }
      const int ml = wm * 64 + i * 16 + l15, nl = wn * 128 + j * 16 + l4 * 4;
      *(u32x2*)(smem + ml * EP + nl * 2) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
  }
  MJV_BARRIER();
  // ---- pass B (row layout: 32 lanes x 16 B = one 512-B output row): residual, coalesced 16-B stores
  const int c8 = (tid & 31) * 8, r0 = tid >> 5;     // 8 rows per pass, 16 passes
  const int n = n0 + c8;
  u32x4 rsv[16];
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int m = m0 + it * 8 + r0;
      rsv[it] = u32x4{0u, 0u, 0u, 0u};
      if (m < p.M && n < p.N) rsv[it] = *(const u32x4*)(p.res + (long)(p.m_base + m) * p.ldr + n);
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int ml = it * 8 + r0, m = m0 + ml;
    u32x4 o = *(const u32x4*)(smem + ml * EP + c8 * 2);
    if constexpr (EPI == MJV_EPI_SCALE_RES) {
      float a[8], b[8];
      unpack8(o, a);
      unpack8(rsv[it], b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
      o = pack8(a);
    }
    if (m < p.M && n < p.N) store16(p.C + (long)(p.m_base + m) * p.ldc + n, o, p.nt_store);
  }
}
}  // namespace t2

struct GemmTune {
  int gm = 0;               // group-M of the 256-tile order: 0 = by shape (pick_gm)
  int skinny_max_m = 128;   // problems with at most this many rows run on the 64 x 32 kernel
  int split_max = 8;        // cap on the K slices per tile
  int split_k = 1;          // split-K for under-filled 128-tile launches when the caller gives a workspace
  int variant = 0;          // 256-tile kernel variant (3: no epilogue, 4: no global stores, 6: s_memtime stamps)
  int split256_min_kt = 8;  // fewest K-tiles per slice of the K-sliced 256-tile launches
  int split256 = 1;         // under-filled problems with deep K run as K slices of 256 x 256 tiles
  int split256_min_nk = 64; // ... deep = at least this many 64-wide K-tiles (bench: 4400 + n)
  int nt = -1;              // streaming output stores: -1 = by shape, 0 / 1 = never / always
  void* stamp_buffer = nullptr;
};
#ifdef MJV_BENCH
GemmTune g_tune;
#define MJV_TUNE(f) (g_tune.f)
#else
constexpr GemmTune g_tune_const{};
#define MJV_TUNE(f) (g_tune_const.f)
#endif

template <int EPI>
int launch(GemmArgs a, hipStream_t s, bool big, bool skinny = false) {
  if (skinny) {
    // (a 64 x 128 tile of the same kernel for the ~1100-row tails of the language tower measured 8 % slower than the
    // 128 x 128 kernel with split-K: 3.67 vs 3.30 ms per step on the residual GEMMs; only NJ = 2 is instantiated)
    a.tiles_m = (a.M + t64::BM - 1) / t64::BM;
    a.tiles_n = (a.N + 31) / 32;
    hipLaunchKernelGGL((t64::gemm_skinny_kernel<EPI, 2>), dim3(a.tiles_m * a.tiles_n), dim3(256), t64::Cfg<2>::LDS_BYTES, s, a);
    return mjv_check_launch("gemm_bf16");
  }
  constexpr int LDS1 = EPI == MJV_EPI_BIAS_GELU ? t256::LDS_BYTES_G : t256::LDS_BYTES;   // (the GELU kernels' map uses all 160 KiB)
  // the dynamic-LDS limit is a per-device function attribute: set it once on every device this instantiation runs on
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)t128::gemm128_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, t128::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)t128::gemm128_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, t128::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
#ifdef MJV_BENCH
    (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<EPI, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
    (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
    (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<EPI, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
#endif
    (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<MJV_EPI_BIAS, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES);
    // (the persistent form is instantiated for the three epilogues that launch it - ADVICE r3: the LayerScale / GELU
    // instantiations were compiled, spilled 13 / 62 registers and were never launched or tested)
    if constexpr (EPI == MJV_EPI_BIAS || EPI == MJV_EPI_BIAS_RELU || EPI == MJV_EPI_SILU_MUL)
      (void)hipFuncSetAttribute((const void*)t256::gemm256p_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES);
    // a norm folded into the GEMM: LayerNorm into the ViT's qkv / fc1 (bias epilogues), RMSNorm into wqkv / w1|w3
    if constexpr (EPI == MJV_EPI_BIAS || EPI == MJV_EPI_BIAS_GELU)
      (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<EPI, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                EPI == MJV_EPI_BIAS_GELU ? t256::LDS_BYTES_G : t256::LDS_BYTES_FUSE);
    if constexpr (EPI == MJV_EPI_SILU_MUL || EPI == MJV_EPI_ROPE_QKV)
      (void)hipFuncSetAttribute((const void*)t256::gemm256_kernel<EPI, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES_FUSE);
    if constexpr (EPI == MJV_EPI_BIAS)
      (void)hipFuncSetAttribute((const void*)t256::gemm256p_kernel<EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES_FUSE_P);
    if constexpr (EPI == MJV_EPI_SILU_MUL)
      (void)hipFuncSetAttribute((const void*)t256::gemm256p_kernel<EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, t256::LDS_BYTES_FUSE_P);
    attr_done.fetch_or(bit, std::memory_order_release);   // racing first calls both set the attributes: idempotent
  }
  const int fuse = a.row_scale ? (a.row_shift ? 2 : 1) : 0;
  if (big && a.split <= 1 && fuse) {
    // the 256 x 256 kernels instantiate the folded-norm epilogue where the model uses it; the small kernels and the K-sliced
    // finish take every combination (store_frags)
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    const int tiles = a.tiles_m * a.tiles_n, nk = a.K / 64;
    const bool persistent = tiles > g_num_cus && nk >= 4 && (nk & 1) == 0;
    constexpr int WANT = (EPI == MJV_EPI_BIAS || EPI == MJV_EPI_BIAS_GELU) ? 2 : (EPI == MJV_EPI_SILU_MUL || EPI == MJV_EPI_ROPE_QKV) ? 1 : 0;
    if constexpr (WANT != 0) {
      if (fuse == WANT) {
        if constexpr (EPI == MJV_EPI_BIAS || EPI == MJV_EPI_SILU_MUL) {
          if (persistent) {
            hipLaunchKernelGGL((t256::gemm256p_kernel<EPI, WANT>), dim3(g_num_cus), dim3(512), t256::LDS_BYTES_FUSE_P, s, a);
            return mjv_check_launch("gemm_bf16");
          }
        }
        hipLaunchKernelGGL((t256::gemm256_kernel<EPI, 0, WANT>), dim3(tiles), dim3(512),
                           EPI == MJV_EPI_BIAS_GELU ? t256::LDS_BYTES_G : t256::LDS_BYTES_FUSE, s, a);
        return mjv_check_launch("gemm_bf16");
      }
    }
    mjv_set_error("gemm: the 256-tile kernel has no instantiation for epilogue %d with %s folded in (bias / bias+GELU take a "
                  "LayerNorm: row_scale + row_shift; SiLU-mul / RoPE-qkv an RMSNorm: row_scale only)", (int)EPI,
                  fuse == 2 ? "row_scale + row_shift" : "row_scale");
    return MJV_E_UNSUPPORTED;
  }
  if (big) {
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    if (a.split > 1) {
      const int tiles = a.tiles_m * a.tiles_n;
      hipLaunchKernelGGL((t256::gemm256_kernel<MJV_EPI_BIAS, 7>), dim3(tiles * a.split), dim3(512), t256::LDS_BYTES, s, a);
      hipLaunchKernelGGL(t256::splitk_finish256_kernel<EPI>, dim3(tiles * 8), dim3(512), 0, s, a);
    }
#ifdef MJV_BENCH
    else if (MJV_TUNE(variant) == 6) {
      a.ws = (float*)MJV_TUNE(stamp_buffer);
      hipLaunchKernelGGL((t256::gemm256_kernel<EPI, 6>), dim3(a.tiles_m * a.tiles_n), dim3(512), LDS1, s, a);
    } else if (MJV_TUNE(variant) == 4)
      hipLaunchKernelGGL((t256::gemm256_kernel<EPI, 4>), dim3(a.tiles_m * a.tiles_n), dim3(512), LDS1, s, a);
    else if (MJV_TUNE(variant) == 3)
      hipLaunchKernelGGL((t256::gemm256_kernel<EPI, 3>), dim3(a.tiles_m * a.tiles_n), dim3(512), LDS1, s, a);
#endif
    else {
      // persistent form (one workgroup per CU walks the tiles, next tile's first K-tile prefetched under the epilogue) when the
      // launch has more tiles than CUs and an even number (>= 4) of K-tiles; the rotary epilogue keeps the one-tile kernel
      const int tiles = a.tiles_m * a.tiles_n, nk = a.K / 64;
      // (measured per epilogue, tools/gemm_bench.py 1000 / 1009 in one process: +1.4 ... +2.1 % for the plain-bias and SiLU
      // epilogues; the LayerScale / residual epilogue loses 6-24 % - its 64 residual registers on top of the persistent
      // loop's state spill - and the GELU epilogue 29 % - half the table gathers in flight per vote - so those keep the
      // one-tile kernel)
      bool persistent = (EPI == MJV_EPI_BIAS || EPI == MJV_EPI_BIAS_RELU || EPI == MJV_EPI_SILU_MUL) && tiles > g_num_cus &&
                        nk >= 4 && (nk & 1) == 0;
#ifdef MJV_BENCH
      if (MJV_TUNE(variant) == 9) persistent = false;   // A/B: the one-tile-per-workgroup kernel
      a.ws = MJV_TUNE(variant) == 8 ? (float*)MJV_TUNE(stamp_buffer) : nullptr;   // 1008: main-loop clock stamps (persistent form)
#endif
      if constexpr (EPI == MJV_EPI_BIAS || EPI == MJV_EPI_BIAS_RELU || EPI == MJV_EPI_SILU_MUL) {
        if (persistent) {
          hipLaunchKernelGGL((t256::gemm256p_kernel<EPI>), dim3(g_num_cus), dim3(512), t256::LDS_BYTES, s, a);
          return mjv_check_launch("gemm_bf16");
        }
      }
      hipLaunchKernelGGL((t256::gemm256_kernel<EPI, 0>), dim3(a.tiles_m * a.tiles_n), dim3(512), LDS1, s, a);
    }
  } else {
    a.tiles_m = (a.M + 127) / 128;
    a.tiles_n = (a.N + 127) / 128;
    const int tiles = a.tiles_m * a.tiles_n;
    if (a.split > 1) {
      hipLaunchKernelGGL((t128::gemm128_kernel<EPI, true>), dim3(tiles * a.split), dim3(256), t128::LDS_BYTES, s, a);
      hipLaunchKernelGGL(t128::splitk_finish_kernel<EPI>, dim3(tiles), dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((t128::gemm128_kernel<EPI, false>), dim3(tiles), dim3(256), t128::LDS_BYTES, s, a);
    }
  }
  return mjv_check_launch("gemm_bf16");
}

// the 128 x 256 two-per-CU kernel (tile code 2): K % 32 == 0, N % 8 == 0, plain output rows, bias / bias + GELU / LayerScale + residual
template <int EPI>
int launch_t2(GemmArgs a, hipStream_t s) {
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)t2::gemm2_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, t2::LDS_BYTES);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  a.tiles_m = (a.M + t2::BM - 1) / t2::BM;
  a.tiles_n = (a.N + t2::BN - 1) / t2::BN;
  hipLaunchKernelGGL((t2::gemm2_kernel<EPI>), dim3(a.tiles_m * a.tiles_n), dim3(256), t2::LDS_BYTES, s, a);
  return mjv_check_launch("gemm_bf16");
}

}  // namespace

// Group-M per shape class, from sweeps on the model's shapes (tools/gemm_bench.py 2003 ... 2016, best of 6 rounds, two
// boxes): the short-K vision GEMMs prefer 5-7 (qkv +2.8 %, proj +6 %, fc1 +2.5 % over 8 - with N = 1024 and GM = 8 a group
// is exactly the 32 tiles one XCD runs at a time, and all eight XCDs then walk the same four weight panels in step), the
// K = 8192 GEMM prefers 4 (+1.6 ... +3.8 %) and so does the 16384-column w1|w3 GEMM (+1.7 %), everything between is flat
// within 1 % and keeps 8.  (Powers of two - 4, 8 - are the worst choices for the K = 1024 shapes on both main loops.)
static int pick_gm(int N, int K) { return K <= 1024 ? 5 : (K >= 8192 || N >= 8192) ? 4 : 8; }

extern "C" int64_t mjv_gemm_workspace_bytes(void) { return 256L * 262144L; }

#ifdef MJV_BENCH
// measurement switches of the bench library (include/mjv_bench.h); process-wide, never part of the product build
extern "C" int mjv_bench_gemm_set(int32_t code) {
  if (code > 4100 && code <= 4108) { g_tune.split_max = code - 4100; return MJV_OK; }
  if (code >= 6000 && code <= 6512) { g_tune.skinny_max_m = code - 6000; return MJV_OK; }   // 6000 switches the skinny kernel off
  if (code > 4300 && code <= 4364) { g_tune.split256_min_kt = code - 4300; return MJV_OK; }
  if (code > 4400 && code <= 4528) { g_tune.split256_min_nk = code - 4400; return MJV_OK; }
  if (code == 4200 || code == 4201) { g_tune.split256 = code - 4200; return MJV_OK; }
  if (code == 4000 || code == 4001) { g_tune.split_k = code - 4000; return MJV_OK; }
  if (code >= 2000 && code < 2100) { g_tune.gm = code - 2000; return MJV_OK; }               // 2000: back to the per-shape choice
  if (code >= 1000 && code < 1010) { g_tune.variant = code - 1000; return MJV_OK; }           // 1000: production kernel
  if (code >= 7000 && code <= 7002) { g_tune.nt = code - 7001; return MJV_OK; }               // 7000: by shape, 7001: never, 7002: always
  if (code == 0) { g_tune = GemmTune{}; return MJV_OK; }
  mjv_set_error("bench_gemm_set: unknown code %d", code);
  return MJV_E_ARG;
}
// diagnostic (variant 1006): 8 x uint64 per workgroup of the following 256-tile launches go to this device buffer
extern "C" int mjv_bench_gemm_stamp_buffer(void* p) {
  g_tune.stamp_buffer = p;
  return MJV_OK;
}
#endif

extern "C" int mjv_gemm_bf16(const mjv_gemm_desc* d, void* stream) {
  MJV_REQUIRE(d && d->A && d->W && d->C, "gemm: null pointer");
  MJV_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  MJV_REQUIRE(d->tile == 0 || d->tile == 2 || d->tile == 64 || d->tile == 128 || d->tile == 256, "gemm: tile %d not in {0, 2, 64, 128, 256}", d->tile);
  if (d->a_format != MJV_FMT_BF16 || d->w_format != MJV_FMT_BF16) return mjv_gemm_mxfp8_dispatch(d, stream);   // gemm_fp8.hip
  MJV_REQUIRE(d->c_format == MJV_FMT_BF16, "gemm: an MXFP8 output needs MXFP8 operands");
  const int force_tile = d->tile;
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
  if (d->epilogue == MJV_EPI_ROPE_QKV) {
    // (rope_group 0, ABI 6: a k | v projection without q heads - the last decoder layer of a scorer needs q for a few rows only)
    MJV_REQUIRE(d->rope_cos && d->rope_sin && d->rope_pos && d->rope_k && d->rope_group >= 0 && (d->rope_q || d->rope_group == 0),
                "gemm: ROPE_QKV needs cos / sin / positions / k / group >= 0 (and q unless group == 0)");
    MJV_REQUIRE(d->N % ((d->rope_group + 2) * 128) == 0 && d->bias == nullptr && !d->out_rows && d->out_group <= 0,
                "gemm: ROPE_QKV needs N %% ((group + 2) * 128) == 0, no bias, plain output rows");
    MJV_REQUIRE(d->rope_ldq % 8 == 0 && d->rope_ldk % 8 == 0 && ((uintptr_t)d->rope_q | (uintptr_t)d->rope_k |
                (uintptr_t)d->rope_cos | (uintptr_t)d->rope_sin) % 16 == 0, "gemm: ROPE_QKV alignment");
  }
  GemmArgs a;
  a.A = d->A; a.lda = d->lda; a.W = d->W; a.ldw = d->ldw; a.C = d->C; a.ldc = d->ldc;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.bias = d->bias; a.scale = d->scale; a.res = d->res; a.ldr = d->ldr;
  a.res_mod = d->res_mod; a.res_off = d->res_off; a.out_group = d->out_group; a.out_pad = d->out_pad;
  a.out_rows = d->out_rows;
  a.tiles_m = a.tiles_n = 0;
  a.m_base = 0;
  a.ws = nullptr;
  a.split = 1;
  a.gm = MJV_TUNE(gm) > 0 ? MJV_TUNE(gm) : pick_gm(d->N, d->K);
  a.rope_cos = d->rope_cos; a.rope_sin = d->rope_sin; a.rope_pos = d->rope_pos; a.rope_q = d->rope_q; a.rope_k = d->rope_k;
  a.rope_ldq = d->rope_ldq; a.rope_ldk = d->rope_ldk; a.rope_group = d->rope_group;
  a.row_scale = d->row_scale; a.row_shift = d->row_shift; a.col_shift = d->col_shift; a.bias_f32 = d->bias_f32;
  if (d->row_scale) {
    MJV_REQUIRE((d->row_shift != nullptr) == (d->col_shift != nullptr) && (d->row_shift != nullptr) == (d->bias_f32 != nullptr),
                "gemm: row_shift, col_shift and bias_f32 go together (a folded LayerNorm) or not at all (a folded RMSNorm)");
    MJV_REQUIRE(!d->row_shift || !d->bias, "gemm: a folded LayerNorm carries its bias in bias_f32");
    MJV_REQUIRE(d->epilogue != MJV_EPI_SCALE_RES, "gemm: no norm is folded into the residual epilogue");
    MJV_REQUIRE(((uintptr_t)d->row_scale | (uintptr_t)d->row_shift | (uintptr_t)d->col_shift | (uintptr_t)d->bias_f32) % 16 == 0,
                "gemm: row / column vectors must be 16-byte aligned");
  } else {
    MJV_REQUIRE(!d->row_shift && !d->col_shift && !d->bias_f32, "gemm: row_shift / col_shift / bias_f32 need row_scale");
  }
  // streaming stores for every large output (round 4, once the stores really were nt - see store16: qkv +8 %, proj +4 %,
  // fc1 +3 %, every other model shape +0.3 ... +0.8 % in tools/gemm_bench.py 7001 7002; in the model, one box, two rounds:
  // never 93.5 ms per step, K <= 1024 only 92.6, always 92.3 - profiles/r04_g_nt_stores.txt); small outputs, which the next
  // kernel reads back from the L2, keep plain stores
  a.nt_store = ((double)d->M * (d->epilogue == MJV_EPI_SILU_MUL ? d->N / 2 : d->N) * 2.0 >= 32.0 * 1024 * 1024) ? 1 : 0;
  if (MJV_TUNE(nt) >= 0) a.nt_store = MJV_TUNE(nt);
  const bool big = (force_tile && force_tile != 2) ? force_tile == 256 : (d->M >= 512 && d->N >= 256);
  hipStream_t s = (hipStream_t)stream;
  g_num_cus = mjv_device_cus();   // tail peeling and split-K plan against the CUs of THIS device (partitioned parts differ)
  const double flops = 2.0 * d->M * (double)d->N * d->K;
  // (algorithmic bytes: A and W read once, the output written once - N / 2 columns for the SiLU-mul epilogue)
  const double bytes = 2.0 * ((double)d->M * d->K + (double)d->N * d->K + (double)d->M * (d->epilogue == MJV_EPI_SILU_MUL ? d->N / 2 : d->N));
  // Wave quantisation: with one 256x256 workgroup per CU a launch runs in ceil(tiles / 256) rounds, and a last round
  // that is mostly empty costs a full tile time (M = 17488, N = 2048: 552 tiles = 2.16 rounds -> 3).  When the last
  // round is under-filled, the trailing m-tile rows are peeled off and run as 128x128 tiles (2 workgroups per CU,
  // quarter-size work items) in a second launch on the same stream; rows are independent, so results are unchanged.
  int m_main = d->M;
  if (big && !force_tile && !d->out_rows) {
    const int tn = (d->N + 255) / 256, tmx = (d->M + 255) / 256;
    const int tiles = tn * tmx;
    const int rem = tiles % g_num_cus;
    const int rows_main_tiles = (tiles - rem) / tn;  // whole m-tile rows inside the full rounds
    if (tiles > g_num_cus && rem > 0 && rem * 4 <= g_num_cus * 2 && rows_main_tiles > 0) {
      m_main = rows_main_tiles * 256;
      if (m_main >= d->M) m_main = d->M;
    }
  }
  static const char* const tags256[] = {"gemm256_bias", "gemm256_bias_gelu", "gemm256_bias_relu", "gemm256_scale_res",
                                        "gemm256_silu_mul", "gemm256_rope_qkv"};
  static const char* const tags256s[] = {"gemm256s_bias", "gemm256s_bias_gelu", "gemm256s_bias_relu", "gemm256s_scale_res",
                                         "gemm256s_silu_mul"};
  static const char* const tags128[] = {"gemm128_bias", "gemm128_bias_gelu", "gemm128_bias_relu", "gemm128_scale_res",
                                        "gemm128_silu_mul"};
  if (d->epilogue < 0 || d->epilogue > MJV_EPI_ROPE_QKV) {
    mjv_set_error("gemm: unknown epilogue %d", d->epilogue);
    return MJV_E_ARG;
  }
  // tile 2: the 128 x 256 two-workgroups-per-CU kernel (short-K Linears; every row in ONE launch: its last m-tile may be partial)
  const bool t2_ok = (d->epilogue == MJV_EPI_BIAS || d->epilogue == MJV_EPI_BIAS_GELU || d->epilogue == MJV_EPI_SCALE_RES) && !d->out_rows &&
                     d->out_group <= 0 && d->res_mod <= 0 && !d->row_scale && (uintptr_t)d->C % 16 == 0 && d->ldc % 8 == 0 &&
                     (d->epilogue != MJV_EPI_SCALE_RES || (d->ldr % 8 == 0 && (uintptr_t)d->res % 16 == 0));
  if (force_tile == 2) {
    if (!t2_ok) {
      mjv_set_error("gemm: tile 2 (128 x 256, two workgroups per CU) takes bias / bias + GELU / LayerScale + residual epilogues on plain, "
                    "16-byte aligned output rows");
      return MJV_E_UNSUPPORTED;
    }
    static const char* const tags2[] = {"gemm2_bias", "gemm2_bias_gelu", "", "gemm2_scale_res"};
    a.gm = MJV_TUNE(gm) > 0 ? MJV_TUNE(gm) : 8;
    MjvProfScope ps(tags2[d->epilogue], s, flops, bytes);
    switch (d->epilogue) {
      case MJV_EPI_BIAS: return launch_t2<MJV_EPI_BIAS>(a, s);
      case MJV_EPI_BIAS_GELU: return launch_t2<MJV_EPI_BIAS_GELU>(a, s);
      default: return launch_t2<MJV_EPI_SCALE_RES>(a, s);
    }
  }
  // one profiler scope per kernel launch, named like the kernel rocprofv3 reports (t256::gemm256_kernel<EPI> / t128::...)
  // A 128-tile launch with fewer workgroups than the chip holds (the peeled tail rows, the batch-sized head GEMMs) is
  // bound by each CU's L2->LDS fill rate (about 45 GB/s for one workgroup): its time is the launch's operand traffic
  // divided by the CUs it occupies.  Splitting K over more workgroups engages the idle CUs; the slices' fp32
  // accumulators go through the caller's workspace and a second, fully parallel launch sums them in slice order.
  auto plan_split = [&](GemmArgs& g) {
    if (!MJV_TUNE(split_k) || !d->workspace) return;
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128), nk = g.K / 64;
    // measured (tools/gemm_bench.py, MJV_BENCH_TAILS=1): the second launch costs about 5 us, so K = 1024 / 2048 tails lose
    // 1-5 us while K = 4096 / 8192 tails gain 12 / 25 us (45 -> 33, 89 -> 64)
    if (tiles >= 2 * g_num_cus || nk < 64) return;
    int sp = (2 * g_num_cus) / tiles;
    if (sp > MJV_TUNE(split_max)) sp = MJV_TUNE(split_max);
    if (sp > nk / 4) sp = nk / 4;
    if (sp < 2 || (long)tiles * sp * 65536L > d->workspace_bytes) return;
    g.split = sp;
    g.ws = (float*)d->workspace;
  };
  static const char* const tags64[] = {"gemm64_bias", "gemm64_bias_gelu", "gemm64_bias_relu", "gemm64_scale_res",
                                       "gemm64_silu_mul"};
  // Under-filled problems with deep K (K >= 4096: the ~1100-row tail of the language tower's w2, and the whole w2 of a
  // single-video forward): K slices of 256 x 256 tiles, one workgroup per CU, at most one round - the 256 kernel's main
  // loop moves 1.4-1.6x the flops per CU-second of the 128 kernel's, which pays for the fp32 round trip of the slices
  // (tools/gemm_bench.py, MJV_BENCH_TAILS=1: 1104 x 2048 x 8192 in 56 us against 64 us on sliced 128 tiles and 141 us on
  // unsliced 256 tiles; 2186 rows: 83 against 145 us).  At K = 2048 it is a wash or a loss (wqkv tail 41 -> 44 us): not used.
  auto plan_split256 = [&](GemmArgs& g) -> bool {
    if (!MJV_TUNE(split256) || force_tile || !d->workspace || d->out_rows) return false;
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256), nk = g.K / 64;
    // (a last m-tile that is mostly empty wastes its share of every slice: 138 rows x 16384 columns ran 10 % slower sliced)
    const bool filled = (long)g.M * 10 >= (long)((g.M + 255) / 256) * 256 * 7;
    if (g.M <= MJV_TUNE(skinny_max_m) || g.N < 256 || nk < MJV_TUNE(split256_min_nk) || tiles * 2 > g_num_cus || !filled) return false;
    int sp = g_num_cus / tiles;
    if (sp > MJV_TUNE(split_max)) sp = MJV_TUNE(split_max);
    if (sp > nk / MJV_TUNE(split256_min_kt)) sp = nk / MJV_TUNE(split256_min_kt);
    if (sp < 2 || (long)tiles * sp * 262144L > d->workspace_bytes) return false;
    g.split = sp;
    g.ws = (float*)d->workspace;
    return true;
  };
  auto run = [&](GemmArgs g, bool use_big) -> int {
    if (plan_split256(g)) use_big = true;
    // skinny problems (peeled tails of a few dozen rows, batch-sized head layers): the 64 x 32 kernel
    const bool skinny = !use_big && (force_tile ? force_tile == 64 : g.M <= MJV_TUNE(skinny_max_m));
    if (!use_big && !skinny) plan_split(g);
    const double frac = (double)g.M / (double)d->M;
    // the rotary epilogue needs a whole 256-column tile staged in LDS: only the unsplit 256^2 kernel has it.  Rows that run
    // on the other kernels (peeled tails, small problems) get the plain Linear into C and the standalone rope_split kernel.
    const int epi = (d->epilogue == MJV_EPI_ROPE_QKV && (!use_big || g.split > 1)) ? (int)MJV_EPI_BIAS : d->epilogue;
    int rc;
    {
      MjvProfScope ps(use_big ? (g.split > 1 ? tags256s[epi] : tags256[epi]) : skinny ? tags64[epi] : tags128[epi], s, flops * frac, bytes * frac);
      switch (epi) {
        case MJV_EPI_BIAS: rc = launch<MJV_EPI_BIAS>(g, s, use_big, skinny); break;
        case MJV_EPI_BIAS_GELU: rc = launch<MJV_EPI_BIAS_GELU>(g, s, use_big, skinny); break;
        case MJV_EPI_BIAS_RELU: rc = launch<MJV_EPI_BIAS_RELU>(g, s, use_big, skinny); break;
        case MJV_EPI_SCALE_RES: rc = launch<MJV_EPI_SCALE_RES>(g, s, use_big, skinny); break;
        case MJV_EPI_ROPE_QKV: rc = launch<MJV_EPI_ROPE_QKV>(g, s, true, false); break;   // unsplit 256 kernel only
        default: rc = launch<MJV_EPI_SILU_MUL>(g, s, use_big, skinny); break;
      }
    }
    if (rc == MJV_OK && epi != d->epilogue) {
      const long r0 = g.m_base;
      rc = mjv_rope_split_bf16(d->C + r0 * d->ldc, d->ldc, d->rope_q + r0 * d->rope_ldq, d->rope_ldq,
                               d->rope_k + r0 * d->rope_ldk, d->rope_ldk, d->rope_cos, d->rope_sin, d->rope_pos + r0, g.M,
                               d->N / ((d->rope_group + 2) * 128), d->rope_group, stream);
    }
    return rc;
  };
  if (m_main == d->M) return run(a, big);
  GemmArgs head = a;
  head.M = m_main;
  int rc = run(head, true);
  if (rc) return rc;
  // tail rows [m_main, M): the kernel indexes A relative to the shifted pointer and adds m_base back for the output /
  // residual row maps, which depend on the absolute row
  GemmArgs tail = a;
  tail.M = d->M - m_main;
  tail.A = a.A + (long)m_main * a.lda;
  tail.m_base = m_main;
  return run(tail, false);
}
