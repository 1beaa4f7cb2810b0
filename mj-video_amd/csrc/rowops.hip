// HBM-bound row kernels: LayerNorm (+pixel-shuffle gather), RMSNorm (+row gather), RoPE/GQA split,
// patchify (im2col), CLS rows, token-embedding gather.  One 64-lane wave per row, 16-byte accesses.
#include "mjv_common.h"
#include <algorithm>
#include "mx8.h"

namespace {

constexpr int WAVES = 4;          // rows per 256-thread workgroup
constexpr int MAX_IT = 8;         // 8 chunks of 512 elements -> rows up to 4096 wide

// ---------------------------------------------------------------------------------------- LayerNorm
// nn.LayerNorm on a bf16 tensor: fp32 mean / biased variance, y = bf16((x - mean) * rstd * g + b).
// gather_grid > 0: the logical input row is the concatenation of 4 rows of the ViT output (pixel shuffle,
// modeling_internvl_chat.py:228-242): out token (tile, a2, b2) <- rows (2a2,2b2) (2a2,2b2+1) (2a2+1,2b2) (2a2+1,2b2+1).
// IT = chunks of 512 elements a lane group covers (row width <= 512 * IT): the narrower instantiations keep fewer values
// in registers, so more rows are in flight per CU.
// MX8: the bf16 result is block-quantised on the way out (y = e4m3 bytes, ldy in bytes, ys = scale records; include/mjv.h
// "MXFP8 operand format"): the input of the fp8 fc1 GEMM.  A lane's 8 columns are a quarter of a 32-element block.
template <int IT, bool MX8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const u16* __restrict__ x, long ldx, u16* __restrict__ y, long ldy,
                                                        const u16* __restrict__ gamma, const u16* __restrict__ beta,
                                                        int rows, int dim, float eps, int grid, uint8_t* __restrict__ ys = nullptr,
                                                        long groups = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int quarter = dim >> 2;
  long src_base[4];
  if (grid > 0) {
    const int h2 = grid >> 1;
    const int per_tile = h2 * h2;
    const int tile = row / per_tile, t = row - tile * per_tile;
    const int a2 = t / h2, b2 = t - a2 * h2;
    const long tb = (long)tile * (grid * grid + 1) + 1;
    src_base[0] = (tb + (2 * a2) * grid + 2 * b2) * ldx;
    src_base[1] = (tb + (2 * a2) * grid + 2 * b2 + 1) * ldx;
    src_base[2] = (tb + (2 * a2 + 1) * grid + 2 * b2) * ldx;
    src_base[3] = (tb + (2 * a2 + 1) * grid + 2 * b2 + 1) * ldx;
  }
  float v[IT][8];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < dim) {
      const u16* src;
      if (grid > 0) {
        const int qd = c / quarter;
        src = x + src_base[qd] + (c - qd * quarter);
      } else {
        src = x + (long)row * ldx + c;
      }
      unpack8(*(const u32x4*)src, v[it]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[it][j];
    }
  }
  const float mean = wave_sum(sum) / (float)dim;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < dim) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[it][j] - mean;
        sq += d * d;
      }
    }
  }
  const float var = wave_sum(sq) / (float)dim;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < dim) {
      float g[8], b[8], o[8];
      unpack8(*(const u32x4*)(gamma + c), g);
      unpack8(*(const u32x4*)(beta + c), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[it][j] - mean) * rstd * g[j] + b[j];
      if constexpr (MX8) {
        unsigned sb;
        const u32x2 q = mx8_quantize_quad(pack8(o), sb);   // (dim % 128 == 0: whole quads are inside the row)
        *(u32x2*)((uint8_t*)y + (long)row * ldy + c) = q;
        if ((lane & 3) == 0) ys[mx8_scale_offset(row, c, groups)] = (uint8_t)sb;
      } else {
        *(u32x4*)(y + (long)row * ldy + c) = pack8(o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ RMSNorm
// modeling_internlm2.py:138-143: h = bf16(x32 * rsqrt(mean(x32^2) + eps)); y = bf16(w * h)
// PARTS > 0 (bench build only, the A/B of DESIGN "Norm fusion"): the row's sum of squares arrives as PARTS per-n-tile partial
// sums from the producing GEMM's epilogue ([rows][PARTS] fp32, summed here in tile order) instead of being reduced in here
template <int IT, int PARTS = 0, bool MX8 = false>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const u16* __restrict__ x, long ldx, u16* __restrict__ y, long ldy,
                                                      const u16* __restrict__ w, const int* __restrict__ row_index,
                                                      int rows, int dim, float eps, const float* __restrict__ partials = nullptr,
                                                      uint8_t* __restrict__ ys = nullptr, long groups = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long srow = row_index ? row_index[row] : row;
  float v[IT][8];
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < dim) {
      unpack8(*(const u32x4*)(x + srow * ldx + c), v[it]);
      if constexpr (PARTS == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sq += v[it][j] * v[it][j];
      }
    }
  }
  float total;
  if constexpr (PARTS == 0) total = wave_sum(sq);
  else {
    total = 0.f;
#pragma unroll
    for (int t = 0; t < PARTS; ++t) total += partials[srow * PARTS + t];
  }
  const float rstd = rsqrtf(total / (float)dim + eps);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < dim) {
      float g[8], o[8];
      unpack8(*(const u32x4*)(w + c), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = g[j] * rbf(v[it][j] * rstd);
      if constexpr (MX8) {   // (see layernorm_kernel: the input of the fp8 w1|w3 GEMM)
        unsigned sb;
        const u32x2 q = mx8_quantize_quad(pack8(o), sb);
        *(u32x2*)((uint8_t*)y + (long)row * ldy + c) = q;
        if ((lane & 3) == 0) ys[mx8_scale_offset(row, c, groups)] = (uint8_t)sb;
      } else {
        *(u32x4*)(y + (long)row * ldy + c) = pack8(o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ row statistics
// The producer half of a norm folded into its consuming GEMM (mjv.h "row_scale"): the same fp32 statistics as the kernels
// above - two-pass mean / biased variance (LayerNorm) or the mean square (RMSNorm, MEAN = false) - and nothing else: 2 B read
// per element, 8 B written per row.
template <int IT, bool MEAN>
__global__ __launch_bounds__(256) void row_stats_kernel(const u16* __restrict__ x, long ldx, float* __restrict__ rstd,
                                                        float* __restrict__ mean_rstd, int rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[IT][8];
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < dim) {
      unpack8(*(const u32x4*)(x + (long)row * ldx + c), v[it]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MEAN) sum += v[it][j];
        else sq += v[it][j] * v[it][j];
      }
    }
  }
  float mean = 0.f;
  if constexpr (MEAN) {
    mean = wave_sum(sum) / (float)dim;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < dim) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[it][j] - mean;
          sq += d * d;
        }
      }
    }
  }
  const float r = rsqrtf(wave_sum(sq) / (float)dim + eps);
  if (lane == 0) {
    rstd[row] = r;
    if constexpr (MEAN) mean_rstd[row] = mean * r;
  }
}

// ------------------------------------------------------------------------------- RoPE + GQA de-interleave
// wqkv output columns are (kv_head, [q_0 .. q_{g-1}, k, v], 128)  (modeling_internlm2.py:361-371).
// q' = bf16(bf16(q*cos) + bf16(rotate_half(q)*sin)) with bf16 tables (modeling_internlm2.py:240-247).
// One thread: 8 elements d..d+7 of the first half and their partners d+64..d+71 of one (row, head slot).
__global__ __launch_bounds__(256) void rope_split_kernel(const u16* __restrict__ qkv, long ldqkv, u16* __restrict__ q, long ldq,
                                                         u16* __restrict__ k, long ldk, const u16* __restrict__ cos_tab,
                                                         const u16* __restrict__ sin_tab, const int* __restrict__ positions,
                                                         int rows, int kv_heads, int group) {
  const int slots = kv_heads * (group + 1);  // rotated head slots per row (q heads + k heads)
  const long total = (long)rows * slots * 8;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int d = (int)(gid & 7) * 8;
  const long rs = gid >> 3;
  const int slot = (int)(rs % slots);
  const long row = rs / slots;
  const int kvh = slot / (group + 1), within = slot - kvh * (group + 1);
  const u16* src = qkv + row * ldqkv + ((long)kvh * (group + 2) + within) * 128;
  u16* dst = (within < group) ? q + row * ldq + ((long)kvh * group + within) * 128 : k + row * ldk + (long)kvh * 128;
  const long pos = positions[row];
  float x1[8], x2[8], c1[8], c2[8], s1[8], s2[8], o1[8], o2[8];
  unpack8(*(const u32x4*)(src + d), x1);
  unpack8(*(const u32x4*)(src + d + 64), x2);
  unpack8(*(const u32x4*)(cos_tab + pos * 128 + d), c1);
  unpack8(*(const u32x4*)(cos_tab + pos * 128 + d + 64), c2);
  unpack8(*(const u32x4*)(sin_tab + pos * 128 + d), s1);
  unpack8(*(const u32x4*)(sin_tab + pos * 128 + d + 64), s2);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o1[j] = rbf(x1[j] * c1[j]) + rbf(-x2[j] * s1[j]);
    o2[j] = rbf(x2[j] * c2[j]) + rbf(x1[j] * s2[j]);
  }
  *(u32x4*)(dst + d) = pack8(o1);
  *(u32x4*)(dst + d + 64) = pack8(o2);
}

// Rotary embedding IN PLACE on n_heads consecutive heads of every row (the Phi-3 layout of BASELINE configs[4]'s language
// tower: qkv_proj's columns are [q heads | k heads | v heads], so ONE launch over the first q + k heads rotates both):
// x' = bf16(bf16(x cos) + bf16(rotate_half(x) sin)) on the first rot_dim elements of a head, bf16 tables [positions][rot_dim]
// that already carry the LongRoPE attention factor (transformers/models/phi3/modeling_phi3.py: apply_rotary_pos_emb,
// Phi3RotaryEmbedding.forward).  One thread: 8 elements j..j+7 of the first half and their partners j + rot_dim / 2.
__global__ __launch_bounds__(256) void rope_heads_kernel(u16* __restrict__ x, long ldx, int head_stride, int n_heads, int rot_dim,
                                                         const u16* __restrict__ cos_tab, const u16* __restrict__ sin_tab,
                                                         const int* __restrict__ positions, int rows) {
  const int half = rot_dim >> 1, cph = half >> 3;   // 16-byte chunks per half
  const long total = (long)rows * n_heads * cph;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int d = (int)(gid % cph) * 8;
  const long rh = gid / cph;
  const int head = (int)(rh % n_heads);
  const long row = rh / n_heads;
  u16* p = x + row * ldx + (long)head * head_stride;
  const long pos = positions[row];
  float x1[8], x2[8], c1[8], c2[8], s1[8], s2[8], o1[8], o2[8];
  unpack8(*(const u32x4*)(p + d), x1);
  unpack8(*(const u32x4*)(p + d + half), x2);
  unpack8(*(const u32x4*)(cos_tab + pos * rot_dim + d), c1);
  unpack8(*(const u32x4*)(cos_tab + pos * rot_dim + d + half), c2);
  unpack8(*(const u32x4*)(sin_tab + pos * rot_dim + d), s1);
  unpack8(*(const u32x4*)(sin_tab + pos * rot_dim + d + half), s2);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o1[j] = rbf(x1[j] * c1[j]) + rbf(-x2[j] * s1[j]);
    o2[j] = rbf(x2[j] * c2[j]) + rbf(x1[j] * s2[j]);
  }
  *(u32x4*)(p + d) = pack8(o1);
  *(u32x4*)(p + d + half) = pack8(o2);
}

// --------------------------------------------------------------------------------------------- patchify
// patches[tile*G*G + a*G + b][c*P*P + i*P + j] = pixels[tile][c][a*P + i][b*P + j]; columns >= 3*P*P are zero.
__global__ __launch_bounds__(256) void patchify_kernel(const u16* __restrict__ px, u16* __restrict__ out, long ldp, int tiles,
                                                       int S, int P) {
  const int G = S / P;
  const int kreal = 3 * P * P;
  const long row = blockIdx.x;  // one workgroup per patch row
  const int tile = (int)(row / (G * G));
  const int t = (int)(row - (long)tile * G * G);
  const int a = t / G, b = t - a * G;
  const u16* base = px + (long)tile * 3 * S * S + (long)(a * P) * S + b * P;
  for (int col = threadIdx.x; col < ldp; col += blockDim.x) {
    u16 v = 0;
    if (col < kreal) {
      const int c = col / (P * P), rem = col - c * P * P;
      const int i = rem / P, j = rem - i * P;
      v = base[(long)c * S * S + (long)i * S + j];
    }
    out[row * ldp + col] = v;
  }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(u16* __restrict__ x, long ldx, const u16* __restrict__ cls,
                                                       const u16* __restrict__ pos0, int tiles, int tpt, int dim) {
  const int tile = blockIdx.x;
  for (int c = threadIdx.x; c < dim; c += blockDim.x)
    x[(long)tile * tpt * ldx + c] = f2bf(bf2f(cls[c]) + bf2f(pos0[c]));
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const int* __restrict__ ids, const u16* __restrict__ table, long ldt,
                                                           u16* __restrict__ x, long ldx, int rows, int dim, int skip_id) {
  const int row = blockIdx.x;
  const int id = ids[row];
  if (id == skip_id) return;
  const u16* src = table + (long)id * ldt;
  for (int c = threadIdx.x * 8; c < dim; c += blockDim.x * 8) *(u32x4*)(x + (long)row * ldx + c) = *(const u32x4*)(src + c);
}

}  // namespace

extern "C" int mjv_layernorm_bf16(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* gamma,
                                  const mjv_bf16* beta, int32_t rows, int32_t dim, float eps, int32_t gather_grid,
                                  void* stream) {
  MJV_REQUIRE(x && y && gamma && beta, "layernorm: null pointer");
  MJV_REQUIRE(rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 512 * MAX_IT, "layernorm: dim %d unsupported", dim);
  MJV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "layernorm: ld alignment");
  if (gather_grid > 0) MJV_REQUIRE(gather_grid % 2 == 0 && dim % 32 == 0, "layernorm: gather needs an even grid");
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps(gather_grid > 0 ? "layernorm_pixshuf" : "layernorm", s, 0, 4.0 * rows * (double)dim);
  const dim3 grid((rows + WAVES - 1) / WAVES);
  if (dim <= 1024)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, gamma, beta, rows, dim, eps, gather_grid);
  else if (dim <= 2048)
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, gamma, beta, rows, dim, eps, gather_grid);
  else
    hipLaunchKernelGGL(layernorm_kernel<MAX_IT>, grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, gamma, beta, rows, dim, eps, gather_grid);
  return mjv_check_launch("layernorm");
}

extern "C" int mjv_rmsnorm_bf16(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* w,
                                const int32_t* row_index, int32_t rows, int32_t dim, float eps, void* stream) {
  MJV_REQUIRE(x && y && w, "rmsnorm: null pointer");
  MJV_REQUIRE(rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 512 * MAX_IT, "rmsnorm: dim %d unsupported", dim);
  MJV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "rmsnorm: ld alignment");
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps("rmsnorm", s, 0, 4.0 * rows * (double)dim);
  const dim3 grid((rows + WAVES - 1) / WAVES);
  if (dim <= 1024)
    hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, w, row_index, rows, dim, eps);
  else if (dim <= 2048)
    hipLaunchKernelGGL(rmsnorm_kernel<4>, grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, w, row_index, rows, dim, eps);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<MAX_IT>, grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, w, row_index, rows, dim, eps);
  return mjv_check_launch("rmsnorm");
}

extern "C" int mjv_row_stats_bf16(const mjv_bf16* x, int64_t ldx, float* rstd, float* mean_rstd, int32_t rows, int32_t dim, float eps,
                                  void* stream) {
  MJV_REQUIRE(x && rstd, "row_stats: null pointer");
  MJV_REQUIRE(rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 512 * MAX_IT, "row_stats: dim %d unsupported", dim);
  MJV_REQUIRE(ldx % 8 == 0, "row_stats: ld alignment");
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps(mean_rstd ? "row_stats_ln" : "row_stats_rms", s, 0, 2.0 * rows * (double)dim);
  const dim3 grid((rows + WAVES - 1) / WAVES);
#define MJV_RS(IT)                                                                                                               \
  do {                                                                                                                           \
    if (mean_rstd) hipLaunchKernelGGL((row_stats_kernel<IT, true>), grid, dim3(256), 0, s, x, (long)ldx, rstd, mean_rstd, rows, dim, eps); \
    else hipLaunchKernelGGL((row_stats_kernel<IT, false>), grid, dim3(256), 0, s, x, (long)ldx, rstd, mean_rstd, rows, dim, eps);         \
  } while (0)
  if (dim <= 1024) MJV_RS(2);
  else if (dim <= 2048) MJV_RS(4);
  else MJV_RS(MAX_IT);
#undef MJV_RS
  return mjv_check_launch("row_stats");
}

extern "C" int mjv_layernorm_mxfp8(const mjv_bf16* x, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* y_scales, const mjv_bf16* gamma,
                                   const mjv_bf16* beta, int32_t rows, int32_t dim, float eps, void* stream) {
  MJV_REQUIRE(x && y && y_scales && gamma && beta, "layernorm_mxfp8: null pointer");
  MJV_REQUIRE(rows > 0 && dim > 0 && dim % 128 == 0 && dim <= 512 * MAX_IT, "layernorm_mxfp8: dim %d unsupported", dim);
  MJV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && (uintptr_t)y % 8 == 0, "layernorm_mxfp8: alignment");
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps("layernorm_mxfp8", s, 0, 3.03 * rows * (double)dim);
  const dim3 grid((rows + WAVES - 1) / WAVES);
  const long groups = (rows + 63) / 64;
  if (dim <= 1024)
    hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, dim3(256), 0, s, x, (long)ldx, (u16*)y, (long)ldy, gamma, beta, rows, dim, eps, 0, y_scales, groups);
  else if (dim <= 2048)
    hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, dim3(256), 0, s, x, (long)ldx, (u16*)y, (long)ldy, gamma, beta, rows, dim, eps, 0, y_scales, groups);
  else
    hipLaunchKernelGGL((layernorm_kernel<MAX_IT, true>), grid, dim3(256), 0, s, x, (long)ldx, (u16*)y, (long)ldy, gamma, beta, rows, dim, eps, 0, y_scales, groups);
  return mjv_check_launch("layernorm_mxfp8");
}

extern "C" int mjv_rmsnorm_mxfp8(const mjv_bf16* x, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* y_scales, const mjv_bf16* w,
                                 int32_t rows, int32_t dim, float eps, void* stream) {
  MJV_REQUIRE(x && y && y_scales && w, "rmsnorm_mxfp8: null pointer");
  MJV_REQUIRE(rows > 0 && dim > 0 && dim % 128 == 0 && dim <= 512 * MAX_IT, "rmsnorm_mxfp8: dim %d unsupported", dim);
  MJV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && (uintptr_t)y % 8 == 0, "rmsnorm_mxfp8: alignment");
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps("rmsnorm_mxfp8", s, 0, 3.03 * rows * (double)dim);
  const dim3 grid((rows + WAVES - 1) / WAVES);
  const long groups = (rows + 63) / 64;
  const int* no_index = nullptr;
  const float* no_parts = nullptr;
  if (dim <= 1024)
    hipLaunchKernelGGL((rmsnorm_kernel<2, 0, true>), grid, dim3(256), 0, s, x, (long)ldx, (u16*)y, (long)ldy, w, no_index, rows, dim, eps, no_parts, y_scales, groups);
  else if (dim <= 2048)
    hipLaunchKernelGGL((rmsnorm_kernel<4, 0, true>), grid, dim3(256), 0, s, x, (long)ldx, (u16*)y, (long)ldy, w, no_index, rows, dim, eps, no_parts, y_scales, groups);
  else
    hipLaunchKernelGGL((rmsnorm_kernel<MAX_IT, 0, true>), grid, dim3(256), 0, s, x, (long)ldx, (u16*)y, (long)ldy, w, no_index, rows, dim, eps, no_parts, y_scales, groups);
  return mjv_check_launch("rmsnorm_mxfp8");
}

#ifdef MJV_BENCH
// A/B for DESIGN "Norm fusion": RMSNorm whose statistics come from the producer (8 per-n-tile partial sums of squares per row,
// what an EPI_SCALE_RES epilogue of a 2048-wide output would emit).  dim 2048 only.
extern "C" int mjv_bench_rmsnorm_prestat(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* w,
                                         const float* partials, int32_t rows, int32_t dim, float eps, void* stream) {
  MJV_REQUIRE(x && y && w && partials && dim == 2048, "rmsnorm_prestat: dim 2048 only");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + WAVES - 1) / WAVES);
  hipLaunchKernelGGL((rmsnorm_kernel<4, 8>), grid, dim3(256), 0, s, x, (long)ldx, y, (long)ldy, w, (const int*)nullptr, rows, dim, eps,
                     partials);
  return mjv_check_launch("rmsnorm_prestat");
}

// The elementwise pass an UN-fused Linear would owe (tools/fused_vs_unfused.py): y = epilogue(x) over a
// [rows][cols] bf16 matrix, the same operations and rounding points as the GEMM epilogues, written as the cheapest correct
// HBM-bound kernel - 16-byte loads and stores, a grid of one wave's worth of rows per workgroup pass, the GELU table in LDS.
//   kind 1: y = gelu(bf16(x + bias))              (fc1)
//   kind 3: y = res + bf16(bf16(x + bias) * scale)   (proj / fc2; scale NULL: y = res + bf16(x + bias); bias NULL: wo / w2)
#include "gelu_table.h"
namespace {
__device__ const u16 g_gelu_table_pass[MJV_GELU_TABLE_LEN] = MJV_GELU_TABLE_INIT;
template <int KIND>
__global__ __launch_bounds__(256) void epilogue_pass_kernel(const u16* __restrict__ x, long ldx, u16* __restrict__ y, long ldy,
                                                            const u16* __restrict__ bias, const u16* __restrict__ scale,
                                                            const u16* __restrict__ res, long ldr, int rows, int cols) {
  __shared__ u16 tab[KIND == 1 ? MJV_GELU_TABLE_LEN : 8];
  if constexpr (KIND == 1) {
    for (int i = threadIdx.x; i < MJV_GELU_TABLE_LEN / 8; i += 256) ((u32x4*)tab)[i] = ((const u32x4*)g_gelu_table_pass)[i];
    __syncthreads();
  }
  const int cpr = cols / 8;                       // 16-byte chunks per row
  const long total = (long)rows * cpr;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < total; c += (long)gridDim.x * 256) {
    const long row = c / cpr;
    const int col = (int)(c - row * cpr) * 8;
    float v[8], b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unpack8(__builtin_nontemporal_load((const u32x4*)(x + row * ldx + col)), v);
    if (bias) unpack8(*(const u32x4*)(bias + col), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + b[e]);
    if constexpr (KIND == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned u = __float_as_uint(v[e]);
        const unsigned mag = (u >> 16) & 0x7fffu, rel = mag - MJV_GELU_LO;
        const bool in_tab = rel < (unsigned)MJV_GELU_R;
        const unsigned t = tab[in_tab ? rel + (u >> 31) * MJV_GELU_NEG_OFF : 0u];
        const unsigned other = mag < MJV_GELU_LO ? __float_as_uint(0.5f * v[e]) : gelu_beyond_table(u, mag);
        v[e] = __uint_as_float(in_tab ? (t << 16) : other);
      }
    } else {
      float r[8], sc[8];
      unpack8(__builtin_nontemporal_load((const u32x4*)(res + row * ldr + col)), r);
      if (scale) {
        unpack8(*(const u32x4*)(scale + col), sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] * sc[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    __builtin_nontemporal_store(pack8(v), (u32x4*)(y + row * ldy + col));
  }
}
}  // namespace
extern "C" int mjv_bench_epilogue_pass(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* bias,
                                       const mjv_bf16* scale, const mjv_bf16* res, int64_t ldr, int32_t rows, int32_t cols,
                                       int32_t kind, void* stream) {
  MJV_REQUIRE(x && y && rows > 0 && cols > 0 && cols % 8 == 0 && (kind == 1 || (kind == 3 && res)), "epilogue_pass: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)rows * (cols / 8);
  const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 16);   // 16 workgroups per CU, grid-stride
  MjvProfScope ps(kind == 1 ? "bench_pass_bias_gelu" : "bench_pass_scale_res", s, 0, (kind == 1 ? 4.0 : 6.0) * rows * (double)cols);
  if (kind == 1) hipLaunchKernelGGL(epilogue_pass_kernel<1>, dim3(blocks), dim3(256), 0, s, x, (long)ldx, y, (long)ldy, bias, scale, res, (long)ldr, rows, cols);
  else hipLaunchKernelGGL(epilogue_pass_kernel<3>, dim3(blocks), dim3(256), 0, s, x, (long)ldx, y, (long)ldy, bias, scale, res, (long)ldr, rows, cols);
  return mjv_check_launch("epilogue_pass");
}
#endif

extern "C" int mjv_rope_split_bf16(const mjv_bf16* qkv, int64_t ldqkv, mjv_bf16* q, int64_t ldq, mjv_bf16* k, int64_t ldk,
                                   const mjv_bf16* cos_tab, const mjv_bf16* sin_tab, const int32_t* positions,
                                   int32_t rows, int32_t kv_heads, int32_t group, void* stream) {
  MJV_REQUIRE(qkv && (q || group == 0) && k && cos_tab && sin_tab && positions, "rope: null pointer");
  MJV_REQUIRE(rows > 0 && kv_heads > 0 && group >= 0, "rope: bad sizes");
  MJV_REQUIRE(ldqkv % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0, "rope: ld alignment");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)rows * kv_heads * (group + 1) * 8;
  MjvProfScope ps("rope_split", s, 0, 4.0 * rows * (double)kv_heads * (group + 1) * 128);
  hipLaunchKernelGGL(rope_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, (long)ldqkv, q, (long)ldq,
                     k, (long)ldk, cos_tab, sin_tab, positions, rows, kv_heads, group);
  return mjv_check_launch("rope_split");
}

extern "C" int mjv_rope_heads_bf16(mjv_bf16* x, int64_t ldx, int32_t head_stride, int32_t n_heads, int32_t rot_dim,
                                   const mjv_bf16* cos_tab, const mjv_bf16* sin_tab, const int32_t* positions, int32_t rows,
                                   void* stream) {
  MJV_REQUIRE(x && cos_tab && sin_tab && positions, "rope_heads: null pointer");
  MJV_REQUIRE(rows > 0 && n_heads > 0 && rot_dim > 0 && rot_dim % 16 == 0 && rot_dim <= head_stride,
              "rope_heads: rot_dim %d must be a positive multiple of 16 within the head stride %d", rot_dim, head_stride);
  MJV_REQUIRE(ldx % 8 == 0 && head_stride % 8 == 0 && ((uintptr_t)x | (uintptr_t)cos_tab | (uintptr_t)sin_tab) % 16 == 0,
              "rope_heads: alignment (16-byte pieces of a head)");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)rows * n_heads * (rot_dim / 16);
  MjvProfScope ps("rope_heads", s, 0, 4.0 * rows * (double)n_heads * rot_dim);
  hipLaunchKernelGGL(rope_heads_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, (long)ldx, head_stride, n_heads,
                     rot_dim, cos_tab, sin_tab, positions, rows);
  return mjv_check_launch("rope_heads");
}

extern "C" int mjv_patchify_bf16(const mjv_bf16* pixels, mjv_bf16* patches, int64_t ldp, int32_t tiles, int32_t image_size,
                                 int32_t patch, void* stream) {
  MJV_REQUIRE(pixels && patches, "patchify: null pointer");
  MJV_REQUIRE(tiles > 0 && patch > 0 && image_size % patch == 0, "patchify: image %d not a multiple of patch %d", image_size, patch);
  MJV_REQUIRE(ldp >= 3 * patch * patch, "patchify: ldp too small");
  hipStream_t s = (hipStream_t)stream;
  const int G = image_size / patch;
  MjvProfScope ps("patchify", s, 0, 2.0 * tiles * 3.0 * image_size * image_size + 2.0 * tiles * G * G * (double)ldp);
  hipLaunchKernelGGL(patchify_kernel, dim3(tiles * G * G), dim3(256), 0, s, pixels, patches, (long)ldp, tiles, image_size, patch);
  return mjv_check_launch("patchify");
}

extern "C" int mjv_cls_rows_bf16(mjv_bf16* x, int64_t ldx, const mjv_bf16* cls, const mjv_bf16* pos0, int32_t tiles,
                                 int32_t tokens_per_tile, int32_t dim, void* stream) {
  MJV_REQUIRE(x && cls && pos0 && tiles > 0 && dim > 0, "cls_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps("cls_rows", s, 0, 2.0 * tiles * dim);
  hipLaunchKernelGGL(cls_rows_kernel, dim3(tiles), dim3(256), 0, s, x, (long)ldx, cls, pos0, tiles, tokens_per_tile, dim);
  return mjv_check_launch("cls_rows");
}

extern "C" int mjv_embed_gather_bf16(const int32_t* ids, const mjv_bf16* table, int64_t ldt, mjv_bf16* x, int64_t ldx,
                                     int32_t rows, int32_t dim, int32_t skip_id, int32_t vocab, void* stream) {
  MJV_REQUIRE(ids && table && x && rows > 0 && dim % 8 == 0, "embed_gather: bad arguments");
  MJV_REQUIRE(ldt % 8 == 0 && ldx % 8 == 0, "embed_gather: ld alignment");
  (void)vocab;
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps("embed_gather", s, 0, 4.0 * rows * (double)dim);
  hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(256), 0, s, ids, table, (long)ldt, x, (long)ldx, rows, dim, skip_id);
  return mjv_check_launch("embed_gather");
}
