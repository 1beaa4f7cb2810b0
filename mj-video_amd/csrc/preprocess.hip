// Frame preprocessing on the GPU: the resize -> tile crop -> ToTensor -> Normalize -> bf16 chain of the reference's
// load_video (scripts/data_processor/data.py:56-64,81-117,158-179), fed with decoded uint8 RGB frames.
//
// The resize reproduces Pillow's 8-bit bicubic resampler BIT FOR BIT: two separable passes (horizontal, then vertical)
// with a uint8 intermediate image, 22-bit fixed-point coefficients and the +0.5 / arithmetic-shift / clip rounding of
// Pillow's ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc.  The per-output-pixel windows (first tap,
// tap count) and the integer coefficients are computed on the host exactly as Pillow's precompute_coeffs /
// normalize_coeffs_8bpc do (mj-video_amd/video.py: pil_resample_coeffs) and passed in.
// Normalisation is fp32 like torchvision's ToTensor + Normalize: ((u8 / 255) - mean) / std, then one bf16 rounding
// (the caller-side .to(torch.bfloat16), eval_genai_mjvideo.py:131).  Byte work, HBM-bound, no MFMA.
#include "mjv_common.h"

namespace {

// byte x fixed-point coefficient (|coefficient| <= 2^22: Pillow's PRECISION_BITS) as a 24-bit multiply
MJV_DEV int mul24(int a, int b) { return __mul24(a, b); }

// horizontal pass: tmp[f][y][x'][c] = clip8((2^21 + sum_k in[f][y][x0 + k][c] * coef[x'][k]) >> 22)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, int H, int W,
                                                       int out_w, const int* __restrict__ bounds, const int* __restrict__ coef,
                                                       int ksize) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, f = blockIdx.z;
  if (x >= out_w) return;
  const int x0 = bounds[2 * x], n = bounds[2 * x + 1];
  const uint8_t* row = in + ((long)f * H + y) * W * 3 + (long)x0 * 3;
  const int* k = coef + (long)x * ksize;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    a0 += row[3 * i] * w;
    a1 += row[3 * i + 1] * w;
    a2 += row[3 * i + 2] * w;
  }
  uint8_t* o = tmp + (((long)f * H + y) * out_w + x) * 3;
  o[0] = (uint8_t)min(max(a0 >> 22, 0), 255);
  o[1] = (uint8_t)min(max(a1 >> 22, 0), 255);
  o[2] = (uint8_t)min(max(a2 >> 22, 0), 255);
}

// vertical pass + tile crop + normalise: out[(f * tiles_per_frame + tile_off + ty * cols + tx)][c][y % S][x % S]
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, u16* __restrict__ out, int H, int out_w,
                                                            int out_h, const int* __restrict__ bounds, const int* __restrict__ coef,
                                                            int ksize, int S, int tiles_per_frame, int tile_off, float m0, float m1,
                                                            float m2, float s0, float s1, float s2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, f = blockIdx.z;
  if (x >= out_w) return;
  const int y0 = bounds[2 * y], n = bounds[2 * y + 1];
  const uint8_t* col = tmp + (((long)f * H + y0) * out_w + x) * 3;
  const int* k = coef + (long)y * ksize;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    const uint8_t* p = col + (long)i * out_w * 3;
    a0 += p[0] * w;
    a1 += p[1] * w;
    a2 += p[2] * w;
  }
  const float v0 = (float)min(max(a0 >> 22, 0), 255), v1 = (float)min(max(a1 >> 22, 0), 255), v2 = (float)min(max(a2 >> 22, 0), 255);
  const int cols = out_w / S;
  const int tx = x / S, ty = y / S;
  const long tile = (long)f * tiles_per_frame + tile_off + ty * cols + tx;
  u16* o = out + tile * 3 * S * S + (long)(y - ty * S) * S + (x - tx * S);
  o[0] = f2bf((v0 / 255.0f - m0) / s0);
  o[(long)S * S] = f2bf((v1 / 255.0f - m1) / s1);
  o[2l * S * S] = f2bf((v2 / 255.0f - m2) / s2);
}

// ---- round 4: the same two passes with coalesced memory traffic (the kernels above stay as the fallback for geometries whose
// tables do not fit LDS or whose rows are not dword-aligned).  Same integer arithmetic, same fp32 expression: bit-identical.
//
// Horizontal pass: one workgroup walks H_ROWS input rows of one frame for a chunk of 512 output pixels, two pixels per thread.
// A pixel's taps are the same for every row: its (up to 4 KS4) coefficients are loaded into registers once per workgroup, zeros
// behind the window's end.  The span of the input row the chunk needs is staged in LDS with 16-byte loads (row starts are arbitrary
// byte addresses: the staged image starts at the 16-byte boundary below the span); a pixel's window - 12 KS4 consecutive bytes
// from an arbitrary byte offset - is read as 3 KS4 + 1 aligned dwords and realigned with v_alignbyte, after which the channel of
// every byte is a compile-time fact (R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 per four taps).  The first form read it byte by
// byte (36 ds_read_u8 + 12 coefficient reads per pixel): LDS-instruction-bound, no faster than the uncoalesced kernel it replaced.
// The output row leaves through LDS as 16-byte stores.
constexpr int H_ROWS = 16;
constexpr int H_CHUNK = 512;
template <int KS4>
__global__ __launch_bounds__(256) void resize_h_lds_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, int H, int W, int out_w,
                                                           const int* __restrict__ bounds, const int* __restrict__ coef, int ksize,
                                                           long in_bytes, int row_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint8_t* lrow = (uint8_t*)smem;                  // row_lds bytes
  uint8_t* lout = lrow + row_lds;                  // H_CHUNK * 3 bytes
  const int tid = threadIdx.x, f = blockIdx.y;
  const int xa = blockIdx.z * H_CHUNK, xb = min(out_w, xa + H_CHUNK) - 1;   // this chunk's output pixels [xa, xb]
  const int lo = bounds[2 * xa];                                             // first input pixel any of them reads
  const int hi = min(W, bounds[2 * xb] + bounds[2 * xb + 1]);                // one past the last
  int w[2][4 * KS4], x0[2];
  bool valid[2];
#pragma unroll
  for (int pq = 0; pq < 2; ++pq) {
    const int x = xa + tid + 256 * pq;
    valid[pq] = x <= xb;
    const int xs = valid[pq] ? x : xb;
    x0[pq] = bounds[2 * xs];
    const int n = bounds[2 * xs + 1];
#pragma unroll
    for (int i = 0; i < 4 * KS4; ++i) w[pq][i] = (i < n && i < ksize) ? coef[(long)xs * ksize + i] : 0;
  }
  const int nb_out = out_w * 3, nb_chunk = (xb - xa + 1) * 3;
  const bool out16 = (nb_out % 16 == 0) && (((uintptr_t)tmp) % 16 == 0);
  // the staged span of row y: NC 16-byte chunks per thread, fetched into registers one row AHEAD (the loads of row r + 1 fly under
  // the arithmetic of row r; fetched where they are stored, every row paid a global round trip between two barriers)
  constexpr int NC = 3;                             // (the host launches this kernel for spans of at most 3 x 256 chunks)
  u32x4 v[NC];
  int mis = 0;
  auto fetch = [&](int y, int& mis_out) __attribute__((always_inline)) {
    const long start = (((long)f * H + y) * W + lo) * 3;
    mis_out = (int)(((uintptr_t)(in + start)) & 15);
    const uint8_t* g16 = in + start - mis_out;
    const int nch = ((hi - lo) * 3 + mis_out + 15) / 16;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int c = tid + 256 * q;
      v[q] = u32x4{0u, 0u, 0u, 0u};
      if (c >= nch) continue;
      const long off = start - mis_out + 16l * c;   // byte offset of the chunk in the input buffer
      if (off >= 0 && off + 16 <= in_bytes) v[q] = *(const u32x4*)(g16 + 16l * c);
      else {                                        // first / last bytes of the buffer: byte by byte, zeros outside
        unsigned w4[4] = {0u, 0u, 0u, 0u};
        for (int b = 0; b < 16; ++b) {
          const long o = off + b;
          if (o >= 0 && o < in_bytes) w4[b >> 2] |= (unsigned)in[o] << (8 * (b & 3));
        }
        v[q] = u32x4{w4[0], w4[1], w4[2], w4[3]};
      }
    }
  };
  const int y_first = blockIdx.x * H_ROWS;
  int mis_next = 0;
  if (y_first < H) fetch(y_first, mis_next);
  for (int r = 0; r < H_ROWS; ++r) {
    const int y = y_first + r;
    if (y >= H) break;                              // workgroup-uniform
    mis = mis_next;
    const int nch = ((hi - lo) * 3 + mis + 15) / 16;
    __syncthreads();                                // the previous row's readers of lrow / lout are done
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int c = tid + 256 * q;
      if (c < nch) *(u32x4*)(lrow + 16 * c) = v[q];
    }
    if (r + 1 < H_ROWS && y + 1 < H) fetch(y + 1, mis_next);
    __syncthreads();
#pragma unroll
    for (int pq = 0; pq < 2; ++pq) {
      if (!valid[pq]) continue;
      const int sb = mis + (x0[pq] - lo) * 3;       // byte offset of the window in the staged span
      const unsigned* dwp = (const unsigned*)(lrow + (sb & ~3));
      const unsigned sh = sb & 3;
      unsigned raw[3 * KS4 + 1];
#pragma unroll
      for (int q = 0; q < 3 * KS4 + 1; ++q) raw[q] = dwp[q];   // (bytes behind the span: whatever LDS holds, times a zero coefficient)
      int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
#pragma unroll
      for (int g = 0; g < KS4; ++g) {
        const unsigned d0 = __builtin_amdgcn_alignbyte(raw[3 * g + 1], raw[3 * g], sh);
        const unsigned d1 = __builtin_amdgcn_alignbyte(raw[3 * g + 2], raw[3 * g + 1], sh);
        const unsigned d2 = __builtin_amdgcn_alignbyte(raw[3 * g + 3], raw[3 * g + 2], sh);
        const int* wg = &w[pq][4 * g];
        // (24-bit multiplies: a byte times a coefficient of at most 2^22 in magnitude - v_mad_i32_i24 runs at full rate, the 32-bit
        // v_mul_lo_u32 the plain product compiles to at a quarter of it, and there are 36 of them per pixel)
        a0 += mul24((int)(d0 & 255u), wg[0]) + mul24((int)(d0 >> 24), wg[1]) + mul24((int)((d1 >> 16) & 255u), wg[2]) + mul24((int)((d2 >> 8) & 255u), wg[3]);
        a1 += mul24((int)((d0 >> 8) & 255u), wg[0]) + mul24((int)(d1 & 255u), wg[1]) + mul24((int)(d1 >> 24), wg[2]) + mul24((int)((d2 >> 16) & 255u), wg[3]);
        a2 += mul24((int)((d0 >> 16) & 255u), wg[0]) + mul24((int)((d1 >> 8) & 255u), wg[1]) + mul24((int)(d2 & 255u), wg[2]) + mul24((int)(d2 >> 24), wg[3]);
      }
      uint8_t* po = lout + (tid + 256 * pq) * 3;
      po[0] = (uint8_t)min(max(a0 >> 22, 0), 255);
      po[1] = (uint8_t)min(max(a1 >> 22, 0), 255);
      po[2] = (uint8_t)min(max(a2 >> 22, 0), 255);
    }
    __syncthreads();
    uint8_t* o = tmp + ((long)f * H + y) * nb_out + (long)xa * 3;     // (a chunk starts at a multiple of 1536 bytes of its row)
    if (out16) {
      for (int c = tid; c < nb_chunk / 16; c += 256) *(u32x4*)(o + 16 * c) = *(const u32x4*)(lout + 16 * c);
      for (int b = (nb_chunk / 16) * 16 + tid; b < nb_chunk; b += 256) o[b] = lout[b];
    } else {
      for (int b = tid; b < nb_chunk; b += 256) o[b] = lout[b];
    }
  }
}

// Vertical pass + tile crop + normalise: a workgroup makes V_ROWS output rows of one frame, one wave per row (the row's window and
// coefficients are wave-uniform).  The vertical filter does not care which channel a byte belongs to: a lane filters 4 consecutive
// BYTES of the row (dword loads, consecutive lanes consecutive dwords), the normalised bf16 values go to LDS as three planes and
// leave as 16-byte stores of the planar [tile][channel][y][x] output.  Needs out_w * 3 % 4 == 0, tile_size % 8 == 0 and aligned bases.
constexpr int V_ROWS = 4;
template <int TB>   // taps fetched together: the row window in one batch when it has at most TB taps
__global__ __launch_bounds__(256) void resize_v_lds_kernel(const uint8_t* __restrict__ tmp, u16* __restrict__ out, int H, int out_w, int out_h,
                                                           const int* __restrict__ bounds, const int* __restrict__ coef, int ksize, int S,
                                                           int tiles_per_frame, int tile_off, float m0, float m1, float m2, float s0,
                                                           float s1, float s2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, f = blockIdx.y;
  const int y = blockIdx.x * V_ROWS + wave;
  u16* lpl = (u16*)smem + (long)wave * 3 * out_w;  // this wave's row: [channel][x]
  const int nb = out_w * 3, ndw = nb / 4;
  if (y < out_h) {
    const int y0 = __builtin_amdgcn_readfirstlane(bounds[2 * y]), n = __builtin_amdgcn_readfirstlane(bounds[2 * y + 1]);
    const int* k = coef + (long)y * ksize;
    const unsigned* base = (const unsigned*)(tmp + ((long)f * H + y0) * nb);
    for (int d = lane; d < ndw; d += 64) {
      int a[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
      // TB taps per step, their loads in flight together (a tap-by-tap loop of unknown length waits for every load in turn);
      // taps behind the window: the window's last row with a zero coefficient
      for (int i = 0; i < n; i += TB) {
        unsigned dw[TB];
        int w[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const int t = min(i + j, n - 1);
          w[j] = (i + j < n) ? k[t] : 0;
          dw[j] = base[(long)t * ndw + d];
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          a[0] += mul24((int)(dw[j] & 255u), w[j]);
          a[1] += mul24((int)((dw[j] >> 8) & 255u), w[j]);
          a[2] += mul24((int)((dw[j] >> 16) & 255u), w[j]);
          a[3] += mul24((int)(dw[j] >> 24), w[j]);
        }
      }
      int b = 4 * d, x = b / 3, c = b - 3 * x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = (float)min(max(a[j] >> 22, 0), 255);
        const float mc = c == 0 ? m0 : (c == 1 ? m1 : m2), sc = c == 0 ? s0 : (c == 1 ? s1 : s2);
        lpl[c * out_w + x] = f2bf((v / 255.0f - mc) / sc);
        if (++c == 3) { c = 0; ++x; }
      }
    }
  }
  __syncthreads();
  if (y >= out_h) return;
  // planes -> global: per channel and tile a run of S bf16 (2 S bytes), 16 bytes per lane
  const int cols = out_w / S, ty = y / S, per_plane = out_w / 8;
  for (int q = lane; q < 3 * per_plane; q += 64) {
    const int c = q / per_plane, x8 = (q - c * per_plane) * 8;
    const int tx = x8 / S;
    const long tile = (long)f * tiles_per_frame + tile_off + ty * cols + tx;
    u16* o = out + tile * 3 * S * S + (long)c * S * S + (long)(y - ty * S) * S + (x8 - tx * S);
    *(u32x4*)o = *(const u32x4*)(lpl + c * out_w + x8);
  }
}

}  // namespace

extern "C" int mjv_resize_normalize_u8(const uint8_t* frames, int32_t n_frames, int32_t height, int32_t width, int32_t out_w,
                                       int32_t out_h, const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                                       const int32_t* ybounds, const int32_t* ycoef, int32_t ky, uint8_t* tmp, mjv_bf16* out,
                                       int32_t tile_size, int32_t tiles_per_frame, int32_t tile_offset, const float* mean,
                                       const float* stdv, void* stream) {
  MJV_REQUIRE(frames && xbounds && xcoef && ybounds && ycoef && tmp && out && mean && stdv, "resize_normalize: null pointer");
  MJV_REQUIRE(n_frames > 0 && height > 0 && width > 0 && out_w > 0 && out_h > 0 && kx > 0 && ky > 0, "resize_normalize: bad sizes");
  MJV_REQUIRE(tile_size > 0 && out_w % tile_size == 0 && out_h % tile_size == 0, "resize_normalize: output %dx%d is not a grid of %d-px tiles",
              out_w, out_h, tile_size);
  MJV_REQUIRE(tile_offset >= 0 && tile_offset + (out_w / tile_size) * (out_h / tile_size) <= tiles_per_frame,
              "resize_normalize: tile grid does not fit tiles_per_frame=%d", tiles_per_frame);
  hipStream_t s = (hipStream_t)stream;
  const double bytes = (double)n_frames * (3.0 * height * width + 2.0 * 3.0 * height * out_w + 3.0 * out_w * out_h * 3.0);
  MjvProfScope ps("resize_normalize", s, 0, bytes);
  // horizontal pass: the LDS form for windows of up to 32 taps (else one thread per output pixel straight from global)
  const int ks4 = (kx + 3) / 4 <= 2 ? 2 : (kx + 3) / 4 == 7 ? 8 : (kx + 3) / 4;   // the instantiation that runs
  const int row_lds = ((width * 3 + 15 + 12 * ks4 + 4 + 15) / 16) * 16 + 16;
  const size_t h_lds = (size_t)row_lds + H_CHUNK * 3;
  const dim3 hgrid((height + H_ROWS - 1) / H_ROWS, n_frames, (out_w + H_CHUNK - 1) / H_CHUNK);
  const long in_bytes = (long)n_frames * height * width * 3;
#define MJV_H_LAUNCH(K) hipLaunchKernelGGL(resize_h_lds_kernel<K>, hgrid, dim3(256), h_lds, s, frames, tmp, height, width, out_w, xbounds, xcoef, kx, in_bytes, row_lds)
  if (h_lds > 64 * 1024 || ks4 > 8 || (width * 3 + 15 + 15) / 16 > 3 * 256)
    hipLaunchKernelGGL(resize_h_kernel, dim3((out_w + 255) / 256, height, n_frames), dim3(256), 0, s, frames, tmp, height, width, out_w,
                       xbounds, xcoef, kx);
  else if (ks4 <= 2) MJV_H_LAUNCH(2);
  else if (ks4 == 3) MJV_H_LAUNCH(3);
  else if (ks4 == 4) MJV_H_LAUNCH(4);
  else if (ks4 == 5) MJV_H_LAUNCH(5);
  else if (ks4 == 6) MJV_H_LAUNCH(6);
  else MJV_H_LAUNCH(8);
#undef MJV_H_LAUNCH
  // vertical pass: dword reads of the intermediate rows and 16-byte planar stores need these alignments
  const long v_lds = (long)V_ROWS * 3 * out_w * 2;
  if (v_lds <= 64 * 1024 && (out_w * 3) % 4 == 0 && tile_size % 8 == 0 && (uintptr_t)tmp % 4 == 0 && (uintptr_t)out % 16 == 0)
  {
#define MJV_V_LAUNCH(T) hipLaunchKernelGGL(resize_v_lds_kernel<T>, dim3((out_h + V_ROWS - 1) / V_ROWS, n_frames), dim3(256), (size_t)v_lds, s, tmp, out, height, out_w, out_h, ybounds, ycoef, ky, tile_size, tiles_per_frame, tile_offset, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2])
    if (ky <= 5) MJV_V_LAUNCH(5);
    else if (ky <= 9) MJV_V_LAUNCH(9);
    else MJV_V_LAUNCH(12);
#undef MJV_V_LAUNCH
  }
  else
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((out_w + 255) / 256, out_h, n_frames), dim3(256), 0, s, tmp, out, height, out_w, out_h,
                       ybounds, ycoef, ky, tile_size, tiles_per_frame, tile_offset, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  return mjv_check_launch("resize_normalize");
}
