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
  hipLaunchKernelGGL(resize_h_kernel, dim3((out_w + 255) / 256, height, n_frames), dim3(256), 0, s, frames, tmp, height, width, out_w,
                     xbounds, xcoef, kx);
  hipLaunchKernelGGL(resize_v_norm_kernel, dim3((out_w + 255) / 256, out_h, n_frames), dim3(256), 0, s, tmp, out, height, out_w, out_h,
                     ybounds, ycoef, ky, tile_size, tiles_per_frame, tile_offset, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  return mjv_check_launch("resize_normalize");
}
