// MXFP8 block quantiser (bf16 -> e4m3 elements + e8m0 block scales in the MFMA's lane layout): the weights of the fp8 FFN
// GEMMs once at load, and the test vehicle for the format (bit-exact against oracle/ref_fp8.py).  HBM-bound byte work:
// 2 B read + 1.03 B written per element, one wave per 512-column chunk of a row, 16-B loads, 8-B stores.
#include "mx8.h"

namespace {
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const u16* __restrict__ x, long ldx, uint8_t* __restrict__ y, long ldy,
                                                             uint8_t* __restrict__ scales, long groups, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int c0 = blockIdx.y * 512; c0 < cols; c0 += gridDim.y * 512) {
    const int c = c0 + lane * 8;
    // (cols % 128 == 0: a quad's block is either wholly inside the row or wholly outside - whole quads leave together)
    if (c >= cols) break;
    const u32x4 v = *(const u32x4*)(x + row * ldx + c);
    unsigned sb;
    const u32x2 q = mx8_quantize_quad(v, sb);
    *(u32x2*)(y + row * ldy + c) = q;
    if ((lane & 3) == 0) scales[mx8_scale_offset(row, c, groups)] = (uint8_t)sb;
  }
}
}  // namespace

extern "C" int64_t mjv_mxfp8_scale_bytes(int64_t rows, int64_t cols) { return (cols / 128) * ((rows + 63) / 64) * 256; }

extern "C" int mjv_quantize_mxfp8(const uint16_t* x, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* scales, int32_t rows,
                                  int32_t cols, void* stream) {
  MJV_REQUIRE(x && y && scales, "quantize_mxfp8: null pointer");
  MJV_REQUIRE(rows > 0 && cols > 0 && cols % 128 == 0, "quantize_mxfp8: rows=%d cols=%d (cols must be a multiple of 128)", rows, cols);
  MJV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= cols && ldy >= cols, "quantize_mxfp8: leading dims");
  MJV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 8 == 0, "quantize_mxfp8: misaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (cols + 511) / 512;
  MjvProfScope ps("quantize_mxfp8", s, 0, (double)rows * cols * 3.03);
  hipLaunchKernelGGL(quantize_mxfp8_kernel, dim3((rows + 3) / 4, chunks < 8 ? chunks : 8), dim3(256), 0, s, x, (long)ldx, y, (long)ldy,
                     scales, (long)((rows + 63) / 64), rows, cols);
  return mjv_check_launch("quantize_mxfp8");
}
