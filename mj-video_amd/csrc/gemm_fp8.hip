// MX-fp8 GEMM for gfx950:  C[M,N] = epilogue(dequant(A8)[M,K] @ dequant(W8)[N,K]^T), fp32 accumulate.
// (SURVEY.md §8(f)4 / BASELINE configs[4]: the fp8 MFMA weight path; opt-in, the default path stays bf16 - gemm.hip.)
//
// Operands are MXFP8 (include/mjv.h "MXFP8 operand format"): e4m3 bytes, K contiguous, one e8m0 scale per 32-element block.
// The kernel is the 256 x 256 tile kernel of gemm.hip with the K-tile doubled to 128 ELEMENTS = the same 128 BYTES per LDS
// row: the same LDS image (2 K-tiles x 4 half-tiles of 16 KiB, chunk swizzle on the DMA source address), the same two-phase
// loop, barriers and counted waits, and v_mfma_scale_f32_16x16x128_f8f6f4 in place of two v_mfma_f32_16x16x32_bf16 - twice
// the cycles for four times the K (measured: tools/micro/mfma_scale_probe.hip, 33.5 vs 17.5 cycles), i.e. the same MFMA time
// and the same LDS fill per K-tile for twice the flops.
//
// What the hardware wants (measured with one-hot operands, tools/micro/mfma_scale_probe2.hip, profiles/r04_b_*): lane
// (r = lane & 15, g = lane >> 4) holds, in its first four operand registers, k = 16 g .. 16 g + 15 of row r and in the last
// four k = 64 + 16 g .. 64 + 16 g + 15 - the 16-byte chunks g and 4 + g of the 128-byte row, exactly the two chunks the
// bf16 kernel's lane reads for its two K = 32 steps - and its scale register carries the e8m0 of (row r, 32-block g).  The
// scales travel as 256-byte records per (K-tile, 64-row group) whose dword [r][g] holds the bytes of the four 16-row
// fragments: one ds_read_b32 per group and K-tile, the fragment picked by op_sel.  They come in by LDS-DMA like the operands
// (one 16-lane instruction per wave and K-tile, a 4-slot ring of 2 KiB past the epilogue's staging area).
//
// Epilogues: the bf16 ones (same code, same rounding points), and - c_format MXFP8 - the bf16 result block-quantised in pass B
// (a row's 32 columns are 4 neighbouring lanes there), so the GELU / SiLU*up outputs go to the next fp8 GEMM without a
// separate quantise pass and at half the store bytes.
//
// Round 5: (i) the m-tile rows of an under-filled last round, and launches that under-fill the chip as a whole, run K-SLICED
// (gemm256f8_kernel<.., SPLIT> writes fp32 accumulator images, splitk_finish256f8_kernel sums them in slice order and runs
// every epilogue incl. the block quantiser) - mjv_gemm_mxfp8_dispatch plans it like gemm.hip plans the bf16 tails; (ii) the
// GELU epilogue takes the split-table form of gemm.hip's round-4 pass A (G8_* below).
#include "mx8.h"
#include "gelu_table.h"
#include <atomic>

namespace {

__device__ const u16 g_gelu_table8[MJV_GELU_TABLE_LEN] = MJV_GELU_TABLE_INIT;   // (gemm.hip has its own copy: no -fgpu-rdc)

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct Gemm8Args {
  const uint8_t* A; long lda;        // e4m3 [M][lda]
  const uint8_t* W; long ldw;        // e4m3 [N][ldw]
  const uint8_t* As; long a_groups;  // scale records [K / 128][a_groups][256]
  const uint8_t* Ws; long w_groups;
  void* C; long ldc;                 // bf16 elements or e4m3 bytes
  uint8_t* Cs; long c_groups;        // scale records of an e4m3 output
  int M, N, K;
  const u16* bias;
  const u16* scale;
  const u16* res; long ldr;
  int tiles_m, tiles_n;
  int gm;
  int nt_store;
  long a_grp_max;                    // last valid 64-row scale group relative to As (As / A may point at peeled tail rows)
  float* ws; int split;              // K-sliced launches: fp32 accumulator images (gemm.hip's layout), slices per tile
};

MJV_DEV float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
MJV_DEV unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// general form of the table GELU (gemm.hip gelu_lut): x exactly representable in bf16
MJV_DEV float gelu_lut(float xf, const u16* tab) {
  const unsigned u = __float_as_uint(xf);
  const unsigned mag = (u >> 16) & 0x7fffu;
  const unsigned rel = mag - MJV_GELU_LO;
  const unsigned sgn = (unsigned)((int)u >> 31);
  const bool in_tab = rel < (unsigned)MJV_GELU_R;
  const unsigned idx = in_tab ? rel + (sgn & MJV_GELU_NEG_OFF) : 0u;
  const unsigned t = tab[idx];
  const unsigned big = gelu_beyond_table(u, mag);               // beyond the table (mjv_common.h)
  const unsigned small = __float_as_uint(0.5f * xf);
  const unsigned other = mag < MJV_GELU_LO ? small : big;
  return __uint_as_float(in_tab ? (t << 16) : other);
}

// XCD-aware tile order (gemm.hip tile_of_vblock): b-th of nwg tiles
MJV_DEV void tile_of_block(const Gemm8Args& p, const int nwg, const int b, int& tm, int& tn) {
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

constexpr int BM = 256, BN = 256, BK = 128;   // BK in elements = bytes
constexpr int HALF_BYTES = 128 * BK;          // 16 KiB: 128 rows x 128 k
constexpr int PIPE_BYTES = 8 * HALF_BYTES;    // 2 K-tiles x {W0, W1, A0, A1}
constexpr int EPI_PITCH = 256 * 2 + 16;
constexpr int EPI_TILE_BYTES = 256 * EPI_PITCH;
constexpr int GELU_BYTES = MJV_GELU_TABLE_LEN * 2;
constexpr int SCALE_OFF = EPI_TILE_BYTES + GELU_BYTES;   // ring of 4 K-tiles' scale records: [W groups 0-3 | A groups 0-3] x 256 B
constexpr int SCALE_SLOT = 2048;
constexpr int LDS_BYTES = SCALE_OFF + 4 * SCALE_SLOT;
static_assert(EPI_TILE_BYTES >= PIPE_BYTES && LDS_BYTES <= 160 * 1024 && SCALE_OFF % 16 == 0, "LDS budget");
// GELU epilogue (round 5: gemm.hip's round-4 form): the table's two sign halves sit 64 KiB apart in LDS - x > 0 at byte 0, x < 0
// at byte 65536 - so that the gather address is the rounded value's own bit pattern, (fp32 bits >> 15) - 2 LO = 65536 sign +
// 2 (magnitude - LO), and the range test is one wave-wide vote on running min / max: 5.5 vector instructions per element where the
// contiguous table took 9.  This kernel's pipeline owns bytes 0 .. 128 Ki during the main loop, so the table waits behind it
// (where the prologue's DMA put it) and is moved to its two windows at the head of the epilogue (960 16-byte chunks, one barrier);
// the staged output tile is split around the second window: rows 0 .. 108 between the windows, rows 109 .. 255 above.
constexpr int G8_NEG = 65536;
constexpr int G8_HALF = MJV_GELU_NEG_OFF * 2;                 // 7 680 bytes per sign half
constexpr int G8_A0 = G8_HALF, G8_AROWS = (G8_NEG - G8_A0) / EPI_PITCH;
constexpr int G8_B0 = G8_NEG + G8_HALF;
static_assert(MJV_GELU_TABLE_LEN == 2 * MJV_GELU_NEG_OFF && MJV_GELU_R <= MJV_GELU_NEG_OFF && G8_HALF % 16 == 0, "table halves");
static_assert(G8_A0 + G8_AROWS * EPI_PITCH <= G8_NEG && G8_B0 + (256 - G8_AROWS) * EPI_PITCH <= LDS_BYTES && G8_B0 % 16 == 0, "GELU LDS map");
template <bool GL>
MJV_DEV int erow8(int ml) {   // LDS byte offset of row ml of the staged output tile
  return GL ? ml * EPI_PITCH + (ml < G8_AROWS ? G8_A0 : G8_B0 - G8_AROWS * EPI_PITCH) : ml * EPI_PITCH;
}
// general form of the table GELU on the split table (gelu_lut's logic; lds = LDS base)
MJV_DEV float gelu_lut_split8(float xf, const char* lds) {
  const unsigned u = __float_as_uint(xf);
  const unsigned mag = (u >> 16) & 0x7fffu;
  const unsigned rel = mag - MJV_GELU_LO;
  const unsigned sgn = (unsigned)((int)u >> 31);
  const bool in_tab = rel < (unsigned)MJV_GELU_R;
  const unsigned t = *(const u16*)(lds + (sgn & G8_NEG) + (in_tab ? rel * 2 : 0u));
  const unsigned big = gelu_beyond_table(u, mag);
  const unsigned small = __float_as_uint(0.5f * xf);
  const unsigned other = mag < MJV_GELU_LO ? small : big;
  return __uint_as_float(in_tab ? (t << 16) : other);
}

struct StagePtrs {
  const uint8_t* src[4][2];
};
MJV_DEV void init_stage_ptrs(StagePtrs& sp, const Gemm8Args& p, int m0, int n0, int wave, int lane) {
#pragma unroll
  for (int which = 0; which < 4; ++which) {
    const bool is_w = which < 2;
    const uint8_t* base = is_w ? p.W : p.A;
    const long ld = is_w ? p.ldw : p.lda;
    const int row0 = (is_w ? n0 : m0) + (which & 1) * 128;
    const int max_row = (is_w ? p.N : p.M) - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);
      int gr = row0 + r;
      gr = gr < max_row ? gr : max_row;
      sp.src[which][i] = base + (long)gr * ld + c * 16;
    }
  }
}
template <int WHICH>
MJV_DEV void stage_half(const StagePtrs& sp, int t, int nk, char* smem, int wave) {
  if (t >= nk) return;
  char* dst = smem + ((t & 1) * 4 + WHICH) * HALF_BYTES + wave * 2048;
  const int k0 = t * BK;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp.src[WHICH][i] + k0),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
}

// streaming (nt) stores as inline assembly: `if (nt) __builtin_nontemporal_store(..) else plain` is merged into one PLAIN
// store by hipcc (ROCm 7.2) - see gemm.hip store16
MJV_DEV void store16(u16* dst, const u32x4& val, int nt) {
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(val) : "memory");
  else *(u32x4*)dst = val;
}
MJV_DEV void store8(uint8_t* dst, const u32x2& val, int nt) {
  if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(dst), "v"(val) : "memory");
  else *(u32x2*)dst = val;
}

#define MJV_BARRIER()                      \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

template <int OA, int OB>
MJV_DEV f32x4 mfma8(const i32x8& a, const i32x8& b, const f32x4& c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, sa, OB, sb);
}

// SPLIT: the launch has p.split K-slices per tile (workgroup b = tile b / split, slice b % split: gemm.hip's VAR 7): the
// accumulators go to the fp32 workspace as they sit in registers and splitk_finish256f8_kernel sums them in slice order
template <int EPI, bool OUT8, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void gemm256f8_kernel(Gemm8Args p) {
  static_assert(!(OUT8 && EPI == MJV_EPI_SCALE_RES), "the residual epilogue writes the bf16 stream");
  static_assert(!SPLIT || (EPI == MJV_EPI_BIAS && !OUT8), "one epilogue-free instantiation serves every sliced launch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  int tm, tn;
  const int tile_id = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x;
  const int slice = SPLIT ? (int)blockIdx.x % p.split : 0;
  tile_of_block(p, SPLIT ? (int)gridDim.x / p.split : (int)gridDim.x, tile_id, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk_all = p.K / BK;
  const int kt0 = SPLIT ? (int)((long)nk_all * slice / p.split) : 0;          // this slice's K-tiles [kt0, kt0 + nk)
  const int nk = (SPLIT ? (int)((long)nk_all * (slice + 1) / p.split) : nk_all) - kt0;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int sw = l15 & 7;
  int a_off[2], w_off[2];  // [kk]: chunk kk * 4 + l4 = bytes 64 kk + 16 l4 .. of the row
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_off[kk] = l15 * 128 + (((kk * 4 + l4) ^ sw) << 4);
    w_off[kk] = ((wc & 1) * 64 + l15) * 128 + (((kk * 4 + l4) ^ sw) << 4);
  }
  const int a_half = 2 + wr;
  const int w_half = wc >> 1;
  // this lane's dword inside a 256-byte scale record, and the records of this wave inside a ring slot
  const int s_lane = l15 * 16 + l4 * 4;
  const int s_w = wc * 256 + s_lane;                     // weight rows 64 wc .. 64 wc + 63 of the tile
  const int s_a0 = 1024 + (wr * 2) * 256 + s_lane;       // activation rows 128 wr .. + 63
  const int s_a1 = s_a0 + 256;                           //                 128 wr + 64 .. + 127

  u32x2 braw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    braw[j] = u32x2{0u, 0u};
    const int nl = wc * 64 + j * 16 + l4 * 4;
    if (EPI != MJV_EPI_SILU_MUL && p.bias && n0 + nl < p.N) braw[j] = *(const u32x2*)(p.bias + n0 + nl);
  }
  constexpr int OUT_COLS_E = (EPI == MJV_EPI_SILU_MUL) ? 128 : 256;
  u32x4 scraw = {0u, 0u, 0u, 0u};
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
    const int ne = n0 + (tid % (OUT_COLS_E / 8)) * 8;
    if (p.scale && ne < p.N) scraw = *(const u32x4*)(p.scale + ne);
  }
  StagePtrs sp;
  init_stage_ptrs(sp, p, m0, n0, wave, lane);
  if constexpr (SPLIT) {
#pragma unroll
    for (int which = 0; which < 4; ++which)
#pragma unroll
      for (int i = 0; i < 2; ++i) sp.src[which][i] += kt0 * BK;
  }
  if constexpr (EPI == MJV_EPI_BIAS_GELU) {
    static_assert(MJV_GELU_TABLE_LEN % 8 == 0 && MJV_GELU_TABLE_LEN / 8 <= 1024, "two 16-byte chunks per thread cover the table");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = (i * 8 + wave) * 64 + lane;
      if (chunk < MJV_GELU_TABLE_LEN / 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const u32x4*)g_gelu_table8 + chunk),
                                         (__attribute__((address_space(3))) void*)(smem + EPI_TILE_BYTES + (i * 8 + wave) * 1024), 16, 0, 0);
    }
  }
  // scale records: waves 0-3 fetch the weight groups 0-3 of the tile, waves 4-7 the activation groups, 16 lanes x 16 B each
  const uint8_t* sc_src;
  long sc_stride;
  {
    const bool is_w = wave < 4;
    const long groups = is_w ? p.w_groups : p.a_groups;
    const long last = is_w ? p.w_groups - 1 : p.a_grp_max;
    long grp = ((is_w ? n0 : m0) >> 6) + (wave & 3);
    grp = grp < last ? grp : last;   // (groups beyond the problem: rows the epilogue never stores)
    sc_stride = groups * 256;
    sc_src = (is_w ? p.Ws : p.As) + grp * 256 + (lane & 15) * 16 + (long)kt0 * sc_stride;
  }
  auto stage_scales = [&](int t) {
    if (t < nk && lane < 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc_src + (long)t * sc_stride),
                                       (__attribute__((address_space(3))) void*)(smem + SCALE_OFF + (t & 3) * SCALE_SLOT + wave * 256), 16, 0, 0);
  };
  // ---- prologue: K-tile 0 completely, W halves of K-tile 1, the scales of both
  stage_half<0>(sp, 0, nk, smem, wave);
  stage_half<1>(sp, 0, nk, smem, wave);
  stage_half<2>(sp, 0, nk, smem, wave);
  stage_half<3>(sp, 0, nk, smem, wave);
  stage_scales(0);
  stage_half<0>(sp, 1, nk, smem, wave);
  stage_half<1>(sp, 1, nk, smem, wave);
  stage_scales(1);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
  if (nk > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // all but the five instructions of K-tile 1
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MJV_BARRIER();
  if (wr == 1) MJV_BARRIER();  // stagger the second M-group by one barrier

  i32x8 af[4], wf[2][2];  // af[i]; wf[ns][j]: 32 bytes = chunks l4 and 4 + l4 of the row
  int sW = 0, sA0 = 0, sA1 = 0;

#define MJV_FRAG(base, off0, off1) \
  __builtin_shufflevector(*(const i32x4*)((base) + (off0)), *(const i32x4*)((base) + (off1)), 0, 1, 2, 3, 4, 5, 6, 7)
#define MJV_LOAD_A(MS) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = MJV_FRAG(abase + ((MS) * 64 + i * 16) * 128, a_off[0], a_off[1]);
#define MJV_LOAD_W(NS) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) wf[NS][j] = MJV_FRAG(wbase + ((NS) * 32 + j * 16) * 128, w_off[0], w_off[1]);
  // the weight fragment (NS, J) is fragment 2 NS + J of the wave's 64-row weight group, the activation fragment I of (MS) is
  // fragment I of group 2 wr + MS: the op_sel bytes of sW / sA<MS>
#define MJV_MF(MS, NS, I, J, SA) \
  acc[(MS) * 4 + (I)][(NS) * 2 + (J)] = mfma8<(NS) * 2 + (J), (I)>(wf[NS][J], af[I], acc[(MS) * 4 + (I)][(NS) * 2 + (J)], sW, SA);
  // The MFMA intrinsic has no side effect the instruction selector has to order: without a tie to the chain of the volatile
  // statements around it (barriers, waits), this kernel's MFMAs were all emitted in ONE clump behind the phase-II reads, both
  // MFMA segments empty (the bf16 kernel's stay put by luck of the same heuristics).  Pinned from both sides: the scale
  // registers pass through an empty volatile asm right after the segment's opening barrier, the accumulators through one
  // right before its closing barrier.
#define MJV_PIN_ACC(MS, NS)                                                                                                   \
  asm volatile("" ::"v"(acc[(MS) * 4 + 0][(NS) * 2]), "v"(acc[(MS) * 4 + 0][(NS) * 2 + 1]), "v"(acc[(MS) * 4 + 1][(NS) * 2]),     \
               "v"(acc[(MS) * 4 + 1][(NS) * 2 + 1]), "v"(acc[(MS) * 4 + 2][(NS) * 2]), "v"(acc[(MS) * 4 + 2][(NS) * 2 + 1]),       \
               "v"(acc[(MS) * 4 + 3][(NS) * 2]), "v"(acc[(MS) * 4 + 3][(NS) * 2 + 1]));
#define MJV_MFMA(MS, NS, SA)                                                                    \
  asm volatile("" : "+v"(sW), "+v"(SA));                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                \
  MJV_MF(MS, NS, 0, 0, SA) MJV_MF(MS, NS, 0, 1, SA) MJV_MF(MS, NS, 1, 0, SA) MJV_MF(MS, NS, 1, 1, SA) \
  MJV_MF(MS, NS, 2, 0, SA) MJV_MF(MS, NS, 2, 1, SA) MJV_MF(MS, NS, 3, 0, SA) MJV_MF(MS, NS, 3, 1, SA) \
  MJV_PIN_ACC(MS, NS)                                                                           \
  __builtin_amdgcn_s_setprio(0);

  // the two-phase loop of gemm256_kernel (gemm.hip: barrier intervals, buffer lifetimes); per K-tile each wave issues
  // 2 + 2 operand DMA and one scale DMA in phase I / II - the counted wait leaves the five of K-tile t + 2 in flight.  The scale
  // records of K-tile t + 2 go to ring slot (t + 2) & 3, last read for K-tile t - 2; they are retired like the W halves issued
  // with them (the wait of iteration t + 1, a barrier, then the reads of iteration t + 2).
  for (int t = 0; t < nk; ++t) {
    const char* abase = smem + ((t & 1) * 4 + a_half) * HALF_BYTES;
    const char* wbase = smem + ((t & 1) * 4 + w_half) * HALF_BYTES;
    const char* sbase = smem + SCALE_OFF + (t & 3) * SCALE_SLOT;
    MJV_LOAD_W(0)
    MJV_LOAD_W(1)
    MJV_LOAD_A(0)
    sW = *(const int*)(sbase + s_w);
    sA0 = *(const int*)(sbase + s_a0);
    stage_half<2>(sp, t + 1, nk, smem, wave);
    stage_half<3>(sp, t + 1, nk, smem, wave);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MJV_BARRIER();
    MJV_MFMA(0, 0, sA0)
    MJV_MFMA(0, 1, sA0)
    MJV_BARRIER();
    MJV_LOAD_A(1)
    sA1 = *(const int*)(sbase + s_a1);
    stage_half<0>(sp, t + 2, nk, smem, wave);
    stage_half<1>(sp, t + 2, nk, smem, wave);
    stage_scales(t + 2);
    if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MJV_BARRIER();
    MJV_MFMA(1, 1, sA1)
    MJV_MFMA(1, 0, sA1)
    MJV_BARRIER();
  }
  if (wr == 0) MJV_BARRIER();
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) in the form the compiler's wait-count pass reads (gemm.hip)
  if constexpr (SPLIT) {
    f32x4* img = (f32x4*)p.ws + ((long)tile_id * p.split + slice) * (32 * 512) + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) img[(i * 4 + j) * 512] = acc[i][j];
    return;
  }
#undef MJV_FRAG
#undef MJV_LOAD_A
#undef MJV_LOAD_W
#undef MJV_MF
#undef MJV_MFMA
#undef MJV_PIN_ACC

  // ---- epilogue: pass A (fragments -> bf16 tile in LDS, bias / activation), pass B (rows -> global), as gemm256_kernel
  char* etile = smem;
  constexpr bool GL = EPI == MJV_EPI_BIAS_GELU;
  if constexpr (GL) {
    // the table from behind the pipeline (contiguous: [x > 0 | x < 0]) to its two windows; every wave is past its last fragment read
    static_assert(MJV_GELU_TABLE_LEN / 8 <= 1024 && G8_HALF / 16 == 480, "two chunks per thread cover the table");
    const u32x4* src = (const u32x4*)(smem + EPI_TILE_BYTES);
    u32x4 c0, c1 = {0u, 0u, 0u, 0u};
    c0 = src[tid];
    if (tid + 512 < MJV_GELU_TABLE_LEN / 8) c1 = src[tid + 512];
    *(u32x4*)(smem + (tid < 480 ? tid * 16 : G8_NEG + (tid - 480) * 16)) = c0;
    if (tid + 512 < MJV_GELU_TABLE_LEN / 8) *(u32x4*)(smem + G8_NEG + (tid + 512 - 480) * 16) = c1;
    __syncthreads();
  }
  constexpr int OUT_COLS = (EPI == MJV_EPI_SILU_MUL) ? 128 : 256;
  constexpr int LANES_PER_ROW = OUT_COLS / 8;
  constexpr int ROWS_PER_PASS = 512 / LANES_PER_ROW;
  constexpr int PASSES = 256 / ROWS_PER_PASS;
  const int c8 = (tid % LANES_PER_ROW) * 8;
  const int nout0 = (EPI == MJV_EPI_SILU_MUL) ? n0 / 2 : n0;
  const int nlim = (EPI == MJV_EPI_SILU_MUL) ? p.N / 2 : p.N;
  const int n = nout0 + c8;
  const int ml0 = tid / LANES_PER_ROW;
  u32x4 rsv[(EPI == MJV_EPI_SCALE_RES) ? PASSES : 1];
  float sc[8];
  if constexpr (EPI == MJV_EPI_SCALE_RES) {
    unpack8(scraw, sc);
    const u16* rp = p.res + (long)(m0 + ml0) * p.ldr + n;
    const long rstep = (long)ROWS_PER_PASS * p.ldr;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      rsv[it] = u32x4{0u, 0u, 0u, 0u};
      if (m0 + it * ROWS_PER_PASS + ml0 < p.M && n < nlim) rsv[it] = *(const u32x4*)(rp + it * rstep);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nl = wc * 64 + j * 16 + l4 * 4;
    const u32x2 bb = braw[j];
    const float b4[4] = {__uint_as_float(bb[0] << 16), __uint_as_float(bb[0] & 0xffff0000u),
                         __uint_as_float(bb[1] << 16), __uint_as_float(bb[1] & 0xffff0000u)};
    if (EPI == MJV_EPI_SILU_MUL && (j & 1)) continue;
    if constexpr (EPI == MJV_EPI_BIAS_GELU) {
      // a whole column group (8 fragments = 32 elements per lane) at a time: ONE wave-wide range test on the running min / max of
      // |x| (IEEE-2019 maximum / minimum: a NaN makes the vote fail), then 32 gathers in flight together (gemm.hip pass A)
      unsigned ubs[8][4];
      float amax = 0.f, amin = __uint_as_float(0x7f000000u);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xr = rbf(acc[i][j][r] + b4[r]);
          ubs[i][r] = __float_as_uint(xr);
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
          const u32x2 o = {t[i][0] | (t[i][1] << 16), t[i][2] | (t[i][3] << 16)};
          *(u32x2*)(etile + erow8<GL>(ml) + nl * 2) = o;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int ml = wr * 128 + i * 16 + l15;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_lut_split8(__uint_as_float(ubs[i][r]), smem);
          const u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
          *(u32x2*)(etile + erow8<GL>(ml) + nl * 2) = o;
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
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(silu(rbf(acc[i][j][r]))) * rbf(acc[i][j + 1][r]);
        col = wc * 32 + (j >> 1) * 16 + l4 * 4;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + b4[r];
        if constexpr (EPI == MJV_EPI_BIAS_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        col = nl;
      }
      const u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
      *(u32x2*)(etile + ml * EPI_PITCH + col * 2) = o;
    }
  }
  __syncthreads();
  u32x4 vals[PASSES];
#pragma unroll
  for (int it = 0; it < PASSES; ++it) vals[it] = *(const u32x4*)(etile + erow8<GL>(it * ROWS_PER_PASS + ml0) + c8 * 2);
  if constexpr (OUT8) {
    // the row's 32-column blocks are the quads of lanes: block-quantise the bf16 values on the way out (every lane takes part
    // in the quad exchange, rows / columns beyond the problem are simply not stored; nlim % 128 == 0 keeps quads whole)
    uint8_t* crow = (uint8_t*)p.C + (long)(m0 + ml0) * p.ldc + n;
    const long cstep = (long)ROWS_PER_PASS * p.ldc;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int ml = it * ROWS_PER_PASS + ml0;
      unsigned sb;
      const u32x2 q = mx8_quantize_quad(vals[it], sb);
      if (m0 + ml >= p.M || n >= nlim) continue;
      store8(crow + it * cstep, q, p.nt_store);
      if ((tid & 3) == 0) p.Cs[mx8_scale_offset(m0 + ml, n, p.c_groups)] = (uint8_t)sb;
    }
  } else {
    u16* crow = (u16*)p.C + (long)(m0 + ml0) * p.ldc + n;
    const long cstep = (long)ROWS_PER_PASS * p.ldc;
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
      store16(crow + it * cstep, val, p.nt_store);
    }
  }
}

// Second half of a K-sliced launch.  A wave finishes 16 rows x 32 output columns (one MXFP8 block per row): lane = row
// (lane & 15) x 8-column group (lane >> 4), so the slices' images are read in 256-byte runs (16 rows of one fragment column
// group) and a block's four column groups sit in lanes l, l + 16, l + 32, l + 48.  Sums the slices in slice order
// (deterministic), then the epilogue of gemm256f8_kernel - same operations, same rounding points - and, for an MXFP8 output,
// the block quantiser.  Workgroup = 4 waves = 16 rows x 128 output columns.
template <int EPI, bool OUT8>
__global__ __launch_bounds__(256) void splitk_finish256f8_kernel(Gemm8Args p) {
  constexpr int OUT_COLS = (EPI == MJV_EPI_SILU_MUL) ? 128 : 256;
  constexpr int CB = OUT_COLS / 128;                // workgroups per 16-row group of a tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_tile = 16 * CB;
  const int tile_id = (int)blockIdx.x / per_tile, rest = (int)blockIdx.x % per_tile;
  int tm, tn;
  tile_of_block(p, (int)gridDim.x / per_tile, tile_id, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int ml = (rest / CB) * 16 + (lane & 15);                       // row inside the tile
  const int oc = ((rest % CB) * 4 + wave) * 32 + (lane >> 4) * 8;      // first of this lane's 8 output columns inside the tile
  const int nout0 = (EPI == MJV_EPI_SILU_MUL) ? n0 / 2 : n0;
  const int nlim = (EPI == MJV_EPI_SILU_MUL) ? p.N / 2 : p.N;
  const int n = nout0 + oc;
  // image index (in f32x4) of (row ml, tile column tc % 4 == 0): fragment (i, j) of thread (wr, wc, l4, l15) of the GEMM workgroup
  auto img_idx = [&](int tc) {
    const int wr = ml >> 7, i = (ml >> 4) & 7, l15 = ml & 15;
    const int wc = tc >> 6, j = (tc >> 4) & 3, l4 = (tc >> 2) & 3;
    return (i * 4 + j) * 512 + (wr * 4 + wc) * 64 + l4 * 16 + l15;
  };
  constexpr int NF = (EPI == MJV_EPI_SILU_MUL) ? 4 : 2;
  int idx[NF];
  if constexpr (EPI == MJV_EPI_SILU_MUL) {
    // output columns 16 a + b of a 32-column group come from tile columns 32 a + b (gate) and 32 a + 16 + b (up)
    const int tg = (oc >> 4) * 32 + (oc & 15);
    idx[0] = img_idx(tg); idx[1] = img_idx(tg + 4); idx[2] = img_idx(tg + 16); idx[3] = img_idx(tg + 20);
  } else {
    idx[0] = img_idx(oc); idx[1] = img_idx(oc + 4);
  }
  const f32x4* img = (const f32x4*)p.ws + (long)tile_id * p.split * (32 * 512);
  f32x4 fr[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) fr[f] = img[idx[f]];
  for (int sl = 1; sl < p.split; ++sl) {
#pragma unroll
    for (int f = 0; f < NF; ++f) fr[f] += img[(long)sl * (32 * 512) + idx[f]];
  }
  const bool live = m0 + ml < p.M && n < nlim;      // (every lane stays for the block exchange of an MXFP8 output)
  float v[8];
  if constexpr (EPI == MJV_EPI_SILU_MUL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rbf(silu(rbf(fr[e >> 2][e & 3]))) * rbf(fr[2 + (e >> 2)][e & 3]);
  } else {
    float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && live) unpack8(*(const u32x4*)(p.bias + n), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fr[e >> 2][e & 3] + b[e];
    if constexpr (EPI == MJV_EPI_BIAS_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_lut(rbf(v[e]), g_gelu_table8);
    }
    if constexpr (EPI == MJV_EPI_BIAS_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if constexpr (EPI == MJV_EPI_SCALE_RES) {
      float sc[8], rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (live) unpack8(*(const u32x4*)(p.res + (long)(m0 + ml) * p.ldr + n), rs);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rbf(v[e]);
      if (p.scale && live) {
        unpack8(*(const u32x4*)(p.scale + n), sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] * sc[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rs[e];
    }
  }
  const u32x4 val = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
  if constexpr (OUT8) {
    unsigned m = mx8_amax8(val);
    m = max(m, (unsigned)__shfl_xor((int)m, 16, 64));
    m = max(m, (unsigned)__shfl_xor((int)m, 32, 64));
    const unsigned sb = mx8_scale_byte(m);
    const u32x2 q = mx8_cvt8(val, mx8_scale_f32(sb));
    if (!live) return;
    *(u32x2*)((uint8_t*)p.C + (long)(m0 + ml) * p.ldc + n) = q;
    if ((lane >> 4) == 0) p.Cs[mx8_scale_offset(m0 + ml, n, p.c_groups)] = (uint8_t)sb;
  } else {
    if (!live) return;
    *(u32x4*)((u16*)p.C + (long)(m0 + ml) * p.ldc + n) = val;
  }
}

template <int EPI, bool OUT8>
int launch8(Gemm8Args a, hipStream_t s) {
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)gemm256f8_kernel<EPI, OUT8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.N + 255) / 256;
  if (a.split > 1) {
    static std::atomic<unsigned long long> attr_split{0};
    if (!(attr_split.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute((const void*)gemm256f8_kernel<MJV_EPI_BIAS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      attr_split.fetch_or(bit, std::memory_order_release);
    }
    const int tiles = a.tiles_m * a.tiles_n;
    constexpr int CB = (EPI == MJV_EPI_SILU_MUL) ? 1 : 2;
    hipLaunchKernelGGL((gemm256f8_kernel<MJV_EPI_BIAS, false, true>), dim3(tiles * a.split), dim3(512), LDS_BYTES, s, a);
    hipLaunchKernelGGL((splitk_finish256f8_kernel<EPI, OUT8>), dim3(tiles * 16 * CB), dim3(256), 0, s, a);
    return mjv_check_launch("gemm_mxfp8 (K-sliced)");
  }
  hipLaunchKernelGGL((gemm256f8_kernel<EPI, OUT8>), dim3(a.tiles_m * a.tiles_n), dim3(512), LDS_BYTES, s, a);
  return mjv_check_launch("gemm_mxfp8");
}

}  // namespace

// called by mjv_gemm_bf16 (gemm.hip) when the descriptor names MXFP8 operands
int mjv_gemm_mxfp8_dispatch(const mjv_gemm_desc* d, void* stream) {
  MJV_REQUIRE(d->a_format == MJV_FMT_MXFP8 && d->w_format == MJV_FMT_MXFP8, "gemm: A and W must both be MXFP8 (or both bf16)");
  MJV_REQUIRE(d->a_scales && d->w_scales, "gemm(mxfp8): null scale pointer");
  MJV_REQUIRE(d->K % 128 == 0, "gemm(mxfp8): K=%d must be a multiple of 128", d->K);
  MJV_REQUIRE(d->N % 8 == 0, "gemm(mxfp8): N=%d must be a multiple of 8", d->N);
  MJV_REQUIRE(d->lda % 16 == 0 && d->ldw % 16 == 0 && d->lda >= d->K && d->ldw >= d->K, "gemm(mxfp8): lda / ldw");
  MJV_REQUIRE(((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->a_scales | (uintptr_t)d->w_scales) % 16 == 0, "gemm(mxfp8): misaligned pointer");
  MJV_REQUIRE(!d->out_rows && d->out_group <= 0 && d->res_mod <= 0, "gemm(mxfp8): plain output / residual rows only");
  MJV_REQUIRE(d->epilogue >= MJV_EPI_BIAS && d->epilogue <= MJV_EPI_SILU_MUL, "gemm(mxfp8): epilogue %d unsupported", d->epilogue);
  const bool out8 = d->c_format == MJV_FMT_MXFP8;
  MJV_REQUIRE(d->c_format == MJV_FMT_BF16 || out8, "gemm(mxfp8): c_format %d", d->c_format);
  const int nout = d->epilogue == MJV_EPI_SILU_MUL ? d->N / 2 : d->N;
  if (out8) {
    MJV_REQUIRE(d->epilogue != MJV_EPI_SCALE_RES, "gemm(mxfp8): the residual epilogue has no MXFP8 output");
    MJV_REQUIRE(d->c_scales && nout % 128 == 0 && d->ldc % 8 == 0 && (uintptr_t)d->C % 8 == 0, "gemm(mxfp8): MXFP8 output needs scales, width %% 128 == 0");
  } else {
    MJV_REQUIRE(d->ldc % 8 == 0 && (uintptr_t)d->C % 16 == 0, "gemm(mxfp8): ldc / C alignment");
  }
  if (d->epilogue == MJV_EPI_SCALE_RES) MJV_REQUIRE(d->res != nullptr && d->ldr % 8 == 0, "gemm(mxfp8): SCALE_RES needs a residual");
  if (d->epilogue == MJV_EPI_SILU_MUL) MJV_REQUIRE(d->N % 32 == 0 && d->bias == nullptr, "gemm(mxfp8): SILU_MUL needs N %% 32 == 0 and no bias");
  Gemm8Args a;
  a.A = (const uint8_t*)d->A; a.lda = d->lda; a.W = (const uint8_t*)d->W; a.ldw = d->ldw;
  a.As = d->a_scales; a.a_groups = (d->M + 63) / 64; a.Ws = d->w_scales; a.w_groups = (d->N + 63) / 64;
  a.a_grp_max = a.a_groups - 1;
  a.C = d->C; a.ldc = d->ldc; a.Cs = d->c_scales; a.c_groups = (d->M + 63) / 64;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.bias = d->bias; a.scale = d->scale; a.res = d->res; a.ldr = d->ldr;
  a.tiles_m = a.tiles_n = 0;
  a.ws = nullptr; a.split = 1;
  a.gm = d->K <= 2048 ? 5 : (d->K >= 16384 || d->N >= 8192) ? 4 : 8;   // (gemm.hip pick_gm by K BYTES per row)
  a.nt_store = ((double)d->M * nout * (out8 ? 1.0 : 2.0) >= 32.0 * 1024 * 1024) ? 1 : 0;   // (gemm.hip: streaming stores for large outputs)
  hipStream_t s = (hipStream_t)stream;
  static const char* const tags[] = {"gemm256f8_bias", "gemm256f8_bias_gelu", "gemm256f8_bias_relu", "gemm256f8_scale_res", "gemm256f8_silu_mul"};
  static const char* const tags_s[] = {"gemm256f8s_bias", "gemm256f8s_bias_gelu", "gemm256f8s_bias_relu", "gemm256f8s_scale_res", "gemm256f8s_silu_mul"};
  const double flops = 2.0 * d->M * (double)d->N * d->K;
  const double bytes = 1.03 * ((double)d->M * d->K + (double)d->N * d->K) + (out8 ? 1.03 : 2.0) * (double)d->M * nout;
  auto run = [&](const Gemm8Args& g, double frac) -> int {
    MjvProfScope ps(g.split > 1 ? tags_s[d->epilogue] : tags[d->epilogue], s, flops * frac, bytes * frac);
    switch (d->epilogue) {
      case MJV_EPI_BIAS: return out8 ? launch8<MJV_EPI_BIAS, true>(g, s) : launch8<MJV_EPI_BIAS, false>(g, s);
      case MJV_EPI_BIAS_GELU: return out8 ? launch8<MJV_EPI_BIAS_GELU, true>(g, s) : launch8<MJV_EPI_BIAS_GELU, false>(g, s);
      case MJV_EPI_BIAS_RELU: return out8 ? launch8<MJV_EPI_BIAS_RELU, true>(g, s) : launch8<MJV_EPI_BIAS_RELU, false>(g, s);
      case MJV_EPI_SCALE_RES: return launch8<MJV_EPI_SCALE_RES, false>(g, s);
      default: return out8 ? launch8<MJV_EPI_SILU_MUL, true>(g, s) : launch8<MJV_EPI_SILU_MUL, false>(g, s);
    }
  };
  // Wave quantisation (gemm.hip does the same for the bf16 kernels): one 256 x 256 workgroup per CU, so a launch runs in
  // ceil(tiles / CUs) rounds and a last round that is mostly empty costs a full tile time - the 64-row tail of the vision
  // tower's 65 600 rows is a 17th round of 16 tiles after fc1's 16 full ones, w2's 552 tiles are 2.16 rounds.  With a
  // workspace, the m-tile rows of an under-filled last round are peeled off and run K-SLICED over all CUs (fp32 slices
  // through the workspace, summed in slice order by splitk_finish256f8_kernel: deterministic); an under-filled launch as a
  // whole (few rows: one video per forward) is sliced the same way.  Rows are independent and 64-row scale groups stay whole
  // (m_main % 256 == 0): results differ from the unsliced launch by fp32 re-association only.
  const int cus = mjv_device_cus();
  const int tn = (d->N + 255) / 256, tmx = (d->M + 255) / 256;
  const int tiles = tn * tmx, nk = d->K / BK;
  auto plan_split = [&](Gemm8Args& g) {
    if (!d->workspace || nk < 8) return;     // (K < 1024: two launches cost more than the idle CUs)
    const int t = ((g.M + 255) / 256) * tn;
    int sp = cus / t;
    if (sp > nk / 2) sp = nk / 2;            // at least two K-tiles per slice (the pipeline's prologue fills two)
    if (sp > 32) sp = 32;
    if (sp < 2 || (long)t * sp * 262144L > d->workspace_bytes) return;
    g.split = sp;
    g.ws = (float*)d->workspace;
  };
  int m_main = d->M;
  if (d->workspace && d->tile == 0) {
    const int rem = tiles % cus;
    const int rows_main_tiles = (tiles - rem) / tn;   // whole m-tile rows inside the full rounds
    if (nk >= 8 && tiles > cus && rem > 0 && rem * 2 <= cus && rows_main_tiles > 0 && rows_main_tiles * 256 < d->M) m_main = rows_main_tiles * 256;
  }
  if (m_main == d->M) {
    if (tiles * 2 <= cus && d->tile == 0) plan_split(a);
    return run(a, 1.0);
  }
  Gemm8Args head = a;
  head.M = m_main;
  int rc = run(head, (double)m_main / d->M);
  if (rc) return rc;
  // tail rows [m_main, M): every row-indexed pointer moves down m_main rows (m_main / 64 whole scale groups; the group COUNTS,
  // which are the scale buffers' strides, stay those of the whole matrix)
  Gemm8Args tail = a;
  tail.M = d->M - m_main;
  tail.A = a.A + (long)m_main * a.lda;
  tail.As = a.As + (long)(m_main / 64) * 256;
  tail.a_grp_max = a.a_groups - 1 - m_main / 64;
  tail.C = out8 ? (void*)((uint8_t*)a.C + (long)m_main * a.ldc) : (void*)((u16*)a.C + (long)m_main * a.ldc);
  if (a.Cs) tail.Cs = a.Cs + (long)(m_main / 64) * 256;
  if (a.res) tail.res = a.res + (long)m_main * a.ldr;
  plan_split(tail);
  return run(tail, (double)tail.M / d->M);
}
