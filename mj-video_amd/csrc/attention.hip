// Flash-style attention for gfx950 over packed variable-length sequences.
//
// One workgroup = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries.  K/V tiles of
// 64 keys go global -> LDS by LDS-DMA into two unpadded buffers with a source-side chunk swizzle (K rows are
// read row-wise with ds_read_b128, V rows column-wise with ds_read_b64_tr_b16, both conflict-free); the
// register-staged, row-padded single-buffer form of round 1 is kept as the DMA = false instantiation.
//
// QK^T is computed SWAPPED, S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows, B = Q^T), so a lane
// owns ONE query (MFMA column) and its 16 accumulator registers are 16 keys: row max / row sum are
// in-register plus one cross-half exchange.  The exponentiated tile is converted to bf16 in place and
// used directly as the B operand of the second product O^T = V^T P^T (cdna_hip_programming.md §3
// "An accumulator tile as the next MFMA's operand"): O^T keeps the query on the lane, so the online
// softmax rescale is a per-lane scalar multiply.
#include "mjv_common.h"
#include <math.h>
#include <type_traits>
#include <utility>
#include <algorithm>

namespace {

struct AttnArgs {
  const u16 *Q, *K, *V;
  u16* O;
  long ldq, ldk, ldv, ldo;
  int qhs, khs, vhs, ohs;
  const int* cu;
  int n_heads, kv_group;
  int causal;
  float scale;
  int round_mode;
  int n_qb, n_seqs;   // q-blocks per sequence (of the launched kernel's block size), sequences
  // ABI 6 (causal, attn2_kernel only): the queries of a sequence are the LAST rows of it (cu_q: their packed row offsets in
  // Q / O; NULL = one query per key row, the same rows), and every sequence's keys are preceded by prefix_len SHARED keys
  // (rows of Kp / Vp: same leading dimensions and head strides as K / V; prefix_len is a multiple of KB, so a key tile
  // lies on one side of the boundary)
  const int* cu_q;
  const u16 *Kp, *Vp;
  int prefix_len;
};

// Workgroup -> (sequence, head, q-block).  The grid is one-dimensional and hardware hands consecutive workgroup ids to
// the 8 XCDs round-robin, each XCD with its own L2; all q-blocks of one (sequence, kv head) - and, under GQA, of every
// q head that shares that kv head - read the same K and V, so they are given consecutive slots of ONE XCD and K / V come
// from HBM once instead of once per q-block (measured before this mapping: 2.29 GB fetched per ViT attention launch
// against 0.40 GB of q, k, v).
struct BlockId {
  int seq, head, qb;
  bool valid;
};
MJV_DEV BlockId decode_block(const AttnArgs& p) {
  const int total = p.n_qb * p.n_heads * p.n_seqs;
  const int per_xcd = (total + 7) >> 3;
  const int b = blockIdx.x;
  const int v = (b & 7) * per_xcd + (b >> 3);
  BlockId id;
  id.valid = v < total;
  id.qb = v % p.n_qb;
  const int r = v / p.n_qb;
  const int g = r % p.kv_group, r2 = r / p.kv_group;
  const int n_kv = p.n_heads / p.kv_group;
  id.head = (r2 % n_kv) * p.kv_group + g;
  id.seq = r2 / n_kv;
  return id;
}

constexpr int QB = 128;  // queries per workgroup
constexpr int KB = 64;   // keys per tile

template <int D>
struct Cfg {
  static constexpr int KP = D * 2 + 16;                 // K row pitch (bytes): odd number of 16-B slots
  static constexpr int VP = (D == 64) ? 192 : 320;      // V row pitch: 64 * odd, so 4 rows tile the 64 banks
  static constexpr int K_BYTES = KB * KP;
  static constexpr int V_BYTES = KB * VP;
  static constexpr int CHUNKS = D / 8;                  // 16-B chunks per row
  static constexpr int LOADS = KB * CHUNKS / 256;       // chunks per thread per tile
};

// score-rounding modes (template parameter): the reference rounds the scores to bf16 before the softmax
//   RM_MUL   s = bf16(acc * scale)                 [(q * scale) @ k^T, generic scale]
//   RM_POW2  s = bf16(acc) * scale                 [same thing when scale is a power of two: exact, one op less]
//   RM_DIV   s = bf16(bf16(acc) * scale)           [q @ k^T, then / sqrt(D) on the bf16 tensor]
//   RM_FLASH s = acc * scale in fp32, never rounded [the reference's flash-attention path, which is what it runs on a GPU:
//            modeling_intern_vit.py:229-244, modeling_internlm2.py:437-561; API score_round_mode 2, attn2_kernel only]
enum { RM_MUL = 0, RM_DIV = 1, RM_POW2 = 2, RM_FLASH = 3 };

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// two fp32 -> packed bf16 pair in one v_cvt_pk_bf16_f32
MJV_DEV unsigned pack_pair(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
// round two fp32 values through bf16: two conversions with the value in the high half (mjv_common.h rbf) instead of one
// packed conversion + shift + mask - 2 vector instructions per pair instead of 3 on the softmax's critical resource
MJV_DEV f32x2 round_pair(f32x2 v) { return f32x2{rbf(v[0]), rbf(v[1])}; }

// value of the partner lane (lane ^ 32) combined with one's own, through one v_permlane32_swap (no LDS round trip)
MJV_DEV float xhalf_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
MJV_DEV float xhalf_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int RM>
MJV_DEV float round_score(float a, float scale) {
  if constexpr (RM == RM_MUL) return rbf(a * scale);
  if constexpr (RM == RM_DIV) return rbf(rbf(a) * scale);
  if constexpr (RM == RM_FLASH) return a;   // unrounded; the scale is folded into the exp2 argument
  return rbf(a);  // RM_POW2: the scale is folded into the exp2 argument
}

// VAR: 0 = production.  Timing experiments (wrong results by construction, tools/attn_bench.py only; selected with
// mjv_attention_set_variant): 1 = K/V staged once (no barriers, no LDS stores, no global loads after tile 0),
// 2 = softmax removed (P = bf16(S)), 3 = MFMAs removed (LDS reads and the softmax kept).
// DMA = true (production since round 2 for sequences of up to 4096 keys): K / V tiles go global -> LDS by LDS-DMA (no staging registers, no ds_write) into
// TWO unpadded buffers (one barrier per key tile, 32 / 64 KB per workgroup: same occupancy as the padded single buffer);
// bit-identical to the register-staged form (DMA = false: variant 4 and the timing variants 1-3), +2 % at both head sizes; bank conflicts are removed by a chunk swizzle applied on the DMA's source
// address instead of by row padding: 16-byte chunk c of K row r sits at chunk c ^ sK(r), sK = (r >> 1) & 7 (D = 64) /
// r & 15 (D = 128); of V row r at c ^ sV(r), sV = ((r >> 1) & 1) * 4 / (r & 3) * 4 - both conflict-free for the lane groups
// of ds_read_b128 (K, row-wise) and ds_read_b64_tr_b16 (V, transposed) by exhaustive check of the access patterns below.
template <int D, bool CAUSAL, int RM, int VAR = 0, bool DMA = false>
__global__ __launch_bounds__(256, (D == 64) ? 4 : 2) void attn_kernel(AttnArgs p) {
  using C = Cfg<D>;
  constexpr int PK = DMA ? D * 2 : C::KP;   // row pitches in bytes
  constexpr int PV = DMA ? D * 2 : C::VP;
  constexpr int KBYTES = KB * PK, VBYTES = KB * PV, TB = KBYTES + VBYTES;
  __shared__ __attribute__((aligned(16))) char smem[DMA ? 2 * TB : TB];
  char* Ks = smem;
  char* Vs = smem + KBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every branch on it below is wave-uniform
  const int l31 = lane & 31, hi = lane >> 5;
  const BlockId bid = decode_block(p);
  if (!bid.valid) return;
  // causal: the q-blocks with the most key tiles are dispatched first, so the launch ends on the light ones
  const int seq = bid.seq, head = bid.head, qb = CAUSAL ? p.n_qb - 1 - bid.qb : bid.qb;
  const int s0 = p.cu[seq];
  const int len = p.cu[seq + 1] - s0;
  if (qb * QB >= len) return;
  const int kvh = head / p.kv_group;

  const int q0 = qb * QB + wave * 32;
  const int qi = q0 + l31;                 // query index within the sequence
  const int qrow = s0 + (qi < len ? qi : len - 1);

  // Q fragments: B operand, lane holds Q[query l31][d = 16*ks + 8*hi + j]
  bf16x8 qf[D / 16];
  {
    const u16* qp = p.Q + (long)qrow * p.ldq + (long)head * p.qhs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  f32x16 oacc[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // running max in units of the ROUNDED score (RM_POW2: before the scale)

  const int kv_end = CAUSAL ? min(len, (qb + 1) * QB) : len;
  const int n_tiles = (kv_end + KB - 1) / KB;
  const u16* Kg = p.K + (long)kvh * p.khs;
  const u16* Vg = p.V + (long)kvh * p.vhs;
  constexpr float LOG2E = 1.4426950408889634f;
  const float c_exp = (RM == RM_POW2) ? p.scale * LOG2E : LOG2E;   // p = exp2(s * c_exp - m * c_exp)

  // register staging of the NEXT tile (issued before the current tile's math, written to LDS after it).
  // Addresses are a wave-uniform tile base (scalar registers) plus a per-lane 32-bit offset computed once; only a
  // tile that straddles the sequence end takes the per-row clamped form.
  u32x4 kreg[C::LOADS], vreg[C::LOADS];
  constexpr int ROWS_PER_LOAD = 256 / C::CHUNKS;
  const int st_row = tid / C::CHUNKS, st_ch = tid % C::CHUNKS;
  const unsigned k_lane_off = (unsigned)(st_row * (int)p.ldk + st_ch * 8) * 2u;
  const unsigned v_lane_off = (unsigned)(st_row * (int)p.ldv + st_ch * 8) * 2u;
  const char* const k_seq = (const char*)(Kg + (long)s0 * p.ldk);   // wave-uniform bases; offsets within a sequence fit 32 bits
  const char* const v_seq = (const char*)(Vg + (long)s0 * p.ldv);
  auto load_tile = [&](int kt) {
    unsigned ko[C::LOADS], vo[C::LOADS];
    if (kt * KB + KB <= len) {
#pragma unroll
      for (int c = 0; c < C::LOADS; ++c) {
        const unsigned r = (unsigned)(kt * KB + c * ROWS_PER_LOAD);   // wave-uniform
        ko[c] = k_lane_off + r * (unsigned)p.ldk * 2u;
        vo[c] = v_lane_off + r * (unsigned)p.ldv * 2u;
      }
    } else {
#pragma unroll
      for (int c = 0; c < C::LOADS; ++c) {
        int kr = kt * KB + st_row + c * ROWS_PER_LOAD;
        kr = kr < len ? kr : len - 1;
        ko[c] = (unsigned)(kr * (int)p.ldk + st_ch * 8) * 2u;
        vo[c] = (unsigned)(kr * (int)p.ldv + st_ch * 8) * 2u;
      }
    }
#pragma unroll
    for (int c = 0; c < C::LOADS; ++c) {
      kreg[c] = *(const u32x4*)(k_seq + ko[c]);
      vreg[c] = *(const u32x4*)(v_seq + vo[c]);
    }
  };
  char* const k_st = Ks + st_row * PK + st_ch * 16;
  char* const v_st = Vs + st_row * PV + st_ch * 16;
  auto store_tile = [&]() {
#pragma unroll
    for (int c = 0; c < C::LOADS; ++c) {
      *(u32x4*)(k_st + c * ROWS_PER_LOAD * PK) = kreg[c];
      *(u32x4*)(v_st + c * ROWS_PER_LOAD * PV) = vreg[c];
    }
  };

  // A q-block with at most 32 queries (the ragged end of a sequence: 1025 = 8 * 128 + 1 in the vision tower) is run by
  // wave 0 alone: the other three waves leave at once and free their SIMD slots, wave 0 stages K / V by itself (no
  // prefetch, no barriers: one wave's LDS operations execute in order) - the block then costs one wave slot instead of four.
  // (D = 64 only: at D = 128 the second loop costs registers - 256 + spills for the causal kernel - and measures slower)
  const bool solo = D == 64 && len - qb * QB <= 32;   // workgroup-uniform
  if (solo && wave > 0) return;
  auto stage_solo = [&](int kt) {
#pragma unroll
    for (int c0 = 0; c0 < KB * C::CHUNKS / 64; c0 += 2) {
      u32x4 a[2], b[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int idx = lane + (c0 + c) * 64;
        const int row = idx / C::CHUNKS, ch = idx % C::CHUNKS;
        int kr = kt * KB + row;
        kr = kr < len ? kr : len - 1;
        a[c] = *(const u32x4*)(k_seq + (unsigned)(kr * (int)p.ldk + ch * 8) * 2u);
        b[c] = *(const u32x4*)(v_seq + (unsigned)(kr * (int)p.ldv + ch * 8) * 2u);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int idx = lane + (c0 + c) * 64;
        const int row = idx / C::CHUNKS, ch = idx % C::CHUNKS;
        *(u32x4*)(Ks + row * PK + ch * 16) = a[c];
        *(u32x4*)(Vs + row * PV + ch * 16) = b[c];
      }
    }
  };

  // prefetch = true: this wave's share of the NEXT tile's global loads is issued after the QK^T MFMAs (the staging
  // registers are then free while the K fragments are read, and still have the softmax and the PV product to land)
  // DMA: per-lane LDS byte offsets of the K fragments (row l31, chunk (hi + 2 ks) ^ sK) and of the V fragments
  // (row 4 hi + trow, chunk ((i >> 2) * 4 + g16 * 2 + (tcol >> 3)) ^ sV): the swizzle terms are lane constants
  int koff[D / 16], vboff[D / 32];
  if constexpr (DMA) {
    const int ksw = (D == 64) ? ((l31 >> 1) & 7) : (l31 & 15);
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) koff[ks] = l31 * PK + (((hi + 2 * ks) ^ ksw) << 4);
    const int li = lane & 15, g16 = (lane >> 4) & 1, trow = li >> 2, tcol = 4 * (li & 3);
    const int svl = (D == 64) ? (((trow >> 1) & 1) * 4) : (trow * 4);
#pragma unroll
    for (int g = 0; g < D / 32; ++g)
      vboff[g] = (4 * hi + trow) * PV + (((g * 4 + g16 * 2 + (tcol >> 3)) ^ svl) << 4) + (tcol & 7) * 2;
  }
  auto tile_math = [&](int kt, bool prefetch, auto bufc) {
    constexpr int BOFF = decltype(bufc)::value * TB;   // byte offset of the LDS buffer this tile sits in (DMA: 0 / TB)
    const int k0 = kt * KB;
    // wave-uniform: tile entirely above this wave's diagonal, or a wave without a query (ragged last q-block): it only
    // helps staging K/V and keeps the barriers balanced
    if ((CAUSAL && k0 > q0 + 31) || q0 >= len) {
      if (prefetch) load_tile(kt + 1);
      return;
    }

    // ---- S^T = K Q^T : two 32-key sub-tiles.  D = 128: the K fragments are read three MFMAs ahead of their use (the compiler's
    // own order hoists all 16 reads, 64 registers); D = 64 keeps the compiler's order - the ring costs it 19 spilled
    // registers under the 128-register cap that four waves per SIMD need, and measures slower
    f32x16 sacc[2];
    if constexpr (D == 128) {
      constexpr int N_QK = 2 * (D / 16), AHEAD = 3, RING = 4;
      const char* kp = Ks + l31 * PK + hi * 16;
      bf16x8 kr[RING];
      auto read_k = [&](int i, bf16x8& f) {   // i = (D / 16) * t2 + ks
        if constexpr (DMA) f = *(const bf16x8*)(Ks + BOFF + (i / (D / 16)) * 32 * PK + koff[i % (D / 16)]);
        else f = *(const bf16x8*)(kp + (i / (D / 16)) * 32 * PK + (i % (D / 16)) * 32);
      };
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) read_k(i, kr[i % RING]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N_QK; ++i) {
        if (i + AHEAD < N_QK) read_k(i + AHEAD, kr[(i + AHEAD) % RING]);
        const int t2 = i / (D / 16), ks = i % (D / 16);
        if (ks == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[t2][r] = 0.f;
        }
        if constexpr (VAR == 3) {
          asm volatile("" ::"v"(kr[i % RING]));
          sacc[t2][ks] += (float)lane * 1e-3f;
        } else {
          sacc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[i % RING], qf[ks], sacc[t2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t2][r] = 0.f;
        const char* kp = Ks + (t2 * 32 + l31) * PK + hi * 16;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
          const bf16x8 kf = DMA ? *(const bf16x8*)(Ks + BOFF + t2 * 32 * PK + koff[ks]) : *(const bf16x8*)(kp + ks * 32);
          if constexpr (VAR == 3) {
            asm volatile("" ::"v"(kf));
            sacc[t2][ks] += (float)lane * 1e-3f;
          } else {
            sacc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[t2], 0, 0, 0);
          }
        }
      }
    }

    if (prefetch) load_tile(kt + 1);

    // ---- masks only where a tile straddles the sequence end or the causal diagonal (wave-uniform test)
    const bool need_mask = (k0 + KB > len) || (CAUSAL && (k0 + KB - 1 > q0));
    if (need_mask) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= len || (CAUSAL && key > qi)) sacc[t2][r] = -INFINITY;
        }
    }
    bf16x8 pf[2][2];
    if constexpr (VAR == 2) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        unsigned pw[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) pw[r >> 1] = pack_pair(f32x2{sacc[t2][r], sacc[t2][r + 1]});
        pf[t2][0] = __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
        pf[t2][1] = __builtin_bit_cast(bf16x8, u32x4{pw[4], pw[5], pw[6], pw[7]});
      }
      l_run = 1.f;
    } else {
    // ---- online softmax.  bf16 rounding is monotone, so the row max is taken on the raw accumulators and rounded once.
    float mx = sacc[0][0];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[t2][r]);
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, round_score<RM>(mx, p.scale));
    if (__any(m_new > m_run)) {            // exact: rescale only when some query's running max moved
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      m_run = m_new;
    }
    // exponentials, two keys per step so that conversions and the affine map use the packed forms
    // (v_cvt_pk_bf16_f32, v_pk_fma_f32, v_pk_add_f32): the softmax is VALU-bound, not MFMA-bound
    const f32x2 c2 = {c_exp, c_exp};
    const f32x2 nmb2 = {-m_run * c_exp, -m_run * c_exp};
    f32x2 psum2 = {0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      unsigned pw[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a2 = {sacc[t2][r], sacc[t2][r + 1]};
        f32x2 sr = round_pair(a2);
        if constexpr (RM == RM_MUL) sr = round_pair(a2 * f32x2{p.scale, p.scale});
        if constexpr (RM == RM_DIV) sr = round_pair(sr * f32x2{p.scale, p.scale});
        const f32x2 e2 = sr * c2 + nmb2;
        const f32x2 pv = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
        psum2 += pv;
        pw[r >> 1] = pack_pair(pv);
      }
      pf[t2][0] = __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
      pf[t2][1] = __builtin_bit_cast(bf16x8, u32x4{pw[4], pw[5], pw[6], pw[7]});
    }
    l_run += xhalf_sum(psum2[0] + psum2[1]);
    }

    // ---- O^T += V^T P^T : A operand = V^T via transposed LDS reads.
    // element j of the fragment <-> key 32*t2 + 16*s2 + 8*(j>>2) + 4*hi + (j&3)
    // The reads run AHEAD of the MFMA that consumes them (a ring of fragments in the registers the score tile just left):
    // left to itself the compiler issues {2 reads, wait, MFMA} eight times and every MFMA eats a full LDS latency.
    {
      const int g16 = (lane >> 4) & 1;     // which 16-column block of the 32-wide d tile
      const int li = lane & 15;
      const int trow = li >> 2, tcol = 4 * (li & 3);
      const char* vbase = Vs + (4 * hi + trow) * PV + (g16 * 16 + tcol) * 2;
      constexpr int N_PV = 4 * (D / 32), AHEAD = 3, RING = 4;
      bf16x8 fr[RING];
      auto read_v = [&](int i, bf16x8& f) {   // i = 4 * dt + 2 * t2 + s2
        const char* vp = DMA ? Vs + BOFF + vboff[i >> 2] + (i & 3) * 16 * PV : vbase + (i & 3) * 16 * PV + (i >> 2) * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
        const s16x4 hi4 =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * PV));
        const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        f = __builtin_bit_cast(bf16x8, v8);
      };
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) read_v(i, fr[i % RING]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N_PV; ++i) {
        if (i + AHEAD < N_PV) read_v(i + AHEAD, fr[(i + AHEAD) % RING]);
        if constexpr (VAR == 3) {
          asm volatile("" ::"v"(fr[i % RING]), "v"(pf[(i >> 1) & 1][i & 1]));
          oacc[i >> 2][i & 3] += l_run;
        } else {
          oacc[i >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % RING], pf[(i >> 1) & 1][i & 1], oacc[i >> 2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  using Buf0 = std::integral_constant<int, 0>;
  using Buf1 = std::integral_constant<int, 1>;
  if constexpr (DMA) {
    // this wave's LDS-DMA share of a tile: slot = 16-byte LDS slot of the unpadded tile image, NI slots per operand
    constexpr int CH = D / 8, NI = KB * CH / 256;
    auto src_off = [&](int slot, int row0, long ld, bool is_k) -> unsigned {
      const int row = slot / CH, cl = slot % CH;
      const int sw = is_k ? ((D == 64) ? ((row >> 1) & 7) : (row & 15)) : ((D == 64) ? (((row >> 1) & 1) * 4) : ((row & 3) * 4));
      int gr = row0 + row;
      gr = gr < len ? gr : len - 1;                      // rows past the sequence end: any valid row (masked later)
      return (unsigned)(gr * (int)ld + ((cl ^ sw) * 8)) * 2u;
    };
    unsigned kso[NI], vso[NI];                            // whole tiles: offsets relative to the tile's first row
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int slot = (j * 4 + wave) * 64 + lane;
      const int row = slot / CH, cl = slot % CH;
      const int sk = (D == 64) ? ((row >> 1) & 7) : (row & 15);
      const int sv = (D == 64) ? (((row >> 1) & 1) * 4) : ((row & 3) * 4);
      kso[j] = (unsigned)(row * (int)p.ldk + ((cl ^ sk) * 8)) * 2u;
      vso[j] = (unsigned)(row * (int)p.ldv + ((cl ^ sv) * 8)) * 2u;
    }
    auto issue = [&](int kt, int boff, int w_lo, int w_hi) {   // DMA shares of waves w_lo .. w_hi - 1 (the solo wave issues all four)
      char* kdst = smem + boff;
      char* vdst = smem + boff + KBYTES;
      const bool whole = kt * KB + KB <= len;
      for (int w = w_lo; w < w_hi; ++w) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          unsigned ko, vo;
          if (whole && w == wave) {
            ko = kso[j] + (unsigned)(kt * KB) * (unsigned)p.ldk * 2u;
            vo = vso[j] + (unsigned)(kt * KB) * (unsigned)p.ldv * 2u;
          } else {
            const int slot = (j * 4 + w) * 64 + lane;
            ko = src_off(slot, kt * KB, p.ldk, true);
            vo = src_off(slot, kt * KB, p.ldv, false);
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k_seq + ko),
                                           (__attribute__((address_space(3))) void*)(kdst + (j * 4 + w) * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(v_seq + vo),
                                           (__attribute__((address_space(3))) void*)(vdst + (j * 4 + w) * 1024), 16, 0, 0);
        }
      }
    };
    if (solo) {
      // one wave: no barriers (its own LDS operations are in order), the next tile's DMA - all four shares - in flight
      // under the current tile's math
      issue(0, 0, 0, 4);
      for (int kt = 0; kt < n_tiles; kt += 2) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        if (kt + 1 < n_tiles) issue(kt + 1, TB, 0, 4);
        tile_math(kt, false, Buf0{});
        if (kt + 1 < n_tiles) {
          __builtin_amdgcn_s_waitcnt(0x0F70);
          if (kt + 2 < n_tiles) issue(kt + 2, 0, 0, 4);
          tile_math(kt + 1, false, Buf1{});
        }
      }
    } else {
      issue(0, 0, wave, wave + 1);
      for (int kt = 0; kt < n_tiles; kt += 2) {
        // tile kt is in buffer 0: every wave's share has landed after the wait + barrier, and every wave has finished tile
        // kt - 1 (buffer 1), which the next DMA overwrites
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (kt + 1 < n_tiles) issue(kt + 1, TB, wave, wave + 1);
        tile_math(kt, false, Buf0{});
        if (kt + 1 < n_tiles) {
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __syncthreads();
          if (kt + 2 < n_tiles) issue(kt + 2, 0, wave, wave + 1);
          tile_math(kt + 1, false, Buf1{});
        }
      }
    }
  } else if (solo) {
    for (int kt = 0; kt < n_tiles; ++kt) {
      stage_solo(kt);
      tile_math(kt, false, Buf0{});
    }
  } else {
    load_tile(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
      if (VAR != 1 || kt == 0) {
        __syncthreads();  // previous tile fully consumed
        store_tile();
        __syncthreads();
      }
      tile_math(kt, VAR != 1 && kt + 1 < n_tiles, Buf0{});
    }
  }

  // ---- epilogue: O[query][d] = O^T / l ; lane = query, register r <-> d = 32*dt + (r&3) + 8*(r>>2) + 4*hi
  if (qi < len) {
    const float inv = 1.0f / l_run;
    u16* op = p.O + (long)(s0 + qi) * p.ldo + (long)head * p.ohs;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 v = {pack2bf(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv),
                   pack2bf(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv)};
        *(u32x2*)(op + dt * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

// Schedules that were built, verified against the same tests and removed because they were not faster (DESIGN.md,
// "Attention: what was measured"): two staggered wave groups per workgroup (one group issues MFMAs while its SIMD partner
// runs the softmax, one s_barrier per segment) and a one-wave software pipeline (PV of tile t-1 and QK of tile t+1
// interleaved instruction by instruction with the softmax of tile t).  On this part the softmax's vector work and the
// MFMAs of a SIMD take close to the SUM of their times in every arrangement tried (tools/micro/coissue*.hip), so the
// levers that paid were the ones that remove work or traffic: XCD-aware placement, leaner staging addresses, scalar
// control flow, heaviest-first causal order, the one-wave ragged block.  Round 2: two K/V buffers in LDS with ONE barrier per
// key tile instead of two (the lever that gave the GEMM main loop +2-7 %) is bit-identical and measures 0 % at D = 128
// (same occupancy) and -4 % at D = 64 (three workgroups per CU instead of four): with several workgroups per CU the barrier
// waits are already covered by the other workgroups' waves.  (With LDS-DMA staging instead of register staging the two-buffer
// form does pay: see DMA below.)  A half-tile path for tiles whose second 32 keys are all masked (the 17th tile of the vision
// tower's 1025 keys; the diagonal tile of every other wave under the causal mask) is bit-identical and measured -1 % at
// D = 64 and -3 % at D = 128: the second copy of the tile code costs registers (2 spills / +40) and instruction cache.


// =====================================================================================================================
// attn2_kernel (round 3): 64 queries per wave as two 32-query sub-blocks A / B, and an IN-WAVE software pipeline.
//
// Counters on the round-2 kernel (profiles/r02_a_attn_pmc_counters.txt) say what it loses: every wave runs {8 QK^T MFMAs |
// ~150 vector instructions of softmax | 8 PV MFMAs} strictly in turn, the matrix pipe is busy 29 % of the time and a vector
// instruction co-executes in only 22 % of those cycles.  The calibration streams of tools/micro/issue_model say how to get
// the overlap: NOT from other waves (phased streams of 2-4 waves per SIMD overlap 0.4-0.6) but from INDEPENDENT vector work
// that follows each MFMA in the SAME wave's stream (0.72 at any occupancy).  One wave therefore owns two query sub-blocks and
// walks a key tile as four units u = (A,h0) (B,h0) (A,h1) (B,h1) (h = 32-key half of the 64-key tile), software-pipelined:
//
//     slot s :  MFMAs  PV(s-2), QK(s)     ||     vector work  softmax(s-1)            s = 0 .. 5
//
// so every MFMA is followed by a slice of the softmax of the PREVIOUS unit (independent registers).  The slices are pinned
// between the MFMAs with sched_barrier (left alone the compiler issues all MFMAs first).
//
// The softmax is "optimistic": the running offset M of a query is an INTEGER in exp2 units (M = ceil of the rounded score in
// log2 units at the time it was last set, + HEADROOM = 64 since round 5: see HEADROOM below) and a unit does NOT compute its row
// max: p = exp2(s c - M) directly, and only if a lane's partial row sum comes out above 2^20 (or NaN: the first tile starts from
// M = -inf) the unit is redone with M <- max(M, ceil(unit max) + HEADROOM).  Because M is an integer every rescale factor is an exact power of two, and bf16 rounding of P
// commutes with it: O / l do not depend on WHEN a query's offset was raised (no rounding-level difference between the lazy and
// the eager schedule), only on the offsets being integers.  This removes the max (16 v_max3 + exchange per unit), the exp of
// the rescale factor and, almost always, the rescale of the O accumulators from the per-tile vector work: 4.7 -> 3.7 vector
// instructions per score at D = 64.  Row sums stay lane-partial until the epilogue (one cross-half exchange per query block
// instead of one per tile).
//
// Sequences whose length is 1 (mod 64) - the vision tower's 1 + 32^2 tokens - would need a 17th key tile and a 9th query block
// for ONE token.  For them (non-causal only) key 0 becomes the INITIAL STATE of the online softmax (M = ceil(s(q, k_0) c) + HEADROOM,
// l = p_0, O = bf16(p_0) v_0: 32 FMAs per lane, once) and the key tiles start at key 1; the query blocks start at query 1 and
// query 0 is run by one wave of an extra block.
// =====================================================================================================================
namespace v2 {

template <class Fn, int... I>
MJV_DEV void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
MJV_DEV void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr float BIGSUM = 1048576.f;   // 2^20: a lane's partial row sum above this redoes the unit with a raised offset
// Whenever an offset is set it is set HEADROOM binades ABOVE the largest score seen (round 5).  The probabilities are powers of
// two smaller for it - nothing else changes: bf16 / fp32 keep their relative precision anywhere in their exponent range, the
// normalisation divides the factor out - but the next raise now needs a score 2^(20 + 64) above the maximum that set the
// offset instead of 2^20.  With the offset AT the maximum, trained-like logits (sigma 10: 14 binades per sigma, later maxima
// 20 - 40 binades above the first unit's) sent 12 % of the units through the raise path and cost the kernels 6 - 9 %
// (tools/stress_stats.py, profiles/r05_e_stress_stats.txt).  A score more than 126 - 64 = 62 binades BELOW the maximum now
// underflows to 0: a weight below 2^-62 of the row's largest, 38 binades under what an fp32 sum resolves.
constexpr float HEADROOM = 64.f;

struct Pos { int slot, kind, unit, f; };   // kind 0 = QK^T, 1 = PV
// MFMA stream of one key tile: slot s holds the F MFMAs of PV(s - 2) (if that unit exists), then the F of QK(s)
template <int F, int U>
constexpr Pos stream_pos(int i) {
  int s = 0;
  while (true) {
    const int npv = (s >= 2 && s - 2 < U) ? F : 0;
    const int nqk = (s < U) ? F : 0;
    if (i < npv) return Pos{s, 1, s - 2, i};
    i -= npv;
    if (i < nqk) return Pos{s, 0, s, i};
    i -= nqk;
    ++s;
  }
}
template <int F, int U>
constexpr int slot_first(int s) {   // stream index of the first MFMA of slot s
  int i = 0;
  for (int t = 0; t < s; ++t) i += ((t >= 2 && t - 2 < U) ? F : 0) + ((t < U) ? F : 0);
  return i;
}
template <int F, int U>
constexpr int slot_size(int s) { return ((s >= 2 && s - 2 < U) ? F : 0) + ((s < U) ? F : 0); }

// Diagnostic build only (-DMJV_ATTN_STAMPS, tools/attn_stamps.py): s_memtime stamps at the segment boundaries of the tile
// loop, summed per wave in scalar registers and written to a buffer of their own after the loop; never in the product build.
#ifdef MJV_ATTN_STAMPS
__device__ unsigned long long* g_stamp_out = nullptr;
#define MJV_STAMP(i)                                                                       \
  do {                                                                                     \
    unsigned long long t_;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    st_acc[i] += t_ - st_prev;                                                             \
    st_prev = t_;                                                                          \
  } while (0)
#else
#define MJV_STAMP(i) do { } while (0)
#endif

// Chunk swizzles of the unpadded LDS image (16-byte chunk c of row r sits at chunk c ^ swz(r)), per head size.  D = 64 / 128: as
// attn_kernel<.., DMA = true>.  D = 96 (ABI 7, Phi-3-mini's heads): rows are 192 bytes = 12 chunks, so the XOR may only touch
// the two low bits of the chunk index: K swz = (r >> 2) & 3 - the 16 lanes of a ds_read_b128 group (rows {0-3, 12-15, 20-27} /
// {4-11, 16-19, 28-31}) differ in (r mod 4, (r >> 2) & 3), and slot mod 16 = 4 (3 r mod 4) + (c & 12) + ((c & 3) ^ swz): all
// distinct; V needs none - four consecutive rows of 48 banks start at banks 0, 48, 32, 16, the 16-bank windows a
// ds_read_b64_tr_b16 half-wave reads from them tile the 64 banks.
template <int D>
MJV_DEV constexpr int swz_k(int row) { return D == 64 ? ((row >> 1) & 7) : (D == 128 ? (row & 15) : ((row >> 2) & 3)); }
template <int D>
MJV_DEV constexpr int swz_v(int row) { return D == 64 ? (((row >> 1) & 1) * 4) : (D == 128 ? ((row & 3) * 4) : 0); }
constexpr int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// NSUB = 32-query sub-blocks per wave: 2 at D = 64; 1 at D = 128 / 96, where two would need ~310 registers (the pipeline then
// runs over the two key halves of the one sub-block: QK(h0) | QK(h1) || softmax(h0) | PV(h0) || softmax(h1) | PV(h1))
template <int D, bool CAUSAL, int RM, int NW, int NSUB>
__global__ __launch_bounds__(64 * NW, 2) void attn2_kernel(AttnArgs p) {
  constexpr int PK = D * 2, PV = D * 2;           // unpadded rows, swizzled chunks (as the DMA form of attn_kernel)
  // TB = distance of the two tile buffers: a power of two (the buffer toggle XORs it into the fragment offsets); at D = 96 the
  // 24 KiB of a tile sit in a 32 KiB slot
  constexpr int KBYTES = KB * PK, VBYTES = KB * PV, TB = pow2_ceil(KBYTES + VBYTES);
  constexpr int QW = 32 * NSUB;                   // queries per wave
  constexpr int QBW = QW * NW;                    // queries per workgroup
  constexpr int F = D / 16;                       // MFMAs per unit and product
  static_assert((TB & (TB - 1)) == 0 && TB >= KBYTES + VBYTES, "the buffer toggle XORs TB into the fragment offsets");
  __shared__ __attribute__((aligned(16))) char smem[2 * TB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const BlockId bid = decode_block(p);
  if (!bid.valid) return;
  const int seq = bid.seq, head = bid.head;
  const int s0 = p.cu[seq];
  const int len = p.cu[seq + 1] - s0;             // the sequence's own key rows
  const int kvh = head / p.kv_group;
  // causal launches (ABI 6): P shared prefix keys in front of the own ones; the queries are the last lenq positions of the
  // P + len keys, rows sq0 ... of Q / O.  Local query i sits at key position qsh + i.  (Non-causal: P = 0, queries = key rows.)
  const int P = CAUSAL ? p.prefix_len : 0;
  const int sq0 = (CAUSAL && p.cu_q) ? p.cu_q[seq] : s0;
  const int lenq = (CAUSAL && p.cu_q) ? p.cu_q[seq + 1] - sq0 : len;
  const int qsh = CAUSAL ? P + len - lenq : 0;
  // first-key-as-initial-state ("peel"): workgroup-uniform
  const bool peel = !CAUSAL && len > 1 && ((len - 1) % KB == 0);
  const int klen = peel ? len - 1 : P + len;      // keys that go through the tiles (rows s0 + peel ...; prefix rows first)
  const int nq_main = peel ? len - 1 : lenq;      // queries of the ordinary blocks (queries peel ...)
  const int nb_main = (nq_main + QBW - 1) / QBW;
  const int qb = CAUSAL ? nb_main - 1 - bid.qb : bid.qb;   // causal: heaviest blocks first
  if (CAUSAL ? (qb < 0) : (qb > nb_main || (qb == nb_main && !peel))) return;
  const bool cls_block = peel && qb == nb_main;   // the block of query 0
  // (all of that block's waves stay and share the staging: run by wave 0 alone - no barriers, the others gone - it measured
  // 3 % slower: the lone wave issues every DMA instruction of the tile itself)

  // this wave's queries: sub-block A = qw0 + l31, B = qw0 + 32 + l31 (sequence-relative indices)
  const int qw0 = cls_block ? 0 : (peel ? 1 : 0) + qb * QBW + wave * QW;
  // (query-0 block: waves 0 and 1 both hold the query and each takes one 32-key half of every tile - the state of wave 1 is
  // merged into wave 0's after the loop; a single wave walking all 1024 keys held the block's slot twice as long)
  const int nq_wave = cls_block ? (wave < 2 ? 1 : 0) : max(0, min(QW, lenq - qw0));   // valid queries of this wave
  const bool hasA = nq_wave > 0, hasB = NSUB == 2 && nq_wave > 32;
  int qi[NSUB];
  qi[0] = cls_block ? 0 : qw0 + l31;
  if constexpr (NSUB == 2) qi[1] = qw0 + 32 + l31;

  constexpr float LOG2E = 1.4426950408889634f;
  const float c_exp = (RM == RM_POW2 || RM == RM_FLASH) ? p.scale * LOG2E : LOG2E;   // exp2 argument = (rounded score) * c_exp - M

  // Q fragments (B operand): lane holds Q[query l31][d = 16 ks + 8 hi + j]
  bf16x8 qf[NSUB][F];
#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb) {
    const int qr = sq0 + (qi[sb] < lenq ? qi[sb] : lenq - 1);
    const u16* qp = p.Q + (long)qr * p.ldq + (long)head * p.qhs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < F; ++ks) qf[sb][ks] = *(const bf16x8*)(qp + ks * 16);
  }

  f32x16 oacc[NSUB][D / 32];
#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[sb][i][r] = 0.f;
  float Mq[NSUB];   // integer offset in exp2 units (same value in the two half-lanes of a query)
  float lsum[NSUB];                       // LANE-PARTIAL row sums (this lane's keys only); halves are added in the epilogue
#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb) { Mq[sb] = -INFINITY; lsum[sb] = 0.f; }
#ifdef MJV_ATTN_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
#endif

  const u16* Kg = p.K + (long)kvh * p.khs;
  const u16* Vg = p.V + (long)kvh * p.vhs;
  // (own rows: the base is shifted down by the P prefix rows, so a tile's row index kt * KB + row addresses prefix and own
  // tiles alike - own tiles start at kt = P / KB and never reach below row s0)
  const char* const k_seq = (const char*)(Kg + (long)(s0 + (peel ? 1 : 0) - P) * p.ldk);   // key row 0 of the tiles
  const char* const v_seq = (const char*)(Vg + (long)(s0 + (peel ? 1 : 0) - P) * p.ldv);
  const char* const kp_seq = P ? (const char*)(p.Kp + (long)kvh * p.khs) : k_seq;           // the shared prefix rows
  const char* const vp_seq = P ? (const char*)(p.Vp + (long)kvh * p.vhs) : v_seq;

  if (peel && hasA && !(cls_block && wave == 1)) {
    // key 0 as the initial state.  s = q . k_0 in fp32 (this lane's 8-element groups, then the other half-lane's), rounded
    // like every score; M = ceil(s c) + HEADROOM; p_0 = exp2(s c - M) in (1/2, 1] 2^-64; l = bf16(p_0) (counted in the hi = 0 lane only: row sums
    // are lane-partial); O = bf16(p_0) * v_0 - what the MFMA would have accumulated for this key.
    const u16* k0p = Kg + (long)s0 * p.ldk + 8 * hi;
    const u16* v0p = Vg + (long)s0 * p.ldv;
    u32x4 kraw[F];
#pragma unroll
    for (int ks = 0; ks < F; ++ks) kraw[ks] = *(const u32x4*)(k0p + ks * 16);
    u32x2 vraw[D / 32][4];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) vraw[dt][g] = *(const u32x2*)(v0p + dt * 32 + 8 * g + 4 * hi);
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb) {
      float dot = 0.f;
#pragma unroll
      for (int ks = 0; ks < F; ++ks) {
        float kf[8], qv[8];
        unpack8(kraw[ks], kf);
        unpack8(__builtin_bit_cast(u32x4, qf[sb][ks]), qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) dot = fmaf(qv[j], kf[j], dot);
      }
      dot = xhalf_sum(dot);
      const float arg = round_score<RM>(dot, p.scale) * c_exp;
      const float m0 = ceilf(arg) + HEADROOM;
      const float p0 = __builtin_amdgcn_exp2f(arg - m0);
      const float pb = rbf(p0);
      Mq[sb] = m0;
      lsum[sb] = hi == 0 ? pb : 0.f;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          oacc[sb][dt][4 * g + 0] = pb * __uint_as_float(vraw[dt][g][0] << 16);
          oacc[sb][dt][4 * g + 1] = pb * __uint_as_float(vraw[dt][g][0] & 0xffff0000u);
          oacc[sb][dt][4 * g + 2] = pb * __uint_as_float(vraw[dt][g][1] << 16);
          oacc[sb][dt][4 * g + 3] = pb * __uint_as_float(vraw[dt][g][1] & 0xffff0000u);
        }
    }
  }

  const int q_last_w = qsh + qw0 + nq_wave - 1;                  // key position of the last valid query of this wave
  const int kv_end = CAUSAL ? min(klen, qsh + qb * QBW + QBW) : klen;  // keys this workgroup needs
  const int n_tiles = (kv_end + KB - 1) / KB;

  // ---- LDS-DMA staging (same image and swizzles as attn_kernel<.., DMA = true>), NW waves share a tile's 2 x CH instructions
  constexpr int CH = D / 8, NI = CH / NW;
  static_assert(CH % NW == 0, "DMA instructions per operand must divide over the waves");
  unsigned kso[NI], vso[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int slot = (j * NW + wave) * 64 + lane;
    const int row = slot / CH, cl = slot % CH;
    const int sk = swz_k<D>(row);
    const int sv = swz_v<D>(row);
    kso[j] = (unsigned)(row * (int)p.ldk + ((cl ^ sk) * 8)) * 2u;
    vso[j] = (unsigned)(row * (int)p.ldv + ((cl ^ sv) * 8)) * 2u;
  }
  // LDS-DMA as inline assembly: issued through the builtin, the compiler's wait-count pass treats the transposed LDS reads of
  // the V fragments as possibly aliasing the DMA in flight and puts an s_waitcnt vmcnt(0) in front of the first of them - the
  // whole flight time of the NEXT tile's DMA (issued a few hundred cycles earlier) exposed once per tile.  The DMA of tile
  // kt + 1 targets the buffer nobody reads during tile kt; its completion is waited for (asm, vmcnt(0)) at the top of the
  // next iteration, before the barrier.  M0 (LDS destination, wave-uniform) is written in the statement that uses it.
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  auto dma16 = [&](const char* base, unsigned voff, unsigned lds_dst) __attribute__((always_inline)) {
    // (M0 is not saved / restored: nothing else in this kernel uses it - every LDS-DMA is one of these statements - and the
    // restoring s_mov right behind the DMA cost more than the DMA's own issue)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
  };
  // piece idx = 2 j + (0: K, 1: V) of this wave's share of tile kt; WHOLE: all 64 key rows exist (no row clamp)
  auto dma_piece = [&](int kt, int boff, int idx, auto wholec) __attribute__((always_inline)) {
    const int j = idx >> 1;
    const bool is_v = idx & 1;
    unsigned off;
    if constexpr (decltype(wholec)::value) {
      off = is_v ? vso[j] + (unsigned)(kt * KB) * (unsigned)p.ldv * 2u : kso[j] + (unsigned)(kt * KB) * (unsigned)p.ldk * 2u;
    } else {
      const int slot = (j * NW + wave) * 64 + lane;
      const int row = slot / CH, cl = slot % CH;
      const int sw = is_v ? swz_v<D>(row) : swz_k<D>(row);
      int gr = kt * KB + row;
      gr = gr < klen ? gr : klen - 1;                   // rows past the end: any valid row (masked later)
      off = (unsigned)(gr * (int)(is_v ? p.ldv : p.ldk) + ((cl ^ sw) * 8)) * 2u;
    }
    const bool pre = CAUSAL && kt * KB < P;             // a tile of the shared prefix (whole by construction); wave-uniform
    dma16(is_v ? (pre ? vp_seq : v_seq) : (pre ? kp_seq : k_seq), off, lds0 + (unsigned)(boff + (is_v ? KBYTES : 0) + (j * NW + wave) * 1024));
  };
  auto issue = [&](int kt, int boff) __attribute__((always_inline)) {
    if (kt * KB + KB <= klen) {
#pragma unroll
      for (int idx = 0; idx < 2 * NI; ++idx) dma_piece(kt, boff, idx, std::true_type{});
    } else {
#pragma unroll
      for (int idx = 0; idx < 2 * NI; ++idx) dma_piece(kt, boff, idx, std::false_type{});
    }
  };

  // per-lane LDS byte offsets of the fragments (the swizzle terms are lane constants)
  int koff[F], vboff[D / 32];
  {
    const int ksw = swz_k<D>(l31);
#pragma unroll
    for (int ks = 0; ks < F; ++ks) koff[ks] = l31 * PK + (((hi + 2 * ks) ^ ksw) << 4);
    const int li = lane & 15, g16 = (lane >> 4) & 1, trow = li >> 2, tcol = 4 * (li & 3);
    const int svl = swz_v<D>(trow);     // (rows 4 hi + trow + 8 k: the V swizzles depend on the row's two low bits only)
#pragma unroll
    for (int g = 0; g < D / 32; ++g)
      vboff[g] = (4 * hi + trow) * PV + (((g * 4 + g16 * 2 + (tcol >> 3)) ^ svl) << 4) + (tcol & 7) * 2;
  }

  const f32x2 c2 = {c_exp, c_exp};
  const f32x2 scale2 = {p.scale, p.scale};

  // one pair of scores of a unit -> two exponentials (in place of nothing: the scores stay live for a possible redo)
  // The row sum takes the ROUNDED probabilities (v_dot2c_f32_bf16 of the packed pair with (1, 1)): numerator and denominator
  // of O = sum p~ v / sum p~ then carry the same rounding, so the weights sum to one exactly - with an integer offset the
  // largest probability of a row is no longer exactly 1, and a sum of the unrounded values left its rounding error in O
  // (rows dominated by one key: short causal rows, peaked heads).  One VOP2 per pair instead of a v_pk_add_f32.
  auto do_pair = [&](const f32x16& S, int r, float nmb, float& psum, unsigned& pw) __attribute__((always_inline)) {
    const f32x2 a2 = {S[r], S[r + 1]};
    f32x2 sr;
    if constexpr (RM == RM_FLASH) sr = a2;   // 2 fma + 2 exp + pack + dot2c per pair: no rounding, no separate scaling
    else if constexpr (RM == RM_MUL) sr = round_pair(a2 * scale2);
    else {
      sr = round_pair(a2);
      if constexpr (RM == RM_DIV) sr = round_pair(sr * scale2);
    }
    const f32x2 e2 = sr * c2 + f32x2{nmb, nmb};
    const f32x2 pv = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
    pw = pack_pair(pv);
    psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, pw), __builtin_bit_cast(bf16x2v, 0x3f803f80u), psum, false);
  };
  // rare path: raise the offset of sub-block sb to cover unit scores S, rescale its state, recompute the unit's P
  auto redo = [&](const f32x16& S, auto sbc, float& psum, unsigned (&pw)[8]) __attribute__((always_inline)) {
    constexpr int sb = decltype(sbc)::value;
#ifdef MJV_ATTN_STAMPS
    st_acc[10] += 1;   // (diagnostic build: units that took the offset-raise path, per wave - tools/stress_stats.py)
#endif
    float mx = S[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
    mx = xhalf_max(mx);
    const float arg = round_score<RM>(mx, p.scale) * c_exp;
    float mn = fmaxf(Mq[sb], ceilf(arg) + HEADROOM);
    if (!(mn > -INFINITY)) mn = 0.f;                       // nothing but masked keys so far
    const float alpha = __builtin_amdgcn_exp2f(Mq[sb] - mn);   // exact power of two (0 from the initial -inf)
    lsum[sb] *= alpha;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[sb][i][r] *= alpha;
    Mq[sb] = mn;
    psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) do_pair(S, r, -mn, psum, pw[r >> 1]);
  };

  const char* const Ksb = smem;
  const char* const Vsb = smem + KBYTES;
  auto read_k = [&](int h, int ks, bf16x8& f) __attribute__((always_inline)) { f = *(const bf16x8*)(Ksb + h * 32 * PK + koff[ks]); };
  auto read_v = [&](int h, int f_, bf16x8& f) __attribute__((always_inline)) {         // f_ = 2 dt + s2  ->  keys 32 h + 16 s2 .., d tile dt
    const char* vp = Vsb + vboff[f_ >> 1] + (2 * h + (f_ & 1)) * 16 * PV;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * PV));
    const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    f = __builtin_bit_cast(bf16x8, v8);
  };
  auto pfrag = [&](const unsigned (&pw)[8], int s2) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, u32x4{pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]});
  };

    // ---- software pipeline over the units of a whole, unmasked tile: U = 4 (unit u: sub-block u & 1, key half u >> 1) for a
    // wave with both sub-blocks, U = 2 (sub-block A, key half u) for a wave with at most 32 queries (the block of query 0)
  auto pipeline = [&](auto uc_, int kt) __attribute__((always_inline)) {
      constexpr int U = decltype(uc_)::value;
      const bool more = (kt + 2) * KB <= klen && kt + 1 < n_tiles;   // the next tile is whole: its DMA is issued in here
      const int nboff = ((kt + 1) & 1) * TB;
      f32x16 S[U];
      unsigned Pw[U][8];
      float psum[U];
      constexpr int NS = 2 * U * F, AHEAD = 3, RING = 4;
      bf16x8 ring[RING];
      auto rd = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr Pos q = stream_pos<F, U>(i);
        constexpr int h = (U == 4) ? (q.unit >> 1) : q.unit;
        if constexpr (q.kind == 0) read_k(h, q.f, ring[i % RING]);
        else read_v(h, q.f, ring[i % RING]);
      };
      static_for<AHEAD>([&](auto ic) __attribute__((always_inline)) { rd(ic); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<NS>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr Pos q = stream_pos<F, U>(i);
        if constexpr (i + AHEAD < NS) rd(std::integral_constant<int, i + AHEAD>{});
        constexpr int sb = (U == 4) ? (q.unit & 1) : 0;
        if constexpr (q.kind == 0) {
          if constexpr (q.f == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[q.unit][r] = 0.f;
          }
          S[q.unit] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % RING], qf[sb][q.f], S[q.unit], 0, 0, 0);
        } else {
          oacc[sb][q.f >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % RING], pfrag(Pw[q.unit], q.f & 1),
                                                                     oacc[sb][q.f >> 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's DMA, one piece behind each of the first MFMAs (slot 0 has no vector work to put there)
        if constexpr (i < 2 * NI) {
          if (more) dma_piece(kt + 1, nboff, i, std::true_type{});
          __builtin_amdgcn_sched_barrier(0);
        }
        // the slice of softmax(slot - 1) that goes behind this MFMA
        constexpr int s = q.slot;
        if constexpr (s >= 1 && s <= U) {
          constexpr int us = s - 1;                                     // unit whose softmax runs in this slot
          constexpr int j = i - slot_first<F, U>(s), cnt = slot_size<F, U>(s);
          constexpr int pr0 = j * 8 / cnt, pr1 = (j + 1) * 8 / cnt;     // pairs [pr0, pr1) of the unit's 8
          if constexpr (j == 0) psum[us] = 0.f;
          constexpr int sbs = (U == 4) ? (us & 1) : 0;
          const float nmb = -Mq[sbs];
#pragma unroll
          for (int pr = pr0; pr < pr1; ++pr) do_pair(S[us], 2 * pr, nmb, psum[us], Pw[us][pr]);
          if constexpr (j == cnt - 1) {
            if (__any(!(psum[us] < BIGSUM))) redo(S[us], std::integral_constant<int, sbs>{}, psum[us], Pw[us]);
            lsum[sbs] += psum[us];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef MJV_ATTN_STAMPS
        if constexpr (i + 1 == slot_first<F, U>(q.slot) + slot_size<F, U>(q.slot)) MJV_STAMP(4 + q.slot);
#endif
      });
      };
  // ---- general path (ragged blocks, tiles on the diagonal or across the sequence end): unit after unit, with masks
  auto general = [&](int kt) __attribute__((always_inline)) {
    const int k0 = kt * KB;
    if (!hasA || (CAUSAL && k0 > q_last_w)) return;       // no query here / tile above this wave's diagonal: staging help only
    static_for<2 * NSUB>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      constexpr int sb = (NSUB == 2) ? (u & 1) : 0, h = (NSUB == 2) ? (u >> 1) : u;
      const int kb = k0 + 32 * h;
      const int q_last = qsh + qw0 + 32 * sb + min(31, nq_wave - 32 * sb - 1);   // key position of the sub-block's last valid query
      if ((sb == 1 && !hasB) || kb >= klen || (CAUSAL && kb > q_last) || (cls_block && h != wave)) return;   // wave-uniform
      // every fragment of the unit is requested up front (K for the scores, V for the second product): read where they
      // are used, each product paid an LDS round trip per MFMA - 1 775 cycles for one unit in the query-0 block
      bf16x8 kf[F], vf[F];
#pragma unroll
      for (int ks = 0; ks < F; ++ks) read_k(h, ks, kf[ks]);
      constexpr bool V_EARLY = D == 64;    // (at D = 128 sixteen fragments up front cost spilled registers: V follows the softmax)
      if constexpr (V_EARLY) {
#pragma unroll
        for (int f_ = 0; f_ < F; ++f_) read_v(h, f_, vf[f_]);
      }
      f32x16 S;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < F; ++ks) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[sb][ks], S, 0, 0, 0);
      const int qmin = qsh + qw0 + 32 * sb;
      if (kb + 32 > klen || (CAUSAL && kb + 31 > qmin)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= klen || (CAUSAL && key > qi[sb] + qsh)) S[r] = -INFINITY;
        }
      }
      unsigned pw[8];
      float ps = 0.f;
      const float nmb = -Mq[sb];
#pragma unroll
      for (int r = 0; r < 16; r += 2) do_pair(S, r, nmb, ps, pw[r >> 1]);
      if (__any(!(ps < BIGSUM))) redo(S, std::integral_constant<int, sb>{}, ps, pw);
      lsum[sb] += ps;
      if constexpr (!V_EARLY) {
#pragma unroll
        for (int f_ = 0; f_ < F; ++f_) read_v(h, f_, vf[f_]);
      }
#pragma unroll
      for (int f_ = 0; f_ < F; ++f_)
        oacc[sb][f_ >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[f_], pfrag(pw, f_ & 1), oacc[sb][f_ >> 1], 0, 0, 0);
    });
  };

  // every load the compiler counts (Q fragments, key / value row 0) is retired HERE, in a form its wait-count pass reads: the
  // loop's DMA is inline assembly, which that pass does not see, and a compiler-counted load still pending at the loop
  // header would get counted waits inside the loop (vmcnt(7) ... vmcnt(0) in front of the first MFMAs that read the Q
  // fragments) that in reality wait for the youngest operations in the queue - the DMA just issued
  __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef MJV_ATTN_STAMPS
  unsigned long long st_rt0;   // the 100 MHz counter beside the shader clock: clock held in the tile loop = d[13] / d[15] x 100 MHz
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev), "=s"(st_rt0)::"memory");
  const unsigned long long st_begin = st_prev;
#endif
  issue(0, 0);
  // One iteration of the tile loop; the loop is written three times, each with ONE kind of tile arithmetic in its body: a body
  // that chooses between the pipelines and the general path per tile makes the register allocator give the O accumulators a
  // different home on every path (64 v_mov per tile between them, and spills in the pipeline).  Whole, unmasked tiles come
  // first in every wave's key order (the ragged last tile and the causal diagonal are at the end), so the split is by position.
  auto step = [&](int kt, auto issues_itself, auto&& math) __attribute__((always_inline)) {
    // tile kt sits in buffer kt & 1: every wave's share has landed after the wait + barrier, and every wave is done with
    // tile kt - 1 (the other buffer), which the next DMA overwrites
    MJV_STAMP(0);                      // loop overhead (address toggles, branch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MJV_STAMP(1);                      // own DMA of this tile landed
    __syncthreads();
    MJV_STAMP(2);                      // barrier
    // (the pipelines issue the DMA of a WHOLE next tile themselves, behind their first MFMAs)
    if (kt + 1 < n_tiles && !(decltype(issues_itself)::value && (kt + 2) * KB <= klen)) issue(kt + 1, ((kt + 1) & 1) * TB);
    MJV_STAMP(3);                      // DMA issue
    math(kt);
    MJV_STAMP(11);                     // (general-path tiles; the pipelines stamp their slots 4 .. 9 themselves)
    // the fragment addresses follow the tile to the other buffer (TB is a power of two above every offset in a buffer)
#pragma unroll
    for (int ks = 0; ks < F; ++ks) koff[ks] ^= TB;
#pragma unroll
    for (int g = 0; g < D / 32; ++g) vboff[g] ^= TB;
  };
  // leading tiles that are whole (all 64 keys exist) and need no mask for any query of this wave
  const int n_whole = (!hasA || cls_block) ? 0 : min(n_tiles, CAUSAL ? max(0, (qsh + qw0 + 1) / KB) : klen / KB);
  int kt = 0;
  if (NSUB == 2 && hasB) {
    if constexpr (NSUB == 2)
      for (; kt < n_whole; ++kt) step(kt, std::true_type{}, [&](int t) __attribute__((always_inline)) { pipeline(std::integral_constant<int, 4>{}, t); });
  } else if (hasA) {
    for (; kt < n_whole; ++kt) step(kt, std::true_type{}, [&](int t) __attribute__((always_inline)) { pipeline(std::integral_constant<int, 2>{}, t); });
  }
  for (; kt < n_tiles; ++kt) step(kt, std::false_type{}, general);

#ifdef MJV_ATTN_STAMPS
  unsigned long long st_rt1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1)::"memory");
  if (g_stamp_out && lane == 0) {
    unsigned long long* d = g_stamp_out + ((long)blockIdx.x * NW + wave) * 16;
#pragma unroll
    for (int i = 0; i < 12; ++i) d[i] = st_acc[i];
    d[12] = (unsigned long long)n_tiles;
    d[13] = st_prev - st_begin;
    d[14] = (unsigned long long)((hasA ? 1 : 0) + ((hasB || (NSUB == 1 && hasA)) ? 1 : 0) + (cls_block ? 4 : 0));
    d[15] = st_rt1 - st_rt0;
  }
#endif
  if (cls_block) {   // workgroup-uniform: merge wave 1's state (the second key half of every tile) into wave 0's
    __syncthreads();                                 // every wave is out of the tile loop: the K / V buffers are free
    float* mb = (float*)smem;
    constexpr int REC = 2 + 16 * (D / 32);
    if (wave == 1 && l31 == 0) {
      mb[hi * REC] = Mq[0];
      mb[hi * REC + 1] = lsum[0];
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mb[hi * REC + 2 + dt * 16 + r] = oacc[0][dt][r];
    }
    __syncthreads();
    if (wave == 0) {
      const float m1 = mb[hi * REC], l1 = mb[hi * REC + 1];
      const float mn = fmaxf(Mq[0], m1);              // integer offsets: both factors are exact powers of two (or 0)
      const float a0 = __builtin_amdgcn_exp2f(Mq[0] - mn), a1 = (m1 > -INFINITY) ? __builtin_amdgcn_exp2f(m1 - mn) : 0.f;
      lsum[0] = lsum[0] * a0 + l1 * a1;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[0][dt][r] = oacc[0][dt][r] * a0 + mb[hi * REC + 2 + dt * 16 + r] * a1;
    }
  }
  // ---- epilogue: O[query][d] = O^T / l ; lane = query, register r <-> d = 32 dt + (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb) {
    const float l = xhalf_sum(lsum[sb]);
    const bool ok = cls_block ? (wave == 0 && sb == 0 && l31 == 0) : (qi[sb] < lenq);   // (every wave of the query-0 block reaches this)
    if (!ok) continue;
    const float inv = 1.0f / l;
    u16* op = p.O + (long)(sq0 + qi[sb]) * p.ldo + (long)head * p.ohs;
    // 16-byte stores (round 4; cdna_hip_programming.md T21): the two half-lanes of a query hold alternating 4-element chunks
    // of its row (d = 8 g + 4 hi ..); one v_permlane32_swap per word hands the hi = 0 lane both halves of the even chunks and
    // the hi = 1 lane both halves of the odd ones, so each stores 8 consecutive elements - half as many store instructions
    // (both half-lanes of a query take the same branch above: the exchange is between active lanes)
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const unsigned a0 = pack2bf(oacc[sb][dt][4 * g] * inv, oacc[sb][dt][4 * g + 1] * inv);
        const unsigned a1 = pack2bf(oacc[sb][dt][4 * g + 2] * inv, oacc[sb][dt][4 * g + 3] * inv);
        const unsigned b0 = pack2bf(oacc[sb][dt][4 * g + 4] * inv, oacc[sb][dt][4 * g + 5] * inv);
        const unsigned b1 = pack2bf(oacc[sb][dt][4 * g + 6] * inv, oacc[sb][dt][4 * g + 7] * inv);
        const auto w0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);   // [0]: own (hi = 0) / partner's b; [1]: partner's a / own b
        const auto w1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 v = {w0[0], w1[0], w0[1], w1[1]};
        *(u32x4*)(op + dt * 32 + 8 * (g + hi)) = v;
      }
  }
}

}  // namespace v2

#ifdef MJV_BENCH
int g_attn_bench = 0;   // bench library only (mjv_bench_attention_set): timing variants 1-3 of the round-2 kernel
#endif

// max_seqlen: the longest sequence in QUERY rows (= key rows unless cu_q is given)
template <int D, bool CAUSAL>
int launch(AttnArgs a, int n_seqs, int max_seqlen, int kernel, hipStream_t s) {
  int e;
  const bool pow2 = a.round_mode == 0 && frexpf(a.scale, &e) == 0.5f;
  a.n_seqs = n_seqs;
  a.n_qb = (max_seqlen + QB - 1) / QB;
  const int total = a.n_qb * a.n_heads * n_seqs;
  const dim3 grid(8 * ((total + 7) / 8));
#ifdef MJV_BENCH
  if constexpr ((D == 64 && !CAUSAL) || (D == 128 && CAUSAL)) {   // timing experiments on the two production shapes
    constexpr int RMX = (D == 64) ? RM_POW2 : RM_DIV;
    if (g_attn_bench == 1) { hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RMX, 1>), grid, dim3(256), 0, s, a); return mjv_check_launch("attention"); }
    if (g_attn_bench == 2) { hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RMX, 2>), grid, dim3(256), 0, s, a); return mjv_check_launch("attention"); }
    if (g_attn_bench == 3) { hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RMX, 3>), grid, dim3(256), 0, s, a); return mjv_check_launch("attention"); }
  }
#endif
  // round 3: attn2_kernel for every length (measured against the round-2 choice in one process, causal D = 128: 8 192 keys
  // 0.578 vs 0.656 ms, 28 810 keys - BASELINE configs[3] - 3.31 vs 3.76 ms = 1 027 vs 903 TFLOP/s); kernel 4 = the
  // register-staged round-1 kernel, 5 = the round-2 choice (its LDS-DMA form up to 4096 keys, register-staged beyond)
  const bool dma = max_seqlen <= 4096 && kernel != 4;
  if (kernel != 4 && kernel != 5) {
    constexpr int NSUB = (D == 64) ? 2 : 1;
    auto go = [&](auto nwc) {
      constexpr int NW = decltype(nwc)::value;
      constexpr int QBW = 32 * NSUB * NW;
      // non-causal launches may peel key / query 0 of a sequence (length = 1 mod 64): one more block for query 0
      const int nqb2 = CAUSAL ? (max_seqlen + QBW - 1) / QBW : std::max((max_seqlen + QBW - 1) / QBW, (max_seqlen - 1 + QBW - 1) / QBW + 1);
      a.n_qb = nqb2;
      const int total2 = nqb2 * a.n_heads * n_seqs;
      const dim3 grid2(8 * ((total2 + 7) / 8));
      if (a.round_mode == 2) hipLaunchKernelGGL((v2::attn2_kernel<D, CAUSAL, RM_FLASH, NW, NSUB>), grid2, dim3(64 * NW), 0, s, a);
      else if (a.round_mode == 1) hipLaunchKernelGGL((v2::attn2_kernel<D, CAUSAL, RM_DIV, NW, NSUB>), grid2, dim3(64 * NW), 0, s, a);
      else if (pow2) hipLaunchKernelGGL((v2::attn2_kernel<D, CAUSAL, RM_POW2, NW, NSUB>), grid2, dim3(64 * NW), 0, s, a);
      else hipLaunchKernelGGL((v2::attn2_kernel<D, CAUSAL, RM_MUL, NW, NSUB>), grid2, dim3(64 * NW), 0, s, a);
    };
    // Four waves per workgroup; two (half as many queries per block) only on request (desc.kernel 6).  Measured for the one
    // launch size where the smaller block could pay - ONE video per forward, 144 causal blocks of 256 queries on 256 CUs:
    // 1.43 ms of causal attention per video with two waves against 1.28 ms with four (tools/single_video_profile.py); every
    // block stages all its keys, so halving the block doubles the staging per query
    // D = 128 has no two-wave form: a block stages 64 KiB of K / V whatever its wave count, so the CU holds two blocks and a
    // two-wave block would leave ONE wave per SIMD (the compiler said so for every such instantiation: "desired occupancy was
    // 2, final occupancy is 1") - refused instead of run at half occupancy
    const bool small = kernel == 6;
    if constexpr (D != 64) {
      if (small) {
        mjv_set_error("attention: kernel 6 (two waves per workgroup) exists for head_dim 64 only - at head_dim 96 / 128 the 64 KiB of "
                      "staged K / V per workgroup would leave one wave per SIMD; use kernel 0 / 7");
        return MJV_E_UNSUPPORTED;
      }
      go(std::integral_constant<int, 4>{});
    } else {
      if (small) go(std::integral_constant<int, 2>{});
      else go(std::integral_constant<int, 4>{});
    }
    return mjv_check_launch("attention");
  }
  if constexpr (D == 96) {   // (ABI 7: head_dim 96 exists in the round-3 kernel only)
    mjv_set_error("attention: head_dim 96 runs on the round-3 kernel only (kernel 0 / 7)");
    return MJV_E_UNSUPPORTED;
  } else {
  // LDS-DMA staging up to 4096 keys per sequence (measured +2 ... +3 % at 1025 / 2186, 0 at 2048 non-causal); beyond that
  // the register-staged form is the faster one (causal, 8192 keys: 0.64 vs 0.66 ms), so the long-context config keeps it
  if (a.round_mode == 1) {
    if (dma) hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RM_DIV, 0, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RM_DIV, 0, false>), grid, dim3(256), 0, s, a);
  } else if (pow2) {
    if (dma) hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RM_POW2, 0, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RM_POW2, 0, false>), grid, dim3(256), 0, s, a);
  } else {
    if (dma) hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RM_MUL, 0, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<D, CAUSAL, RM_MUL, 0, false>), grid, dim3(256), 0, s, a);
  }
  return mjv_check_launch("attention");
  }
}

}  // namespace

#ifdef MJV_ATTN_STAMPS
extern "C" int mjv_attention_stamp_buffer(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  return hipMemcpyToSymbol(HIP_SYMBOL(v2::g_stamp_out), &q, sizeof(q)) == hipSuccess ? MJV_OK : MJV_E_LAUNCH;
}
#endif

#ifdef MJV_BENCH
// bench library only: 0 = production; 1-3 = timing variants of the round-2 kernel on the two production shapes (K/V staged
// once / softmax removed / MFMAs removed: wrong results by construction)
extern "C" int mjv_bench_attention_set(int32_t v) {
  if (v != 0 && v != 1 && v != 2 && v != 3) {
    mjv_set_error("bench_attention_set: %d not in {0, 1, 2, 3}", v);
    return MJV_E_ARG;
  }
  g_attn_bench = v;
  return MJV_OK;
}
#endif

extern "C" int mjv_attention_bf16(const mjv_attn_desc* d, void* stream) {
  MJV_REQUIRE(d && d->Q && d->K && d->V && d->O && d->cu_seqlens, "attention: null pointer");
  MJV_REQUIRE(d->head_dim == 64 || d->head_dim == 96 || d->head_dim == 128, "attention: head_dim %d not in {64, 96, 128}", d->head_dim);
  MJV_REQUIRE(d->kernel == 0 || (d->kernel >= 4 && d->kernel <= 7), "attention: kernel %d not in {0, 4, 5, 6, 7}", d->kernel);
  MJV_REQUIRE(d->score_round_mode >= 0 && d->score_round_mode <= 2, "attention: score_round_mode %d not in {0, 1, 2}", d->score_round_mode);
  MJV_REQUIRE(d->score_round_mode != 2 || (d->kernel != 4 && d->kernel != 5),
              "attention: score_round_mode 2 (unrounded fp32 scores) exists in the round-3 kernel only (kernel 0, 6 or 7)");
  MJV_REQUIRE(d->n_seqs > 0 && d->max_seqlen > 0 && d->n_heads > 0 && d->kv_group > 0, "attention: bad sizes");
  MJV_REQUIRE(d->n_heads % d->kv_group == 0, "attention: n_heads %% kv_group != 0");
  // (O: the round-3 kernel writes 16-byte pieces of a row - kernels 0 / 6 / 7; the older kernels' 8-byte stores are held to the
  // same contract so that one sentence in mjv.h covers every choice)
  MJV_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 8 == 0, "attention: ld alignment (multiples of 8 elements)");
  MJV_REQUIRE(d->q_head_stride % 8 == 0 && d->k_head_stride % 8 == 0 && d->v_head_stride % 8 == 0 &&
                  d->o_head_stride % 8 == 0, "attention: head stride alignment (multiples of 8 elements)");
  MJV_REQUIRE(((uintptr_t)d->Q | (uintptr_t)d->K | (uintptr_t)d->V | (uintptr_t)d->O) % 16 == 0,
              "attention: misaligned pointer (Q, K, V, O must be 16-byte aligned)");
  // K / V staging offsets inside one sequence are 32-bit byte offsets (row * ld * 2)
  MJV_REQUIRE(((double)d->max_seqlen + (double)(d->prefix_len > 0 ? d->prefix_len : 0)) * (double)(d->ldk > d->ldv ? d->ldk : d->ldv) * 2.0 < 2147483648.0,
              "attention: (max_seqlen + prefix_len) * max(ldk, ldv) * 2 = %.0f bytes does not fit the 32-bit staging offsets",
              ((double)d->max_seqlen + (double)d->prefix_len) * (double)(d->ldk > d->ldv ? d->ldk : d->ldv) * 2.0);
  // ABI 6: suffix queries / shared key prefix - causal launches of the round-3 kernel only
  const bool ext = d->cu_seqlens_q || d->prefix_len || d->prefix_k || d->prefix_v || d->max_seqlen_q;
  if (ext) {
    MJV_REQUIRE(d->prefix_len >= 0 && d->max_seqlen_q >= 0, "attention: negative prefix_len / max_seqlen_q");
    if (!d->causal || d->kernel == 4 || d->kernel == 5) {
      mjv_set_error("attention: cu_seqlens_q / prefix_k / prefix_v exist for causal launches of the round-3 kernel (kernel 0 / 7; 6 at head_dim 64)");
      return MJV_E_UNSUPPORTED;
    }
    MJV_REQUIRE(d->prefix_len % KB == 0, "attention: prefix_len %d must be a multiple of %d (a key tile lies on one side of the boundary)", d->prefix_len, KB);
    MJV_REQUIRE((d->prefix_len > 0) == (d->prefix_k != nullptr) && (d->prefix_len > 0) == (d->prefix_v != nullptr),
                "attention: prefix_k and prefix_v go with prefix_len > 0");
    MJV_REQUIRE(((uintptr_t)d->prefix_k | (uintptr_t)d->prefix_v) % 16 == 0, "attention: misaligned prefix pointer");
    MJV_REQUIRE((d->cu_seqlens_q != nullptr) == (d->max_seqlen_q > 0), "attention: cu_seqlens_q and max_seqlen_q go together");
  }
  AttnArgs a;
  a.cu_q = d->cu_seqlens_q; a.Kp = d->prefix_k; a.Vp = d->prefix_v; a.prefix_len = d->prefix_len;
  a.Q = d->Q; a.K = d->K; a.V = d->V; a.O = d->O;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.qhs = d->q_head_stride; a.khs = d->k_head_stride; a.vhs = d->v_head_stride; a.ohs = d->o_head_stride;
  a.cu = d->cu_seqlens; a.n_heads = d->n_heads; a.kv_group = d->kv_group; a.causal = d->causal;
  a.scale = d->scale; a.round_mode = d->score_round_mode;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic flops: 4 * D per (query, key) pair, halved under the causal mask (upper bound via max_seqlen)
  const int max_q = d->max_seqlen_q > 0 ? d->max_seqlen_q : d->max_seqlen;   // query rows of the longest sequence
  const double lq = max_q, lk = (double)d->max_seqlen + d->prefix_len;
  const double pairs = (double)d->n_seqs * d->n_heads * (d->causal ? lq * (lk - lq) + 0.5 * lq * lq : lq * lk);
  const double flops = 4.0 * d->head_dim * pairs;
  if (d->head_dim == 64) {
    if (d->causal) { MjvProfScope ps("attn_d64_causal", s, flops, 0); return launch<64, true>(a, d->n_seqs, max_q, d->kernel, s); }
    MjvProfScope ps("attn_d64", s, flops, 0);
    return launch<64, false>(a, d->n_seqs, d->max_seqlen, d->kernel, s);
  }
  if (d->head_dim == 96) {
    if (d->causal) { MjvProfScope ps("attn_d96_causal", s, flops, 0); return launch<96, true>(a, d->n_seqs, max_q, d->kernel, s); }
    MjvProfScope ps("attn_d96", s, flops, 0);
    return launch<96, false>(a, d->n_seqs, d->max_seqlen, d->kernel, s);
  }
  if (d->causal) { MjvProfScope ps("attn_d128_causal", s, flops, 0); return launch<128, true>(a, d->n_seqs, max_q, d->kernel, s); }
  MjvProfScope ps("attn_d128", s, flops, 0);
  return launch<128, false>(a, d->n_seqs, d->max_seqlen, d->kernel, s);
}
