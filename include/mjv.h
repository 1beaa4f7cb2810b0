/*
 * mjv.h - C ABI of libmjv_hip.so: the MI355X (gfx950) kernels under the MJ-VIDEO reward-scoring path.
 *
 * The reference (aiming-lab/MJ-Video) has no FFI: its boundary is the Python class API
 * (scripts/model/moe_reward.py:137-297).  This library sits UNDER the Python mirror of that API
 * (mj-video_amd/modeling.py) and replaces the PyTorch/cuBLAS/flash-attn ops the reference's forward
 * reaches.  Every entry point cites the reference code it replaces (paths relative to
 * /root/reference/scripts/model).
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers borrowed from the caller (PyTorch tensors); the library
 *     allocates nothing; the caller keeps buffers alive until the stream has been synchronised;
 *   - bf16 tensors are passed as `const uint16_t*` (raw bf16 bit patterns), row-major, with an explicit
 *     leading dimension in ELEMENTS;
 *   - every call only ENQUEUES on `stream` (a hipStream_t passed as void*; 0 = null stream) and never
 *     synchronises; the library keeps no setting between calls (ABI 4: what a caller may choose - the tile kernel of a
 *     GEMM, the attention kernel - travels in the call's descriptor; there is no process-wide setter), so calls from
 *     different threads on different streams do not interact; the only shared state is the opt-in profiler (mutex);
 *     the measurement switches of earlier ABI versions live in the separate bench build (include/mjv_bench.h);
 *   - return value 0 = ok, negative = error (MJV_E_*); `mjv_last_error()` gives a thread-local message.
 *   - rounding points follow the reference's bf16 eager path (bf16 result after every torch op), with ONE stated exception:
 *     attention's score_round_mode 2 keeps the scores in fp32 up to the softmax, as the reference's flash-attention path does
 *     on a GPU (modeling_intern_vit.py:229-244, modeling_internlm2.py:437-561).  Modes 0 / 1 are the eager rounding points;
 *     the Python model's DEFAULT is mode 2 (model.attention_scores = "flash"; "eager" selects 0 / 1; DESIGN 4 "Attention,
 *     round 4" holds the fixtures' verdict on the two).
 *
 * ABI 5 adds the MX-fp8 operand format for the GEMMs (BASELINE configs[4]; opt-in, the default path
 * is bf16 as before): the `*_format` / `*_scales` fields at the END of the GEMM descriptor - all zero = bf16 everywhere -,
 * mjv_quantize_mxfp8, mjv_layernorm_mxfp8, mjv_rmsnorm_mxfp8.
 * ABI 6 adds, at the END of the attention descriptor (all zero = ABI 5): suffix queries (cu_seqlens_q) and a shared key / value
 * prefix (prefix_k / prefix_v) for causal launches - the two pieces of work a scorer can leave out of the language tower - and
 * lets MJV_EPI_ROPE_QKV run with rope_group 0 (a k | v projection without q heads).
 * ABI 7 adds what the Phi-3-mini language tower of BASELINE configs[4] needs beyond the InternLM2 one: head_dim 96 in
 * mjv_attention_bf16 and mjv_rope_heads_bf16 (rotary embedding in place on a [q | k | v] projection).
 */
#ifndef MJV_H_
#define MJV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJV_ABI_VERSION 7

enum {
  MJV_OK = 0,
  MJV_E_ARG = -1,     /* bad argument (shape/alignment/null) */
  MJV_E_LAUNCH = -2,  /* HIP launch error */
  MJV_E_UNSUPPORTED = -3
};

typedef uint16_t mjv_bf16;

/* ---------------------------------------------------------------------------------------------
 * MXFP8 operand format (OCP Microscaling v1.0 "MXFP8 E4M3": 32-element blocks along K with one shared power-of-two
 * scale; the format gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 consumes).  The reference has no fp8 path (its README
 * names only the 2B bf16 checkpoint; BASELINE configs[4] asks for an fp8 MFMA weight path): this is a definition, pinned by
 * oracle/ref_fp8.py, not a restatement.
 *   elements : uint8 OCP e4m3fn, row-major [rows][ld] (K contiguous), K % 128 == 0, ld % 16 == 0
 *   scales   : uint8 e8m0 (value 2^(b - 127)), one per (row, 32-element block).  For a block of bf16 values with largest
 *              magnitude amax (bf16 bits u = E:8 | m:7):  b = max(1, E - 8 + (m > 0x60));  the smallest power of two with
 *              amax / 2^(b-127) <= 448, so no element saturates;  element = e4m3_rne(x / 2^(b-127)).
 *   scale layout ("lane layout" of the MFMA's scale operand): byte offset of (row, block kb of K-tile kt = k / 128)
 *              = ((kt * groups + row / 64) * 256) + (row % 16) * 16 + kb * 4 + (row / 16) % 4,   groups = ceil(rows / 64);
 *              i.e. per K-tile and 64-row group one 256-byte record whose dword [row % 16][kb] holds the four 16-row
 *              fragments' bytes - the dword a lane hands to the MFMA with op_sel = fragment.  mjv_mxfp8_scale_bytes().
 * ------------------------------------------------------------------------------------------- */
enum mjv_format { MJV_FMT_BF16 = 0, MJV_FMT_MXFP8 = 1 };
int64_t mjv_mxfp8_scale_bytes(int64_t rows, int64_t cols);
/* bf16 [rows][ldx] -> MXFP8 elements [rows][ldy] + scales (weights once at load; tests).  cols % 128 == 0. */
int mjv_quantize_mxfp8(const uint16_t* x, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* scales, int32_t rows, int32_t cols,
                       void* stream);

int mjv_abi_version(void);
const char* mjv_last_error(void);
/* name of the device architecture the code object was built for ("gfx950") */
const char* mjv_arch(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM:  C[M,N] = epilogue( A[M,K] @ W[N,K]^T )      (torch.nn.Linear layout: W is [out, in])
 * replaces every nn.Linear / Conv2d-as-GEMM on the path:
 *   internvl2/modeling_intern_vit.py:145-147,164 (patch embed), :196,208 (qkv, proj), :256-257 (fc1, fc2)
 *   internvl2/modeling_internvl_chat.py:135-140 (mlp1)
 *   internvl2/modeling_internlm2.py:301-307 (wqkv, wo), :256-258 (w1, w3, w2)
 *   moe_reward.py:24-26 (gating hidden layers)
 * Requirements: K % 64 == 0, N % 8 == 0, lda/ldw/ldc/ldr % 8 == 0, 16-byte aligned base pointers.
 * ------------------------------------------------------------------------------------------- */
enum mjv_epilogue {
  MJV_EPI_BIAS = 0,       /* C = bf16(acc + bias)                       bias may be NULL            */
  MJV_EPI_BIAS_GELU = 1,  /* C = bf16(gelu_erf(bf16(acc + bias)))       (F.gelu on the bf16 Linear)  */
  MJV_EPI_BIAS_RELU = 2,  /* C = bf16(max(acc + bias, 0))                                           */
  MJV_EPI_SCALE_RES = 3,  /* v = bf16(acc + bias); if scale: v = bf16(v * scale[n]); C = bf16(res + v) */
  MJV_EPI_SILU_MUL = 4,   /* W rows interleaved [16 x w1 | 16 x w3]...; C[M,N/2] =
                             bf16( bf16(silu(bf16(acc_w1))) * bf16(acc_w3) )   (modeling_internlm2.py:262) */
  MJV_EPI_ROPE_QKV = 5    /* InternLM2 wqkv (modeling_internlm2.py:359-381): columns are kv groups of (rope_group + 2) heads
                             of 128 = [q heads of the group | k | v].  v = bf16(acc) goes to C (its own columns, the other
                             columns of C are left untouched); the q and k heads get the rotary embedding on the bf16 Linear
                             output, out = bf16(bf16(x cos) + bf16(rotate_half(x) sin)) with cos / sin rows taken at
                             rope_pos[m], and go de-interleaved to rope_q [M][heads * 128] / rope_k [M][kv_heads * 128].
                             Needs N % ((rope_group + 2) * 128) == 0, no bias. */
};

typedef struct mjv_gemm_desc {
  const mjv_bf16* A; int64_t lda;   /* activations [M][lda] */
  const mjv_bf16* W; int64_t ldw;   /* weights     [N][ldw] */
  mjv_bf16* C; int64_t ldc;         /* output rows            */
  int32_t M, N, K;
  int32_t epilogue;                 /* enum mjv_epilogue */
  const mjv_bf16* bias;             /* [N] or NULL */
  const mjv_bf16* scale;            /* [N] or NULL (LayerScale ls1/ls2, modeling_intern_vit.py:291,293) */
  const mjv_bf16* res; int64_t ldr; /* residual rows or NULL */
  int32_t res_mod, res_off;         /* res_mod > 0: residual row = res_off + (m % res_mod)   (pos-emb add) */
  int32_t out_group, out_pad;       /* out_group > 0: out row = (m / out_group) * (out_group + out_pad) + out_pad
                                       + m % out_group   (skip the CLS slot of every tile)               */
  const int32_t* out_rows;          /* optional explicit output row per m (splice into <IMG_CONTEXT> rows,
                                       modeling_internvl_chat.py:176-179); overrides out_group            */
  void* workspace;                  /* optional scratch (device, 16-B aligned) private to this stream while the call is in
                                       flight; lets under-filled launches split K (fp32 partial tiles, summed in a fixed
                                       order: results stay deterministic).  NULL: never split.  mjv_gemm_workspace_bytes() */
  int64_t workspace_bytes;
  int32_t tile;                     /* 0 = automatic (256x256 8-wave kernel for M >= 512 and N >= 256; else the 64x32 skinny
                                       kernel up to 128 rows, 128x128 above); 64 / 128 / 256 force one kernel - every choice
                                       computes the same roundings, only the fp32 summation order differs (the parity tests
                                       run every case on all three) */
  /* MJV_EPI_ROPE_QKV only */
  const mjv_bf16 *rope_cos, *rope_sin;   /* [positions][128] tables (bf16, modeling_internlm2.py:147-180) */
  const int32_t* rope_pos;               /* [M] position of every row (device) */
  mjv_bf16 *rope_q, *rope_k;             /* outputs */
  int64_t rope_ldq, rope_ldk;
  int32_t rope_group;                    /* q heads per kv head; 0 (ABI 6) = columns are [k | v] pairs only, rope_q unused */
  /* ---- ABI 5: operand / output formats (enum mjv_format; all zero = bf16 everywhere, the ABI-4 behaviour) ----
   * a_format == w_format == MJV_FMT_MXFP8: A and W point to e4m3 bytes (lda / ldw in BYTES = elements), a_scales / w_scales
   * to their scale records (layout above; groups = ceil(M / 64) and ceil(N / 64)); K % 128 == 0; the product of the
   * dequantised operands is accumulated in fp32 by v_mfma_scale_f32_16x16x128_f8f6f4 and the epilogue is the bf16 one.
   * Epilogues: BIAS, BIAS_GELU, BIAS_RELU, SCALE_RES, SILU_MUL (plain output rows only; no ROPE_QKV).
   * c_format == MJV_FMT_MXFP8 (BIAS, BIAS_GELU, BIAS_RELU, SILU_MUL): the epilogue's bf16 result is block-quantised on
   * the way out - C points to e4m3 bytes (ldc in bytes), c_scales to its scale records (groups = ceil(M / 64)); output
   * width (N, or N / 2 for SILU_MUL) % 128 == 0.  Bit-identical to the bf16 output followed by mjv_quantize_mxfp8. */
  int32_t a_format, w_format, c_format;
  const uint8_t *a_scales, *w_scales;
  uint8_t* c_scales;
  /* ---- ABI 5: a norm folded into this GEMM (north_star "RMSNorm / LayerNorm -> GEMM fusion"; bf16 operands) ----
   * The Linear's value before its bf16 rounding becomes   row_scale[m] * acc - row_shift[m] * col_shift[n] + bias_f32[n]
   * (row_shift / col_shift / bias_f32 all NULL: row_scale[m] * acc + bias[n]).  With A = the UN-normalised rows x and
   * W' = W * gain (folded once), row_scale = rstd, row_shift = mean * rstd, col_shift[n] = sum_k W'[n][k],
   * bias_f32 = bias + W beta this is Linear(LayerNorm(x)) (modeling_intern_vit.py:291-293) / Linear(RMSNorm(x))
   * (modeling_internlm2.py:138-143,653,669) without the normalised rows ever reaching HBM; mjv_row_stats_bf16 writes the two
   * row vectors.  Rounding points DIFFER from the reference's (no bf16 rounding of the normalised rows, the gain rounded into
   * W'): opt-in, behind model.norm_fusion, judged by the fixtures (DESIGN "Norm fusion, round 4").
   * fp32 vectors, 16-byte aligned; row vectors readable up to ceil(M / 256) * 256 entries, column vectors up to
   * ceil(N / 256) * 256 (the 256-tile kernel fetches whole tiles of them by LDS-DMA).  Not with MJV_EPI_SCALE_RES.  The
   * 256-tile kernel instantiates BIAS / BIAS_GELU with a folded LayerNorm and SILU_MUL / ROPE_QKV with a folded RMSNorm (the
   * model's four call sites); every other combination runs on the small-tile kernels or returns MJV_E_UNSUPPORTED. */
  const float *row_scale, *row_shift, *col_shift, *bias_f32;
} mjv_gemm_desc;

/* a workspace of this size is enough for every problem shape (256 partial 256x256 fp32 tiles = 64 MiB) */
int64_t mjv_gemm_workspace_bytes(void);

int mjv_gemm_bf16(const mjv_gemm_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Flash-style attention over packed variable-length sequences (no N x N scores in HBM).
 *   non-causal D=64 : internvl2/modeling_intern_vit.py:210-227 (_naive_attn) / :229-244 (_flash_attn)
 *   causal GQA D=128: internvl2/modeling_internlm2.py:383-411 (eager) / :437-561 (flash varlen)
 * Q/K/V/O are addressed as base + row * ld + head * head_stride (elements); K/V head = q_head / kv_group.
 * Alignment: Q, K, V, O 16-byte aligned; ldq / ldk / ldv / ldo and the four head strides multiples of 8 elements (every
 * kernel moves 16-byte pieces of a row; O included since round 4).
 * score rounding: mode 0: s = bf16(acc * scale)   [(q*scale) @ k^T with scale a power of two]
 *                 mode 1: s = bf16(bf16(acc) * scale)   [matmul, then / sqrt(D) in bf16]
 *                 mode 2: s = acc * scale in fp32, never rounded   [the reference's flash-attention numerics, the path it takes on
 *                         a GPU: modeling_intern_vit.py:229-244, modeling_internlm2.py:437-561; kernel 0 / 6 / 7 only]
 * ------------------------------------------------------------------------------------------- */
typedef struct mjv_attn_desc {
  const mjv_bf16 *Q, *K, *V;
  mjv_bf16* O;
  int64_t ldq, ldk, ldv, ldo;
  int32_t q_head_stride, k_head_stride, v_head_stride, o_head_stride;
  const int32_t* cu_seqlens;  /* [n_seqs + 1] packed row offsets (device) */
  int32_t n_seqs, max_seqlen;
  int32_t n_heads, kv_group;
  int32_t head_dim;           /* 64, 128, or (ABI 7; kernel 0 / 7 only) 96 - Phi-3-mini's heads: dense 192-byte K / V rows in LDS */
  int32_t causal;
  float scale;
  int32_t score_round_mode;
  int32_t kernel;             /* 0 = automatic (round 3: the two-sub-block pipelined kernel); 4 = the register-staged round-1
                                 kernel for every shape;
                                 5 = the round-2 choice (its LDS-DMA form up to 4096 keys); 6 / 7 = the round-3 kernel with two /
                                 four waves (128 / 256 queries at head_dim 64; 128 at 128) per workgroup (automatic = four;
                                 bit-identical results).  6 exists at head_dim 64 only: at 128 a workgroup stages 64 KiB of
                                 K / V, two fit a CU, and a two-wave block would run one wave per SIMD - MJV_E_UNSUPPORTED.
                                 Every accepted choice gives correct results: the tests A/B them.  Other values: MJV_E_ARG */
  /* ---- ABI 6 (all zero = the ABI-5 behaviour): what a SCORER may leave out of a causal tower (kernel 0 / 6 / 7, causal only;
   * otherwise MJV_E_UNSUPPORTED).  The reward heads read two late rows per sample (moe_reward.py:226-243) and every prompt
   * starts with the same system tokens (conversation.py:354-365, eval_genai_mjvideo.py:132-137):
   *   cu_seqlens_q / max_seqlen_q : the queries of sequence s are only its LAST cu_seqlens_q[s+1] - cu_seqlens_q[s] positions,
   *       packed at those rows of Q and O (cu_seqlens still describes the K / V rows); local query i sits at key position
   *       prefix_len + keys(s) - queries(s) + i.  max_seqlen_q >= the longest query count.
   *   prefix_k / prefix_v / prefix_len : prefix_len keys SHARED by every sequence precede its own: rows of prefix_k / prefix_v
   *       [prefix_len][ldk / ldv] with K's / V's head strides (a cache of the constant prompt prefix's keys / values);
   *       prefix_len % 64 == 0.  Key j < prefix_len comes from the prefix rows, key j >= prefix_len from row j - prefix_len
   *       of the sequence.  Same arithmetic, same key-tile boundaries as one launch over the concatenated rows. */
  const int32_t* cu_seqlens_q;
  int32_t max_seqlen_q;
  int32_t prefix_len;
  const mjv_bf16 *prefix_k, *prefix_v;
} mjv_attn_desc;

int mjv_attention_bf16(const mjv_attn_desc* d, void* stream);
/* max_seqlen MUST be >= the longest sequence of cu_seqlens: query rows beyond it are not computed (their O rows are left
 * untouched). */

/* LayerNorm over the last dim, fp32 statistics, bf16 out (nn.LayerNorm on bf16:
 * modeling_intern_vit.py:291,293; modeling_internvl_chat.py:136).
 * gather_grid > 0 selects the pixel-shuffle gather of modeling_internvl_chat.py:228-242,255-260:
 * output row (tile, a2, b2) is the concatenation of the 4 source rows
 * (2a2,2b2) (2a2,2b2+1) (2a2+1,2b2) (2a2+1,2b2+1) of a [tile][1 + grid*grid][dim/4] buffer (CLS skipped). */
int mjv_layernorm_bf16(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* gamma,
                       const mjv_bf16* beta, int32_t rows, int32_t dim, float eps, int32_t gather_grid,
                       void* stream);

/* InternLM2RMSNorm (modeling_internlm2.py:138-143): y = w * bf16(x * rsqrt(mean(x^2) + eps)).
 * row_index (optional, device) gathers input rows: y[i] = norm(x[row_index[i]]). */
int mjv_rmsnorm_bf16(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* w,
                     const int32_t* row_index, int32_t rows, int32_t dim, float eps, void* stream);

/* the same two norms with an MXFP8 output (elements y [rows][ldy] bytes + scale records): bf16(norm(x)) exactly as above,
 * then block-quantised - what the fp8 FFN GEMMs (fc1, w1|w3) read.  dim % 128 == 0.  No gathers. */
int mjv_layernorm_mxfp8(const mjv_bf16* x, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* y_scales, const mjv_bf16* gamma,
                        const mjv_bf16* beta, int32_t rows, int32_t dim, float eps, void* stream);
int mjv_rmsnorm_mxfp8(const mjv_bf16* x, int64_t ldx, uint8_t* y, int64_t ldy, uint8_t* y_scales, const mjv_bf16* w,
                      int32_t rows, int32_t dim, float eps, void* stream);

/* Row statistics of a norm that is folded into the consuming GEMM (mjv_gemm_desc.row_scale): one pass over x,
 * rstd[m] = rsqrt(var + eps) with var = mean((x - mean)^2) (LayerNorm) or mean(x^2) (mean_rstd == NULL: RMSNorm),
 * mean_rstd[m] = mean * rstd.  8 bytes written per row instead of the normalised row. */
int mjv_row_stats_bf16(const mjv_bf16* x, int64_t ldx, float* rstd, float* mean_rstd, int32_t rows, int32_t dim, float eps,
                       void* stream);

/* GQA de-interleave + rotary embedding (modeling_internlm2.py:361-381,233-247):
 * qkv [rows][kv_heads * (group + 2) * 128] -> q [rows][kv_heads*group*128], k [rows][kv_heads*128], both rotated
 * with bf16 cos/sin tables [max_pos][128]; V stays in place (read strided by the attention kernel). */
int mjv_rope_split_bf16(const mjv_bf16* qkv, int64_t ldqkv, mjv_bf16* q, int64_t ldq, mjv_bf16* k, int64_t ldk,
                        const mjv_bf16* cos_tab, const mjv_bf16* sin_tab, const int32_t* positions,
                        int32_t rows, int32_t kv_heads, int32_t group, void* stream);

/* ABI 7 - rotary embedding in place on `n_heads` consecutive heads of every row (head h at x + row * ldx + h * head_stride): the
 * language tower of BASELINE configs[4] (InternVL2-4B = InternViT + Phi-3-mini; the reference cannot build it -
 * internvl2/modeling_internvl_chat.py:125-130 - so this restates transformers/models/phi3/modeling_phi3.py, the module the
 * upstream checkpoint's code is generated from): apply_rotary_pos_emb on qkv_proj's [q heads | k heads | v heads] columns, one
 * launch over the q + k heads.  x' = bf16(bf16(x cos) + bf16(rotate_half(x) sin)) on the first rot_dim elements of a head;
 * cos / sin [positions][rot_dim] bf16 tables that already carry the LongRoPE attention factor (Phi3RotaryEmbedding.forward).
 * rot_dim % 16 == 0, head_stride % 8 == 0, 16-byte aligned pointers. */
int mjv_rope_heads_bf16(mjv_bf16* x, int64_t ldx, int32_t head_stride, int32_t n_heads, int32_t rot_dim, const mjv_bf16* cos_tab,
                        const mjv_bf16* sin_tab, const int32_t* positions, int32_t rows, void* stream);

/* im2col for the patch-embedding conv (modeling_intern_vit.py:145-147,164): pixels [tiles][3][S][S] ->
 * patches [tiles * (S/P)^2][ldp], column = c*P*P + i*P + j, zero-padded to ldp. */
int mjv_patchify_bf16(const mjv_bf16* pixels, mjv_bf16* patches, int64_t ldp, int32_t tiles, int32_t image_size,
                      int32_t patch, void* stream);

/* CLS row of every tile: x[tile * tokens_per_tile] = bf16(cls + pos0)   (modeling_intern_vit.py:166-173) */
int mjv_cls_rows_bf16(mjv_bf16* x, int64_t ldx, const mjv_bf16* cls, const mjv_bf16* pos0, int32_t tiles,
                      int32_t tokens_per_tile, int32_t dim, void* stream);

/* tok_embeddings gather (modeling_internvl_chat.py:163): x[t] = table[ids[t]] unless ids[t] == skip_id
 * (those rows are written by the projector GEMM through out_rows). */
int mjv_embed_gather_bf16(const int32_t* ids, const mjv_bf16* table, int64_t ldt, mjv_bf16* x, int64_t ldx,
                          int32_t rows, int32_t dim, int32_t skip_id, int32_t vocab, void* stream);

/* Reward / gating heads (moe_reward.py:226-285).  hr / hg = post-final-norm reward rows h_r and gating rows h_g
 * ([B][hidden] each, row stride ldh).  ga/gc = outputs of the last hidden layer (post-ReLU) of the aspect / criteria
 * gating MLPs.  Outputs follow CustomOutput (moe_reward.py:287-297). */
typedef struct mjv_heads_desc {
  const mjv_bf16* hr; const mjv_bf16* hg; int64_t ldh; int32_t hidden;
  const mjv_bf16* ga; const mjv_bf16* gc; int64_t ldg; int32_t gate_hidden;
  const mjv_bf16* w_reg;      /* [n_obj][hidden]  regression_layer.weight */
  const mjv_bf16* w_transform;/* [n_obj][n_obj]   reward_transform_matrix */
  const mjv_bf16* wa; const mjv_bf16* ba;  /* aspect_gating last layer   [n_asp][gate_hidden], [n_asp] */
  const mjv_bf16* wc; const mjv_bf16* bc;  /* criteria_gating last layer [n_obj][gate_hidden], [n_obj] */
  const mjv_bf16* ls_a; const mjv_bf16* ls_c; /* logit_scale[0] of each net */
  float temperature;
  int32_t batch, n_obj, n_asp;
  const int32_t* group_offsets; /* [n_asp + 1] into group_index */
  const int32_t* group_index;   /* [n_obj] criteria ids in aspect2criteria dict order */
  mjv_bf16* rewards;            /* [B][n_obj] */
  mjv_bf16* criteria_gating;    /* [B][n_obj] pre-softmax */
  mjv_bf16* aspect_gating;      /* [B][n_asp] */
  mjv_bf16* aspect_weights;     /* [B][n_obj] concatenated in dict order */
  mjv_bf16* weighted_last;      /* [B] last aspect's bf16 weighted sum (moe_reward.py:273,294) */
  float* aspect_scores;         /* [B][n_asp] */
  float* score;                 /* [B] */
  float* packed34;              /* optional [B][1 + n_asp + n_obj] = (score, aspect_scores, rewards) for the
                                   RCCL all-gather of SURVEY.md §8(e); may be NULL */
} mjv_heads_desc;

int mjv_reward_heads_bf16(const mjv_heads_desc* d, void* stream);

/* Frame preprocessing of load_video on the GPU (scripts/data_processor/data.py:56-64,81-117,158-179):
 * decoded uint8 RGB frames [n_frames][height][width][3] -> Pillow-bit-exact bicubic resize to out_w x out_h (two passes,
 * uint8 intermediate `tmp` [n_frames][height][out_w][3]) -> crop into tile_size^2 tiles (row-major grid) ->
 * ((u8/255) - mean) / std in fp32 -> bf16, written to out[(frame * tiles_per_frame + tile_offset + tile)][3][S][S].
 * xbounds/ybounds = (first tap, tap count) per output column / row, xcoef/ycoef = 22-bit fixed-point taps
 * [out][kx|ky], computed on the host exactly as Pillow does (mj-video_amd/video.py: pil_resample_coeffs).
 * mean / stdv are HOST pointers to 3 floats. */
int mjv_resize_normalize_u8(const uint8_t* frames, int32_t n_frames, int32_t height, int32_t width, int32_t out_w,
                            int32_t out_h, const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                            const int32_t* ybounds, const int32_t* ycoef, int32_t ky, uint8_t* tmp, mjv_bf16* out,
                            int32_t tile_size, int32_t tiles_per_frame, int32_t tile_offset, const float* mean,
                            const float* stdv, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Opt-in profiler: when enabled, every launch is bracketed by hipEvents recorded on ITS stream.
 * mjv_prof_collect synchronises those events and accumulates per-kernel-tag time; used by bench.py
 * for the roofline line (kernel average launch duration measured on the launch stream).
 * ------------------------------------------------------------------------------------------- */
int mjv_prof_enable(int32_t on);
/* record only launches with this tag (NULL or "" = every tag): lets a timed region carry events for one kernel only */
int mjv_prof_filter(const char* tag);
int mjv_prof_reset(void);
int mjv_prof_collect(void);
/* number of distinct tags seen; tag i: name, launches, total ms, total algorithmic flops, total algorithmic bytes */
int mjv_prof_count(void);
int mjv_prof_get(int32_t i, const char** name, int64_t* launches, double* ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* MJV_H_ */
