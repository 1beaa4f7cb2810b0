/*
 * mjv_bench.h - measurement switches of the BENCH build of the library (make -C mj-video_amd/csrc bench ->
 * mj-video_amd/libmjv_hip_bench.so, compiled with -DMJV_BENCH).  The product library (libmjv_hip.so) exports none of
 * these and does not contain the kernel variants they select; nothing under mj-video_amd/ loads the bench build (only
 * tools/: gemm_bench.py, gemm_stamps.py, attn_bench.py with variant lists).  Process-wide, not thread-safe, and several
 * of the variants skip work on purpose - wrong results by construction - to attribute a kernel's time.
 */
#ifndef MJV_BENCH_H_
#define MJV_BENCH_H_
#include "mjv.h"
#ifdef __cplusplus
extern "C" {
#endif

/* GEMM: 0 resets everything.  4000 / 4001 split-K of 128-tile launches off / on; 4200 / 4201 K-sliced 256-tile launches off / on;
 * 4100 + s caps the slices per tile at s (1..8); 4300 + n sets the fewest K-tiles per 256-tile slice; 4400 + n the fewest K-tiles
 * of a problem that may run K-sliced on 256 tiles at all (default 64 = K >= 4096); 2000 + g forces the
 * group-M of the tile order (2000 = per shape); 6000 + m = largest M the skinny kernel takes (6000 = never);
 * 7000 / 7001 / 7002 streaming (nt) output stores by shape / never / always;
 * 1000 / 1003 / 1004 / 1006 select the 256-tile kernel's variants (1003: no epilogue, 1004: no global stores - wrong results;
 * 1006: s_memtime stamps of wave 0 to the stamp buffer; 1008: the persistent kernel with shader-clock / 100 MHz stamps around
 * every tile's main loop, summed per workgroup into slots 2 / 6 of its 8 x uint64 and the tile count into slot 7 - zero the
 * buffer first; 1009: never the persistent form). */
int mjv_bench_gemm_set(int32_t code);
/* 1006: wave 0 of every workgroup writes 8 x uint64 {start stamp, prologue, main loop, epilogue pass A, pass B, total
 * cycles, main loop in 100 MHz ticks (s_memrealtime)} to this device buffer; NULL = off.  Clock held in the main loop =
 * slot 2 / slot 6 x 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6). */
int mjv_bench_gemm_stamp_buffer(void* device_buffer);
/* attention: 0 = production; on the two production shapes of the round-2 kernel (desc.kernel = 5) 1 = K/V staged once,
 * 2 = softmax removed, 3 = MFMAs removed - wrong results. */
int mjv_bench_attention_set(int32_t variant);

/* RMSNorm with the row statistics supplied by the producer: partials = [rows][8] fp32 per-n-tile sums of squares (dim 2048),
 * summed in tile order.  Exists to measure what emitting the statistics from the EPI_SCALE_RES epilogue could return
 * (tools/norm_ab.py; DESIGN "Norm fusion"). */
int mjv_bench_rmsnorm_prestat(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* w,
                              const float* partials, int32_t rows, int32_t dim, float eps, void* stream);

/* The elementwise pass an UN-fused Linear would owe after a plain vendor GEMM (tools/fused_vs_unfused.py): same operations and
 * rounding points as the fused epilogues, HBM-bound.  kind 1: y = gelu(bf16(x + bias)); kind 3: y = res + bf16(bf16(x + bias)
 * * scale) (scale / bias may be NULL). */
int mjv_bench_epilogue_pass(const mjv_bf16* x, int64_t ldx, mjv_bf16* y, int64_t ldy, const mjv_bf16* bias, const mjv_bf16* scale,
                            const mjv_bf16* res, int64_t ldr, int32_t rows, int32_t cols, int32_t kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MJV_BENCH_H_ */
