#!/usr/bin/env python3
"""Writes KERNELS.md - the one-page state of the kernels - from a bench.py line (the per-kernel table of its profiled step and
the `secondary` legs) plus the fixed text below.  usage: kernels_md.py profiles/rNN_x_bench_default.json [driver BENCH_rNN.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_line(path):
    txt = open(path).read()
    try:
        j = json.loads(txt)
        if isinstance(j, dict) and "parsed" in j:      # a driver record (BENCH_rNN.json)
            return j["parsed"]
    except ValueError:
        pass
    return json.loads([ln for ln in txt.splitlines() if ln.startswith('{"metric')][-1])


# kernel tag -> (what it is, shapes at the headline workload, bound, peak for `frac`, the ONE reason it is not faster, DESIGN anchor)
ROWS = [
    ("gemm256_scale_res", "256² tile GEMM, bias·LayerScale + residual epilogue", "ViT proj 65 536×1024×1024, fc2 65 536×1024×4096; LLM wo 16 384×2048×2048, w2 16 384×2048×8192 (main rows)", "MFMA", 2500,
     "main loop at 0.85-0.96 of what the 1.7-1.9 GHz the chip holds allows; the residual epilogue (64 residual registers, LDS round trip) is not overlapped by MFMA work: 1 workgroup per CU owns all 160 KiB (round 6: the 128 × 256 two-per-CU form that does overlap it - epilogue 4 % of a tile - was built and measured 23-36 % slower: its main loop runs at 0.68 ×)", "docs/rounds/r04.md 'the clock'; r06.md §3"),
    ("gemm256_silu_mul", "256² GEMM, SiLU(w1 x)·(w3 x) epilogue, persistent form", "LLM w1∣w3 16 976×16 384×2048", "MFMA", 2500,
     "same main loop; last round 75 % full (67 × 64 tiles = 16.75 rounds) and 5.5 × the algorithmic fabric bytes (tile order floor)", "docs/rounds/r03.md 'persistent'; DESIGN §9"),
    ("gemm256_bias_gelu", "256² GEMM, bias + exact-erf GELU by table", "ViT fc1 65 536×4096×1024", "MFMA", 2500,
     "K = 1024 is 16 K-tiles per tile, so the epilogue is ⅓ of a tile's life; 128 LDS table gathers per lane (≈ 9 LDS cycles each: random banks) + 5.5 vector instr. per element that no MFMA overlaps (beside a neighbour's main loop, in the two-per-CU form, the GELU still costs 20 %: round 6)", "docs/rounds/r04.md 'GELU'; r06.md §3"),
    ("gemm256_bias", "256² GEMM, bias epilogue, persistent form", "ViT qkv 65 536×3072×1024; projector", "MFMA", 2500,
     "short K (16 K-tiles): prologue + epilogue ≈ 20 % of a tile even with the next tile's first K-tile prefetched under the epilogue", "docs/rounds/r03.md 'persistent'; DESIGN §9"),
    ("gemm256_rope_qkv", "256² GEMM, rotary embedding + GQA de-interleave epilogue", "LLM wqkv 16 384×4096×2048 (23 layers), last layer k∣v 16 384×2048×2048", "MFMA", 2500,
     "two dependent table fetches (position → cos / sin rows) per pass in the epilogue; no persistent form (its LDS window is the whole tile)", "docs/rounds/r02.md"),
    ("attn_d64", "flash-style attention, D = 64, non-causal, 64 seq × 16 heads × 1025", "ViT, 24 layers", "MFMA (VALU-issue limited)", 2500,
     "6 vector instructions per score pair, 4 of the 8 issue slots are the two v_exp_f32: the softmax, not the MFMA, sets the pace at D = 64 (moving the row sum onto the MFMA pipe makes the matrix pipe the longer one: round 6); + one extra block per (sequence, head) for query 0 (0.8 ms per step gross)", "DESIGN §4.3; docs/rounds/r03.md, r04.md, r06.md §4"),
    ("attn_d128_causal", "flash-style attention, D = 128, causal GQA, 8 seq × 16/8 heads × 2122 (+64 cached prefix keys)", "LLM, 24 layers (last: 5 queries per sequence)", "MFMA + VALU", 2500,
     "one 32-query sub-block per wave (two would need ≈ 310 registers): MFMA and softmax of one unit serialise more than at D = 64; diagonal tiles run the masked general path", "DESIGN §4.3; docs/rounds/r03.md, r04.md, r06.md §4"),
    ("layernorm", "LayerNorm 1024, one wave per row", "ViT norm1 / norm2, 65 600 rows × 48", "HBM", 8000,
     "4 B per element of traffic at 5.7-5.9 TB/s = the 6.3 TB/s this chip's HBM delivers to any kernel; only folding it into the GEMM removes it (built: model.norm_fusion, off for parity)", "docs/rounds/r04.md 'Norm fusion'"),
    ("rmsnorm", "RMSNorm 2048 (cast before gain), one wave per row", "LLM attention_norm / ffn_norm, 16 976 rows × 48", "HBM", 8000,
     "as layernorm (5.0 TB/s: shorter launch, 35 MB per launch sits partly in the Infinity Cache)", "same"),
    ("gemm256s_scale_res", "K-sliced 256² GEMM + finish kernel", "LLM w2 tail: 592 rows × 2048 × 8192 in 8 slices", "MFMA / fabric", 2500,
     "a tail: 24 tiles cannot fill 256 CUs without slicing K, and the slices' fp32 images make a 50 MB round trip", "docs/rounds/r03.md, r05.md 'Tails'; DESIGN §9"),
    ("gemm64_scale_res", "64×32 skinny GEMM", "ViT proj / fc2 tails: the 64 CLS rows of M = 65 600 = 256·256 + 64 (48 launches), last-layer rows", "latency", 2500,
     "32 workgroups each walk K / 64 dependent LDS fills (≈ 20 µs at K = 4096): latency-bound by construction", "docs/rounds/r02.md"),
    ("gemm128_bias", "128² GEMM (+ rope_split)", "LLM wqkv tail 592 × 4096 × 2048; heads", "L2→LDS fill", 2500,
     "160 workgroups, one per CU, double-buffered only: ≈ 45 GB/s of LDS fill per CU; K-slicing at K = 2048 is a wash (measured r3)", "docs/rounds/r03.md, r05.md 'Tails'; DESIGN §9"),
    ("gemm128_scale_res", "128² GEMM, residual epilogue", "LLM wo tail 592 × 2048 × 2048", "L2→LDS fill", 2500, "as gemm128_bias", "same"),
    ("gemm64_bias_gelu", "64×32 skinny GEMM", "ViT fc1 tail (64 CLS rows)", "latency", 2500, "as gemm64_scale_res", "same"),
    ("gemm64_bias", "64×32 skinny GEMM", "ViT qkv tail (64 CLS rows), gating layers", "latency", 2500, "as gemm64_scale_res", "same"),
]
F8 = [
    ("gemm256f8_scale_res", "MXFP8 256² GEMM (v_mfma_scale_f32_16x16x128_f8f6f4), LayerScale / residual epilogue", "fc2, w2 main rows", 5000,
     "the bf16 kernel's loop with twice the flops per LDS byte: 2 450 TFLOP/s is the same 0.85-0.95 of the held clock; bf16 residual stream in the epilogue"),
    ("gemm256f8_silu_mul", "MXFP8 GEMM, SiLU·mul → MXFP8 output (block quantiser in pass B)", "w1∣w3", 5000, "as above; last round 75 % full"),
    ("gemm256f8_bias_gelu", "MXFP8 GEMM, bias + GELU → MXFP8", "fc1", 5000,
     "8 K-tiles per tile: a 13 µs main loop against 3 µs of fixed costs + 4.2 µs of GELU table gathers (LDS-bound) + 0.5-1.5 µs of block quantiser (profiles/r05_v_gemm_fp8_bench.txt: epilogue decomposition)"),
    ("gemm256f8s_scale_res", "K-sliced MXFP8 GEMM + finish kernel (round 5)", "fc2 64-row tail (16 slices), w2 592-row tail (10 slices)", 5000,
     "tails: before round 5 they were an extra, mostly empty round of full tiles (fc2: 5 rounds for 4.02 rounds of work)"),
]


def mfma_busy(suffix=""):
    """tag -> MFMA-pipe busy share from the committed PMC summary (tools/mfma_busy_summary.py), if there is one
    (suffix "_4b": the passes over bench.py --backbone 4b)"""
    import glob
    import re
    out = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_mfma_pipe_busy{suffix}.txt")))
    if not files:
        return out
    out["_file"] = os.path.relpath(files[-1], ROOT)
    epi = ["bias", "bias_gelu", "bias_relu", "scale_res", "silu_mul", "rope_qkv"]
    for ln in open(files[-1]):
        m = re.match(r"t256::gemm256p?_kernel<(\d), 0.*\s(\d\.\d+)\s*$", ln)
        if m:
            out["gemm256_" + epi[int(m.group(1))]] = float(m.group(2))
        m = re.match(r"v2::attn2_kernel<(\d+), (true|false).*\s(\d\.\d+)\s*$", ln)
        if m:
            out["attn_d" + m.group(1) + ("_causal" if m.group(2) == "true" else "")] = float(m.group(3))
    return out


def main():
    line = load_line(sys.argv[1])
    busy = mfma_busy()
    drv = load_line(sys.argv[2]) if len(sys.argv) > 2 else None
    k = line["kernels"]
    sec = line.get("secondary", {})
    out = []
    w = out.append
    w("# KERNELS — state of the kernels on one page\n")
    w(f"Headline workload (BASELINE.json configs[1]: 4 pairs = 8 videos × 8 frames @448², N = 2186 tokens per video, bf16, 1 MI355X): "
      f"**{line['value']:.2f} pairs/s, {line['ms_per_step']:.2f} ms per step = {line['frac_of_mfma_roofline']:.3f} of the 2.5 PFLOP/s MFMA peak on the "
      f"{line.get('executed_tflop_per_pair', 25.8)} TFLOP per pair it EXECUTES** (25.8 algorithmic: 96.9 pairs/s at the peak; prefix cache "
      f"{'on' if line['config'].get('prefix_cache') else 'off'}, last-layer query trimming {'on' if line['config'].get('trim_last_layer') else 'off'}); "
      f"every row in every forward (`value_all_work`): **{line.get('value_all_work')} pairs/s = {line.get('frac_all_work')}** "
      f"(`{os.path.relpath(sys.argv[1], ROOT)}`, a box of this round's pool)."
      + (f" The driver's own record: **{drv['value']:.2f} pairs/s** (`{os.path.basename(sys.argv[2])}`)." if drv else "") + "\n")
    w("Every kernel is hand-written HIP for gfx950 (`mj-video_amd/csrc/`). Columns: time per 4-pair step from the profiled step of that "
      "bench line (HIP events around every launch), achieved rate, fraction of the peak that bounds it (2.5 PFLOP/s dense bf16 MFMA, "
      "8 TB/s HBM), the MFMA pipe's busy share by the PMC counters (of the cycles the chip actually ran: it holds 1.7-2.0 GHz of its 2.4 "
      "under these kernels), the ONE reason it is not faster, and where DESIGN.md holds the measurements behind that sentence.\n")
    w("| kernel (profiler tag) | what / where | ms per step | achieved | of peak | MFMA busy | bound | the one reason it is not faster | DESIGN |")
    w("|---|---|---|---|---|---|---|---|---|")
    seen = 0.0
    for tag, what, shapes, bound, peak, why, ref in ROWS:
        r = k.get(tag)
        if not r:
            continue
        seen += r["ms_per_step"]
        if bound.startswith("HBM"):
            ach, frac = f"{r['gbs']:.0f} GB/s", r["gbs"] / peak
        else:
            ach, frac = (f"{r['tflops']:.0f} TFLOP/s", r["tflops"] / peak) if r.get("tflops") else ("-", 0.0)
        w(f"| `{tag}` | {what}; {shapes} | {r['ms_per_step']:.2f} | {ach} | {frac:.2f} | {busy.get(tag, '-')} | {bound} | {why} | {ref} |")
    rest = sum(v["ms_per_step"] for v in k.values()) - seen
    w(f"| (everything else) | patchify, CLS rows, embedding gather, pixel-shuffle LayerNorm, rope_split of tail rows, gating GEMMs, reward heads | "
      f"{rest:.2f} | - | - | - | launch latency / HBM | small launches; < 1 % of the step | §4 table |\n")
    f8 = sec.get("fp8_ffn") or {}
    if f8.get("value"):
        w(f"## fp8 FFN path (`model.set_ffn_format('mxfp8')`; `secondary.fp8_ffn` of the same line): {f8['value']:.2f} pairs/s, {f8['ms_per_step']:.2f} ms per step, "
          f"{f8['mixed_roofline']['frac']:.3f} of its mixed roofline ({f8['mixed_roofline']['ceiling_pairs_per_s']} pairs/s: FFN flops at 5 PFLOP/s, the rest at 2.5)\n")
        w("| kernel | what / where | ms per step | TFLOP/s | of 5 PFLOP/s | the one reason it is not faster |")
        w("|---|---|---|---|---|---|")
        for tag, what, shapes, peak, why in F8:
            r = f8["fp8_kernels"].get(tag)
            if r:
                w(f"| `{tag}` | {what}; {shapes} | {r['ms_per_step']:.2f} | {r['tflops']:.0f} | {r['tflops'] / peak:.2f} | {why} |")
        w("\nThe four attention-side Linears stay bf16: each of them on MXFP8 operands adds noise and none keeps 0 flips "
          "(`profiles/r05_c_fp8_per_linear_study.txt`).  Which FFN Linear carries the rank noise, and the presets that follow "
          "(`profiles/r06_a_fp8_ffn_subset_study.txt`, DESIGN §5):\n")
        w("| preset | fp8 Linears | pairs/s | ms per step | Spearman vs the reference's bf16 scores @224² / @448² | decisive flips |")
        w("|---|---|---|---|---|---|")
        for leg, preset, lin, rho in (("fp8_rank999", "mxfp8-rank999", "fc1, fc2, w2", "0.99920 / 0.99912 (north_star's bar: ≥ 0.999)"),
                                      ("fp8_ffn", "mxfp8", "fc1, fc2, w1∣w3, w2", "0.99863 / 0.99820 (its own stated tolerance)")):
            r = sec.get(leg) or {}
            if r.get("value"):
                w(f"| `{preset}` | {lin} | {r['value']:.2f} | {r['ms_per_step']:.1f} | {rho} | 0 / 0 |")
        w("")
    w("## The other configs of BASELINE.json, same process (`secondary`)\n")
    w("| leg | config | pairs/s | ms per step | of its MFMA roofline |")
    w("|---|---|---|---|---|")
    for name, label in (("prefix_cache_off", "configs[1] with prefix cache and last-layer trimming OFF (every row, every forward)"),
                        ("pairs8", "configs[2]'s shard: 8 pairs = 16 videos on one GPU (the weak-scaling baseline)"),
                        ("c4_112_tiles", "configs[3]: 16 frames × 7 tiles = 112 tiles per video, N = 28 810")):
        r = sec.get(name) or {}
        if r.get("value"):
            w(f"| `{name}` | {label} | {r['value']:.2f} | {r['ms_per_step']:.1f} | {r.get('frac_of_mfma_roofline', '-')} |")
    c5 = sec.get("c5_4b") or {}
    if c5.get("value"):
        w(f"\n## BASELINE configs[4]: the InternVL2-4B backbone (InternViT + Phi-3-mini, {c5['parameters_G']} G parameters; `secondary.c5_4b`)\n")
        w(f"4 pairs × 8 frames @448², N = {c5['N']}, one MI355X; {c5['algorithmic_tflop_per_pair']} algorithmic TFLOP per pair = {c5['roofline_pairs_per_s_bf16']} pairs/s at "
          "the bf16 MFMA peak.  Oracle: `oracle/ref_phi3.py`, pinned to transformers 5.15's `Phi3ForCausalLM` run inside the reference's reward model.\n")
        w("| format | pairs/s | ms per step | of the bf16 MFMA roofline | largest kernels (ms per step, TFLOP/s) |")
        w("|---|---|---|---|---|")
        for name, label in (("bf16", "bf16"), ("fp8_rank999", "`mxfp8-rank999`"), ("fp8_ffn", "`mxfp8`")):
            r = c5.get(name) or {}
            if r.get("value"):
                top = "; ".join(f"`{t}` {v['ms_per_step']:.1f}" + (f" @ {v['tflops']:.0f}" if v.get('tflops') else "") for t, v in list(r["kernels"].items())[:6])
                w(f"| {label} | {r['value']:.2f} | {r['ms_per_step']:.1f} | {r['frac_of_bf16_mfma_roofline']} | {top} |")
        b4 = mfma_busy("_4b")
        if b4:
            f4 = b4.pop("_file")
            w(f"\nMFMA pipe busy share by the PMC counters on this configuration's bf16 line (`{f4}`): "
              + ", ".join(f"`{k}` {v:.2f}" for k, v in b4.items()) + ".")
        w("\n`attn_d96_causal` = the head_dim 96 instantiation of the attention kernel (ABI 7); `rope_heads` = the rotary embedding in place on "
          "`qkv_proj`'s [q ∣ k ∣ v] columns (HBM-bound, not fused: DESIGN §5, docs/rounds/r06.md §2).")
    w("\nNot on this page because they are not on the scoring path's clock: `preprocess.hip` (device-side `load_video` resize: 0.44 ms per "
      "128 720p frames, docs/rounds/r04.md), `mxfp8.hip` (weights quantised once).")
    open(os.path.join(ROOT, "KERNELS.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
