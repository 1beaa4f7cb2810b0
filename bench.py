#!/usr/bin/env python3
"""Headline benchmark: video-pairs scored/sec, MJ-VIDEO-2B, 8 frames @448^2 (max_num=1), bf16.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Called WITHOUT a launcher and with --gpus N > 1 it starts the N ranks itself (a child `python -m torch.distributed.run`
on 127.0.0.1, started before this process touches a GPU; the parent only waits and passes rank 0's JSON line through).

One "step" = one pass of the reward-scoring hot path over one batch of synthetic pairs, inputs already resident in
HBM, random-init weights of the exact MJ-VIDEO-2B architecture.  --gpus 1: BASELINE.json configs[1] (4 pairs = 8 videos
x 8 tiles @448^2, N = 2186 tokens per video).  --gpus N > 1: BASELINE.json configs[2]'s shard size, 8 pairs per GPU per
step (64 pairs at 8 GPUs), weak scaling; the step goes through mj_video_amd.parallel.score_pairs_dp, whose only
collective is one RCCL all-gather of the [pairs, 2, 34] fp32 score block.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     dominant kernel's algorithmic TFLOP/s (HIP events on the launch stream, recorded inside the
                  timed region by the library's opt-in profiler) against the 2.5 PFLOP/s dense bf16 MFMA peak,
  "cpu_baseline": the oracle (CPU restatement of the reference forward, bf16, LM head included as the
                  reference executes it) timed on this host: 1 warm-up + 2 timed forwards of ONE video, for the C2
                  (headline) and the C1 (224^2) shapes, CPU model / physical cores / threads stated (N=1 only),
  "latency":      one video per forward (the reference's real call pattern, eval_genai_mjvideo.py:140-141), ms,
  "secondary":    (N=1 default run only) the BASELINE.json configs this command line is not quoted on, measured in the same
                  process AFTER the headline's timed region and never mixed into `value`: `fp8_ffn` (configs[4]'s weight path
                  on the 2B stand-in), `pairs8` (configs[2]'s per-GPU shard: the weak-scaling baseline at N = 1),
                  `c4_112_tiles` (configs[3]).  A failing leg reports its error and leaves the headline alone.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mj_video_amd  # noqa: E402,F401
from mj_video_amd import configuration as C, ops, synth  # noqa: E402
from mj_video_amd.chat_input import num_image_tokens_per_tile  # noqa: E402
from mj_video_amd.modeling import InternVLChatRewardModeling  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (block-scaled K = 128 MFMA), same table
HBM_PEAK_GBS = 8000.0
ALGO_TFLOP_PER_PAIR = 25.8       # 12.9 TFLOP per video at C2 (causal-halved, LM head skipped)


def algorithmic_tflop_per_video(cfg, n_tiles, n_tokens, image_size):
    """SURVEY.md §8(d) / Appendix A as a function of the config: every contraction of the scoring path for ONE video (causal
    attention halved, LM head skipped).  MJ-VIDEO-2B, 8 tiles @448^2, N = 2181: 12.9 TFLOP (the survey's figure, which the
    headline keeps quoting as ALGO_TFLOP_PER_PAIR); used for the configs the survey gives no figure for (the 4B backbone)."""
    v, l = cfg.vision_config, cfg.llm_config
    d, ff, P = v.hidden_size, v.intermediate_size, v.patch_size
    G = image_size // P
    T = G * G + 1
    vit = n_tiles * (2.0 * G * G * 3 * P * P * d + v.num_hidden_layers * (2.0 * T * (4 * d * d + 2 * d * ff) + 4.0 * T * T * d))
    h = l.hidden_size
    ntok = n_tiles * (G // 2) ** 2
    proj = 2.0 * ntok * (4 * d * h + h * h)
    hd = h // l.num_attention_heads
    qkv_out = (l.num_attention_heads + 2 * l.num_key_value_heads) * hd
    per_tok = 2.0 * (h * qkv_out + l.num_attention_heads * hd * h + 3 * h * l.intermediate_size)
    llm = l.num_hidden_layers * (per_tok * n_tokens + 2.0 * l.num_attention_heads * hd * float(n_tokens) ** 2)
    return (vit + proj + llm) / 1e12


def executed_tflop_per_video(cfg, n_tokens, prefix_tokens, trim_queries):
    """What the language tower of a forward really EXECUTES per video, relative to the algorithmic figure:
    returns the TFLOP left out.  Always left out (since round 1, sanctioned by SURVEY.md §7.5): wo / FFN of the last decoder layer
    on every row but the two the heads read.  ``prefix_tokens`` P > 0 (model.prefix_cache): P rows leave every Linear of every
    layer.  ``trim_queries`` (model.trim_last_layer): the last layer projects k | v only (half of wqkv) and runs no attention
    beyond a handful of queries."""
    l = cfg.llm_config
    h, ff = l.hidden_size, l.intermediate_size
    hd = h // l.num_attention_heads
    qkv_out = (l.num_attention_heads + 2 * l.num_key_value_heads) * hd
    kv_out = 2 * l.num_key_value_heads * hd
    per_tok = 2.0 * (h * qkv_out + l.num_attention_heads * hd * h + 3 * h * ff)
    attn_layer = 2.0 * l.num_attention_heads * hd * float(n_tokens) ** 2
    full = l.num_hidden_layers * (per_tok * n_tokens + attn_layer)
    rows = n_tokens - prefix_tokens
    done = (l.num_hidden_layers - 1) * (per_tok * rows + attn_layer)
    done += 2.0 * h * (kv_out if trim_queries else qkv_out) * rows + (0.0 if trim_queries else attn_layer)
    return (full - done) / 1e12


def random_init_on_device(model, config, device, seed):
    """Random-init weights of the architecture, generated directly in HBM (same distributions as synth)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ls0 = float(config.vision_config.initializer_factor)
    params = dict(model.named_parameters())
    for key, shape, kind in synth.state_dict_spec(config):
        p = params[key]
        assert tuple(p.shape) == tuple(shape), key
        if kind in ("w", "b"):
            p.normal_(0.0, 0.02, generator=g)
        elif kind == "g":
            p.normal_(1.0, 0.05, generator=g)
        elif kind == "ls":
            p.normal_(ls0, 0.05 * ls0, generator=g)
        elif kind == "emb":
            p.normal_(0.0, 1.0, generator=g)
        elif kind in ("head", "gate"):
            p.normal_(0.0, 0.05, generator=g)
        elif kind == "eye":
            p.copy_(torch.eye(shape[0], device=device))
        elif kind == "one":
            p.fill_(1.0)
        elif kind == "lmhead":
            p.zero_()


def host_cpu_info():
    """(model name, physical cores, logical cpus) of this host from /proc/cpuinfo."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("processor"):
                logical += 1
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or (os.cpu_count() or 1)), (logical or (os.cpu_count() or 1))


def cpu_baseline(image_size, n_tiles, threads, iters=2):
    """Oracle forward of one video of the workload on the host CPU (checker code, timed only as a baseline):
    1 warm-up forward, then ``iters`` timed ones."""
    from oracle import ref_cpu
    torch.set_num_threads(threads)
    config = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(image_size), **C.mjvideo_head_kwargs())
    sd = synth.synth_state_dict(config, seed=0)
    px = synth.synth_pixel_values(300, 0, n_tiles, image_size)
    ids = synth.synth_input_ids(num_image_tokens_per_tile(config) * n_tiles, 0)
    times = []
    for it in range(1 + iters):
        t0 = time.time()
        ref_cpu.reward_forward(sd, config, px, ids, torch.ones_like(ids), synth.IMG_CONTEXT_ID, synth.PAD_ID, lm_head=True)
        if it:
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return 0.5 / dt, dt, int(ids.shape[1])


METRIC = "video-pairs scored/sec, MJ-VIDEO-2B 8-frame bf16, 1/2/4/8 MI355X"   # BASELINE.json "metric", verbatim

# SURVEY.md Appendix A, per video at C2: the five FFN Linears = fc1 + fc2 (2 x 1650.9 GF) + w1|w3 (3512.7) + w2 (1756.4)
FFN_TFLOP_PER_PAIR = 2 * (2 * 1650.9 + 3512.7 + 1756.4) / 1e3          # 17.14 of the 25.8 TFLOP per pair
ALGO_TFLOP_PER_PAIR_C4 = 2 * 250.6                                      # 250.6 TFLOP per 112-tile video


def synthetic_batch(cfg, dev, pairs, S, F, seed):
    """``pairs`` pairs of synthetic videos of ``F`` tiles @``S``^2, resident in HBM: (pixel_values, input_ids, mask, N)"""
    per_tile = num_image_tokens_per_tile(cfg)
    g = torch.Generator(device=dev).manual_seed(seed)
    px = torch.randn(2 * pairs * F, 3, S, S, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    rows = []
    for p in range(pairs):
        row = synth.synth_input_ids(per_tile * F, caption_seed=seed * 1000 + p)
        rows += [row, row]
    ids, mask = synth.pad_batch(rows)
    return px, ids.to(dev), mask.to(dev), int(ids.shape[1])


def time_forwards(model, px, ids, mask, pairs, steps, warmup, profile=False):
    """``warmup`` untimed + ``steps`` timed forwards of one batch (fresh id / mask tensors each, as the headline does).  With
    ``profile`` the LAST warm-up forward carries HIP events around every kernel launch (its per-kernel table is returned)."""
    def one():
        model.forward(px, ids.clone(), mask.clone())
        return model.last_packed34
    res = None
    for w in range(warmup):
        if profile and w == warmup - 1:
            torch.cuda.synchronize()
            ops.prof_filter(None); ops.prof_reset(); ops.prof_enable(True)
            one()
            torch.cuda.synchronize()
            ops.prof_enable(False)
            res = ops.prof_results()
            ops.prof_reset()
        else:
            one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not torch.isfinite(out).all():
        raise RuntimeError("non-finite scores")
    return {"value": round(pairs * steps / dt, 4), "unit": "pairs/s", "ms_per_step": round(1e3 * dt / steps, 3),
            "steps": steps, "warmup": warmup, "pairs_per_step": pairs}, res


def secondary_legs(model, cfg, dev, px, ids, mask):
    """The configs the default command line is not quoted on (module docstring "secondary"), each in its own try block."""
    sec = {}

    def leg(name, fn):
        t0 = time.perf_counter()
        try:
            sec[name] = fn()
        except Exception as e:   # a secondary leg never costs the headline
            sec[name] = {"value": None, "error": repr(e)}
        sec[name]["leg_wall_s"] = round(time.perf_counter() - t0, 2)

    def fp8_ffn(fmt="mxfp8"):
        model.set_ffn_format(fmt)
        try:
            r, res = time_forwards(model, px, ids, mask, pairs=4, steps=10, warmup=3, profile=True)
        finally:
            model.set_ffn_format("bf16")
        f8 = {k: v for k, v in res.items() if k.startswith("gemm256f8") and v["ms"] > 0}
        name, k = max(f8.items(), key=lambda kv: kv[1]["ms"])
        tfl = k["flops"] / (k["ms"] * 1e-3) / 1e12
        ceiling = 1.0 / (FFN_TFLOP_PER_PAIR / MFMA_FP8_PEAK_TFLOPS + (ALGO_TFLOP_PER_PAIR - FFN_TFLOP_PER_PAIR) / MFMA_BF16_PEAK_TFLOPS)
        covered = sorted(model_fp8_linears(fmt))
        ffn_tf = 2 * sum({"fc1": 1650.9, "fc2": 1650.9, "w13": 3512.7, "w2": 1756.4}[n] for n in covered) / 1e3
        ceiling = 1.0 / (ffn_tf / MFMA_FP8_PEAK_TFLOPS + (ALGO_TFLOP_PER_PAIR - ffn_tf) / MFMA_BF16_PEAK_TFLOPS)
        r.update({
            "baseline_config": "configs[4]'s fp8 MFMA weight path on the 2B model (secondary.c5_4b runs it on the 4B backbone); same batch as the headline",
            "preset": fmt, "fp8_linears": covered,
            "rank_agreement_vs_reference_bf16": ({"spearman_224": 0.99863, "spearman_448": 0.99820, "decisive_flips": "0 / 478, 0 / 240",
                                                  "tolerance": "its own, looser than north_star's 0.999 (tests/test_fp8_gpu.py)"} if fmt == "mxfp8" else
                                                 {"spearman_224": 0.99920, "spearman_448": 0.99912, "decisive_flips": "0 / 478, 0 / 240",
                                                  "tolerance": "north_star's: rho >= 0.999 and 0 flips (tests/test_fp8_gpu.py)"}),
            "rank_agreement_source": "profiles/r06_a_fp8_ffn_subset_study.txt (engineered rank sets, all 15 subsets of the four Linears)",
            "dtype": f"fp8-e4m3 (MXFP8, block-32 e8m0 scales) operands + fp32 accumulate in the FFN GEMMs {covered}, bf16 elsewhere",
            "roofline": {"kernel": name, "bound": "mfma", "achieved": round(tfl, 2), "peak": MFMA_FP8_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tfl / MFMA_FP8_PEAK_TFLOPS, 4), "launches": k["launches"],
                         "avg_launch_ms": round(k["ms"] / max(k["launches"], 1), 4)},
            "mixed_roofline": {"ceiling_pairs_per_s": round(ceiling, 2), "frac": round(r["value"] / ceiling, 4),
                               "note": f"{ffn_tf:.2f} TFLOP of fp8 Linears per pair at the 5 PFLOP/s fp8 peak + "
                                       f"{ALGO_TFLOP_PER_PAIR - ffn_tf:.2f} TFLOP at the 2.5 PFLOP/s bf16 peak"},
            "fp8_kernels": {n: {"ms_per_step": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                            for n, v in sorted(f8.items(), key=lambda kv: -kv[1]["ms"])}})
        return r

    def pairs8():
        px8, ids8, mask8, n8 = synthetic_batch(cfg, dev, 8, 448, 8, seed=7)
        r, _ = time_forwards(model, px8, ids8, mask8, pairs=8, steps=5, warmup=1)
        r.update({"baseline_config": "configs[2]'s per-GPU shard (8 pairs = 16 videos per GPU per step) on ONE GPU: the N = 1 point "
                                     "a weak-scaling efficiency of `--gpus N` (8 pairs per GPU) divides by",
                  "dtype": "bf16", "N": n8, "frac_of_mfma_roofline": round(r["value"] * ALGO_TFLOP_PER_PAIR / MFMA_BF16_PEAK_TFLOPS, 4)})
        return r

    def c4():
        px4, ids4, mask4, n4 = synthetic_batch(cfg, dev, 1, 448, 112, seed=9)
        r, _ = time_forwards(model, px4, ids4, mask4, pairs=1, steps=2, warmup=1)
        ceiling = MFMA_BF16_PEAK_TFLOPS / ALGO_TFLOP_PER_PAIR_C4
        r.update({"baseline_config": "configs[3]: 16 frames x 7 dynamic tiles = 112 tiles per video (long-context image tokens), "
                                     "one pair = 2 videos per step",
                  "dtype": "bf16", "N": n4, "roofline_pairs_per_s": round(ceiling, 3),
                  "frac_of_mfma_roofline": round(r["value"] / ceiling, 4)})
        return r

    def prefix_cache_off():
        keep = (model.prefix_cache, model.trim_last_layer)
        model.prefix_cache = model.trim_last_layer = False
        try:
            r, _ = time_forwards(model, px, ids, mask, pairs=4, steps=10, warmup=2)
        finally:
            model.prefix_cache, model.trim_last_layer = keep
        r.update({"baseline_config": "configs[1], the headline's batch, with model.prefix_cache = model.trim_last_layer = False: every "
                                     "prompt row through all 24 decoder layers in every forward, every query in the last layer",
                  "dtype": "bf16", "frac_of_mfma_roofline": round(r["value"] * ALGO_TFLOP_PER_PAIR / MFMA_BF16_PEAK_TFLOPS, 4)})
        return r

    def c5_4b():
        """BASELINE configs[4]: the InternVL2-4B backbone (InternViT + Phi-3-mini: hidden 3072, 32 heads x 96, ff 8192, 32 layers) +
        the 28-criteria MoE heads, random-init weights, the headline's batch shape; bf16, then the fp8 weight path (both presets)."""
        cfg4 = C.InternVLChatRewardModelingConfig(**C.internvl2_4b_config_dict(448), **C.mjvideo_head_kwargs())
        m4 = InternVLChatRewardModeling.from_config(cfg4, dtype=torch.bfloat16, device=dev)
        random_init_on_device(m4, cfg4, dev, seed=4321)
        tk = synth.PHI3_TOKENS
        m4.config.pad_token_id = tk.pad
        m4.model.img_context_token_id = tk.img_context
        m4.eval()
        per_tile = num_image_tokens_per_tile(cfg4)
        rows = []
        for p_ in range(4):
            row = synth.synth_input_ids(per_tile * 8, caption_seed=5000 + p_, tokens=tk)
            rows += [row, row]
        ids4, mask4 = synth.pad_batch(rows, pad_id=tk.pad)
        ids4, mask4 = ids4.to(dev), mask4.to(dev)
        n4 = int(ids4.shape[1])
        algo = 2 * algorithmic_tflop_per_video(cfg4, 8, n4, 448)
        out = {"baseline_config": "configs[4]: InternVL2-4B backbone (InternViT-300M + Phi-3-mini) + 28-criteria MoE heads, 4 pairs of 8 frames "
                                  "@448^2 on ONE MI355X (the 8-GPU form is this shard under score_pairs_dp)",
               "N": n4, "parameters_G": round(sum(p_.numel() for p_ in m4.parameters()) / 1e9, 3),
               "algorithmic_tflop_per_pair": round(algo, 2), "roofline_pairs_per_s_bf16": round(MFMA_BF16_PEAK_TFLOPS / algo, 2),
               "oracle": "oracle/ref_phi3.py, pinned to transformers 5.15's Phi3ForCausalLM inside the reference's reward model "
                         "(tests/golden/make_golden_phi3.py); no reference implementation of a 4B MJ-VIDEO exists (README.md:18)"}
        try:
            for name, fmt in (("bf16", "bf16"), ("fp8_ffn", "mxfp8"), ("fp8_rank999", "mxfp8-rank999")):
                m4.set_ffn_format(fmt)
                r, res = time_forwards(m4, px, ids4, mask4, pairs=4, steps=5, warmup=2, profile=True)
                # (on EXECUTED flops, like the headline: the prefix cache and the last-layer trimming are on for this tower too)
                left = 2 * executed_tflop_per_video(cfg4, n4, (m4._prefix or {}).get("P", 0) if m4.prefix_cache else 0, bool(m4.trim_last_layer))
                r["executed_tflop_per_pair"] = round(algo - left, 2)
                r["frac_of_bf16_mfma_roofline"] = round(r["value"] * (algo - left) / MFMA_BF16_PEAK_TFLOPS, 4)
                top = sorted(res.items(), key=lambda kv: -kv[1]["ms"])[:8]
                r["kernels"] = {k: {"ms_per_step": round(v["ms"], 3),
                                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] else None}
                                for k, v in top}
                out[name] = r
            out["value"] = out["bf16"]["value"]
        finally:
            del m4
            torch.cuda.empty_cache()
        return out

    if model.prefix_cache or model.trim_last_layer:
        leg("prefix_cache_off", prefix_cache_off)
    leg("fp8_ffn", fp8_ffn)
    leg("fp8_rank999", lambda: fp8_ffn("mxfp8-rank999"))
    leg("pairs8", pairs8)
    leg("c4_112_tiles", c4)
    leg("c5_4b", c5_4b)
    torch.cuda.empty_cache()
    return sec


def model_fp8_linears(fmt):
    from mj_video_amd.modeling import FFN_LINEARS, FP8_PRESETS
    return FP8_PRESETS.get(fmt, FFN_LINEARS)


def dp_step(score_local, global_pairs, device):
    """One step of the benchmark at any N: the batch's pairs are sharded over the ranks by ``parallel.score_pairs_dp`` (the
    product's one collective: an all-gather of the [pairs, 2, 34] fp32 block) and every rank ends up with the whole block in
    pair order.  ``score_local(local_pair_ids) -> [n_local, 2, 34]``.  Without a process group it is a plain call.
    (tests/test_dp_gloo.py runs this very function at world size 2 over gloo.)"""
    from mj_video_amd import parallel
    return parallel.score_pairs_dp(score_local, global_pairs, device=device)


def kernel_sources_sha1():
    """SHA-1 over the kernel sources: the PMC traffic file names the sources it was measured on (tools/pmc_summary.py writes
    the same hash), and a file measured on other sources is not quoted."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mj-video_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def main():
    # dmabuf IPC is the only kind this pool's host driver supports (RCCL init across processes fails without it); exported on the
    # GPU boxes already - set here too, before the first HIP call, for a launcher that starts the ranks with a scrubbed environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=None,
                    help="pairs per GPU per step; default 4 at --gpus 1 (BASELINE.json configs[1]: batch = 4 pairs on 1 MI355X) and "
                         "8 at --gpus N > 1 (configs[2]: batch = 64 pairs sharded DP over 8 MI355X = 8 pairs per GPU)")
    ap.add_argument("--image-size", type=int, default=448)
    ap.add_argument("--frames", type=int, default=8, help="tiles per video (frames x tiles per frame; configs[3]: 16 x 7 = 112)")
    ap.add_argument("--gemm-code", type=int, action="append", default=[],
                    help="A/B switches of the GEMM library (mjv_gemm_set_tile codes, e.g. 4000 = no split-K, 6000 = no skinny "
                         "kernel); not for reported numbers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-video latency section (profiling runs: keeps small-batch launches out of the "
                         "per-kernel statistics)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-prefix-cache", action="store_true",
                    help="recompute the constant prompt prefix's rows in every forward and every query of the last decoder layer, as "
                         "the reference does (model.prefix_cache = model.trim_last_layer = False); the default line states both "
                         "switches in `config` and carries this setting as `secondary.prefix_cache_off`")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` legs (fp8 FFN path, the 8-pair shard, the 112-tile config) of the default N = 1 run")
    ap.add_argument("--backbone", choices=["2b", "4b"], default="2b",
                    help="4b: BASELINE configs[4]'s backbone (InternVL2-4B = InternViT + Phi-3-mini) with the 28-criteria MoE heads instead of "
                         "MJ-VIDEO-2B - its own line (metric, config.baseline_config, its own algorithmic flops), never the headline; with "
                         "--gpus 8 it is configs[4]'s 8-GPU form (8 pairs per GPU, one RCCL all-gather); combine with --fp8 / --fp8-preset")
    ap.add_argument("--fp8-preset", default="mxfp8", help="with --fp8: 'mxfp8' (all four FFN Linears), 'mxfp8-rank999', 'mxfp8-vit' or 'mxfp8:<a>+<b>'")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE configs[4]'s weight path on the 2B stand-in: the five FFN Linears of both towers on MXFP8 (e4m3, "
                         "block-32 e8m0 scales) operands, fp32 accumulate (model.set_ffn_format('mxfp8')).  Its own line, its own "
                         "dtype and tolerance (DESIGN §7.4) - never the default, never the headline")
    args = ap.parse_args()
    if args.pairs is None:
        args.pairs = 4 if args.gpus == 1 else 8

    # MJV_BENCH_FORCE_LAUNCHER=1: take the N > 1 path (child torch.distributed.run, RCCL init, score_pairs_dp's all-gather,
    # ranks_seen) at --gpus 1 too - so a 1-GPU box can execute, and a -m gpu test can cover, the code an 8-GPU run goes through
    force_dist = bool(os.environ.get("MJV_BENCH_FORCE_LAUNCHER"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        # No launcher: start the N ranks as fresh children (one process per GPU, RCCL rendezvous on 127.0.0.1) BEFORE
        # this process has made any GPU call - it never does: it waits, passes the children's output (rank 0's JSON line)
        # through and exits with their code.  Never re-exec a process that has initialised the GPU.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs: the scoring path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (force_dist and "WORLD_SIZE" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    S, F = args.image_size, args.frames
    four_b = args.backbone == "4b"
    tk = synth.PHI3_TOKENS if four_b else synth.INTERNLM2_TOKENS
    cfg = C.InternVLChatRewardModelingConfig(**(C.internvl2_4b_config_dict(S) if four_b else C.mjvideo_2b_config_dict(S)), **C.mjvideo_head_kwargs())
    model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
    random_init_on_device(model, cfg, dev, seed=1234)
    model.config.pad_token_id = tk.pad
    model.model.img_context_token_id = tk.img_context
    model.eval()
    if args.fp8:
        model.set_ffn_format(args.fp8_preset)
    if args.no_prefix_cache:
        model.prefix_cache = model.trim_last_layer = False
    if os.environ.get("MJV_BENCH_VIT_CHUNK"):     # A/B (round 6): the vision tower over this many tiles at a time
        model.vit_chunk_tiles = int(os.environ["MJV_BENCH_VIT_CHUNK"])
    if os.environ.get("MJV_BENCH_NORM_FUSION"):   # A/B of DESIGN "Norm fusion, round 4" (never a reported headline: the line says so)
        model.norm_fusion = True
    for code in args.gemm_code:   # (bench build of the library only: MJV_LIBRARY=.../libmjv_hip_bench.so)
        ops.gemm_set_tile(code)

    n_videos = 2 * args.pairs
    per_tile = num_image_tokens_per_tile(cfg)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    px = torch.randn(n_videos * F, 3, S, S, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    ids_list = []
    for p in range(args.pairs):
        row = synth.synth_input_ids(per_tile * F, caption_seed=rank * 1000 + p, tokens=tk)
        ids_list += [row, row]  # both videos of a pair share the caption
    ids, mask = synth.pad_batch(ids_list, pad_id=tk.pad)
    ids, mask = ids.to(dev), mask.to(dev)
    seq_len = int(ids.shape[1])
    global_pairs = list(range(world * args.pairs))   # pair ids of the whole batch; rank r owns a contiguous block of them

    def score_local(local_pairs):
        # this rank's shard (its own synthetic inputs, resident in HBM).  Fresh id / mask tensors every step, as every real
        # batch brings: the forward pays its device->host copy of the ids (the model caches the host copy only for the
        # SAME unmodified tensor objects)
        assert len(local_pairs) == args.pairs
        model.forward(px, ids.clone(), mask.clone())
        return model.last_packed34.view(args.pairs, 2, 34)

    def step():
        return dp_step(score_local, global_pairs, dev)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    prof = not args.no_prof
    for _ in range(max(args.warmup - (1 if prof else 0), 0)):
        step()
    fence()
    res_all = None
    if prof:
        # Last warm-up step (untimed): HIP events around EVERY kernel launch give the per-kernel table and name the
        # dominant kernel.  The timed steps then carry events around that kernel's launches only - an event pair per
        # launch on every kernel costs about 1.4 % of the step (the marker packets serialise the queue).
        ops.prof_filter(None)
        ops.prof_reset()
        ops.prof_enable(True)
        step()
        fence()
        ops.prof_enable(False)
        res_all = ops.prof_results()
        dominant = max(res_all.items(), key=lambda kv: kv[1]["ms"])[0]
        ops.prof_reset()
        ops.prof_filter(dominant)
        ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if prof:
        ops.prof_enable(False)
        ops.prof_filter(None)
    if not torch.isfinite(out).all():
        raise SystemExit("non-finite scores")
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ranks_seen = 1
    rank_ms = [1e3 * elapsed / args.steps]
    if use_dist:
        # every rank's own clock over the timed region (a straggler shows as max >> min), then the contract's MAX over ranks
        every = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, t)
        rank_ms = [1e3 * float(x) / args.steps for x in every.tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        who = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(who, torch.tensor([rank], dtype=torch.int64, device=dev))   # RCCL all-gather
        ranks_seen = int(torch.unique(who).numel())
    elapsed = float(t.item())

    if rank == 0:
        total_pairs = args.pairs * world * args.steps
        value = total_pairs / elapsed
        algo_pair = 2 * algorithmic_tflop_per_video(cfg, F, seq_len, S) if four_b else ALGO_TFLOP_PER_PAIR
        line = {
            "metric": (METRIC if not four_b else "video-pairs scored/sec, InternVL2-4B backbone + 28-criteria MoE heads "
                                                 "(BASELINE configs[4]; NOT the headline metric), 8-frame, 1/2/4/8 MI355X"),
            "value": round(value, 4), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": (f"fp8-e4m3 (MXFP8, block-32 e8m0 scales) operands + fp32 accumulate in the FFN GEMMs (preset {args.fp8_preset}), bf16 elsewhere"
                      if args.fp8 else "bf16"),
            "data": "synthetic", "ranks_seen": ranks_seen,
            "ms_per_step_by_rank": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3),
                                    "all": [round(x, 3) for x in rank_ms]},
            "norm_fusion": bool(model.norm_fusion), "vit_chunk_tiles": getattr(model, "vit_chunk_tiles", None),
            "attention_scores": model.attention_scores,
            "process_group": ("nccl" if use_dist else None),
            "config": {"workload": (("InternVL2-4B backbone (InternViT + Phi-3-mini) + 28-criteria MoE heads" if four_b else "MJ-VIDEO-2B")
                                    + f", batch={args.pairs * world} pairs"
                                    + (f" sharded DP over {world} MI355X ({args.pairs} pairs per GPU), RCCL all-gather rewards"
                                       if world > 1 else " on 1 MI355X")
                                    + f", {F} frames @{S}^2 max_num=1, N={seq_len} tokens/video, random-init weights, inputs "
                                      "resident in HBM"),
                       "baseline_config": (("configs[4]: InternVL2-4B backbone" + (", fp8 MFMA weight path" if args.fp8 else ", bf16")
                                            + (", 8 MI355X" if world == 8 else f", {world} MI355X") + "; NOT the headline") if four_b else
                                           "configs[4] weight path on the 2B model: FFN GEMMs only; NOT the headline"
                                           if args.fp8 else "configs[1]" if (world, args.pairs, S, F) == (1, 4, 448, 8) else
                                           "configs[2] shard size (8 pairs per GPU; 64 pairs at 8 GPUs)"
                                           if (args.pairs, S, F) == (8, 448, 8) and world > 1 else "other"),
                       "pairs_per_gpu_per_step": args.pairs, "global_pairs_per_step": args.pairs * world,
                       "scaling_note": ("weak scaling at a fixed per-GPU batch; the default per-GPU batch is 4 pairs at --gpus 1 "
                                        "(configs[1]) and 8 at --gpus N > 1 (configs[2]): an efficiency against N = 1 needs "
                                        "`--gpus 1 --pairs 8` as its baseline"),
                       "ids": "fresh id / mask tensors every step (one device->host copy of the ids per forward)",
                       # steady-state work removal a scorer is entitled to, STATED: the keys / values of the
                       # constant prompt prefix (a multiple of 64 tokens: system prompt + "Frame1: <img>") come from a cache filled by
                       # the first forward, and the last decoder layer computes queries only from the first selected row on.
                       # secondary.prefix_cache_off is the same workload with both off (every row, every forward)
                       "prefix_cache_note": ("steady-state work removal, stated: the keys / values of the prompt's constant first tokens "
                                             "(system turn + 'Frame1: <img>', cut to a multiple of 64: prefix_cache_tokens) come from a per-model cache filled once by a pass over those "
                                             "tokens alone, and the last decoder layer computes queries only from the first row the heads "
                                             "read; outputs equal the full computation up to fp32 re-association (bit for bit when no GEMM "
                                             "slices K); secondary.prefix_cache_off = both switches off, every row in every forward"),
                       "prefix_cache": bool(model.prefix_cache), "prefix_cache_tokens": (model._prefix or {}).get("P", 0),
                       "prefix_cache_hits": model.prefix_cache_hits, "trim_last_layer": bool(model.trim_last_layer),
                       "parallelism": f"dp{world} (replicated weights, one all-gather of [pairs,2,34] fp32 per step)"},
        }
        if (S, F) == (448, 8) and not args.fp8:
            # the roofline fraction is quoted on EXECUTED flops (VERDICT r5 item 9 / ADVICE r5): the algorithmic 25.8 TFLOP per
            # pair minus what this run's switches leave out of the language tower
            left_out = 2 * executed_tflop_per_video(cfg, seq_len, (model._prefix or {}).get("P", 0) if model.prefix_cache else 0,
                                                    bool(model.trim_last_layer))
            line["executed_tflop_per_pair"] = round(algo_pair - left_out, 3)
            line["frac_of_mfma_roofline"] = round(value * (algo_pair - left_out) / (MFMA_BF16_PEAK_TFLOPS * world), 4)
            line["frac_of_mfma_roofline_note"] = (
                f"value x executed TFLOP per pair / 2.5 PFLOP/s: {algo_pair:.2f} algorithmic (SURVEY.md §8(d)) minus {left_out:.3f} "
                "this run leaves out of the language tower (cached prompt-prefix rows in every layer; last decoder layer: q projection + "
                "attention beyond the rows the heads read, wo / FFN on every other row).  `value_all_work` = the same batch with "
                "prefix_cache / trim_last_layer off (every row in every forward), `frac_all_work` its fraction on ITS executed flops")
            line["pairs_per_s_roofline"] = round(MFMA_BF16_PEAK_TFLOPS * world / algo_pair, 2)
            line["value_all_work"] = None
        else:
            line["frac_of_mfma_roofline"] = None
        from mj_video_amd import _lib
        lib_path = os.environ.get("MJV_LIBRARY") or _lib.LIB_PATH
        line["library"] = {"path": os.path.relpath(lib_path, ROOT), "bench_build": hasattr(_lib.load_library(), "mjv_bench_gemm_set")
                           and getattr(_lib.load_library().mjv_bench_gemm_set, "argtypes", None) is not None}
        if prof:
            res = ops.prof_results()

            traffic, traffic_note = {}, None
            # (the 2B workload's passes are profiles/rNN_pmc_traffic.json; the 4B backbone's launch shapes have their own file,
            # profiles/rNN_4b_pmc_traffic.json - same kernel names, other shapes)
            tpaths = sorted(p_ for p_ in os.listdir(os.path.join(ROOT, "profiles"))
                            if p_.endswith("_pmc_traffic.json") and p_.endswith("_4b_pmc_traffic.json") == four_b)
            if four_b and (args.fp8 or not tpaths):   # (the 4B passes ran the bf16 line: the fp8 line's bf16 GEMMs take other inputs)
                traffic_note = "not collected for this line's launch shapes (profiles/*_pmc_traffic.json: the 2B workload; *_4b_*: the 4B backbone in bf16)"
            elif (S, F, args.pairs) == (448, 8, 4) and tpaths:   # the latest round's committed PMC passes of this workload
                tj = json.load(open(os.path.join(ROOT, "profiles", tpaths[-1])))
                if tj.get("source_sha1") == kernel_sources_sha1():
                    traffic = tj.get("per_launch_bytes", {})
                else:   # kernels changed since the counters were collected: a stale figure is worse than none
                    traffic_note = (f"profiles/{tpaths[-1]} was measured on other kernel sources (source_sha1 differs): "
                                    "not quoted; re-run tools/collect_profiles.sh")

            def roofline(res, steps, share_from=None):
                share_from = share_from or res
                tot = sum(r["ms"] for r in share_from.values())
                name, r = max(res.items(), key=lambda kv: kv[1]["ms"])
                tfl = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
                peak = MFMA_FP8_PEAK_TFLOPS if name.startswith("gemm256f8") else MFMA_BF16_PEAK_TFLOPS
                return {"kernel": name, "bound": "mfma", "achieved": round(tfl, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(tfl / peak, 4),
                        # HBM/fabric bytes per launch from the committed rocprofv3 PMC passes of this same workload
                        # (profiles/rNN_pmc_traffic.json; a PMC run cannot be nested inside this process)
                        "traffic": traffic.get(name), **({"traffic_note": traffic_note} if traffic_note else {}),
                        "algorithmic_bytes_per_launch": round(r["bytes"] / max(r["launches"], 1)),
                        "launches": r["launches"], "avg_launch_ms": round(r["ms"] / max(r["launches"], 1), 4),
                        "share_of_kernel_time": round(share_from[name]["ms"] / tot, 4) if tot else None}

            def table(res, steps):
                return {k: {"ms_per_step": round(v["ms"] / steps, 3),
                            "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] else None,
                            "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] and v["ms"] else None}
                        for k, v in sorted(res.items(), key=lambda kv: -kv[1]["ms"])}

            line["roofline"] = roofline(res, args.steps, share_from=res_all)
            line["roofline"]["note"] = "events around this kernel's launches on the launch stream inside the timed region"
            line["kernels"] = table(res_all, 1)
        if world == 1 and not args.no_latency and not args.fp8 and not four_b:
            # the reference's real call pattern: ONE video per forward (eval_genai_mjvideo.py:140-141), fresh ids each time
            px1, ids1, mask1 = px[:F].contiguous(), ids[:1].contiguous(), mask[:1].contiguous()
            for _ in range(2):
                model.forward(px1, ids1.clone(), mask1.clone())
            torch.cuda.synchronize()
            n_lat = 10
            t1 = time.perf_counter()
            for _ in range(n_lat):
                model.forward(px1, ids1.clone(), mask1.clone())
            torch.cuda.synchronize()
            lat_ms = 1e3 * (time.perf_counter() - t1) / n_lat
            t1 = time.perf_counter()
            model.forward(px1, ids1.clone(), mask1.clone())
            host_ms = 1e3 * (time.perf_counter() - t1)   # enqueue time of one forward (the GPU is still running it)
            torch.cuda.synchronize()
            line["latency"] = {"one_video_per_forward_ms": round(lat_ms, 3), "host_enqueue_ms": round(host_ms, 3),
                               "pairs_per_s_at_batch_1_video": round(0.5e3 / lat_ms, 3),
                               "note": "back-to-back single-video forwards, not part of `value`"}
        if (world == 1 and not use_dist and not args.no_secondary and not args.fp8 and not four_b and (S, F, args.pairs) == (448, 8, 4)
                and not os.environ.get("MJV_BENCH_NORM_FUSION") and not os.environ.get("MJV_BENCH_VIT_CHUNK") and not args.gemm_code):
            line["secondary"] = secondary_legs(model, cfg, dev, px, ids, mask)
            off = line["secondary"].get("prefix_cache_off")
            if off and off.get("value"):
                lo = 2 * executed_tflop_per_video(cfg, seq_len, 0, False)
                line["value_all_work"] = off["value"]
                line["frac_all_work"] = round(off["value"] * (ALGO_TFLOP_PER_PAIR - lo) / MFMA_BF16_PEAK_TFLOPS, 4)
                off["frac_of_mfma_roofline"] = line["frac_all_work"]
        if world == 1 and (args.fp8 or four_b):
            line["cpu_baseline"] = {"value": None, "note": "timed on the headline line only (default `python bench.py`: the oracle's bf16 forward of "
                                                            "MJ-VIDEO-2B on the host cores); this line is a secondary configuration"}
        if world == 1 and not args.no_cpu_baseline and not args.fp8 and not four_b:
            # oneDNN bf16 GEMMs stop scaling (and oversubscribe NUMA domains) far below a 256-thread host: cap at 32
            cpu_model, phys, logical = host_cpu_info()
            threads = min(phys, 32)
            try:
                v, dt, n2 = cpu_baseline(S, F, threads)
                line["cpu_baseline"] = {"value": round(v, 5), "unit": "pairs/s", "cores": threads, "kind": "port",
                                        "sample": f"1 video ({F} tiles @{S}^2, N={n2}) = half a pair per oracle forward in bf16 "
                                                  f"incl. the reference's unused LM-head GEMM; 1 warm-up + 2 timed forwards, "
                                                  f"{dt:.1f}s each",
                                        "cpu_model": cpu_model, "physical_cores": phys, "logical_cpus": logical,
                                        "threads": threads}
                if (S, F) == (448, 8):
                    v1, dt1, n1 = cpu_baseline(224, 8, threads)
                    line["cpu_baseline"]["c1_224"] = {"value": round(v1, 5), "unit": "pairs/s",
                                                      "sample": f"1 video (8 tiles @224^2, N={n1}), 1 warm-up + 2 timed, {dt1:.1f}s each"}
            except Exception as e:  # the baseline is informational; never lose the GPU number over it
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
