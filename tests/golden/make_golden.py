#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the REFERENCE ITSELF.

Build-container only (needs /root/reference; never runs on the GPU box).  For every case it
  1. builds seed-defined weights / pixels / token ids with mj_video_amd.synth (so the GPU box can
     regenerate the identical inputs from the seeds stored in the fixture),
  2. runs the reference's own ``InternVLChatRewardModeling.forward`` (imported unmodified through
     oracle/reference_shim.py) on CPU in bf16,
  3. runs oracle/ref_cpu.py on the same inputs and REQUIRES bit-identical outputs (this is what
     pins the oracle),
  4. runs the oracle once more in fp32 (same bf16-representable weights) to record the
     reference's own bf16 noise floor, which calibrates the tolerances written in the tests,
  5. stores outputs (+ intermediate probes) as float32 arrays (every bf16 value is exact in f32).

Usage:  python tests/golden/make_golden.py tiny | full_c1 | full_c2 | rankset_c1 [--pairs P] | rankset_c2
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import mj_video_amd  # noqa: E402
from mj_video_amd import configuration as C, synth  # noqa: E402
from oracle import ref_cpu, reference_shim as RS  # noqa: E402

FIELDS = ref_cpu.OUTPUT_FIELDS
PROBE_VIT_LAYERS, PROBE_VIT_ROWS = (0, 5, 11, 17, 23), 48     # first tile: CLS + 47 patch rows, all 1024 columns
PROBE_LLM_LAYERS, PROBE_LLM_ROWS = (0, 7, 15, 23), 32         # the last 32 token rows (caption + pattern), all 2048 columns


def _bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> its uint16 bit patterns."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def make_cfg(kind: str, image_size: int, vit_image_size=None):
    cd = C.tiny_config_dict(image_size) if kind == "tiny" else C.mjvideo_2b_config_dict(image_size)
    if vit_image_size is not None:
        cd["vision_config"]["image_size"] = vit_image_size
    hk = C.mjvideo_head_kwargs()
    return cd, hk, C.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **copy.deepcopy(hk))


def run_reference(model, px, ids, mask):
    with torch.no_grad():
        out = model.forward(px, ids, mask)
    return {k: getattr(out, k) for k in FIELDS}


def to_np(d):
    return {k: v.detach().float().numpy() for k, v in d.items()}


def check_equal(a, b, what):
    for k in FIELDS:
        if not (a[k].dtype == b[k].dtype and torch.equal(a[k], b[k])):
            raise SystemExit(f"ORACLE != REFERENCE for {what}:{k} max|d|="
                             f"{(a[k].float() - b[k].float()).abs().max().item()}")


def n_img_tokens(cfg, n_tiles):
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    return num_image_tokens_per_tile(cfg) * n_tiles


def gen_tiny():
    cases = []
    arrays = {}
    specs = [
        # name, image_size, vit_image_size, weight_seed, [(video_idx, n_tiles, caption_seed, interleave)], pad_to
        ("single", 56, None, 11, [(0, 4, 1, None)], None),
        ("posinterp", 56, 112, 12, [(1, 3, 2, None)], None),
        ("batch2pad", 56, None, 13, [(2, 4, 3, None), (3, 2, 4, None)], None),
        ("interleave", 56, None, 14, [(4, 4, 5, 4)], None),
        ("img84", 84, None, 15, [(5, 2, 6, None)], None),
        # round 6: masks other than right padding (modeling_internlm2.py:96-125,893-913: positions ignore the mask, keys obey it)
        ("leftpad", 56, None, 17, [(6, 4, 7, None), (7, 2, 8, None)], "left"),
        ("holes", 56, None, 18, [(8, 3, 9, None)], "holes"),
    ]
    for name, S, vS, wseed, vids, mask_mode in specs:
        cd, hk, cfg = make_cfg("tiny", S, vS)
        sd32 = synth.synth_state_dict(cfg, seed=wseed, dtype=torch.float32)
        sd = {k: v.to(torch.bfloat16) for k, v in sd32.items()}
        model = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
        px_list, ids_list = [], []
        for (vi, nt, cs, il) in vids:
            px_list.append(synth.synth_pixel_values(100, vi, nt, S))
            ids_list.append(synth.synth_input_ids(n_img_tokens(cfg, nt), cs, interleave_frames=il))
        px = torch.cat(px_list)
        ids, mask = synth.pad_batch(ids_list)
        ids, mask = synth.remask(ids, mask, mask_mode)
        ref = run_reference(model, px, ids, mask)
        probes = {}
        orc = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID, probes=probes)
        check_equal(ref, orc, name)
        # per-sample batch-1 forwards must equal the padded batch rows
        if len(vids) > 1 and mask_mode is None:
            off = 0
            for b, (vi, nt, cs, il) in enumerate(vids):
                one = run_reference(model, px[off:off + nt], ids_list[b], torch.ones_like(ids_list[b]))
                off += nt
                for k in FIELDS:
                    assert torch.equal(one[k][0], ref[k][b]), (name, k, b)
        sd_f = {k: v.float() for k, v in sd.items()}
        f32 = ref_cpu.reward_forward(sd_f, cfg, px.float(), ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
        for k, v in to_np(ref).items():
            arrays[f"{name}/{k}"] = v
        for k, v in to_np(f32).items():
            arrays[f"{name}/fp32/{k}"] = v
        for k, v in probes.items():
            arrays[f"{name}/probe/{k}"] = v.float().numpy()
        cases.append(dict(name=name, kind="tiny", image_size=S, vit_image_size=vS, weight_seed=wseed, mask_mode=mask_mode,
                          pixel_seed=100, videos=[dict(video_idx=a, n_tiles=b, caption_seed=c, interleave=d)
                                                  for a, b, c, d in vids]))
        print("tiny case", name, "score", ref["score"].tolist(), "fp32", f32["score"].tolist())
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **arrays)
    json.dump(dict(cases=cases, cpu_threads=torch.get_num_threads()), open(os.path.join(HERE, "tiny.json"), "w"), indent=1)


def gen_full(tag: str, S: int, n_videos: int, wseed: int, pixel_seed: int, n_tiles: int = 8, check_oracle: int = 1):
    cd, hk, cfg = make_cfg("2b", S)
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=wseed)
    model = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    print(f"[{tag}] model ready in {time.time() - t0:.1f}s")
    arrays, vids = {}, []
    sd_f = None
    for v in range(n_videos):
        px = synth.synth_pixel_values(pixel_seed, v, n_tiles, S)
        ids = synth.synth_input_ids(n_img_tokens(cfg, n_tiles), caption_seed=v // 2)
        mask = torch.ones_like(ids)
        t0 = time.time()
        ref = run_reference(model, px, ids, mask)
        dt = time.time() - t0
        if v < check_oracle:
            probes = {}
            orc = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID, probes=probes)
            check_equal(ref, orc, f"{tag}:{v}")
            arrays[f"v{v}/probe/vit_embeds_head"] = probes["vit_embeds"][:, :4, :16].float().numpy()
            arrays[f"v{v}/probe/vit_layer0_head"] = probes["vit_layer0"][:, :4, :16].float().numpy()
            arrays[f"v{v}/probe/vit_embed_head"] = probes["vit_embed"][:, :4, :16].float().numpy()
            nl = cfg.llm_config.num_hidden_layers
            arrays[f"v{v}/probe/llm_layer0_tail"] = probes["llm_layer0"][0, -4:, :16].float().numpy()
            arrays[f"v{v}/probe/llm_last_tail"] = probes[f"llm_layer{nl - 1}"][0, -4:, :16].float().numpy()
            # whole-row samples of several layers of both towers (bf16 bit patterns): per-layer relative-L2 checks at
            # MJ-VIDEO-2B dims need thousands of elements, not a 4 x 16 corner
            for L in PROBE_VIT_LAYERS:
                arrays[f"v{v}/probe/vit_layer{L}_rows"] = _bits(probes[f"vit_layer{L}"][0, :PROBE_VIT_ROWS, :])
            for L in PROBE_LLM_LAYERS:
                arrays[f"v{v}/probe/llm_layer{L}_rows"] = _bits(probes[f"llm_layer{L}"][0, -PROBE_LLM_ROWS:, :])
            if sd_f is None:
                sd_f = {k: t.float() for k, t in sd.items()}
            f32 = ref_cpu.reward_forward(sd_f, cfg, px.float(), ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
            for k, t in to_np(f32).items():
                arrays[f"v{v}/fp32/{k}"] = t
        for k, t in to_np(ref).items():
            arrays[f"v{v}/{k}"] = t
        vids.append(dict(video_idx=v, n_tiles=n_tiles, caption_seed=v // 2, seconds=round(dt, 2)))
        print(f"[{tag}] video {v}: {dt:.1f}s score={ref['score'].item():+.5f}")
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **arrays)
    json.dump(dict(kind="2b", image_size=S, weight_seed=wseed, pixel_seed=pixel_seed, videos=vids,
                   cpu_threads=torch.get_num_threads(),
                   row_probes=dict(vit_layers=list(PROBE_VIT_LAYERS), vit_rows=PROBE_VIT_ROWS,
                                   llm_layers=list(PROBE_LLM_LAYERS), llm_rows=PROBE_LLM_ROWS,
                                   layout="v{i}/probe/vit_layer{L}_rows: tile 0, rows [0, vit_rows), all columns; "
                                          "v{i}/probe/llm_layer{L}_rows: last llm_rows token rows, all columns; bf16 bit patterns")),
              open(os.path.join(HERE, f"{tag}.json"), "w"), indent=1)


RESUME = False


def gen_rankset(tag: str, S: int, pairs: int, wseed: int, pixel_seed: int, fp32_every: int, n_tiles: int = 8):
    """P pairs; both videos of pair p share caption seed 1000+p.  Stores the 34 numbers per video
    SURVEY.md §8(e) names (score, 5 aspect scores, 28 rewards) from the reference in bf16, and the
    fp32 run of the oracle for every ``fp32_every``-th pair (noise floor)."""
    cd, hk, cfg = make_cfg("2b", S)
    sd = synth.synth_state_dict(cfg, seed=wseed)
    model = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    sd_f = {k: t.float() for k, t in sd.items()} if fp32_every else None
    out = np.zeros((pairs, 2, 34), dtype=np.float32)
    out32 = np.full((pairs, 2, 34), np.nan, dtype=np.float32)
    path = os.path.join(HERE, f"{tag}.npz")
    first = 0
    if RESUME and os.path.isfile(path):   # extend an existing set (same seeds / thread count): keep what is there
        old = np.load(path)
        first = min(old["ref_bf16"].shape[0], pairs)
        out[:first], out32[:first] = old["ref_bf16"][:first], old["ref_fp32"][:first]
        print(f"[{tag}] resuming after {first} pairs", flush=True)
    t_start = time.time()
    for p in range(first, pairs):
        ids = synth.synth_input_ids(n_img_tokens(cfg, n_tiles), caption_seed=1000 + p)
        mask = torch.ones_like(ids)
        for j in range(2):
            px = synth.synth_pixel_values(pixel_seed, 2 * p + j, n_tiles, S)
            r = run_reference(model, px, ids, mask)
            out[p, j, 0] = r["score"].item()
            out[p, j, 1:6] = r["aspect_scores"][0].numpy()
            out[p, j, 6:] = r["rewards"][0].float().numpy()
            if fp32_every and p % fp32_every == 0:
                f = ref_cpu.reward_forward(sd_f, cfg, px.float(), ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
                out32[p, j, 0] = f["score"].item()
                out32[p, j, 1:6] = f["aspect_scores"][0].numpy()
                out32[p, j, 6:] = f["rewards"][0].numpy()
        print(f"[{tag}] pair {p}: {out[p, 0, 0]:+.5f} {out[p, 1, 0]:+.5f}  ({time.time() - t_start:.0f}s)", flush=True)
        if p % 8 == 7 or p == pairs - 1:
            np.savez_compressed(path + ".tmp.npz", ref_bf16=out[:p + 1], ref_fp32=out32[:p + 1])
            os.replace(path + ".tmp.npz", path)   # atomic: a snapshot of the tree never sees a half-written fixture
            json.dump(dict(kind="2b", image_size=S, weight_seed=wseed, pixel_seed=pixel_seed, pairs=p + 1,
                           n_tiles=n_tiles, caption_seed_base=1000, fp32_every=fp32_every,
                           cpu_threads=torch.get_num_threads(),
                           layout="[pair, video, (score, aspect_scores[5], rewards[28])]"),
                      open(os.path.join(HERE, f"{tag}.json"), "w"), indent=1)


def gen_c4lite(tag: str = "c4lite", S: int = 448, n_tiles: int = 48, wseed: int = 0, pixel_seed: int = 600):
    """BASELINE.json configs[3] (16 frames x max_num = 6 dynamic tiles, long-context image tokens) at the largest size the
    eager reference fits in this container: 16 frames x 3 tiles = 48 tiles -> N = 48 * 256 + 138 = 12 426 tokens (the
    [1, 16, N, N] scores are 4.9 GB in bf16 and 9.9 GB as the fp32 softmax input / output; the full 112-tile case, N = 28 810,
    would need 26.6 + 53 GB per layer).  One video, bf16 reference only (an fp32 run does not fit), all nine output fields
    incl. the two full hidden-state rows."""
    cd, hk, cfg = make_cfg("2b", S)
    sd = synth.synth_state_dict(cfg, seed=wseed)
    model = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    del sd
    px = synth.synth_pixel_values(pixel_seed, 0, n_tiles, S)
    ids = synth.synth_input_ids(n_img_tokens(cfg, n_tiles), caption_seed=77)
    mask = torch.ones_like(ids)
    t0 = time.time()
    ref = run_reference(model, px, ids, mask)
    dt = time.time() - t0
    arrays = {f"v0/{k}": t for k, t in to_np(ref).items()}
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **arrays)
    json.dump(dict(kind="2b", image_size=S, weight_seed=wseed, pixel_seed=pixel_seed, n_tiles=n_tiles, caption_seed=77,
                   seq_len=int(ids.shape[1]), seconds=round(dt, 1), cpu_threads=torch.get_num_threads(),
                   note="reference bf16 only; no fp32 run at this size"),
              open(os.path.join(HERE, f"{tag}.json"), "w"), indent=1)
    print(f"[{tag}] N={ids.shape[1]} {dt:.0f}s score={ref['score'].item():+.5f}", flush=True)


def gen_rankhid(tag: str, S: int, pairs: int, wseed: int, pixel_seed: int, fp32_every: int, n_tiles: int = 8,
                old_tag: str = None):
    """BACKBONE pass of the reference over the rank set's videos (same seeds as ``gen_rankset``): stores the two
    hidden-state rows the heads read (``hidden_state`` = h_r, ``prompt_embedding`` = h_g; bf16 bit patterns), the 34
    output numbers under the default synthetic heads, and the same from an fp32 run of every ``fp32_every``-th pair.
    Everything downstream of (h_r, h_g) is a function of the head weights only (moe_reward.py:239-297), so other
    head weights can be scored against the reference later by running the reference's own head code on these rows
    (``rankset_eng``) without repeating the backbone.  For every video it also REQUIRES
    ``ref_cpu.reward_heads(h_r, h_g)`` == the reference's outputs bit for bit."""
    cd, hk, cfg = make_cfg("2b", S)
    sd = synth.synth_state_dict(cfg, seed=wseed)
    model = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    sd_f = {k: t.float() for k, t in sd.items()} if fp32_every else None
    H = cfg.llm_config.hidden_size
    n32 = (pairs + fp32_every - 1) // fp32_every if fp32_every else 0
    A = dict(hr_bf16=np.zeros((pairs, 2, H), np.uint16), hg_bf16=np.zeros((pairs, 2, H), np.uint16),
             out34=np.zeros((pairs, 2, 34), np.float32),
             hr_fp32=np.zeros((n32, 2, H), np.float32), hg_fp32=np.zeros((n32, 2, H), np.float32),
             out34_fp32=np.zeros((n32, 2, 34), np.float32))
    path = os.path.join(HERE, f"{tag}.npz")
    first = 0
    if RESUME and os.path.isfile(path):
        old = np.load(path)
        first = min(int(old["done"]), pairs)
        for k in A:
            n = min(old[k].shape[0], A[k].shape[0])
            A[k][:n] = old[k][:n]
        print(f"[{tag}] resuming after {first} pairs", flush=True)
    prev = None
    if old_tag and os.path.isfile(os.path.join(HERE, f"{old_tag}.npz")):
        prev = np.load(os.path.join(HERE, f"{old_tag}.npz"))["ref_bf16"]
    t_start = time.time()
    mism = 0

    def pack34(r, row):
        row[0] = r["score"].item()
        row[1:6] = r["aspect_scores"][0].float().numpy()
        row[6:] = r["rewards"][0].float().numpy()

    for p in range(first, pairs):
        ids = synth.synth_input_ids(n_img_tokens(cfg, n_tiles), caption_seed=1000 + p)
        mask = torch.ones_like(ids)
        for j in range(2):
            px = synth.synth_pixel_values(pixel_seed, 2 * p + j, n_tiles, S)
            r = run_reference(model, px, ids, mask)
            heads = ref_cpu.reward_heads(sd, cfg, r["hidden_state"], r["prompt_embedding"])
            check_equal(r, heads, f"{tag}: heads-from-hidden-rows, pair {p} video {j}")
            A["hr_bf16"][p, j] = _bits(r["hidden_state"][0])
            A["hg_bf16"][p, j] = _bits(r["prompt_embedding"][0])
            pack34(r, A["out34"][p, j])
            if prev is not None and p < prev.shape[0] and not np.array_equal(prev[p, j], A["out34"][p, j]):
                mism += 1
            if fp32_every and p % fp32_every == 0:
                f = ref_cpu.reward_forward(sd_f, cfg, px.float(), ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
                q = p // fp32_every
                A["hr_fp32"][q, j] = f["hidden_state"][0].numpy()
                A["hg_fp32"][q, j] = f["prompt_embedding"][0].numpy()
                pack34(f, A["out34_fp32"][q, j])
        print(f"[{tag}] pair {p}: {A['out34'][p, 0, 0]:+.5f} {A['out34'][p, 1, 0]:+.5f}  "
              f"({time.time() - t_start:.0f}s, {mism} videos differ from {old_tag})", flush=True)
        if p % 8 == 7 or p == pairs - 1:
            np.savez_compressed(path + ".tmp.npz", done=np.int64(p + 1), **A)
            os.replace(path + ".tmp.npz", path)
            json.dump(dict(kind="2b", image_size=S, weight_seed=wseed, pixel_seed=pixel_seed, pairs=p + 1,
                           n_tiles=n_tiles, caption_seed_base=1000, fp32_every=fp32_every,
                           cpu_threads=torch.get_num_threads(), videos_differing_from_old_set=mism,
                           layout="hr/hg: [pair, video, hidden] (bf16 bit patterns / fp32 of every fp32_every-th pair); "
                                  "out34: [pair, video, (score, aspect_scores[5], rewards[28])] under the default heads"),
                      open(os.path.join(HERE, f"{tag}.json"), "w"), indent=1)


def _from_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.int32) << 16).view(torch.float32).to(torch.bfloat16)


def _accuracy_by_reference_code(votes, scores):
    """prefer_truth / prefer_total / truth / total by EXECUTING the reference's own bookkeeping lines
    (scripts/eval/eval_genai_mjvideo.py:142-163, read from /root/reference at generation time; only the counts are
    stored)."""
    import textwrap
    src_path = os.path.join(RS.REFERENCE_ROOT, "scripts", "eval", "eval_genai_mjvideo.py")
    lines = open(src_path).read().split("\n")
    first = next(i for i, l in enumerate(lines) if l.strip().startswith('left_judge = "good"'))
    last = next(i for i, l in enumerate(lines) if l.strip().startswith('print(f"prefer_Acc'))
    body = compile(textwrap.dedent("\n".join(lines[first:last])), src_path, "exec")
    ns = dict(prefer_truth=0, prefer_total=0, truth=0, total=0)
    for vote, (sl, sr) in zip(votes, scores):
        ns.update(vote_type=vote, score_left=torch.tensor(float(sl)), score_right=torch.tensor(float(sr)))
        exec(body, ns)
    return {k: int(ns[k]) for k in ("prefer_truth", "prefer_total", "truth", "total")}


def gen_rankset_eng(hid_tag: str, out_tag: str, S: int, n_dirs: int = 4, margin_k: float = 12.0, check_full: int = 2,
                    n_gate_dirs: int = 8):
    check_full = int(os.environ.get("MJV_CHECK_FULL", check_full))
    """The ENGINEERED rank set (SURVEY.md §7 "hard parts", north_star ">= 0.999 rank agreement on a fixed synthetic set"):
    same backbone, same videos as the rank set, but a regression head fit the way a trained head is - its 28 rows lie in
    the span of the leading principal directions of the backbone's reward-row hidden state over the set (made orthogonal
    to the mean state, so rewards are centred).  Score differences between videos then come from feature directions
    whose inter-video spread is two orders of magnitude above the bf16 noise (random rows, as in ``rankset_*``: about 13),
    and every stored pair is decisive by construction: a pair is kept only if its reference margin exceeds ``margin_k`` x
    the rms bf16-vs-fp32 deviation of the engineered scores.
    The reference scores come from the reference's OWN head code (moe_reward.py:213-297) replayed on the hidden-state
    rows its backbone produced in ``gen_rankhid`` (oracle/reference_shim.build_reference_heads); ``check_full`` videos are
    also pushed through a full reference forward with the new head to prove the composition is what one forward gives."""
    hid = np.load(os.path.join(HERE, f"{hid_tag}.npz"))
    meta = json.load(open(os.path.join(HERE, f"{hid_tag}.json")))
    P = int(hid["done"])
    cd, hk, cfg = make_cfg("2b", S)
    H = cfg.llm_config.hidden_size
    hr, hg = _from_bits(hid["hr_bf16"][:P]), _from_bits(hid["hg_bf16"][:P])          # [P, 2, H] bf16
    every = meta["fp32_every"]
    idx32 = np.arange(0, P, every)
    hr32 = torch.from_numpy(hid["hr_fp32"][:len(idx32)]).double().reshape(-1, H)
    mu = hr32.mean(0)
    _, sv, vt = torch.linalg.svd(hr32 - mu, full_matrices=False)
    mhat = mu / mu.norm()
    dirs = vt[:n_dirs] - (vt[:n_dirs] @ mhat)[:, None] * mhat[None, :]               # orthogonal to the mean state
    dirs, _ = torch.linalg.qr(dirs.T)
    dirs = dirs.T                                                                    # [n_dirs, H], orthonormal
    spread = ((hr32 - mu) @ dirs.T).std(0)                                           # inter-video std along each
    g = np.random.Generator(np.random.Philox(key=[1234, 5678]))
    # every criterion reads the leading direction with a positive weight near 1 (so the gating-weighted sums of rewards do
    # not turn the gating nets' own bf16 noise, ~4e-3 relative, into score noise: sum_c w_c r_c is insensitive to w when the
    # r_c agree) plus a smaller criterion-specific mix of the other directions (so that the 28 rewards, the 5 aspect scores
    # and the gating weights all still matter to the score)
    mix = torch.from_numpy(g.standard_normal((cfg.num_objectives, n_dirs)))
    mix[:, 0] = 1.0 + 0.25 * mix[:, 0]
    mix[:, 1:] *= 0.35
    W = (mix / spread[None, :]) @ dirs
    W = W.to(torch.bfloat16)
    # first gating layers: rows in the span of the leading principal directions of the GATING-row hidden state (what is
    # left of the score noise after the regression head is fit comes from the gating nets amplifying h_g's bf16 noise)
    hg32d = torch.from_numpy(hid["hg_fp32"][:len(idx32)]).double().reshape(-1, H)
    mu_g = hg32d.mean(0)
    _, _, vt_g = torch.linalg.svd(hg32d - mu_g, full_matrices=False)
    gdirs = vt_g[:n_gate_dirs]
    gate_dirs = (gdirs / ((hg32d - mu_g) @ gdirs.T).std(0)[:, None]).numpy().astype(np.float32)
    head_sd = synth.engineered_head_state_dict(cfg, meta["weight_seed"], W, gate_dirs)
    import hashlib
    checksum = hashlib.sha1(b"".join(_bits(head_sd[k]).tobytes() for k in sorted(head_sd))).hexdigest()
    ref_model = RS.build_reference_heads(cd, hk, head_sd, torch.bfloat16, synth.PAD_ID)
    out = np.zeros((P, 2, 34), np.float32)
    t0 = time.time()
    for p in range(P):
        for j in range(2):
            o = ref_model.replay(hr[p, j][None], hg[p, j][None])
            orc = ref_cpu.reward_heads(head_sd, cfg, hr[p, j][None], hg[p, j][None])
            for k in FIELDS:
                assert torch.equal(getattr(o, k), orc[k]), (p, j, k)
            out[p, j, 0] = o.score.item()
            out[p, j, 1:6] = o.aspect_scores[0].float().numpy()
            out[p, j, 6:] = o.rewards[0].float().numpy()
    print(f"[{out_tag}] {2 * P} head replays through the reference's head code in {time.time() - t0:.0f}s (oracle heads identical)")
    head32 = {k: v.float() for k, v in head_sd.items()}
    out32 = np.zeros((len(idx32), 2, 34), np.float32)
    hg32 = torch.from_numpy(hid["hg_fp32"][:len(idx32)])
    hr32f = torch.from_numpy(hid["hr_fp32"][:len(idx32)])
    for q in range(len(idx32)):
        for j in range(2):
            f = ref_cpu.reward_heads(head32, cfg, hr32f[q, j][None], hg32[q, j][None])
            out32[q, j, 0] = f["score"].item()
            out32[q, j, 1:6] = f["aspect_scores"][0].numpy()
            out32[q, j, 6:] = f["rewards"][0].numpy()
    noise = np.abs(out[idx32][..., 0] - out32[..., 0])
    noise_rms, noise_max = float(np.sqrt((noise ** 2).mean())), float(noise.max())
    margin = np.abs(out[:, 0, 0] - out[:, 1, 0])
    keep = margin > margin_k * noise_rms
    sigma = float(out[..., 0].std())
    print(f"[{out_tag}] score spread {sigma:.4f}; bf16-vs-fp32 noise rms {noise_rms:.5f} max {noise_max:.5f} (spread / rms = "
          f"{sigma / noise_rms:.0f}); pairs kept {int(keep.sum())}/{P} (margin > {margin_k} x rms = {margin_k * noise_rms:.4f}); "
          f"smallest kept margin {margin[keep].min():.4f}")
    if check_full:   # composition == one full reference forward with the engineered head (a few videos)
        sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"])
        sd.update(head_sd)
        full = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
        for v in range(check_full):
            p, j = v // 2, v % 2
            ids = synth.synth_input_ids(n_img_tokens(cfg, meta["n_tiles"]), caption_seed=meta["caption_seed_base"] + p)
            px = synth.synth_pixel_values(meta["pixel_seed"], 2 * p + j, meta["n_tiles"], S)
            r = run_reference(full, px, ids, torch.ones_like(ids))
            assert r["score"].item() == out[p, j, 0] and np.array_equal(r["rewards"][0].float().numpy(), out[p, j, 6:]), (p, j)
        print(f"[{out_tag}] {check_full} full reference forwards with the engineered head == backbone rows + head replay, bit for bit")
        del full, sd
    votes_cycle = ["leftvote", "rightvote", "tievote", "bothbad_vote"]
    votes = [votes_cycle[i % 4] for i in range(P)]
    acc = _accuracy_by_reference_code([v for v, k in zip(votes, keep) if k], out[keep][..., 0])
    np.savez_compressed(os.path.join(HERE, f"{out_tag}.npz"), regression_weight_bits=_bits(W), gate_dirs=gate_dirs,
                        ref_bf16=out, ref_fp32=out32, fp32_pairs=idx32, keep=keep)
    json.dump(dict(kind="2b", image_size=S, weight_seed=meta["weight_seed"], pixel_seed=meta["pixel_seed"], pairs=P,
                   n_tiles=meta["n_tiles"], caption_seed_base=meta["caption_seed_base"], hidden_rows_from=hid_tag,
                   n_dirs=n_dirs, n_gate_dirs=n_gate_dirs, head_weights_sha1=checksum, margin_k=margin_k, noise_rms=noise_rms, noise_max=noise_max, score_spread=sigma,
                   pairs_kept=int(keep.sum()), votes="vote_type of pair i = [leftvote, rightvote, tievote, bothbad_vote][i % 4]",
                   accuracy_on_kept_pairs=acc, cpu_threads=torch.get_num_threads(),
                   layout="ref_bf16 / ref_fp32: [pair, video, (score, aspect_scores[5], rewards[28])] under the engineered "
                          "heads = synth.engineered_head_state_dict(cfg, weight_seed, regression_weight_bits as bf16, gate_dirs); "
                          "head_weights_sha1 = sha1 over the bf16 bits of every head tensor in key order; backbone as synth "
                          "seed weight_seed"),
              open(os.path.join(HERE, f"{out_tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["tiny", "full_c1", "full_c2", "rankset_c1", "rankset_c2", "rankhid_c1", "rankhid_c2", "c4lite", "rankeng_c1", "rankeng_c2"])
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--fp32-every", type=int, default=4, help="rankhid: fp32 run of every n-th pair")
    ap.add_argument("--resume", action="store_true", help="rank sets: keep the pairs already in the fixture and append")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    RESUME = a.resume
    if a.what == "tiny":
        gen_tiny()
    elif a.what == "full_c1":
        gen_full("full_c1", 224, n_videos=4, wseed=0, pixel_seed=200, check_oracle=4)
    elif a.what == "full_c2":
        gen_full("full_c2", 448, n_videos=4, wseed=0, pixel_seed=300, check_oracle=2)
    elif a.what == "rankset_c1":
        gen_rankset("rankset_c1", 224, a.pairs, wseed=0, pixel_seed=400, fp32_every=4)
    elif a.what == "rankset_c2":
        gen_rankset("rankset_c2", 448, a.pairs, wseed=0, pixel_seed=500, fp32_every=8)
    elif a.what == "rankeng_c1":
        gen_rankset_eng("rankhid_c1", "rankeng_c1", 224)
    elif a.what == "rankeng_c2":
        gen_rankset_eng("rankhid_c2", "rankeng_c2", 448)
    elif a.what == "c4lite":
        gen_c4lite()
    elif a.what == "rankhid_c1":
        gen_rankhid("rankhid_c1", 224, a.pairs, wseed=0, pixel_seed=400, fp32_every=a.fp32_every, old_tag="rankset_c1")
    elif a.what == "rankhid_c2":
        gen_rankhid("rankhid_c2", 448, a.pairs, wseed=0, pixel_seed=500, fp32_every=a.fp32_every, old_tag="rankset_c2")
