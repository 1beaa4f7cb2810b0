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
    ]
    for name, S, vS, wseed, vids, _ in specs:
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
        ref = run_reference(model, px, ids, mask)
        probes = {}
        orc = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID, probes=probes)
        check_equal(ref, orc, name)
        # per-sample batch-1 forwards must equal the padded batch rows
        if len(vids) > 1:
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
        cases.append(dict(name=name, kind="tiny", image_size=S, vit_image_size=vS, weight_seed=wseed,
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
                   cpu_threads=torch.get_num_threads()),
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


def _bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> its uint16 bit patterns."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["tiny", "full_c1", "full_c2", "rankset_c1", "rankset_c2", "rankhid_c1", "rankhid_c2"])
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
    elif a.what == "rankhid_c1":
        gen_rankhid("rankhid_c1", 224, a.pairs, wseed=0, pixel_seed=400, fp32_every=a.fp32_every, old_tag="rankset_c1")
    elif a.what == "rankhid_c2":
        gen_rankhid("rankhid_c2", 448, a.pairs, wseed=0, pixel_seed=500, fp32_every=a.fp32_every, old_tag="rankset_c2")
