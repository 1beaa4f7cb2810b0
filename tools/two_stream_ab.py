#!/usr/bin/env python3
"""A/B: the 4-pair step as ONE forward over its 8 videos on one HIP stream (production) against the same 8 videos as TWO
forwards of 4 videos, each on its own stream and its own model instance (own activation buffers and K-slicing scratch), so that
the hardware runs the two launch sequences side by side: does a second queue fill the under-filled launches (tails, last rounds,
attention drain) of the first?  Also 16 videos as 1 x 16 against 2 x 8.  Every video is an independent forward, so
the split changes no result; tensors are passed unmodified (no per-step ids copy) in both arms.

usage: two_stream_ab.py [steps=6] [rounds=3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mj_video_amd import configuration as C, synth  # noqa: E402
from mj_video_amd.modeling import InternVLChatRewardModeling  # noqa: E402
from mj_video_amd.chat_input import num_image_tokens_per_tile  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
S, F = 448, 8
cfg = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(S), **C.mjvideo_head_kwargs())


def make():
    m = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
    bench.random_init_on_device(m, cfg, dev, seed=1)     # same seed: the two instances hold the same weights
    m.config.pad_token_id = synth.PAD_ID
    m.model.img_context_token_id = synth.IMG_CONTEXT_ID
    return m.eval()


models = [make(), make()]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
per_tile = num_image_tokens_per_tile(cfg)


def batch(n_videos, first):
    px = torch.randn(n_videos * F, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + first)).to(torch.bfloat16)
    ids, mask = synth.pad_batch([synth.synth_input_ids(per_tile * F, caption_seed=(first + p) // 2) for p in range(n_videos)])
    return px, ids.to(dev), mask.to(dev)


def run(parts, use_streams):
    """parts: [(model, stream, (px, ids, mask))]; one step = every part's forward, enqueued in order from this thread"""
    def one():
        for m, s, (px, ids, mask) in parts:
            if use_streams:
                with torch.cuda.stream(s):
                    m.forward(px, ids, mask)
            else:
                m.forward(px, ids, mask)
    for _ in range(2):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for total in (8, 16):
    whole = batch(total, 0)
    h = total // 2
    halves = [tuple(t[: h * F] if i == 0 else t[:h] for i, t in enumerate(whole)),
              tuple(t[h * F:] if i == 0 else t[h:] for i, t in enumerate(whole))]
    halves = [tuple(t.contiguous() for t in hv) for hv in halves]
    # the split changes no score: each video of the halves against the same video of the whole batch
    models[0].forward(*whole)
    ref = models[0].last_packed34.clone()
    torch.cuda.synchronize()      # (one model instance = one set of buffers: never two of its forwards in flight on two streams)
    got = []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            models[i].forward(*halves[i])
            got.append(models[i].last_packed34.clone())
    torch.cuda.synchronize()
    scale = ref.float().abs().mean().item()
    same = torch.equal(torch.cat(got), ref)
    md = (torch.cat(got).float() - ref.float()).abs().max().item()
    arms = {
        f"1 x {total} videos, one stream": lambda: run([(models[0], None, whole)], False),
        f"2 x {h} videos, one stream (back to back)": lambda: run([(models[0], None, halves[0]), (models[1], None, halves[1])], False),
        f"2 x {h} videos, two streams": lambda: run([(models[0], streams[0], halves[0]), (models[1], streams[1], halves[1])], True),
    }
    res = {k: [] for k in arms}
    for _ in range(rounds):                       # interleaved: box drift hits every arm alike
        for k, fn in arms.items():
            res[k].append(fn())
    print(f"--- {total} videos = {total // 2} pairs per step, {steps} steps x {rounds} rounds; split results bit-identical to the whole batch: {same} (max |d| {md:.3g} on values of mean magnitude {scale:.3g}: the halves' GEMMs slice K differently)")
    for k, v in res.items():
        best = min(v)
        print(f"{k:44s} {best:8.2f} ms per step  {total / 2 / best * 1e3:7.2f} pairs/s   (all rounds: {' '.join(f'{x:.2f}' for x in v)})")
