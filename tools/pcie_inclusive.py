#!/usr/bin/env python3
"""Pairs/s with the inputs starting in HOST memory every step (BASELINE configs[1] shape: 4 pairs = 8 videos x 8 tiles @448^2):
(a) preprocessed bf16 pixel tensors in pinned host memory -> H2D -> forward (what the reference's callers hand over after
load_video); (b) decoded uint8 720p frames in pinned host memory -> H2D -> device preprocessing -> forward (the eval driver's
default path).  bench.py's `value` starts with the pixels resident in HBM; this is the PCIe-inclusive figure next to it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mj_video_amd import configuration as C, synth, video  # noqa: E402
from mj_video_amd.modeling import InternVLChatRewardModeling  # noqa: E402
from mj_video_amd.chat_input import num_image_tokens_per_tile  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
S, F, n = 448, 8, 8
cfg = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(S), **C.mjvideo_head_kwargs())
model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
bench.random_init_on_device(model, cfg, dev, seed=1)
model.config.pad_token_id = synth.PAD_ID
model.model.img_context_token_id = synth.IMG_CONTEXT_ID
model.eval()
ids, mask = synth.pad_batch([synth.synth_input_ids(num_image_tokens_per_tile(cfg) * F, caption_seed=p // 2) for p in range(n)])
px_host = torch.randn(n * F, 3, S, S).to(torch.bfloat16).pin_memory()
frames_host = torch.randint(0, 256, (n * F, 720, 1280, 3), dtype=torch.uint8).pin_memory()
px_dev = px_host.to(dev)


def timed(fn, steps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def resident():
    model.forward(px_dev, ids.to(dev), mask.to(dev))


def from_host_pixels():
    model.forward(px_host.to(dev, non_blocking=True), ids.to(dev), mask.to(dev))


def from_host_frames():
    pv, _ = video.load_frames_device(frames_host.to(dev, non_blocking=True), input_size=S, max_num=1)
    model.forward(pv, ids.to(dev), mask.to(dev))


def prefetched(host_batch, to_px, steps=8):
    """the same step fed through harness.prefetch_to_device: batch i + 1 uploads on a copy stream while batch i is scored"""
    from mj_video_amd import harness

    def run(n):
        for t in harness.prefetch_to_device((host_batch for _ in range(n)), dev):
            model.forward(to_px(t), ids.to(dev), mask.to(dev))
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for name, fn, mb in (("pixels resident in HBM", resident, 0.0),
                     ("bf16 pixel tensors from pinned host memory", from_host_pixels, px_host.numel() * 2 / 1e6),
                     ("uint8 720p frames from pinned host memory + device preprocessing", from_host_frames, frames_host.numel() / 1e6)):
    ms = timed(fn)
    print(f"{name:70s} {ms:8.2f} ms per 4-pair step = {4e3 / ms:6.2f} pairs/s   ({mb:6.1f} MB over PCIe per step)")
for name, host, to_px, mb in (("bf16 pixel tensors from pinned host memory, uploads prefetched on a copy stream", px_host, lambda t: t, px_host.numel() * 2 / 1e6),
                             ("uint8 720p frames from pinned host memory, uploads prefetched + device preprocessing", frames_host,
                              lambda t: video.load_frames_device(t, input_size=S, max_num=1)[0], frames_host.numel() / 1e6)):
    ms = prefetched(host, to_px)
    print(f"{name:90s} {ms:8.2f} ms per 4-pair step = {4e3 / ms:6.2f} pairs/s   ({mb:6.1f} MB over PCIe per step)")
