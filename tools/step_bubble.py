#!/usr/bin/env python3
"""How much of a scoring step is the start-of-forward bubble: the forward needs the ids on the HOST (row lists, cu_seqlens), a
device->host copy that waits for the previous step's GPU work, after which the GPU idles until the first launch arrives.
Compares back-to-back steps with fresh id tensors (every real batch) against steps that reuse the same id tensors (the model then
reuses its host copy and never synchronises)."""
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

dev = torch.device("cuda:0")
S, F, n = 448, 8, 8
cfg = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(S), **C.mjvideo_head_kwargs())
model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
bench.random_init_on_device(model, cfg, dev, seed=1)
model.config.pad_token_id = synth.PAD_ID
model.model.img_context_token_id = synth.IMG_CONTEXT_ID
model.eval()
px = torch.randn(n * F, 3, S, S, device=dev).to(torch.bfloat16)
ids, mask = synth.pad_batch([synth.synth_input_ids(num_image_tokens_per_tile(cfg) * F, caption_seed=p // 2) for p in range(n)])
ids, mask = ids.to(dev), mask.to(dev)


def run(fresh, steps=10):
    for _ in range(3):
        model.forward(px, ids.clone() if fresh else ids, mask.clone() if fresh else mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.forward(px, ids.clone() if fresh else ids, mask.clone() if fresh else mask)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rnd in range(3):
    a, b = run(True), run(False)
    print(f"round {rnd}: fresh id tensors {a:.3f} ms per step, same id tensors {b:.3f} ms per step, difference {a - b:+.3f} ms")
