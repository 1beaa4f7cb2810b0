#!/usr/bin/env python3
"""Per-kernel time of ONE video per forward (the reference's own call pattern, 8 tiles @448^2, N = 2186) next to the batched
step (8 videos per forward), per video: where the small batch loses."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mj_video_amd import configuration as C, ops, synth  # noqa: E402
from mj_video_amd.modeling import InternVLChatRewardModeling  # noqa: E402
from mj_video_amd.chat_input import num_image_tokens_per_tile  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
S, F = 448, 8
cfg = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(S), **C.mjvideo_head_kwargs())
model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16, device=dev)
bench.random_init_on_device(model, cfg, dev, seed=1)
model.config.pad_token_id = synth.PAD_ID
model.model.img_context_token_id = synth.IMG_CONTEXT_ID
model.eval()
res = {}
walls = {}
for n in (1, 2, 8):
    px = torch.randn(n * F, 3, S, S, device=dev).to(torch.bfloat16)
    ids, mask = synth.pad_batch([synth.synth_input_ids(num_image_tokens_per_tile(cfg) * F, caption_seed=p // 2) for p in range(n)])
    ids, mask = ids.to(dev), mask.to(dev)
    for _ in range(3):
        model.forward(px, ids, mask)
    torch.cuda.synchronize()
    ops.prof_filter(None); ops.prof_reset(); ops.prof_enable(True)
    reps = 8 // n * 2
    for _ in range(reps):
        model.forward(px, ids, mask)
    torch.cuda.synchronize()
    ops.prof_enable(False)
    res[n] = {k: v["ms"] / (reps * n) for k, v in ops.prof_results().items()}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)   # wall clock of back-to-back forwards, no per-kernel events
    t0.record()
    for _ in range(reps):
        model.forward(px, ids.clone(), mask.clone())
    t1.record(); torch.cuda.synchronize()
    walls[n] = t0.elapsed_time(t1) / (reps * n)
keys = sorted(set(res[1]) | set(res[2]) | set(res[8]), key=lambda k: -(res[1].get(k, 0)))
print(f"{'kernel':28s} {'1 video/forward':>16s} {'2 (one pair)':>14s} {'8 videos/forward':>17s}   (ms per VIDEO)")
for k in keys:
    print(f"{k:28s} {res[1].get(k, 0):16.3f} {res[2].get(k, 0):14.3f} {res[8].get(k, 0):17.3f}")
print(f"{'sum of kernels':28s} {sum(res[1].values()):16.3f} {sum(res[2].values()):14.3f} {sum(res[8].values()):17.3f}")
print(f"{'wall clock, back to back':28s} {walls[1]:16.3f} {walls[2]:14.3f} {walls[8]:17.3f}")
