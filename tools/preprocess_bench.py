#!/usr/bin/env python3
"""Device preprocessing (Pillow-exact resize -> tiles -> normalise -> bf16) timed on the GPU box: frames per second and
algorithmic bytes (input + intermediate written and read + output) per second against the HBM roofline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mj_video_amd import video

dev = torch.device("cuda:0")
for (F, H, W, max_num) in [(128, 720, 1280, 1), (64, 720, 1280, 1), (128, 480, 854, 1), (32, 896, 1344, 6), (16, 1080, 1920, 12)]:
    frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev)
    for _ in range(3):
        out, counts = video.load_frames_device(frames, input_size=448, max_num=max_num)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        out, counts = video.load_frames_device(frames, input_size=448, max_num=max_num)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    S = 448
    cols, rows = video.target_grid(W, H, 1, max_num, S)
    passes = [(S * cols, S * rows)] + ([(S, S)] if cols * rows != 1 else [])
    byts = sum(F * (3.0 * H * W + 2 * 3.0 * H * ow + 3.0 * ow * oh * 2) for ow, oh in passes)
    print(f"{F:4d} frames {W}x{H} max_num={max_num} -> {out.shape[0]} tiles: {ms:7.3f} ms  {F / ms * 1e3:9.0f} frames/s  "
          f"{byts / ms / 1e9:6.2f} TB/s algorithmic (HBM ~8 peak)", flush=True)
