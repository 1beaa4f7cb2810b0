#!/usr/bin/env python3
"""K-slicing choices for the under-filled GEMMs of a ONE- or TWO-video forward: the language tower's wo and w2
at M = 2122 / 4244 rows (72 / 136 tiles of 256^2 on 256 CUs) under every (slices cap, fewest K-tiles per slice, K threshold)
setting of the bench build, HIP-event timings, rotated rounds.  Prints the table a shape-indexed choice would be read from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MJV_LIBRARY", os.path.join(ROOT, "mj-video_amd", "libmjv_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch
from mj_video_amd import ops
from mj_video_amd._lib import EPI_SCALE_RES, load_library, check
dev, BF = "cuda", torch.bfloat16
lib = load_library()
ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
ops.set_gemm_workspace(ws)
SHAPES = [("wo   1 video", 2122, 2048, 2048), ("w2   1 video", 2122, 2048, 8192), ("wo   2 videos", 4244, 2048, 2048), ("w2   2 videos", 4244, 2048, 8192),
          ("proj 1 video", 8200, 1024, 1024), ("fc2  1 video", 8200, 1024, 4096)]
# (label, codes): 0 resets; 4400 + n = K threshold in K-tiles; 4100 + s = slices cap; 4300 + n = fewest K-tiles per slice; 4200 = off
SETTINGS = [("default", []), ("slicing off", [4200]), ("nk>=16 cap2", [4416, 4102, 4304]), ("nk>=16 cap3", [4416, 4103, 4304]),
            ("nk>=16 cap4 kt4", [4416, 4104, 4304]), ("cap2", [4102]), ("cap4", [4104]), ("cap8 kt8", [4108]), ("cap8 kt16", [4108, 4316])]
iters, rounds = 30, 3
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.03).to(BF); x = torch.randn(M, N, device=dev).to(BF)
    times = {lab: [] for lab, _ in SETTINGS}
    for rnd in range(rounds):
        order = SETTINGS[rnd % len(SETTINGS):] + SETTINGS[:rnd % len(SETTINGS)]
        for lab, codes in order:
            check(lib.mjv_bench_gemm_set(0), "reset")
            for c in codes:
                check(lib.mjv_bench_gemm_set(c), "set")
            for _ in range(3):
                ops.gemm(a, w, x, EPI_SCALE_RES, res=x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm(a, w, x, EPI_SCALE_RES, res=x)
            e1.record(); torch.cuda.synchronize()
            times[lab].append(e0.elapsed_time(e1) / iters * 1e3)
    check(lib.mjv_bench_gemm_set(0), "reset")
    fl = 2.0 * M * N * K
    best = min(times, key=lambda k: sorted(times[k])[1])
    print(f"{name:14s} {M} x {N} x {K}: " + "; ".join(f"{lab} {sorted(t)[1]:.1f} us" for lab, t in times.items()) + f"  -> best: {best} ({fl / sorted(times[best])[1] / 1e6:.0f} TF/s)", flush=True)
