#!/usr/bin/env python3
"""cpu_baseline thread sweep: the oracle forward of ONE video at the headline shape (8 tiles @448^2,
N = 2186, bf16, LM head included as the reference executes it) at 16 / 32 / 64 / 128 threads of the GPU box's host - which
thread count is the best the reference's CPU path does there?  (bench.py uses 32.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

model, phys, logical = bench.host_cpu_info()
print(f"host: {model}, {phys} physical cores, {logical} logical CPUs")
for threads in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    if threads > logical:
        continue
    t0 = time.time()
    v, dt, n = bench.cpu_baseline(448, 8, threads, iters=2)
    print(f"threads {threads:4d}: {dt:6.2f} s per video forward = {v:.5f} pairs/s  (N = {n}; 1 warm-up + 2 timed, {time.time() - t0:.0f} s in all)", flush=True)
