#!/usr/bin/env python3
"""A/B asked by the round-2 review (item 8): RMSNorm at the language tower's shape (17 488 x 2 048) as it runs today (one pass:
read x, reduce, normalise, round, gain, write h) against a consumer that gets the row statistics from the producing GEMM's
epilogue (8 per-n-tile partial sums of squares per row, summed in tile order - bench build, mjv_bench_rmsnorm_prestat).
Both read x once and write h once; the difference is the in-wave reduction only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MJV_LIBRARY", os.path.join(ROOT, "mj-video_amd", "libmjv_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mj_video_amd import _lib, ops  # noqa: E402

lib = _lib.load_library()
dev, BF = torch.device("cuda", 0), torch.bfloat16
rows, dim, eps = 17488, 2048, 1e-5
torch.manual_seed(0)
x = torch.randn(rows, dim, device=dev).to(BF)
w = (1 + 0.1 * torch.randn(dim, device=dev)).to(BF)
parts = x.float().view(rows, 8, 256).pow(2).sum(-1).contiguous()       # what the epilogue would emit, one value per n-tile
ya, yb = torch.empty_like(x), torch.empty_like(x)
stream = torch.cuda.current_stream(dev).cuda_stream


def a():
    ops.rmsnorm(x, w, ya, eps)


def b():
    _lib.check(lib.mjv_bench_rmsnorm_prestat(x.data_ptr(), dim, yb.data_ptr(), dim, w.data_ptr(), parts.data_ptr(), rows, dim, eps,
                                             stream), "prestat")


def timeit(f, n=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rnd in range(3):
    ta, tb = timeit(a), timeit(b)
    gb = 4.0 * rows * dim / 1e6      # MB; MB / us = TB/s
    print(f"round {rnd}: today {ta:6.2f} us ({gb / ta:5.2f} TB/s)   statistics supplied {tb:6.2f} us ({gb / tb:5.2f} TB/s)   "
          f"difference {ta - tb:+.2f} us per launch = {(ta - tb) * 49 / 1e3:+.3f} ms per step (49 launches)")
a(); b(); torch.cuda.synchronize()
ne = (ya != yb).sum().item()
print(f"outputs differing between the two (fp32 sum order of the statistics): {ne} of {ya.numel()}")
