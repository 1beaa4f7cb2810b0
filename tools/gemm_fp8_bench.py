#!/usr/bin/env python3
"""MXFP8 GEMM micro-benchmark at the FFN shapes of the model (HIP events, random data, interleaved with the bf16 kernel of the
same shape): TFLOP/s of the fp8 path with its producing epilogue (MXFP8 output where the model uses one)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mj_video_amd import ops  # noqa: E402

dev, BF = torch.device("cuda:0"), torch.bfloat16
ROUNDS = int(os.environ.get("MJV_BENCH_ROUNDS", 5))
ITERS = int(os.environ.get("MJV_BENCH_ITERS", 10))


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


shapes = [("vit_fc1  gelu -> mxfp8", 65600, 4096, 1024, ops.EPI_BIAS_GELU, True),
          ("vit_fc2  ls + res", 65600, 1024, 4096, ops.EPI_SCALE_RES, False),
          ("llm_w13  silu*up -> mxfp8", 17488, 16384, 2048, ops.EPI_SILU_MUL, True),
          ("llm_w2   + res", 17488, 2048, 8192, ops.EPI_SCALE_RES, False),
          ("vit_fc1  main rows", 65536, 4096, 1024, ops.EPI_BIAS_GELU, True),
          ("vit_fc2  main rows", 65536, 1024, 4096, ops.EPI_SCALE_RES, False),
          ("llm_w13  main rows", 17408, 16384, 2048, ops.EPI_SILU_MUL, True),
          ("llm_w2   main rows", 16384, 2048, 8192, ops.EPI_SCALE_RES, False),
          ("square8k bias", 8192, 8192, 8192, ops.EPI_BIAS, False)]
ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
print(f"{'shape':28s} {'M':>6s} {'N':>6s} {'K':>5s} | mxfp8: ms, TFLOP/s median (best) | bf16 kernel: ms, TFLOP/s | speed-up")
for name, M, N, K, epi, out8 in shapes:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    a8, w8 = ops.quantize_mxfp8(a), ops.quantize_mxfp8(w)
    nout = N // 2 if epi == ops.EPI_SILU_MUL else N
    bias = torch.randn(N, device=dev).to(BF) if epi != ops.EPI_SILU_MUL else None
    res = torch.randn(M, nout, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
    scale = torch.randn(N, device=dev).to(BF) if epi == ops.EPI_SCALE_RES and "vit" in name else None
    o16 = torch.empty(M, nout, device=dev, dtype=BF)
    o8 = ops.MX8.empty(M, nout, dev) if out8 else None
    f8 = lambda: ops.gemm(a8, w8, o8 if out8 else o16, epi, bias=bias, scale=scale, res=res)          # noqa: E731
    f16 = lambda: ops.gemm(a, w, o16, epi, bias=bias, scale=scale, res=res, workspace=ws)             # noqa: E731
    t8, t16 = [], []
    for rnd in range(ROUNDS):
        for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            (t8 if k == 0 else t16).append(timed(f8 if k == 0 else f16))
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:28s} {M:6d} {N:6d} {K:5d} | {np.median(t8):7.3f} ms {fl / np.median(t8):6.0f} ({fl / min(t8):5.0f}) | "
          f"{np.median(t16):7.3f} ms {fl / np.median(t16):6.0f} | {np.median(t16) / np.median(t8):.2f}x", flush=True)
