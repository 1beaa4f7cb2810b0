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
    f8 = lambda: ops.gemm(a8, w8, o8 if out8 else o16, epi, bias=bias, scale=scale, res=res, workspace=ws)   # noqa: E731  (round 5: tails K-sliced)
    f16 = lambda: ops.gemm(a, w, o16, epi, bias=bias, scale=scale, res=res, workspace=ws)             # noqa: E731
    t8, t16 = [], []
    for rnd in range(ROUNDS):
        for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            (t8 if k == 0 else t16).append(timed(f8 if k == 0 else f16))
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:28s} {M:6d} {N:6d} {K:5d} | {np.median(t8):7.3f} ms {fl / np.median(t8):6.0f} ({fl / min(t8):5.0f}) | "
          f"{np.median(t16):7.3f} ms {fl / np.median(t16):6.0f} | {np.median(t16) / np.median(t8):.2f}x", flush=True)


# ---- round 5: where the time of fc1 + GELU -> MXFP8 goes.  The same main rows (65 536 x 4096 x 1024: 16 full
# rounds of 256 tiles, 8 K-tiles of 128 per tile) under four epilogues, interleaved; per TILE = launch time x 256 CUs / 4096 tiles.
print("\nfc1 (65536 x 4096 x 1024, mxfp8 operands): epilogue decomposition, ms per launch median | TFLOP/s | us per 256 x 256 tile")
M, N, K = 65536, 4096, 1024
a = torch.randn(M, K, device=dev).to(BF)
w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
a8, w8 = ops.quantize_mxfp8(a), ops.quantize_mxfp8(w)
bias = torch.randn(N, device=dev).to(BF)
o16, o8 = torch.empty(M, N, device=dev, dtype=BF), ops.MX8.empty(M, N, dev)
cases = [("bias -> bf16", ops.EPI_BIAS, o16), ("bias -> mxfp8", ops.EPI_BIAS, o8), ("bias + gelu -> bf16", ops.EPI_BIAS_GELU, o16),
         ("bias + gelu -> mxfp8 (the model's)", ops.EPI_BIAS_GELU, o8)]
ts = {c[0]: [] for c in cases}
for rnd in range(ROUNDS):
    for name, epi, out in cases[rnd % 4:] + cases[:rnd % 4]:
        ts[name].append(timed(lambda: ops.gemm(a8, w8, out, epi, bias=bias)))
fl = 2.0 * M * N * K / 1e9
for name, _, _ in cases:
    t = np.median(ts[name])
    print(f"  {name:36s} {t:7.3f} ms | {fl / t:6.0f} | {t * 1e3 * 256 / 4096:6.2f}")
print("  (the MFMA time of a tile at the 5 PFLOP/s peak: 6.9 us; at the rate the K = 8192 main loop holds, 2 600 TFLOP/s: 13.2 us)")
