#!/usr/bin/env python3
"""Where a 256x256 GEMM tile spends its life (diagnostic build 1006 of the kernel: s_memtime stamps of wave 0 of every
workgroup): prologue (first LDS fills) / main loop / epilogue pass A (registers -> LDS tile) / pass B (LDS -> HBM), and how
much of the launch's span the CUs spend between workgroups."""
import os
os.environ.setdefault("MJV_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mj-video_amd", "libmjv_hip_bench.so"))   # bench build: make -C mj-video_amd/csrc bench
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mj_video_amd import ops
from mj_video_amd._lib import load_library

dev = torch.device("cuda:0")
BF = torch.bfloat16
lib = load_library()
shapes = [("vit_qkv", 65536, 3072, 1024, ops.EPI_BIAS), ("vit_proj", 65536, 1024, 1024, ops.EPI_SCALE_RES),
          ("vit_fc1", 65536, 4096, 1024, ops.EPI_BIAS_GELU), ("vit_fc2", 65536, 1024, 4096, ops.EPI_SCALE_RES),
          ("llm_wqkv", 16384, 4096, 2048, ops.EPI_BIAS), ("llm_w13", 16384, 16384, 2048, ops.EPI_SILU_MUL),
          ("llm_w2", 16384, 2048, 8192, ops.EPI_SCALE_RES)]
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    nout = N // 2 if epi == ops.EPI_SILU_MUL else N
    out = torch.empty(M, nout, device=dev, dtype=BF)
    bias = torch.randn(N, device=dev).to(BF) if epi != ops.EPI_SILU_MUL else None
    res = torch.randn(M, nout, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
    scale = torch.randn(N, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
    tiles = (M // 256) * (N // 256)
    buf = torch.zeros(tiles, 8, dtype=torch.int64, device=dev)
    for _ in range(2):
        ops.gemm(a, w, out, epi, bias=bias, scale=scale, res=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(a, w, out, epi, bias=bias, scale=scale, res=res)
    e1.record()
    torch.cuda.synchronize()
    plain_ms = e0.elapsed_time(e1)
    ops.gemm_set_tile(1006)
    lib.mjv_bench_gemm_stamp_buffer(buf.data_ptr())
    ops.gemm(a, w, out, epi, bias=bias, scale=scale, res=res)
    torch.cuda.synchronize()
    lib.mjv_bench_gemm_stamp_buffer(None)
    ops.gemm_set_tile(0)
    d = buf.cpu().double()
    span = (d[:, 0] + d[:, 5]).max().item() - d[:, 0].min().item()
    life = d[:, 5].sum().item() / 256.0
    nk = K // 64
    print(f"{name:9s} M={M} N={N} K={K}: {plain_ms:.3f} ms; per tile cycles: prologue {d[:, 1].mean():6.0f}  main {d[:, 2].mean():7.0f} "
          f"({d[:, 2].mean() / nk:5.0f} per K-tile)  passA {d[:, 3].mean():6.0f}  passB {d[:, 4].mean():6.0f}  total {d[:, 5].mean():7.0f}; "
          f"launch span {span:.0f} cycles, of which workgroups alive on a CU {life / span:.3f}", flush=True)
