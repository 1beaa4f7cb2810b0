#!/usr/bin/env python3
"""The clock the chip holds INSIDE the main loops (MI355X_MICROARCH.md "DVFS give-back" item 6): delta s_memtime / delta
s_memrealtime x 100 MHz, stamped around the loop in the diagnostic builds (never in the product library), after >= 2 s of
back-to-back launches of the same kernel on random data; median over workgroups (GEMM) / waves (attention).

  python tools/inkernel_clock.py gemm   # libmjv_hip_bench.so: gemm256<3,0> (variant 1006) and the persistent gemm256p (1008)
  python tools/inkernel_clock.py attn   # libmjv_hip_stamps.so: attn2<64> and attn2<128, causal>
Replaces the GRBM_GUI_ACTIVE / dispatch-time table of round 3 (that quotient reads high on < 0.3 ms dispatches)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
WARM_S = float(os.environ.get("MJV_CLOCK_WARM_S", 2.0))
os.environ["MJV_LIBRARY"] = os.path.join(ROOT, "mj-video_amd", "libmjv_hip_bench.so" if what == "gemm" else "libmjv_hip_stamps.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mj_video_amd import _lib, ops  # noqa: E402

lib = _lib.load_library()
dev, BF = torch.device("cuda:0"), torch.bfloat16


def keep_busy(fn):
    """>= WARM_S seconds of back-to-back launches (the queue never runs dry: 50 launches are enqueued between host checks)"""
    t0 = time.time()
    n = 0
    while time.time() - t0 < WARM_S:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    return n


if what == "gemm":
    shapes = [("vit_proj  gemm256<3,0>", 65536, 1024, 1024, ops.EPI_SCALE_RES, 1006),
              ("vit_fc2   gemm256<3,0>", 65536, 1024, 4096, ops.EPI_SCALE_RES, 1006),
              ("llm_wo    gemm256<3,0>", 16384, 2048, 2048, ops.EPI_SCALE_RES, 1006),
              ("llm_w2    gemm256<3,0>", 16384, 2048, 8192, ops.EPI_SCALE_RES, 1006),
              ("vit_fc1   gemm256<1,0>", 65536, 4096, 1024, ops.EPI_BIAS_GELU, 1006),
              ("vit_qkv   gemm256p<0>", 65536, 3072, 1024, ops.EPI_BIAS, 1008),
              ("llm_w13   gemm256p<4>", 17408, 16384, 2048, ops.EPI_SILU_MUL, 1008)]
    for name, M, N, K, epi, var in shapes:
        a = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        nout = N // 2 if epi == ops.EPI_SILU_MUL else N
        out = torch.empty(M, nout, device=dev, dtype=BF)
        bias = torch.randn(N, device=dev).to(BF) if epi != ops.EPI_SILU_MUL else None
        res = torch.randn(M, nout, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
        scale = torch.randn(N, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
        tiles = (M // 256) * (N // 256)
        buf = torch.zeros(max(tiles, 256), 8, dtype=torch.int64, device=dev)
        ops.gemm_set_tile(var)
        lib.mjv_bench_gemm_stamp_buffer(buf.data_ptr())
        run = lambda: ops.gemm(a, w, out, epi, bias=bias, scale=scale, res=res)   # noqa: E731
        n = keep_busy(run)
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        lib.mjv_bench_gemm_stamp_buffer(None)
        ops.gemm_set_tile(0)
        d = buf.cpu().numpy().astype(np.float64)
        d = d[d[:, 6] > 0]
        clk = d[:, 2] / d[:, 6] * 0.1   # GHz
        print(f"{name}  M={M} N={N} K={K}: {n} warm launches, stamped launch {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s; "
              f"in-kernel main-loop clock median {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}; "
              f"{len(clk)} workgroups); main loop {np.median(d[:, 2] / np.maximum(d[:, 7], 1)) / (K // 64):.0f} cycles per K-tile", flush=True)
else:
    lib.mjv_attention_stamp_buffer.restype = C.c_int
    lib.mjv_attention_stamp_buffer.argtypes = [C.c_void_p]
    for name, n_seq, L, H, G, D, causal, mode in (("attn2<64>  vit", 64, 1025, 16, 1, 64, False, 0), ("attn2<128> llm causal", 8, 2186, 16, 2, 128, True, 1)):
        N = n_seq * L
        q = torch.randn(N, H * D, device=dev).to(BF)
        k = torch.randn(N, (H // G) * D, device=dev).to(BF)
        v = torch.randn(N, (H // G) * D, device=dev).to(BF)
        o = torch.empty(N, H * D, device=dev, dtype=BF)
        cu = torch.arange(0, (n_seq + 1) * L, L, dtype=torch.int32, device=dev)
        nblk = 8 * ((((L + 63) // 64 + 2) * H * n_seq + 7) // 8) + 64
        buf = torch.zeros(nblk * 4 * 16, dtype=torch.int64, device=dev)
        run = lambda: ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode)   # noqa: E731
        assert lib.mjv_attention_stamp_buffer(buf.data_ptr()) == 0
        n = keep_busy(run)
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        lib.mjv_attention_stamp_buffer(None)
        b = buf.view(-1, 16).cpu().numpy().astype(np.float64)
        rows = b[(b[:, 14] == 2) & (b[:, 12] > 0) & (b[:, 15] > 0)]
        clk = rows[:, 13] / rows[:, 15] * 0.1
        fl = 4.0 * D * n_seq * H * L * L * (0.5 if causal else 1.0)
        print(f"{name}: {n} warm launches, stamped launch {ms:.3f} ms = {fl / ms / 1e9:.0f} TFLOP/s (stamps cost ~10 %); in-kernel tile-loop clock "
              f"median {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}; {len(clk)} waves)", flush=True)
