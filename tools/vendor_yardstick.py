#!/usr/bin/env python3
"""External yardstick for the "plateau" statements of DESIGN.md (tools only, never product): the vendor's
kernels on the SAME box, same shapes, same random data, interleaved rounds in one process -
  * GEMM: torch.nn.functional.linear (hipBLASLt / rocBLAS, bf16, no bias, un-fused) against the hand-written 256 x 256 kernel
    with its production epilogue and with the epilogue compiled out (bench variant 1003: main loop only);
  * attention: torch.nn.functional.scaled_dot_product_attention against mjv_attention_bf16 at the two model shapes.
Prints median / best of the rounds.  Any shape where the vendor GEMM beats the hand-written main loop by > 5 % is a target."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MJV_LIBRARY", os.path.join(ROOT, "mj-video_amd", "libmjv_hip_bench.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mj_video_amd import ops  # noqa: E402

dev, BF = torch.device("cuda:0"), torch.bfloat16
ROUNDS = int(os.environ.get("MJV_BENCH_ROUNDS", 5))
ITERS = int(os.environ.get("MJV_BENCH_ITERS", 10))


def timed(fn, iters=ITERS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_table():
    shapes = [("vit_qkv", 65536, 3072, 1024, ops.EPI_BIAS), ("vit_proj", 65536, 1024, 1024, ops.EPI_SCALE_RES),
              ("vit_fc1", 65536, 4096, 1024, ops.EPI_BIAS_GELU), ("vit_fc2", 65536, 1024, 4096, ops.EPI_SCALE_RES),
              ("llm_wqkv", 16384, 4096, 2048, ops.EPI_BIAS), ("llm_wo", 16384, 2048, 2048, ops.EPI_SCALE_RES),
              ("llm_w13", 17408, 16384, 2048, ops.EPI_SILU_MUL), ("llm_w2", 16384, 2048, 8192, ops.EPI_SCALE_RES),
              ("mlp1_1", 16384, 2048, 4096, ops.EPI_BIAS_GELU), ("square4k", 4096, 4096, 4096, ops.EPI_BIAS),
              ("square8k", 8192, 8192, 8192, ops.EPI_BIAS)]
    print(f"GEMM, bf16, random data, {ROUNDS} interleaved rounds x {ITERS} launches; TFLOP/s median (best)")
    print(f"{'shape':10s} {'M':>6s} {'N':>6s} {'K':>5s} | {'hand-written + epilogue':>24s} | {'hand-written, main loop only':>28s} | {'torch F.linear (vendor)':>24s} | vendor / main loop")
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        nout = N // 2 if epi == ops.EPI_SILU_MUL else N
        out = torch.empty(M, nout, device=dev, dtype=BF)
        bias = torch.randn(N, device=dev).to(BF) if epi != ops.EPI_SILU_MUL else None
        res = torch.randn(M, nout, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
        scale = torch.randn(N, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
        vout = torch.empty(M, N, device=dev, dtype=BF)

        def ours():
            ops.gemm(a, w, out, epi, bias=bias, scale=scale, res=res)

        def ours_noepi():
            ops.gemm_set_tile(1003)
            ops.gemm(a, w, out, epi, bias=bias, scale=scale, res=res)
            ops.gemm_set_tile(1000)

        def vendor():
            torch.mm(a, w.t(), out=vout)

        fns = [("ours", ours), ("noepi", ours_noepi), ("vendor", vendor)]
        ts = {k: [] for k, _ in fns}
        for rnd in range(ROUNDS):
            for k, fn in fns[rnd % 3:] + fns[:rnd % 3]:
                ts[k].append(timed(fn))
        fl = 2.0 * M * N * K / 1e9
        cell = lambda k: f"{fl / np.median(ts[k]):7.0f} ({fl / min(ts[k]):5.0f})"   # noqa: E731
        print(f"{name:10s} {M:6d} {N:6d} {K:5d} | {cell('ours'):>24s} | {cell('noepi'):>28s} | {cell('vendor'):>24s} | "
              f"{np.median(ts['noepi']) / np.median(ts['vendor']):.3f}", flush=True)
    ops.gemm_set_tile(0)


def attn_table():
    print(f"\nattention, bf16, random data, {ROUNDS} interleaved rounds x {ITERS} launches; TFLOP/s median (best)")
    for name, n_seq, L, H, G, D, causal, mode in (("vit  64 x 1025, 16 heads, D = 64", 64, 1025, 16, 1, 64, False, 0),
                                                  ("llm  8 x 2186 causal, 16 / 8 heads, D = 128", 8, 2186, 16, 2, 128, True, 1),
                                                  ("     64 x 1024, D = 64", 64, 1024, 16, 1, 64, False, 0),
                                                  ("     8 x 2048 causal, D = 128", 8, 2048, 16, 2, 128, True, 1)):
        N = n_seq * L
        q = torch.randn(N, H * D, device=dev).to(BF)
        k = torch.randn(N, (H // G) * D, device=dev).to(BF)
        v = torch.randn(N, (H // G) * D, device=dev).to(BF)
        o = torch.empty(N, H * D, device=dev, dtype=BF)
        cu = torch.arange(0, (n_seq + 1) * L, L, dtype=torch.int32, device=dev)
        q4 = q.view(n_seq, L, H, D).transpose(1, 2).contiguous()
        k4 = k.view(n_seq, L, H // G, D).transpose(1, 2).contiguous()
        v4 = v.view(n_seq, L, H // G, D).transpose(1, 2).contiguous()
        if G > 1:   # (GQA: expanded heads, as the reference's repeat_kv does before its attention call)
            k4e = k4.repeat_interleave(G, dim=1).contiguous()
            v4e = v4.repeat_interleave(G, dim=1).contiguous()
        else:
            k4e, v4e = k4, v4
        fns = [("ours", lambda: ops.attention(q, k, v, o, cu, L, H, G, D, causal, D ** -0.5, mode))]
        try:
            F.scaled_dot_product_attention(q4, k4e, v4e, is_causal=causal)
            torch.cuda.synchronize()
            fns.append(("sdpa", lambda: F.scaled_dot_product_attention(q4, k4e, v4e, is_causal=causal)))
        except Exception as e:   # noqa: BLE001
            print(f"  scaled_dot_product_attention failed at this shape: {type(e).__name__}: {e}")
        ts = {kk: [] for kk, _ in fns}
        for rnd in range(ROUNDS):
            for kk, fn in fns[rnd % len(fns):] + fns[:rnd % len(fns)]:
                ts[kk].append(timed(fn))
        fl = 4.0 * D * n_seq * H * L * L * (0.5 if causal else 1.0) / 1e9
        line = f"{name:46s}"
        for kk, _ in fns:
            line += f" | {kk}: {np.median(ts[kk]):7.3f} ms {fl / np.median(ts[kk]):6.0f} ({fl / min(ts[kk]):5.0f}) TF/s"
        print(line, flush=True)
    try:
        from torch.backends.cuda import flash_sdp_enabled, mem_efficient_sdp_enabled, math_sdp_enabled
        print(f"  sdpa backends enabled: flash {flash_sdp_enabled()}, mem_efficient {mem_efficient_sdp_enabled()}, math {math_sdp_enabled()}")
    except Exception:   # noqa: BLE001
        pass


print(torch.__version__, torch.cuda.get_device_name(0))
which = sys.argv[1:] or ["gemm", "attn"]
if "gemm" in which:
    gemm_table()
if "attn" in which:
    attn_table()
