#!/usr/bin/env python3
"""Is fusing the epilogue into the hand-written GEMM worth it? (tools only, bench build of the library.)
Round 4 showed the vendor's UN-fused GEMM beating the fused hand-written kernel on five model shapes and answered "the vendor
would still owe the elementwise pass" without measuring it.  Here both sides of that sentence, same box, same data, interleaved:
  fused    = mjv_gemm_bf16 with the production epilogue (one launch)
  unfused  = torch.mm (hipBLASLt / rocBLAS, bf16 in, bf16 out) + mjv_bench_epilogue_pass, the cheapest correct elementwise kernel
             (16-byte streaming loads / stores, GELU table in LDS; same operations and rounding points as the fused epilogue)
for proj, fc1, fc2 (65 536 vision rows), wo, w2 (16 384 language rows).  The pass alone is reported with its GB/s.  A shape where
unfused wins is a target with that number; where fused wins everywhere the topic is closed."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MJV_LIBRARY", os.path.join(ROOT, "mj-video_amd", "libmjv_hip_bench.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mj_video_amd import ops, _lib  # noqa: E402

dev, BF = torch.device("cuda:0"), torch.bfloat16
ROUNDS = int(os.environ.get("MJV_BENCH_ROUNDS", 7))
ITERS = int(os.environ.get("MJV_BENCH_ITERS", 10))
lib = _lib.load_library()
assert _lib.is_bench_build(), "needs the bench build (make -C mj-video_amd/csrc bench)"


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


def epilogue_pass(x, y, bias, scale, res, kind):
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())   # noqa: E731
    _lib.check(lib.mjv_bench_epilogue_pass(p(x), x.stride(0), p(y), y.stride(0), p(bias), p(scale), p(res),
                                           0 if res is None else res.stride(0), x.shape[0], x.shape[1], kind,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mjv_bench_epilogue_pass")


shapes = [("vit_proj", 65536, 1024, 1024, ops.EPI_SCALE_RES, True, True), ("vit_fc1", 65536, 4096, 1024, ops.EPI_BIAS_GELU, True, False),
          ("vit_fc2", 65536, 1024, 4096, ops.EPI_SCALE_RES, True, True), ("llm_wo", 16384, 2048, 2048, ops.EPI_SCALE_RES, False, False),
          ("llm_w2", 16384, 2048, 8192, ops.EPI_SCALE_RES, False, False)]
print(f"fused hand-written GEMM vs vendor GEMM + elementwise pass; bf16, random data, {ROUNDS} interleaved rounds x {ITERS} launches; "
      f"ms median (best)")
print(f"{'shape':9s} {'M':>6s} {'N':>5s} {'K':>5s} | {'fused':>15s} | {'vendor mm':>15s} | {'pass':>15s} {'GB/s':>6s} | {'mm + pass':>15s} | "
      f"unfused / fused")
for name, M, N, K, epi, has_bias, has_scale in shapes:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    bias = torch.randn(N, device=dev).to(BF) if has_bias else None
    scale = torch.randn(N, device=dev).to(BF) if has_scale else None
    res = torch.randn(M, N, device=dev).to(BF) if epi == ops.EPI_SCALE_RES else None
    out_f = torch.empty(M, N, device=dev, dtype=BF)
    lin = torch.empty(M, N, device=dev, dtype=BF)
    out_u = torch.empty(M, N, device=dev, dtype=BF)
    kind = 1 if epi == ops.EPI_BIAS_GELU else 3

    def fused():
        ops.gemm(a, w, out_f, epi, bias=bias, scale=scale, res=res)

    def mm():
        torch.mm(a, w.t(), out=lin)

    def pass_only():
        epilogue_pass(lin, out_u, bias, scale, res, kind)

    def unfused():
        mm()
        pass_only()

    # the two paths compute the same function: identical up to the fp32 summation order of the two GEMMs
    fused(); unfused(); torch.cuda.synchronize()
    # (the un-fused path rounds the Linear to bf16 BEFORE the bias - the vendor GEMM's output - and again after it: one rounding
    # more than the fused epilogue and the reference; with a LayerScale of |s| up to 4 and a residual that can cancel the term, the
    # honest yardstick is the magnitude of the terms, not of their sum)
    diff = (out_f.float() - out_u.float()).abs()
    mag = out_f.float().abs() + (res.float().abs() if res is not None else 0.0)
    ok = bool((diff <= 0.05 + 0.03 * mag).all())
    fns = [("fused", fused), ("mm", mm), ("pass", pass_only), ("unfused", unfused)]
    ts = {k: [] for k, _ in fns}
    for rnd in range(ROUNDS):
        for k, fn in fns[rnd % 4:] + fns[:rnd % 4]:
            ts[k].append(timed(fn))
    cell = lambda k: f"{np.median(ts[k]):6.3f} ({min(ts[k]):6.3f})"   # noqa: E731
    gbs = (4.0 if kind == 1 else 6.0) * M * N / np.median(ts["pass"]) / 1e6
    print(f"{name:9s} {M:6d} {N:5d} {K:5d} | {cell('fused'):>15s} | {cell('mm'):>15s} | {cell('pass'):>15s} {gbs:6.0f} | {cell('unfused'):>15s} | "
          f"{np.median(ts['unfused']) / np.median(ts['fused']):.3f}{'' if ok else '   (RESULTS DIFFER)'}", flush=True)
