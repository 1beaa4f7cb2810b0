#!/usr/bin/env python3
"""Exactness screen of the GEMM kernel variants (tile codes 1000 + v) on integer data + every epilogue vs variant 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mj_video_amd import ops
BF = torch.bfloat16; dev = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [1000, 1002]
for var in variants:
    ops.gemm_set_tile(var)
    ok = True
    for (M, N, K) in [(513, 520, 64), (700, 256, 128), (256, 1024, 320), (2200, 768, 2048), (4133, 2048, 1024), (70000, 1024, 256), (17488, 2048, 512)]:
        g = torch.Generator().manual_seed(5)
        a = torch.randint(-3, 4, (M, K), generator=g).float().to(BF).to(dev)
        w = torch.randint(-2, 3, (N, K), generator=g).float().to(BF).to(dev)
        ref = (a.float() @ w.float().t()).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        for it in range(4):
            out.zero_(); ops.gemm(a, w, out, ops.EPI_BIAS)
            good = torch.equal(out, ref)
            if not good:
                print("  MISMATCH", var, (M, N, K), int((out != ref).sum()))
            ok &= good
    print("variant", var, "integer-exact:", ok, flush=True)
# epilogues: every variant must be bit-identical to variant 0 on random data
M, N, K = 3000, 1024, 256
a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.1).to(BF)
b = torch.randn(N, device=dev).to(BF); ls = torch.randn(N, device=dev).to(BF); res = torch.randn(M, N, device=dev).to(BF)
outs = {}
for var in variants:
    ops.gemm_set_tile(var)
    o = []
    for epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU, ops.EPI_BIAS_RELU):
        t = torch.empty(M, N, dtype=BF, device=dev); ops.gemm(a, w, t, epi, bias=b); o.append(t)
    t = res.clone(); ops.gemm(a, w, t, ops.EPI_SCALE_RES, bias=b, scale=ls, res=t); o.append(t)
    t = torch.empty(M, N // 2, dtype=BF, device=dev); ops.gemm(a, w, t, ops.EPI_SILU_MUL); o.append(t)
    outs[var] = o
for var in variants[1:]:
    print("variant", var, "epilogues identical to", variants[0], [torch.equal(x, y) for x, y in zip(outs[var], outs[variants[0]])])
ops.gemm_set_tile(1000)
