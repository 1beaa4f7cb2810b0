#!/usr/bin/env python3
"""Exactness screen for an experimental 256-tile kernel variant (tile code 1000 + v): exact-integer problems of the model's
shapes, repeated; every run must equal the production kernel's result bit for bit.  usage: var_check.py <v> [repeats]"""
import os, sys
os.environ.setdefault("MJV_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mj-video_amd", "libmjv_hip_bench.so"))   # bench build: make -C mj-video_amd/csrc bench
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mj_video_amd
from mj_video_amd import ops

v = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(5)
bad = 0
for (M, N, K, epi) in [(4133, 2048, 8192, ops.EPI_SCALE_RES), (8200, 1024, 1024, ops.EPI_SCALE_RES), (8200, 4096, 1024, ops.EPI_BIAS_GELU),
                       (17488, 2048, 2048, ops.EPI_SCALE_RES), (2186, 16384, 2048, ops.EPI_SILU_MUL), (8200, 3072, 1024, ops.EPI_BIAS),
                       (2304, 1024, 4096, ops.EPI_SCALE_RES), (512, 256, 64, ops.EPI_BIAS), (768, 512, 128, ops.EPI_BIAS), (600, 520, 192, ops.EPI_BIAS_RELU)]:
    a = (torch.randint(-2, 3, (M, K), generator=g).float() * (torch.rand(M, K, generator=g) < 0.15)).to(BF).to(dev)
    w = torch.randint(-2, 3, (N, K), generator=g).float().to(BF).to(dev)
    nout = N // 2 if epi == ops.EPI_SILU_MUL else N
    bias = None if epi == ops.EPI_SILU_MUL else torch.randint(-2, 3, (N,), generator=g).float().to(BF).to(dev)
    res = torch.randint(-8, 9, (M, nout), generator=g).float().to(BF).to(dev) if epi == ops.EPI_SCALE_RES else None
    ops.gemm_set_tile(256)
    ref = torch.empty(M, nout, dtype=BF, device=dev)
    ops.gemm(a, w, ref, epi, bias=bias, res=res)
    ops.gemm_set_tile(1000 + v)
    ops.gemm_set_tile(256) if False else None
    for it in range(reps):
        out = torch.full((M, nout), 7.0, dtype=BF, device=dev)
        ops.gemm_set_tile(256); ops.gemm_set_tile(1000 + v)   # 1000 + v keeps the automatic tile choice: force 256 via M >= 512 shapes
        ops.gemm(a, w, out, epi, bias=bias, res=res)
        n_bad = int((out != ref).sum())
        if n_bad:
            bad += 1
            print(f"M={M} N={N} K={K} epi={epi} run {it}: {n_bad} elements differ", flush=True)
    ops.gemm_set_tile(0)
print("variant", v, "bad runs:", bad)
sys.exit(1 if bad else 0)
