#!/usr/bin/env python3
"""Random-data comparison of split-K vs unsplit 128-tile GEMM launches (ulp statistics) on the model's tail shapes."""
import sys, os
os.environ.setdefault("MJV_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mj-video_amd", "libmjv_hip_bench.so"))   # bench build: make -C mj-video_amd/csrc bench
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mj_video_amd import ops
BF = torch.bfloat16
dev = "cuda"
ws = torch.empty(ops.gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
torch.manual_seed(0)
for (M, N, K, m_off) in [(16, 1024, 4096, 0), (64, 1024, 4096, 0), (1104, 2048, 8192, 0), (16400, 1024, 4096, 0), (17488, 2048, 8192, 0)]:
    a = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.03).to(BF)
    bias = (torch.randn(N, device=dev) * 0.1).to(BF)
    ls = (torch.randn(N, device=dev) * 0.3).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    outs = []
    for use in (False, True):
        ops.set_gemm_workspace(ws if use else None)
        x = res.clone()
        ops.gemm(a, w, x, ops.EPI_SCALE_RES, bias=bias, scale=ls, res=x)
        outs.append(x.float())
    ops.set_gemm_workspace(None)
    ref = (res.float() + ((a.float() @ w.float().t() + bias.float()).to(BF).float() * ls.float()).to(BF).float()).to(BF).float()
    d = (outs[0] - outs[1]).abs()
    rows_changed = (d.max(dim=1).values > 0).sum().item()
    print(f"M={M} N={N} K={K}: split vs unsplit: differing elements {int((d>0).sum())} / {d.numel()} in {rows_changed} rows, max abs diff {d.max().item():.4g};"
          f"  vs fp32 ref: unsplit max {((outs[0]-ref).abs().max().item()):.4g}, split max {((outs[1]-ref).abs().max().item()):.4g}")
