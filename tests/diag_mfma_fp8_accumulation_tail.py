#!/usr/bin/env python3
"""The TAIL of the accumulation error of v_mfma_scale_f32_16x16x128_f8f6f4 (tests/diag_mfma_fp8_accumulation.py looked at 65 k outputs
per setting; tools/fuzz_kernels.py found one output in 3.7e11 a little beyond the bound that sample suggested).  Here: ~1e9 outputs
per setting, exact reference in fp64 on the GPU, the part of the error one bf16 ulp does not explain relative to (a) sum|a||w| and
(b) 128 x the largest |a w| product of the row - the quantity a fixed-point adder aligned to the largest term would be bounded by."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mj_video_amd import ops
sys.path.insert(0, os.path.join(ROOT, "tools"))
BF = torch.bfloat16
dev = torch.device("cuda:0")
torch.manual_seed(11)


from oracle import ref_fp8
for K, astd, wstd, reps in ((128, 1.0, 0.08, 120), (128, 1.0, 1.0, 60), (1024, 1.0, 0.05, 12)):
    M = N = 2048
    worst_T, worst_P, n_cells, over15, over16 = 0.0, 0.0, 0, 0, 0
    for rep in range(reps):
        a = (torch.randn(M, K, device=dev) * astd).to(BF); w = (torch.randn(N, K, device=dev) * wstd).to(BF)
        a8, w8 = ops.quantize_mxfp8(a), ops.quantize_mxfp8(w)
        aq = ref_fp8.mx_fake_quant_fast(a).double(); wq = ref_fp8.mx_fake_quant_fast(w).double()   # (runs on the GPU: plain torch ops)
        S = aq @ wq.t(); T = aq.abs() @ wq.abs().t()
        out = torch.empty(M, N, dtype=BF, device=dev); ops.gemm(a8, w8, out, ops.EPI_BIAS); torch.cuda.synchronize()
        err = (out.double() - S).abs()
        excess = (err - S.abs() * 2.0 ** -8).clamp_min(0)
        relT = excess / T
        worst_T = max(worst_T, relT.max().item())
        over15 += int((relT > 2.0 ** -15).sum()); over16 += int((relT > 2.0 ** -16).sum())
        n_cells += M * N
        # largest single product per output: max_k |a_mk w_nk| <= max|a_m| max|w_n| (bound), exact for the worst cell only
        i = int(relT.argmax()); m, n = i // N, i % N
        pmax = (aq[m].abs() * wq[n].abs()).max().item()
        worst_P = max(worst_P, excess[m, n].item() / (K * pmax))
    print(f"K={K} std {astd}/{wstd}: {n_cells:.2e} outputs: max excess / sum|a||w| = 2^{np.log2(worst_T):.2f}; outputs beyond 2^-16: {over16}, beyond 2^-15: {over15}; "
          f"worst cell's excess / (K x its largest product) = 2^{np.log2(max(worst_P, 1e-30)):.2f}", flush=True)
