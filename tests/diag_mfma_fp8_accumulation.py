"""How exact is the accumulation inside v_mfma_scale_f32_16x16x128_f8f6f4?  Random MXFP8 operands through mjv_gemm_bf16 (MXFP8
operands, bf16 output), the exact sum of the dequantised operands in fp64, and the part of the error that one bf16 ulp of the
output does not explain, relative to sum|a||w| (profiles/r04_c_mfma_fp8_accumulation.txt; a calibration script of the fp8 tests: it lives under tests/ because it calls the oracle)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mj_video_amd import ops
from oracle import ref_fp8
BF = torch.bfloat16
torch.manual_seed(0)
for K in (128, 1024):
    for astd, wstd in ((1.0, 0.05), (1.0, 1.0)):
        M = N = 256
        a = (torch.randn(M, K) * astd).to(BF); w = (torch.randn(N, K) * wstd).to(BF)
        a8, w8 = ops.quantize_mxfp8(a.cuda()), ops.quantize_mxfp8(w.cuda())
        aq, wq = ref_fp8.mx_fake_quant(a).double(), ref_fp8.mx_fake_quant(w).double()
        S = aq @ wq.t(); T = aq.abs() @ wq.abs().t()
        out = torch.empty(M, N, dtype=BF, device="cuda"); ops.gemm(a8, w8, out, ops.EPI_BIAS); torch.cuda.synchronize()
        got = out.double().cpu()
        err = (got - S).abs()
        excess = (err - S.abs() * 2.0 ** -8).clamp_min(0)     # beyond a full bf16 ulp of the exact sum
        rel = (excess / T)
        # the same sum in float32 sequential order
        S32 = (aq.float() @ wq.float().t()).double()
        print(f"K={K} std {astd}/{wstd}: max excess error / sum|a||w| = {rel.max().item():.3e} (2^{np.log2(max(rel.max().item(),1e-30)):.1f}); "
              f"fraction of outputs with excess > 2^-16 T: {(rel > 2.0**-16).float().mean().item():.4f}; "
              f"torch fp32 matmul vs exact: {((S32 - S).abs() / T).max().item():.2e}")
