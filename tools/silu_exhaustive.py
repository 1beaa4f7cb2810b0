#!/usr/bin/env python3
"""How often does the SiLU of the w1|w3 epilogue (x * v_rcp_f32(1 + exp(-x)), rounded to bf16 once) differ from torch's CPU
bf16 SiLU (an IEEE division)?  All finite bf16 inputs as the gate, up = 1, through the three tile kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mj_video_amd import ops  # noqa: E402

dev, BF = torch.device("cuda", 0), torch.bfloat16
vals = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(BF)
vals = vals[torch.isfinite(vals.float())]
M = vals.numel()
K = 64
a = torch.zeros(M, K, dtype=BF)
a[:, 0] = vals
a[:, 1] = 1.0
FFD = 16                                   # one 16-row group of gate rows and one of up rows
w1 = torch.zeros(FFD, K, dtype=BF); w1[:, 0] = 1.0
w3 = torch.zeros(FFD, K, dtype=BF); w3[:, 1] = 1.0
w13 = torch.stack([w1.view(1, 16, K), w3.view(1, 16, K)], dim=1).reshape(2 * FFD, K).contiguous()
ref = F.silu(vals)                          # bf16 CPU kernel (what the reference's act_fn computes), times up = 1
for tile in (64, 128, 256):
    out = torch.empty(M, FFD, dtype=BF, device=dev)
    ops.gemm(a.to(dev), w13.to(dev), out, ops.EPI_SILU_MUL, tile=tile)
    got = out[:, 0].cpu()
    normal = (vals.float().abs() >= 2.0 ** -125) & (vals.float().abs() < 2.0 ** 127)
    diff = (got.view(torch.int16) != ref.view(torch.int16)) & ~((got.float() == 0) & (ref.float() == 0)) & normal
    print(f"tile {tile}: {int(diff.sum())} of {int(normal.sum())} inputs differ from torch's bf16 SiLU; "
          f"largest difference {int((got.view(torch.int16).int() - ref.view(torch.int16).int()).abs()[diff].max()) if diff.any() else 0} bf16 ulp")
    if diff.any():
        i = diff.nonzero().view(-1)[:6]
        print("   e.g. x =", vals[i].float().tolist(), "ours", got[i].float().tolist(), "torch", ref[i].float().tolist())
