#!/usr/bin/env python3
"""What would MXFP8 operands on ALL trunk Linears (the four attention-side ones on top of the five FFN ones) do to the scores?
Runs the engineered rank set @224^2 on the GPU three ways - bf16, mxfp8 FFN (the product's opt-in path), mxfp8 FFN + attention-side
Linears through unfused launches (``model._exp_fp8_attn_side``, measurement only) - against the reference's bf16 scores.
Decides whether the fused kernels of an all-Linear fp8 path (RoPE epilogue on fp8 operands, MXFP8 output of the attention
kernel) are worth writing (DESIGN §4 "The fp8 weight path")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scipy.stats import spearmanr
import test_e2e_gpu as T
from util import load_golden

dev = torch.device("cuda:0")
enpz, emeta = load_golden("rankeng_c1")
ref, keep = enpz["ref_bf16"], enpz["keep"]
f32, idx32 = enpz["ref_fp32"], enpz["fp32_pairs"]
noise = float(np.sqrt(((ref[idx32][..., 0] - f32[..., 0]) ** 2).mean()))
orig = T.build_hip_model
for label, fmt, flag in (("bf16", "bf16", False), ("mxfp8 FFN", "mxfp8", False), ("mxfp8 FFN + attention-side Linears", "mxfp8", True)):
    def patched(cfg, sd, d, _f=flag):
        m = orig(cfg, sd, d)
        m._exp_fp8_attn_side = _f
        return m
    T.build_hip_model = patched
    T._RANK_CACHE.clear()
    run = T._rank_run(dev, "rankset_c1", 8, ffn_format=fmt)
    got = run["eng"][: ref.shape[0]]
    d = (got[..., 0] - ref[..., 0]).ravel()
    rms = float(np.sqrt((d ** 2).mean()))
    agree = np.sign(got[:, 0, 0] - got[:, 1, 0]) == np.sign(ref[:, 0, 0] - ref[:, 1, 0])
    rho = spearmanr(got[..., 0].ravel(), ref[..., 0].ravel()).correlation
    margins = np.abs(ref[:, 0, 0] - ref[:, 1, 0])
    flipped = margins[keep & ~agree]
    print(f"{label:38s} |hip - ref| rms {rms:.5f} = {rms / noise:5.1f} x the bf16 noise ({noise:.5f}), max {np.abs(d).max():.4f}; "
          f"decisive pairs {int(keep.sum())}: agreement {agree[keep].mean():.5f} ({int((~agree[keep]).sum())} flips"
          f"{', reference margins ' + ', '.join(f'{m:.3f}' for m in sorted(flipped)) if flipped.size else ''}); all {len(agree)} pairs "
          f"{agree.mean():.5f}; spearman {rho:.6f}", flush=True)
