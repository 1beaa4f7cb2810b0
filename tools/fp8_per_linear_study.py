#!/usr/bin/env python3
"""Per-Linear sensitivity of the attention side to MXFP8 operands.  Round 4 measured all four attention-side
Linears at once (1 flip of 478, rms 11.8 x the reference's bf16 noise against 7.8 x for the FFN-only path) and stopped there.
Here each choice separately, through the UNFUSED measurement launches of ``model._exp_fp8_attn_side`` (a set of names):
  proj + wo   - the two output projections: they feed the residual stream, not a softmax
  qkv         - the vision tower's q / k / v projection
  wqkv        - the language tower's
on the engineered rank set @224^2 against the REFERENCE's bf16 scores: rms deviation in units of its bf16-vs-fp32 noise, flips
on the decisive pairs, Spearman.  A fused kernel is worth writing only for a subset that keeps 0 flips and rho >= 0.999 - or
at least what the FFN-only path has."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scipy.stats import spearmanr
import test_e2e_gpu as T
from util import load_golden

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "rankset_c1"
enpz, emeta = load_golden(name.replace("rankset", "rankeng"))
ref, keep = enpz["ref_bf16"], enpz["keep"]
f32, idx32 = enpz["ref_fp32"], enpz["fp32_pairs"]
noise = float(np.sqrt(((ref[idx32][..., 0] - f32[..., 0]) ** 2).mean()))
orig = T.build_hip_model
CASES = (("bf16", "bf16", False), ("mxfp8 FFN (the opt-in path)", "mxfp8", False), ("+ proj + wo", "mxfp8", {"proj", "wo"}),
         ("+ proj only", "mxfp8", {"proj"}), ("+ wo only", "mxfp8", {"wo"}), ("+ qkv only", "mxfp8", {"qkv"}),
         ("+ wqkv only", "mxfp8", {"wqkv"}), ("+ all four", "mxfp8", True))
for label, fmt, flag in CASES:
    def patched(cfg, sd, d, _f=flag):
        m = orig(cfg, sd, d)
        m._exp_fp8_attn_side = _f
        return m
    T.build_hip_model = patched
    T._RANK_CACHE.clear()
    run = T._rank_run(dev, name, 8 if name.endswith("c1") else 4, ffn_format=fmt)
    got = run["eng"][: ref.shape[0]]
    d = (got[..., 0] - ref[..., 0]).ravel()
    rms = float(np.sqrt((d ** 2).mean()))
    agree = np.sign(got[:, 0, 0] - got[:, 1, 0]) == np.sign(ref[:, 0, 0] - ref[:, 1, 0])
    rho = spearmanr(got[..., 0].ravel(), ref[..., 0].ravel()).correlation
    margins = np.abs(ref[:, 0, 0] - ref[:, 1, 0])
    flipped = margins[keep & ~agree]
    print(f"{label:30s} |hip - ref| rms {rms:.5f} = {rms / noise:5.1f} x the bf16 noise ({noise:.5f}), max {np.abs(d).max():.4f}; "
          f"decisive pairs {int(keep.sum())}: {int((~agree[keep]).sum())} flips"
          f"{' (reference margins ' + ', '.join(f'{m:.3f}' for m in sorted(flipped)) + ')' if flipped.size else ''}; all {len(agree)} pairs "
          f"{agree.mean():.5f}; spearman {rho:.6f}", flush=True)
