#!/usr/bin/env python3
"""Which of the four FFN Linears carries the fp8 path's rank noise.  The full MXFP8 FFN set (fc1, fc2 of
the vision tower, w1|w3, w2 of the language tower) sits at Spearman 0.9986 / 0.9979 against the reference's bf16 scores on the
engineered rank sets - under north_star's 0.999.  Round 5 only ADDED Linears to that set.  Here every non-empty subset of the
four (``model.set_ffn_format("mxfp8:<a>+<b>")``: product code path, the Linears outside the subset run the bf16 kernels) on
both engineered sets (@224^2: 512 pairs / 478 decisive; @448^2: 256 / 240): rms deviation in units of the reference's own
bf16-vs-fp32 noise, flips on the decisive pairs, Spearman over all scores.  The largest subset with rho >= 0.999 and 0 flips on
BOTH sets becomes a named preset (modeling.FP8_PRESETS); if none passes, this table is the deliverable.

usage: fp8_ffn_subset_study.py [rankset_c1|rankset_c2|both] [subset ...]      (a subset = fc1+w13; default: all 15)"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scipy.stats import spearmanr
import test_e2e_gpu as T
from util import load_golden

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
names = ("rankset_c1", "rankset_c2") if which == "both" else (which,)
LIN = ("fc1", "fc2", "w13", "w2")
subsets = [tuple(a.split("+")) for a in sys.argv[2:]] or [c for r in (1, 2, 3, 4) for c in itertools.combinations(LIN, r)]
table = {}
for name in names:
    enpz, emeta = load_golden(name.replace("rankset", "rankeng"))
    ref, keep = enpz["ref_bf16"], enpz["keep"]
    f32, idx32 = enpz["ref_fp32"], enpz["fp32_pairs"]
    noise = float(np.sqrt(((ref[idx32][..., 0] - f32[..., 0]) ** 2).mean()))
    print(f"== {name}: {ref.shape[0]} pairs, {int(keep.sum())} decisive; reference bf16-vs-fp32 noise rms {noise:.5f}", flush=True)
    for sub in [None] + subsets:
        fmt = "bf16" if sub is None else "mxfp8:" + "+".join(sub)
        T._RANK_CACHE.clear()
        run = T._rank_run(dev, name, 8 if name.endswith("c1") else 4, ffn_format=fmt)
        got = run["eng"][: ref.shape[0]]
        d = (got[..., 0] - ref[..., 0]).ravel()
        rms = float(np.sqrt((d ** 2).mean()))
        agree = np.sign(got[:, 0, 0] - got[:, 1, 0]) == np.sign(ref[:, 0, 0] - ref[:, 1, 0])
        rho = spearmanr(got[..., 0].ravel(), ref[..., 0].ravel()).correlation
        flips = int((~agree[keep]).sum())
        margins = np.abs(ref[:, 0, 0] - ref[:, 1, 0])[keep & ~agree]
        table.setdefault(fmt, {})[name] = (rms / noise, flips, rho)
        print(f"{fmt:24s} |hip - ref| rms {rms:.5f} = {rms / noise:5.2f} x noise, max {np.abs(d).max():.4f}; decisive flips {flips}"
              f"{' (margins ' + ', '.join(f'{m:.3f}' for m in sorted(margins)) + ')' if flips else ''}; all pairs {agree.mean():.5f}; "
              f"spearman {rho:.6f} {'PASS' if rho >= 0.999 and flips == 0 else ''}", flush=True)
if len(names) == 2:
    print("\n== subsets that hold rho >= 0.999 and 0 decisive flips on BOTH sets (largest first)")
    ok = [(f, v) for f, v in table.items() if f != "bf16" and all(v[n][2] >= 0.999 and v[n][1] == 0 for n in names)]
    for f, v in sorted(ok, key=lambda t: -t[0].count("+")):
        print(f"{f:24s} " + "; ".join(f"{n}: {v[n][0]:.2f} x noise, rho {v[n][2]:.6f}" for n in names))
    if not ok:
        print("(none)")
