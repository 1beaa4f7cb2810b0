#!/usr/bin/env python3
"""Does a forward read scratch memory it has not written?  Runs the same batch twice and, between the two, fills the model's
scratch buffers (model._ws) - all of them, then one at a time - with NaN bit patterns: any field that changes (or turns NaN)
names a buffer whose stale contents reach the result.  (Found the cause of round 5's history-dependent test failures.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from mj_video_amd import synth
from util import FIELDS, build_hip_model, case_inputs, make_cfg

dev = torch.device("cuda:0")
cfg = make_cfg("2b", 224)
sd = synth.synth_state_dict(cfg, seed=0, lm_head=False)
sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
model = build_hip_model(cfg, sd, dev)
for k in sys.argv[1:]:
    name, val = k.split("=")
    setattr(model, name, {"True": True, "False": False}.get(val, val))
vids = [dict(video_idx=i, n_tiles=t, caption_seed=i) for i, t in enumerate([8, 6, 8, 3, 5])]
px, ids, mask, _ = case_inputs(cfg, vids, 77, 224)
px, ids, mask = px.to(dev), ids.to(dev), mask.to(dev)


def poison(names):
    for n in names:
        t = model._ws[n]
        if not torch.is_tensor(t) or t.dtype == torch.int32:
            continue
        t.view(torch.uint8).fill_(0xFF)          # bf16 / fp32 NaN patterns, e4m3 NaN, scale byte 255


def run():
    o = model.forward(px, ids.clone(), mask.clone())
    torch.cuda.synchronize()
    return {f: getattr(o, f).clone() for f in FIELDS}


ref = run()
ref2 = run()
print("repeat without poison identical:", all(torch.equal(ref[f], ref2[f]) for f in FIELDS))
names = [n for n in model._ws if isinstance(n, str) and "gemm_ws" not in n]
poison(names)
got = run()
bad = [f for f in FIELDS if not torch.equal(ref[f], got[f])]
print("all scratch poisoned ->", "identical" if not bad else f"fields differ: {bad}; NaN in score: {bool(torch.isnan(got['score']).any())}")
if bad:
    for n in names:
        run()
        poison([n])
        g = run()
        b = [f for f in FIELDS if not torch.equal(ref[f], g[f])]
        if b:
            print(f"  poisoning {n!r} alone changes {b}")
# the same for the split-K workspace
run()
poison([n for n in model._ws if isinstance(n, str) and "gemm_ws" in n])
g = run()
print("split-K workspace poisoned ->", "identical" if all(torch.equal(ref[f], g[f]) for f in FIELDS) else "DIFFERS")
