#!/usr/bin/env python3
"""Prints |hip - ref| of every video of the full_c2 fixture with split-K on / off (diagnostic for the tolerance margin)."""
import sys, os
os.environ.setdefault("MJV_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mj-video_amd", "libmjv_hip_bench.so"))   # bench build: make -C mj-video_amd/csrc bench
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from util import load_golden, make_cfg, build_hip_model, case_inputs
from mj_video_amd import ops, synth
tag, S = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("full_c2", 448)
npz, meta = load_golden(tag)
cfg = make_cfg("2b", S)
sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"], lm_head=False)
sd["model.language_model.output.weight"] = torch.zeros(1, dtype=torch.bfloat16).expand(cfg.llm_config.vocab_size, cfg.llm_config.hidden_size)
model = build_hip_model(cfg, sd, "cuda")
px, ids, mask, _ = case_inputs(cfg, meta["videos"], meta["pixel_seed"], S)
for split in (8, 4, 3, 2, 0):
    ops.gemm_set_tile(4001 if split else 4000)
    if split:
        ops.gemm_set_tile(4100 + split)
    for probes in (False,):
        model.debug_probes = {} if probes else None
        out = model.forward(px.cuda(), ids.cuda(), mask.cuda())
        torch.cuda.synchronize()
        for i, v in enumerate(meta["videos"]):
            p = f"v{v['video_idx']}"
            ref = float(npz[f"{p}/score"][0]); f32 = float(npz[f"{p}/fp32/score"][0]) if f"{p}/fp32/score" in npz.files else float("nan")
            got = float(out.score[i])
            print(f"split={split} probes={probes} {p}: hip={got:+.4f} ref_bf16={ref:+.4f} ref_fp32={f32:+.4f}  |hip-ref|={abs(got-ref):.4f} |hip-fp32|={abs(got-f32):.4f} |ref-fp32|={abs(ref-f32):.4f}")
ops.gemm_set_tile(4001); ops.gemm_set_tile(4108)
