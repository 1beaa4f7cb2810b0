#!/usr/bin/env python3
"""What would an fp8 (OCP e4m3) MFMA weight path cost in parity?  

Build-container study, not a fixture generator and not product code: runs the CPU oracle on videos of the engineered
rank set with the operands of the trunk GEMMs (ViT qkv/proj/fc1/fc2, LLM wqkv/wo/w1/w3/w2) rounded to e4m3 the way an
fp8-MFMA kernel would see them (weights: per-output-channel scale; activations: per-tensor scale; products and sums
exact in fp32, output rounded to bf16 as today), and compares the scores with the reference bf16 scores stored in
``rankeng_c1.npz``.  The bf16 path's own deviation from those scores (|hip - ref| rms 0.0087 = the reference's
bf16-vs-fp32 noise 0.0079) is the yardstick.

Usage: python tests/golden/fp8_study.py [--pairs 8] [--mode all|no_attn_out|weights_only]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import mj_video_amd  # noqa: E402,F401
from mj_video_amd import synth  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from make_golden import make_cfg, n_img_tokens, _from_bits  # noqa: E402

E4M3_MAX = 448.0


def fq_weight(w: torch.Tensor) -> torch.Tensor:
    s = w.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / E4M3_MAX
    return ((w.float() / s).to(torch.float8_e4m3fn).float() * s)


def fq_act(x: torch.Tensor) -> torch.Tensor:
    s = x.float().abs().amax().clamp_min(1e-12) / E4M3_MAX * 2.0     # static scale with a 2x margin
    return ((x.float() / s).to(torch.float8_e4m3fn).float() * s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--mode", default="all", choices=["all", "no_attn_out", "weights_only", "ffn_only"])
    a = ap.parse_args()
    fx = np.load(os.path.join(HERE, "rankeng_c1.npz"))
    meta = json.load(open(os.path.join(HERE, "rankeng_c1.json")))
    S = meta["image_size"]
    cd, hk, cfg = make_cfg("2b", S)
    sd = synth.synth_state_dict(cfg, seed=meta["weight_seed"])
    sd.update(synth.engineered_head_state_dict(cfg, meta["weight_seed"], _from_bits(fx["regression_weight_bits"]),
                                               torch.from_numpy(fx["gate_dirs"])))
    names = ["attn.qkv.weight", "mlp.fc1.weight", "mlp.fc2.weight", "attention.wqkv.weight",
             "feed_forward.w1.weight", "feed_forward.w3.weight", "feed_forward.w2.weight"]
    if a.mode in ("all", "weights_only"):
        names += ["attn.proj.weight", "attention.wo.weight"]
    if a.mode == "ffn_only":
        names = ["mlp.fc1.weight", "mlp.fc2.weight", "feed_forward.w1.weight", "feed_forward.w3.weight",
                 "feed_forward.w2.weight"]
    table = {}
    for k, w in sd.items():
        if any(k.endswith(n) for n in names) and ("vision_model.encoder" in k or "language_model.model.layers" in k):
            table[id(w)] = fq_weight(w)
    print(f"{len(table)} weight matrices in e4m3 ({a.mode})", flush=True)
    plain = F.linear

    def linear(x, w, b=None):
        wq = table.get(id(w))
        if wq is None:
            return plain(x, w, b)
        xq = x.float() if a.mode == "weights_only" else fq_act(x)
        y = plain(xq, wq)                                   # fp32 accumulate of exact products
        if b is not None:
            y = y + b.float()
        return y.to(x.dtype)

    ref_cpu.F.linear = linear
    dev, ref_scores, got_scores = [], [], []
    for p in range(a.pairs):
        ids = synth.synth_input_ids(n_img_tokens(cfg, meta["n_tiles"]), caption_seed=meta["caption_seed_base"] + p)
        mask = torch.ones_like(ids)
        for j in range(2):
            px = synth.synth_pixel_values(meta["pixel_seed"], 2 * p + j, meta["n_tiles"], S)
            r = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID)
            s, s_ref = r["score"].item(), float(fx["ref_bf16"][p, j, 0])
            got_scores.append(s)
            ref_scores.append(s_ref)
            dev.append(s - s_ref)
        m_ref = ref_scores[-2] - ref_scores[-1]
        m_got = got_scores[-2] - got_scores[-1]
        print(f"pair {p}: ref {ref_scores[-2]:+.4f} {ref_scores[-1]:+.4f}  fp8 {got_scores[-2]:+.4f} {got_scores[-1]:+.4f}  "
              f"margin ref {m_ref:+.4f} fp8 {m_got:+.4f} {'same' if m_ref * m_got > 0 else 'FLIPPED'}", flush=True)
    dev = np.array(dev)
    print(f"mode {a.mode}: score deviation rms {np.sqrt((dev ** 2).mean()):.4f} max {np.abs(dev).max():.4f} over {len(dev)} videos; "
          f"bf16 yardstick: noise rms {meta['noise_rms']:.4f}, score spread {meta['score_spread']:.3f}")


if __name__ == "__main__":
    main()
