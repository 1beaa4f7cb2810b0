#!/usr/bin/env python3
"""Golden vectors of BASELINE configs[4]'s backbone (InternVL2-4B = InternViT + Phi-3-mini) - build-container only.

The reference cannot build that model (modeling_internvl_chat.py:125-130), so "the reference" here is the reference's OWN
reward-model / chat-model code around the decoder's authoritative implementation - transformers' ``Phi3ForCausalLM``
(transformers 5.15, imported here, never shipped) handed in through the ``language_model`` argument:
oracle/reference_shim.py:build_reference_model_phi3.  For every case this script
  1. builds seed-defined weights / pixels / token ids with mj_video_amd.synth (the GPU box regenerates them from the seeds),
  2. runs that model on CPU in bf16 (eager attention),
  3. runs oracle/ref_phi3.py on the same inputs and REQUIRES bit-identical outputs - this pins the oracle,
  4. runs the oracle in fp32 for the reference's own bf16 noise floor,
  5. stores outputs (+ probes) under tests/golden/phi3_*.npz.

Usage:  python tests/golden/make_golden_phi3.py tiny | layers | full [--size 224|448] [--videos N]
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mj_video_amd  # noqa: E402,F401
from mj_video_amd import configuration as C, synth  # noqa: E402
from oracle import ref_phi3, reference_shim as RS  # noqa: E402
from make_golden import FIELDS, _bits, check_equal, n_img_tokens, run_reference, to_np  # noqa: E402

TK = synth.PHI3_TOKENS
PROBE_VIT_LAYERS, PROBE_VIT_ROWS = (0, 11, 23), 48
PROBE_LLM_LAYERS, PROBE_LLM_ROWS = (0, 10, 21, 31), 32


def make_cfg(kind: str, image_size: int, **llm_over):
    cd = C.tiny_phi3_config_dict(image_size) if kind == "tiny" else C.internvl2_4b_config_dict(image_size)
    cd["llm_config"].update(llm_over)
    hk = C.mjvideo_head_kwargs()
    return cd, hk, C.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **copy.deepcopy(hk))


def oracle(sd, cfg, px, ids, mask, probes=None):
    """(the fp32 runs keep the bf16 model's rotary frequencies - ref_phi3.rope_tables ``buffer_dtype`` - so that they measure
    the bf16 run's rounding noise, not another positional encoding)"""
    return ref_phi3.reward_forward(sd, cfg, px, ids, mask, TK.img_context, TK.pad, cfg.gating_token_pattern, probes=probes,
                                   rope_buffer_dtype=torch.bfloat16)


def gen_tiny():
    """tiny dims, full per-layer probes.  original_max_position_embeddings = 128: 'short' (N = 88 + ...) exercises the SHORT
    LongRoPE factors, every other case (N > 128) the LONG ones; 'batch2pad' is a right-padded batch whose PADDED width decides."""
    cases, arrays = [], {}
    specs = [
        # name, image_size, weight_seed, [(video_idx, n_tiles, caption_seed, interleave)], llm overrides
        ("single", 56, 31, [(0, 4, 1, None)], {}),
        ("shortrope", 56, 32, [(1, 1, 2, None)], {"original_max_position_embeddings": 256}),
        ("batch2pad", 56, 33, [(2, 4, 3, None), (3, 2, 4, None)], {}),
        ("interleave", 56, 34, [(4, 4, 5, 4)], {}),
        ("norope_scaling", 56, 35, [(5, 3, 6, None)], {"rope_scaling": None, "max_position_embeddings": 4096,
                                                     "original_max_position_embeddings": 4096}),
    ]
    for name, S, wseed, vids, over in specs:
        cd, hk, cfg = make_cfg("tiny", S, **over)
        sd32 = synth.synth_state_dict(cfg, seed=wseed, dtype=torch.float32)
        sd = {k: v.to(torch.bfloat16) for k, v in sd32.items()}
        model = RS.build_reference_model_phi3(cd, hk, sd, torch.bfloat16, TK.img_context, TK.pad, cfg.gating_token_pattern)
        try:
            px_list, ids_list = [], []
            for (vi, nt, cs, il) in vids:
                px_list.append(synth.synth_pixel_values(300, vi, nt, S))
                ids_list.append(synth.synth_input_ids(n_img_tokens(cfg, nt), cs, interleave_frames=il, tokens=TK))
            px = torch.cat(px_list)
            ids, mask = synth.pad_batch(ids_list, pad_id=TK.pad)
            ref = run_reference(model, px, ids, mask)
            probes = {}
            orc = oracle(sd, cfg, px, ids, mask, probes)
            check_equal(ref, orc, "phi3:" + name)
            if len(vids) > 1:   # per-sample batch-1 forwards == the padded batch's rows, also for this decoder
                off = 0
                for b, (vi, nt, cs, il) in enumerate(vids):
                    one = run_reference(model, px[off:off + nt], ids_list[b], torch.ones_like(ids_list[b]))
                    off += nt
                    for k in FIELDS:
                        if k in ("hidden_state", "prompt_embedding") or True:
                            same = torch.equal(one[k][0], ref[k][b])
                            if not same:   # (a shorter PADDED width may pick the other LongRoPE factor list: report, do not assume)
                                print(f"  note: {name} sample {b} {k}: batch-1 forward differs from the padded batch's row "
                                      f"(N {ids_list[b].shape[1]} vs padded {ids.shape[1]}; original window "
                                      f"{cfg.llm_config.original_max_position_embeddings})")
                                break
        finally:
            model.restore_token_pattern()
        sd_f = {k: v.float() for k, v in sd.items()}
        f32 = oracle(sd_f, cfg, px.float(), ids, mask)
        for k, v in to_np(ref).items():
            arrays[f"{name}/{k}"] = v
        for k, v in to_np(f32).items():
            arrays[f"{name}/fp32/{k}"] = v
        for k, v in probes.items():
            arrays[f"{name}/probe/{k}"] = v.float().numpy()
        cases.append(dict(name=name, kind="tiny", image_size=S, weight_seed=wseed, pixel_seed=300, llm_overrides=over,
                          n_tokens=int(ids.shape[1]),
                          videos=[dict(video_idx=a, n_tiles=b, caption_seed=c, interleave=d) for a, b, c, d in vids]))
        print("phi3 tiny case", name, "N", int(ids.shape[1]), "score", ref["score"].tolist(), "fp32", f32["score"].tolist())
    np.savez_compressed(os.path.join(HERE, "phi3_tiny.npz"), **arrays)
    json.dump(dict(cases=cases, cpu_threads=torch.get_num_threads(), transformers=model.transformers_version),
              open(os.path.join(HERE, "phi3_tiny.json"), "w"), indent=1)


def gen_layers(wseed=0, xseed=43):
    """ONE Phi3DecoderLayer of transformers at InternVL2-4B dims on [1, 2186, 3072] seed-defined rows (layers 0 and 31; short
    factors) and on [1, 4224, 3072] rows (layer 0; N > 4096: the LONG factors), bf16 and fp32: sampled whole output rows."""
    from transformers import Phi3Config
    from transformers.models.phi3.modeling_phi3 import Phi3DecoderLayer, Phi3RotaryEmbedding
    from transformers.masking_utils import create_causal_mask
    from util import layer_input_rows as input_rows, layer_tensors
    cd, hk, cfg = make_cfg("4b", 448)
    lc = cd["llm_config"]
    pc = Phi3Config(**{k: v for k, v in lc.items() if k not in ("architectures", "attn_implementation")})
    pc._attn_implementation = "eager"
    arrays, cases = {}, []
    for li, N in ((0, 2186), (31, 2186), (0, 4224)):
        w = layer_tensors(cfg, f"model.language_model.model.layers.{li}.", wseed)
        tag = f"phi3_{li}_{N}"
        x = input_rows(xseed, tag, (1, N, lc["hidden_size"]))
        rows = np.sort(np.random.Generator(np.random.Philox(key=[xseed, 200 + li + N])).choice(N, size=24, replace=False))
        rows[0], rows[-1] = 0, N - 1
        outs = {}
        for dt in (torch.bfloat16, torch.float32):
            layer = Phi3DecoderLayer(pc, 0)
            layer.load_state_dict({k: v.to(dt) for k, v in w.items()}, strict=True)
            layer = layer.to(dt).eval()
            # model.to(bfloat16) casts the rotary module's inv_freq buffers too (ref_phi3.inv_freq); the fp32 run keeps those
            # bf16-rounded frequencies: it measures the bf16 run's rounding noise, not another positional encoding
            rot = Phi3RotaryEmbedding(pc).to(torch.bfloat16)
            xd = x.to(dt)
            pos = torch.arange(N).unsqueeze(0)
            mask = create_causal_mask(config=pc, inputs_embeds=xd, attention_mask=torch.ones(1, N, dtype=torch.long),
                                      past_key_values=None, position_ids=pos)
            with torch.no_grad():
                outs[dt] = layer(xd, attention_mask=mask, position_ids=pos, position_embeddings=rot(xd, position_ids=pos))
            if dt == torch.bfloat16:   # the restatement on the very same rows / weights: bit for bit
                sd = {f"model.language_model.model.layers.0.{k}": v for k, v in w.items()}
                c, s = ref_phi3.rope_tables(cfg, N, dt)
                m = ref_phi3.causal_padding_mask(torch.ones(1, N, dtype=torch.bool), dt)
                mine = ref_phi3.layer(sd, cfg, 0, xd, m, c, s)
                if not torch.equal(mine, outs[dt]):
                    raise SystemExit(f"ORACLE != transformers Phi3DecoderLayer for {tag}: max|d| {(mine.float() - outs[dt].float()).abs().max()}")
        y, y32 = outs[torch.bfloat16][:, rows], outs[torch.float32][:, rows]
        name = f"phi3_layer{li}_n{N}"
        arrays[f"{name}/rows"], arrays[f"{name}/out"], arrays[f"{name}/fp32"] = rows, _bits(y), y32.numpy()
        noise = float((y.float() - y32).norm() / y32.norm())
        cases.append(dict(name=name, layer=li, shape=list(x.shape), input_tag=tag, ref_bf16_vs_fp32=noise))
        print(name, "transformers bf16 vs fp32 rel-L2", noise)
    np.savez_compressed(os.path.join(HERE, "phi3_layers.npz"), **arrays)
    json.dump(dict(weight_seed=wseed, input_seed=xseed, image_size=448, cases=cases, cpu_threads=torch.get_num_threads()),
              open(os.path.join(HERE, "phi3_layers.json"), "w"), indent=1)


def gen_full(S: int, n_videos: int, wseed: int = 3, pixel_seed: int = 400, n_tiles: int = 8):
    """InternVL2-4B dims end to end (4.1 G parameters: about 8 GB in bf16, 16 GB for the fp32 oracle run)."""
    tag = f"phi3_full_{S}"
    cd, hk, cfg = make_cfg("4b", S)
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=wseed)
    model = RS.build_reference_model_phi3(cd, hk, sd, torch.bfloat16, TK.img_context, TK.pad, cfg.gating_token_pattern)
    print(f"[{tag}] model ready in {time.time() - t0:.1f}s", flush=True)
    arrays, vids = {}, []
    try:
        for v in range(n_videos):
            px = synth.synth_pixel_values(pixel_seed, v, n_tiles, S)
            ids = synth.synth_input_ids(n_img_tokens(cfg, n_tiles), caption_seed=v // 2, tokens=TK)
            mask = torch.ones_like(ids)
            t0 = time.time()
            ref = run_reference(model, px, ids, mask)
            dt = time.time() - t0
            if v == 0:
                probes = {}
                orc = oracle(sd, cfg, px, ids, mask, probes)
                check_equal(ref, orc, f"{tag}:{v}")
                for L in PROBE_VIT_LAYERS:
                    arrays[f"v{v}/probe/vit_layer{L}_rows"] = _bits(probes[f"vit_layer{L}"][0, :PROBE_VIT_ROWS, :])
                for L in PROBE_LLM_LAYERS:
                    arrays[f"v{v}/probe/llm_layer{L}_rows"] = _bits(probes[f"llm_layer{L}"][0, -PROBE_LLM_ROWS:, :])
                del probes
            for k, t in to_np(ref).items():
                arrays[f"v{v}/{k}"] = t
            vids.append(dict(video_idx=v, n_tiles=n_tiles, caption_seed=v // 2, seconds=round(dt, 2), n_tokens=int(ids.shape[1])))
            print(f"[{tag}] video {v}: {dt:.1f}s score={ref['score'].item():+.5f}", flush=True)
    finally:
        model.restore_token_pattern()
    tv = model.transformers_version
    del model
    sd_f = {k: t.float() for k, t in sd.items()}
    del sd
    for v in range(n_videos):   # the reference's own bf16 noise at these dims: the oracle in fp32 on the same inputs
        px = synth.synth_pixel_values(pixel_seed, v, n_tiles, S)
        ids = synth.synth_input_ids(n_img_tokens(cfg, n_tiles), caption_seed=v // 2, tokens=TK)
        probes = {} if v == 0 else None
        f32 = oracle(sd_f, cfg, px.float(), ids, torch.ones_like(ids), probes)
        for k, t in to_np(f32).items():
            arrays[f"v{v}/fp32/{k}"] = t
        if probes is not None:   # the same rows from the fp32 run: the reference's own bf16 noise LAYER BY LAYER (the tests' bounds)
            for L in PROBE_VIT_LAYERS:
                arrays[f"v{v}/probe/fp32/vit_layer{L}_rows"] = probes[f"vit_layer{L}"][0, :PROBE_VIT_ROWS, :].numpy()
            for L in PROBE_LLM_LAYERS:
                arrays[f"v{v}/probe/fp32/llm_layer{L}_rows"] = probes[f"llm_layer{L}"][0, -PROBE_LLM_ROWS:, :].numpy()
            del probes
        print(f"[{tag}] video {v} fp32 score={f32['score'].item():+.5f} (bf16 {arrays[f'v{v}/score'].item():+.5f})", flush=True)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **arrays)
    json.dump(dict(kind="4b", image_size=S, weight_seed=wseed, pixel_seed=pixel_seed, videos=vids, transformers=tv,
                   cpu_threads=torch.get_num_threads(),
                   row_probes=dict(vit_layers=list(PROBE_VIT_LAYERS), vit_rows=PROBE_VIT_ROWS, llm_layers=list(PROBE_LLM_LAYERS),
                                   llm_rows=PROBE_LLM_ROWS)),
              open(os.path.join(HERE, f"{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["tiny", "layers", "full"])
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--videos", type=int, default=2)
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.what == "tiny":
        gen_tiny()
    elif a.what == "layers":
        gen_layers()
    else:
        gen_full(a.size, a.videos)
