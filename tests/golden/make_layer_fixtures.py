#!/usr/bin/env python3
"""Single-layer fixtures at production shape + the dynamic-NTK rotary sequence, from the reference's own code.

Build-container only (needs /root/reference).  Two things the end-to-end fixtures cannot show:

1. ONE layer at MJ-VIDEO-2B dimensions and the headline sequence lengths, without 24 layers of compounding (the backbone is
   chaotic at bf16: whole-tower probes need 3-4 % bounds).  The reference's ``InternVisionEncoderLayer.forward``
   (internvl2/modeling_intern_vit.py:283-295) for vision layers 0 and 23 on [2, 1025, 1024] rows and its
   ``InternLM2DecoderLayer.forward`` (internvl2/modeling_internlm2.py:621-681, eager attention, the model's own
   ``_prepare_decoder_attention_mask``) for language layers 0 and 23 on [1, 2186, 2048] rows are executed on seed-defined
   bf16 inputs with the synthetic weights of those layers, once in bf16 and once in fp32; sampled whole output rows of both
   runs are stored (the fp32 run gives the reference's OWN one-layer bf16 noise, which sets the test's bound).
2. The rotary cache is STATEFUL in the reference (modeling_internlm2.py:169-176,204-229): a tiny config with
   ``max_position_embeddings`` below the sequence length and ``rope_scaling = dynamic`` is run on short -> long -> short
   inputs with ONE model; the third call differs from the first because the rescaled ``inv_freq`` sticks.  The oracle
   (stateful ``rope_tables``) must reproduce all three calls bit for bit.

Stores tests/golden/layers.npz / layers.json (outputs only).   Usage:  python tests/golden/make_layer_fixtures.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mj_video_amd  # noqa: E402,F401
from mj_video_amd import configuration as C, synth  # noqa: E402
from oracle import ref_cpu, reference_shim as RS  # noqa: E402
from make_golden import FIELDS, check_equal, make_cfg, n_img_tokens, run_reference, to_np  # noqa: E402
from util import layer_input_rows as input_rows, layer_tensors  # noqa: E402


def bits(t):
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm())


def gen_layers(arrays, meta, wseed=0, xseed=41):
    RS.load_reference()
    from internvl2.modeling_intern_vit import InternVisionEncoderLayer
    from internvl2.modeling_internlm2 import InternLM2DecoderLayer, InternLM2Model
    from internvl2 import InternVLChatConfig
    cd, hk, cfg = make_cfg("2b", 448)
    rcfg = InternVLChatConfig(**cd)
    rcfg.llm_config.attn_implementation = "eager"
    cases = []
    for li in (0, 23):
        w = layer_tensors(cfg, f"model.vision_model.encoder.layers.{li}.", wseed)
        x = input_rows(xseed, f"vit{li}", (2, 1025, cfg.vision_config.hidden_size))
        rows = np.sort(np.random.Generator(np.random.Philox(key=[xseed, li])).choice(1025, size=24, replace=False))
        rows[0] = 0   # the CLS row is always in
        outs = {}
        for dt in (torch.bfloat16, torch.float32):
            layer = InternVisionEncoderLayer(rcfg.vision_config, 0.0)
            layer.load_state_dict({k: v.to(dt) for k, v in w.items()}, strict=True)
            layer = layer.to(dt).eval()
            with torch.no_grad():
                outs[dt] = layer(x.to(dt))
        y, y32 = outs[torch.bfloat16][:, rows], outs[torch.float32][:, rows]
        name = f"vit_layer{li}"
        arrays[f"{name}/rows"], arrays[f"{name}/out"], arrays[f"{name}/fp32"] = rows, bits(y), y32.numpy()
        noise = rel_l2(y, y32)
        cases.append(dict(name=name, tower="vit", layer=li, shape=list(x.shape), input_tag=f"vit{li}", ref_bf16_vs_fp32=noise))
        print(name, "reference bf16 vs fp32 rel-L2", noise)
    for li in (0, 23):
        w = layer_tensors(cfg, f"model.language_model.model.layers.{li}.", wseed)
        N = 2186
        x = input_rows(xseed, f"llm{li}", (1, N, cfg.llm_config.hidden_size))
        rows = np.sort(np.random.Generator(np.random.Philox(key=[xseed, 100 + li])).choice(N, size=24, replace=False))
        rows[0], rows[-1] = 0, N - 1
        outs = {}
        for dt in (torch.bfloat16, torch.float32):
            layer = InternLM2DecoderLayer(rcfg.llm_config)
            layer.load_state_dict({k: v.to(dt) for k, v in w.items()}, strict=True)
            layer = layer.to(dt).eval()
            xd = x.to(dt)
            mask = InternLM2Model._prepare_decoder_attention_mask(None, torch.ones(1, N, dtype=torch.bool), (1, N), xd, 0)
            pos = torch.arange(N).unsqueeze(0)
            with torch.no_grad():
                outs[dt] = layer(xd, attention_mask=mask, position_ids=pos)[0]
        y, y32 = outs[torch.bfloat16][:, rows], outs[torch.float32][:, rows]
        name = f"llm_layer{li}"
        arrays[f"{name}/rows"], arrays[f"{name}/out"], arrays[f"{name}/fp32"] = rows, bits(y), y32.numpy()
        noise = rel_l2(y, y32)
        cases.append(dict(name=name, tower="llm", layer=li, shape=list(x.shape), input_tag=f"llm{li}", ref_bf16_vs_fp32=noise))
        print(name, "reference bf16 vs fp32 rel-L2", noise)
    meta["layers"] = dict(weight_seed=wseed, input_seed=xseed, image_size=448, cases=cases)


def gen_ntk(arrays, meta, wseed=16):
    cd, hk, cfg = make_cfg("tiny", 56)
    cd["llm_config"]["max_position_embeddings"] = 48
    cd["llm_config"]["rope_scaling"] = {"type": "dynamic", "factor": 2.0}
    cfg = C.InternVLChatRewardModelingConfig(**cd, **hk)
    sd32 = synth.synth_state_dict(cfg, seed=wseed, dtype=torch.float32)
    sd = {k: v.to(torch.bfloat16) for k, v in sd32.items()}
    model = RS.build_reference_model(cd, hk, sd, torch.bfloat16, synth.IMG_CONTEXT_ID, synth.PAD_ID)
    calls = [("short_first", 6, 1, 7), ("long", 7, 8, 8), ("short_again", 6, 1, 7), ("batch_padded", None, None, None)]
    state, recs = {}, []
    for name, vi, nt, cs in calls:
        if name == "batch_padded":   # two sequences of different length: the PADDED width decides the regrowth
            px = torch.cat([synth.synth_pixel_values(100, 8, 10, 56), synth.synth_pixel_values(100, 9, 2, 56)])
            ids, mask = synth.pad_batch([synth.synth_input_ids(n_img_tokens(cfg, 10), 9), synth.synth_input_ids(n_img_tokens(cfg, 2), 10)])
            vids = [dict(video_idx=8, n_tiles=10, caption_seed=9), dict(video_idx=9, n_tiles=2, caption_seed=10)]
        else:
            px = synth.synth_pixel_values(100, vi, nt, 56)
            ids, mask = synth.pad_batch([synth.synth_input_ids(n_img_tokens(cfg, nt), cs)])
            vids = [dict(video_idx=vi, n_tiles=nt, caption_seed=cs)]
        ref = run_reference(model, px, ids, mask)
        orc = ref_cpu.reward_forward(sd, cfg, px, ids, mask, synth.IMG_CONTEXT_ID, synth.PAD_ID, rope_state=state)
        check_equal(ref, orc, "ntk/" + name)
        f32 = ref_cpu.reward_forward({k: v.float() for k, v in sd.items()}, cfg, px.float(), ids, mask, synth.IMG_CONTEXT_ID,
                                     synth.PAD_ID, rope_state=dict(state))   # same rotary state as the bf16 call just made
        for k, v in to_np(ref).items():
            arrays[f"ntk/{name}/{k}"] = v
        for k, v in to_np(f32).items():
            arrays[f"ntk/{name}/fp32/{k}"] = v
        recs.append(dict(name=name, videos=vids, n_tokens=int(ids.shape[1])))
        print("ntk call", name, "N =", int(ids.shape[1]), "score", ref["score"].tolist())
    a, b = arrays["ntk/short_first/hidden_state"], arrays["ntk/short_again/hidden_state"]
    assert not np.array_equal(a, b), "the rescaled inv_freq should have stuck"
    meta["ntk"] = dict(weight_seed=wseed, pixel_seed=100, image_size=56, max_position_embeddings=48,
                       rope_scaling={"type": "dynamic", "factor": 2.0}, calls=recs)


def main():
    torch.manual_seed(0)
    arrays, meta = {}, dict(cpu_threads=torch.get_num_threads())
    gen_ntk(arrays, meta)
    gen_layers(arrays, meta)
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "layers.json"), "w"), indent=1)
    print("wrote layers.npz", os.path.getsize(os.path.join(HERE, "layers.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
