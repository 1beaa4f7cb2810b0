"""CPU: oracle/ref_phi3.py (the Phi-3 language tower of BASELINE configs[4], restated from transformers 5.15's modeling_phi3.py)
against the fixtures the REFERENCE's reward-model code produced around transformers' own Phi3ForCausalLM
(tests/golden/make_golden_phi3.py, build container) - and the host-side pieces of that backbone (config, prompt template,
parameter layout, gating pattern)."""
import copy

import numpy as np
import pytest
import torch

from util import FIELDS, load_golden
from mj_video_amd import configuration as C, synth
from oracle import ref_phi3

TK = synth.PHI3_TOKENS


def _cfg(case):
    cd = C.tiny_phi3_config_dict(case["image_size"])
    cd["llm_config"].update(case["llm_overrides"])
    return C.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **C.mjvideo_head_kwargs())


def _inputs(cfg, case):
    from mj_video_amd.chat_input import num_image_tokens_per_tile
    px, ids = [], []
    for v in case["videos"]:
        px.append(synth.synth_pixel_values(case["pixel_seed"], v["video_idx"], v["n_tiles"], case["image_size"]))
        ids.append(synth.synth_input_ids(num_image_tokens_per_tile(cfg) * v["n_tiles"], v["caption_seed"],
                                         interleave_frames=v.get("interleave"), tokens=TK))
    ids_b, mask = synth.pad_batch(ids, pad_id=TK.pad)
    return torch.cat(px), ids_b, mask, ids


def test_phi3_oracle_reproduces_reference_tiny():
    npz, meta = load_golden("phi3_tiny")
    torch.set_num_threads(int(meta.get("cpu_threads", torch.get_num_threads())))
    names = [c["name"] for c in meta["cases"]]
    n_exact = 0
    for case in meta["cases"]:
        cfg = _cfg(case)
        sd = {k: v.to(torch.bfloat16) for k, v in synth.synth_state_dict(cfg, seed=case["weight_seed"], dtype=torch.float32).items()}
        px, ids, mask, _ = _inputs(cfg, case)
        assert ids.shape[1] == case["n_tokens"]
        probes = {}
        out = ref_phi3.reward_forward(sd, cfg, px, ids, mask, TK.img_context, TK.pad, cfg.gating_token_pattern, probes=probes)
        exact = True
        for f in FIELDS:
            got, ref = out[f].float().numpy(), npz[f"{case['name']}/{f}"]
            assert got.shape == ref.shape
            if np.array_equal(got, ref):
                continue
            exact = False   # (another host's oneDNN may pick other bf16 GEMM kernels: then half the reference's own noise floor)
            floor = max(float(np.abs(npz[f"{n}/{f}"] - npz[f"{n}/fp32/{f}"]).max()) for n in names)
            assert float(np.abs(got - ref).max()) <= 0.5 * floor + 4e-3 * float(np.abs(ref).max()), (case["name"], f)
        n_exact += exact
        for k, v in probes.items():
            ref = npz[f"{case['name']}/probe/{k}"]
            assert v.shape == ref.shape
            assert np.abs(v.float().numpy() - ref).max() <= 0.05 * np.abs(ref).max() + 1e-3, (case["name"], k)
        assert out["score"].dtype == torch.float32 and out["rewards"].dtype == torch.bfloat16
    print(f"bit-exact cases: {n_exact}/{len(meta['cases'])}")


def test_phi3_rope_tables_follow_the_padded_width():
    """LongRoPE: the factor list is chosen by the PADDED sequence width against original_max_position_embeddings, per forward;
    the short / unscaled frequencies are bf16-rounded (module buffers under model.to(bfloat16)), the long ones are not"""
    cfg = _cfg(dict(image_size=56, llm_overrides={}))
    l = cfg.llm_config
    assert l.original_max_position_embeddings == 128
    short, long = ref_phi3.inv_freq(l, 128, torch.bfloat16), ref_phi3.inv_freq(l, 129, torch.bfloat16)
    assert not torch.equal(short, long)
    assert torch.equal(short, short.to(torch.bfloat16).float())
    assert not torch.equal(long, long.to(torch.bfloat16).float())
    assert torch.equal(ref_phi3.inv_freq(l, 128, torch.float32), 1.0 / (torch.tensor(l.rope_scaling["short_factor"]) * 10000.0 ** (torch.arange(0, 96, 2).float() / 96)))
    a = ref_phi3.attention_factor(l)
    assert abs(a - np.sqrt(1 + np.log(4096 / 128) / np.log(128))) < 1e-12
    c, s = ref_phi3.rope_tables(cfg, 200, torch.bfloat16)
    assert c.shape == (200, 96) and c.dtype == torch.bfloat16 and float(c[0, 0]) == float(torch.tensor(a).to(torch.bfloat16))


def test_phi3_config_and_parameter_layout():
    cd = C.internvl2_4b_config_dict(448)
    cfg = C.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **C.mjvideo_head_kwargs())
    l = cfg.llm_config
    assert type(l).__name__ == "Phi3Config" and l.hidden_size == 3072 and l.hidden_size // l.num_attention_heads == 96
    assert cfg.template == "phi3-chat" and cfg.gating_token_pattern == [32007, 32001, 13]
    # round trip through to_dict (what save_pretrained writes), and the transformers-5 spelling of the rope parameters
    again = C.InternVLChatRewardModelingConfig(**cfg.to_dict())
    assert again.llm_config.rope_scaling == l.rope_scaling and again.gating_token_pattern == cfg.gating_token_pattern
    lc = copy.deepcopy(cd["llm_config"])
    rs = lc.pop("rope_scaling")
    lc["rope_parameters"] = dict(rope_type="longrope", rope_theta=lc.pop("rope_theta"), short_factor=rs["short_factor"],
                                 long_factor=rs["long_factor"], original_max_position_embeddings=lc.pop("original_max_position_embeddings"))
    new = C.Phi3Config(**{k: v for k, v in lc.items() if k != "architectures"})
    assert new.rope_scaling["short_factor"] == rs["short_factor"] and new.original_max_position_embeddings == 4096 and new.rope_theta == 10000.0
    with pytest.raises(ValueError, match="length 48"):
        C.Phi3Config(rope_scaling={"type": "longrope", "short_factor": [1.0] * 3, "long_factor": [1.0] * 48})
    # the InternLM2 default pattern is the reference's module constant
    two_b = C.InternVLChatRewardModelingConfig(**C.mjvideo_2b_config_dict(448), **C.mjvideo_head_kwargs())
    assert two_b.gating_token_pattern == [92542, 92543, 525, 11353, 364]
    # parameter names / shapes of the tiny skeleton = transformers' Phi3ForCausalLM under model.language_model
    from mj_video_amd.modeling import InternVLChatRewardModeling
    tiny = _cfg(dict(image_size=56, llm_overrides={}))
    m = InternVLChatRewardModeling.from_config(tiny, dtype=torch.bfloat16)
    keys = dict(m.state_dict())
    p = "model.language_model.model.layers.1."
    assert keys[p + "self_attn.qkv_proj.weight"].shape == (3 * 192, 192) and keys[p + "mlp.gate_up_proj.weight"].shape == (1024, 192)
    assert "model.language_model.lm_head.weight" in keys and "model.language_model.model.embed_tokens.weight" in keys
    assert set(keys) == {k for k, _, _ in synth.state_dict_spec(tiny)}
    with pytest.raises(RuntimeError):   # no CPU path for this tower either
        m.to(torch.bfloat16).forward(torch.zeros(1, 3, 56, 56, dtype=torch.bfloat16), torch.zeros(1, 8, dtype=torch.long))


def test_phi3_prompt_template():
    """conversation.py:368-379 (phi3-chat) through build_query: MPT separator style"""
    from mj_video_amd.chat_input import build_query, num_image_tokens_per_tile, video_prefix
    cfg = _cfg(dict(image_size=56, llm_overrides={}))
    n = 2 * num_image_tokens_per_tile(cfg)
    q = build_query(cfg, video_prefix(2) + "a cat", 2)
    assert q.startswith("<|system|>\n") and q.endswith("a cat<|end|><|assistant|>\n")
    assert "<|end|><|user|>\nFrame1: <img>" + "<IMG_CONTEXT>" * n + "</img>\nFrame2: <image>\na cat" in q
