"""Shared helpers for the parity tests (test infrastructure only)."""
import copy
import json
import os

import numpy as np
import torch

import mj_video_amd  # noqa: F401
from mj_video_amd import configuration as C, synth
from mj_video_amd.chat_input import num_image_tokens_per_tile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIELDS = ("rewards", "hidden_state", "prompt_embedding", "criteria_gating_output", "aspect_gating_output",
          "aspect_weights", "score", "weighted_scores", "aspect_scores")


def make_cfg(kind, image_size, vit_image_size=None):
    cd = C.tiny_config_dict(image_size) if kind == "tiny" else C.mjvideo_2b_config_dict(image_size)
    if vit_image_size is not None:
        cd["vision_config"]["image_size"] = vit_image_size
    return C.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **C.mjvideo_head_kwargs())


def load_golden(tag):
    return np.load(os.path.join(GOLDEN, f"{tag}.npz")), json.load(open(os.path.join(GOLDEN, f"{tag}.json")))


def case_inputs(cfg, case_videos, pixel_seed, image_size):
    px, ids = [], []
    for v in case_videos:
        px.append(synth.synth_pixel_values(pixel_seed, v["video_idx"], v["n_tiles"], image_size))
        ids.append(synth.synth_input_ids(num_image_tokens_per_tile(cfg) * v["n_tiles"], v["caption_seed"],
                                         interleave_frames=v.get("interleave")))
    ids_b, mask = synth.pad_batch(ids)
    return torch.cat(px), ids_b, mask, ids


def case_inputs_masked(cfg, case, pixel_seed=None):
    """``case_inputs`` + the fixture case's mask arrangement (tests/golden/make_golden.py "mask_mode": left padding, holes)"""
    px, ids_b, mask, ids = case_inputs(cfg, case["videos"], case["pixel_seed"] if pixel_seed is None else pixel_seed, case["image_size"])
    ids_b, mask = synth.remask(ids_b, mask, case.get("mask_mode"))
    return px, ids_b, mask, ids


def apply_test_overrides(model):
    """MJV_TEST_ATTENTION_SCORES=eager|flash runs the model-level parity tests under the other attention numerics (the gate of
    DESIGN §4 "Attention, round 4": both settings are held to the same fixtures); unset = the model's default"""
    v = os.environ.get("MJV_TEST_ATTENTION_SCORES")
    if v:
        assert v in ("eager", "flash")
        model.attention_scores = v
    nf = os.environ.get("MJV_TEST_NORM_FUSION")     # the gate of DESIGN "Norm fusion, round 4": same fixtures, "1" folded / "0" not
    if nf:
        model.norm_fusion = nf == "1"
    return model


def build_hip_model(cfg, sd, device):
    from mj_video_amd.modeling import InternVLChatRewardModeling
    model = InternVLChatRewardModeling.from_config(cfg, dtype=torch.bfloat16)
    model.load_state_dict(sd, strict=True)
    model.config.pad_token_id = synth.PAD_ID
    model = model.to(torch.bfloat16).to(device)
    model.model.img_context_token_id = synth.IMG_CONTEXT_ID
    model.eval()
    return apply_test_overrides(model)


def bf16_ulps(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """distance in bf16 units-in-the-last-place between two bf16-representable float tensors"""
    ai = a.to(torch.bfloat16).view(torch.int16).to(torch.int32)
    bi = b.to(torch.bfloat16).view(torch.int16).to(torch.int32)
    ai = torch.where(ai < 0, -(ai & 0x7fff), ai)
    bi = torch.where(bi < 0, -(bi & 0x7fff), bi)
    return (ai - bi).abs()


def layer_tensors(cfg, prefix, seed):
    """the synthetic checkpoint tensors under ``prefix`` with the prefix stripped (same values as synth.synth_state_dict:
    one Philox stream per key) - for tests that need a few layers of a 2B-dims model, not all 2.2 G parameters"""
    ls0 = float(cfg.vision_config.initializer_factor)
    out = {}
    for key, shape, kind in synth.state_dict_spec(cfg):
        if not key.startswith(prefix):
            continue
        if kind in ("w", "b"):
            t = synth._normal(seed, key, shape, 0.02)
        elif kind == "g":
            t = synth._normal(seed, key, shape, 0.05, 1.0)
        elif kind == "ls":
            t = synth._normal(seed, key, shape, 0.05 * ls0, ls0)
        else:
            raise AssertionError((key, kind))
        out[key[len(prefix):]] = t.to(torch.bfloat16)
    return out


def layer_input_rows(seed, tag, shape):
    """seed-defined bf16 hidden states of the single-layer fixtures (tests/golden/make_layer_fixtures.py)"""
    return synth._normal(seed, "layer-input/" + tag, shape, 1.0).to(torch.bfloat16)
