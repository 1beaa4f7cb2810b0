"""Deterministic synthetic weights and inputs (no checkpoints, tokenizer or videos exist offline).

Everything here is a pure function of integer seeds through numpy's counter-based Philox
generator, so this container (where golden vectors are produced by importing the reference) and
the GPU box (where the HIP path is checked against them) regenerate bit-identical tensors
without shipping gigabytes.  Workload layout: SURVEY.md §8(d).
"""
from __future__ import annotations

import os
import zlib
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

# special token ids of the InternVL2 / InternLM2 tokenizer.  92542/92543/525/11353/364 are pinned by
# scripts/model/moe_reward.py:48; the others are the published ids of the checkpoint's tokenizer.
IM_END, IM_START = 92542, 92543
IMG_START_ID, IMG_END_ID, IMG_CONTEXT_ID = 92544, 92545, 92546
BOS_ID, PAD_ID = 1, 2
GATING_PATTERN = (92542, 92543, 525, 11353, 364)
N_TEXT_TOKENS = 138  # non-image tokens of the synthetic prompt (N = 2186 at 8x256 image tokens)


class TokenProfile:
    """the special-token ids a synthetic prompt is built from (one per tokenizer family)"""

    def __init__(self, name, im_end, im_start, img_start, img_end, img_context, bos, pad, nl, gating_pattern, text_hi):
        self.name, self.im_end, self.im_start = name, im_end, im_start
        self.img_start, self.img_end, self.img_context = img_start, img_end, img_context
        self.bos, self.pad, self.nl, self.gating_pattern, self.text_hi = bos, pad, nl, tuple(gating_pattern), text_hi


INTERNLM2_TOKENS = TokenProfile("internlm2", IM_END, IM_START, IMG_START_ID, IMG_END_ID, IMG_CONTEXT_ID, BOS_ID, PAD_ID, 364,
                                GATING_PATTERN, 60000)
# stand-in ids of the InternVL2-4B (Phi-3) tokenizer [recalled, unpinned - no tokenizer offline]: `<|end|>` 32007 closes a turn,
# `<|system|>` 32006 / `<|user|>` 32010 / `<|assistant|>` 32001 open one, `<img>` 32011, `</img>` 32012, `<IMG_CONTEXT>` 32013,
# pad = `<|endoftext|>` 32000; the gating pattern is `<|end|><|assistant|>\n` (configuration.PHI3_GATING_PATTERN).  The same
# prompt STRUCTURE as the InternLM2 profile (same positions of every run), so sequence lengths match: im_start stands for the
# role marker of a turn.
PHI3_TOKENS = TokenProfile("phi3", 32007, 32010, 32011, 32012, 32013, 1, 32000, 13, (32007, 32001, 13), 30000)


def token_profile(config) -> TokenProfile:
    return PHI3_TOKENS if type(config.llm_config).__name__ == "Phi3Config" else INTERNLM2_TOKENS


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def _normal(seed: int, name: str, shape, std: float, mean: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape))
    a = _rng(seed, name).standard_normal(n, dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return torch.from_numpy(a).reshape(tuple(shape))


def state_dict_spec(config) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every tensor of the reward model's checkpoint (SURVEY.md §8(b) layout)."""
    v, l = config.vision_config, config.llm_config
    d, ff, P = v.hidden_size, v.intermediate_size, v.patch_size
    npos = (v.image_size // P) ** 2 + 1
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    nobj = config.num_objectives
    spec.append(("reward_transform_matrix", (nobj, nobj), "eye"))
    p = "model.vision_model.embeddings."
    spec += [(p + "class_embedding", (1, 1, d), "emb"), (p + "position_embedding", (1, npos, d), "emb"),
             (p + "patch_embedding.weight", (d, 3, P, P), "w"), (p + "patch_embedding.bias", (d,), "b")]
    for i in range(v.num_hidden_layers):
        p = f"model.vision_model.encoder.layers.{i}."
        spec += [(p + "ls1", (d,), "ls"), (p + "ls2", (d,), "ls"),
                 (p + "attn.qkv.weight", (3 * d, d), "w")]
        if v.qkv_bias:
            spec.append((p + "attn.qkv.bias", (3 * d,), "b"))
        spec += [(p + "attn.proj.weight", (d, d), "w"), (p + "attn.proj.bias", (d,), "b"),
                 (p + "mlp.fc1.weight", (ff, d), "w"), (p + "mlp.fc1.bias", (ff,), "b"),
                 (p + "mlp.fc2.weight", (d, ff), "w"), (p + "mlp.fc2.bias", (d,), "b"),
                 (p + "norm1.weight", (d,), "g"), (p + "norm1.bias", (d,), "b"),
                 (p + "norm2.weight", (d,), "g"), (p + "norm2.bias", (d,), "b")]
    h, lf = l.hidden_size, l.intermediate_size
    hd = h // l.num_attention_heads
    qkv_out = (l.num_attention_heads + 2 * l.num_key_value_heads) * hd
    if type(l).__name__ == "Phi3Config":   # transformers/models/phi3/modeling_phi3.py parameter names (the upstream 4B checkpoint's)
        spec.append(("model.language_model.model.embed_tokens.weight", (l.vocab_size, h), "w"))
        for i in range(l.num_hidden_layers):
            p = f"model.language_model.model.layers.{i}."
            spec += [(p + "self_attn.o_proj.weight", (h, l.num_attention_heads * hd), "w"),
                     (p + "self_attn.qkv_proj.weight", (qkv_out, h), "w"),
                     (p + "mlp.gate_up_proj.weight", (2 * lf, h), "w"), (p + "mlp.down_proj.weight", (h, lf), "w"),
                     (p + "input_layernorm.weight", (h,), "g"), (p + "post_attention_layernorm.weight", (h,), "g")]
        spec.append(("model.language_model.model.norm.weight", (h,), "g"))
        spec.append(("model.language_model.lm_head.weight", (l.vocab_size, h), "lmhead"))
    else:
        spec.append(("model.language_model.model.tok_embeddings.weight", (l.vocab_size, h), "w"))
        for i in range(l.num_hidden_layers):
            p = f"model.language_model.model.layers.{i}."
            spec += [(p + "attention.wqkv.weight", (qkv_out, h), "w"), (p + "attention.wo.weight", (h, h), "w"),
                     (p + "feed_forward.w1.weight", (lf, h), "w"), (p + "feed_forward.w3.weight", (lf, h), "w"),
                     (p + "feed_forward.w2.weight", (h, lf), "w"),
                     (p + "attention_norm.weight", (h,), "g"), (p + "ffn_norm.weight", (h,), "g")]
        spec.append(("model.language_model.model.norm.weight", (h,), "g"))
        spec.append(("model.language_model.output.weight", (l.vocab_size, h), "lmhead"))
    c4 = d * int(1 / config.downsample_ratio) ** 2
    spec += [("model.mlp1.0.weight", (c4,), "g"), ("model.mlp1.0.bias", (c4,), "b"),
             ("model.mlp1.1.weight", (h, c4), "w"), ("model.mlp1.1.bias", (h,), "b"),
             ("model.mlp1.3.weight", (h, h), "w"), ("model.mlp1.3.bias", (h,), "b")]
    spec.append(("regression_layer.weight", (nobj, h), "head"))
    gh, gn = config.gating_hidden_dim, config.gating_n_hidden
    for net, nout in (("aspect_gating", config.num_aspects), ("criteria_gating", nobj)):
        spec.append((f"{net}.logit_scale", (1,), "one"))
        fin = h
        for j in range(gn):
            spec += [(f"{net}.layers.{j}.weight", (gh, fin), "gate"), (f"{net}.layers.{j}.bias", (gh,), "b")]
            fin = gh
        spec += [(f"{net}.layers.{gn}.weight", (nout, fin), "gate"), (f"{net}.layers.{gn}.bias", (nout,), "b")]
    return spec


def synth_state_dict(config, seed: int = 0, dtype=torch.bfloat16, head_std: float = 0.05,
                     gate_std: float = 0.05, lm_head: bool = True) -> Dict[str, torch.Tensor]:
    """Random-init weights with exact checkpoint keys/shapes.

    Linear/conv/embedding weights N(0, 0.02) (the reference's HF init,
    modeling_internlm2.py:714-723); biases N(0, 0.02) and norm gains 1+N(0, 0.05) instead of the
    init's 0/1 so that every bias/gain code path is numerically exercised; reward/gating heads
    use a larger sigma so scores spread well above bf16 noise (SURVEY.md §7 "hard parts").
    The LM head is zeros: the reward path never reads it (only ``strict=True`` loading does).
    """
    ls0 = float(config.vision_config.initializer_factor)

    def make(item):
        key, shape, kind = item
        if kind == "w":
            t = _normal(seed, key, shape, 0.02)
        elif kind == "b":
            t = _normal(seed, key, shape, 0.02)
        elif kind == "g":
            t = _normal(seed, key, shape, 0.05, 1.0)
        elif kind == "ls":
            t = _normal(seed, key, shape, 0.05 * ls0, ls0)
        elif kind == "emb":
            t = _normal(seed, key, shape, 1.0)
        elif kind == "head":
            t = _normal(seed, key, shape, head_std)
        elif kind == "gate":
            t = _normal(seed, key, shape, gate_std)
        elif kind == "eye":
            t = torch.eye(shape[0], dtype=torch.float32)
        elif kind == "one":
            t = torch.ones(shape, dtype=torch.float32)
        elif kind == "lmhead":
            t = torch.zeros(shape, dtype=dtype)
        else:
            raise AssertionError(kind)
        return key, t.to(dtype)

    items = [it for it in state_dict_spec(config) if lm_head or it[2] != "lmhead"]
    # every tensor has its own Philox stream, so generation order / threading cannot change values
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        return dict(pool.map(make, items))


def synth_head_state_dict(config, seed: int = 0, dtype=torch.bfloat16, head_std: float = 0.05,
                          gate_std: float = 0.05) -> Dict[str, torch.Tensor]:
    """Only the reward / gating head tensors of ``synth_state_dict`` (identical values: every tensor has its own
    Philox stream), i.e. every checkpoint key outside ``model.*``."""
    out = {}
    for key, shape, kind in state_dict_spec(config):
        if key.startswith("model."):
            continue
        if kind == "head":
            t = _normal(seed, key, shape, head_std)
        elif kind == "gate":
            t = _normal(seed, key, shape, gate_std)
        elif kind == "b":
            t = _normal(seed, key, shape, 0.02)
        elif kind == "eye":
            t = torch.eye(shape[0], dtype=torch.float32)
        elif kind == "one":
            t = torch.ones(shape, dtype=torch.float32)
        else:
            raise AssertionError((key, kind))
        out[key] = t.to(dtype)
    return out


def engineered_head_state_dict(config, seed: int, regression_weight: torch.Tensor, gate_dirs: np.ndarray,
                               dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Head weights of the ENGINEERED rank sets (tests/golden/rankeng_*.npz, made by make_golden.gen_rankset_eng): the
    synthetic heads of ``synth_head_state_dict`` with (a) the stored regression matrix and (b) first gating layers whose
    rows are random mixes of ``gate_dirs`` ([n, hidden]: leading principal directions of the gating-row hidden state,
    each divided by its inter-video spread).  The mix is a pure function of ``seed`` and the product is formed
    elementwise in float64 in a fixed order, so the build container and the GPU box get bit-identical bf16 weights
    (the fixture stores their checksum)."""
    sd = synth_head_state_dict(config, seed=seed, dtype=dtype)
    sd["regression_layer.weight"] = regression_weight.to(dtype)
    d = np.asarray(gate_dirs, dtype=np.float64)
    for net in ("aspect_gating", "criteria_gating"):
        key = f"{net}.layers.0.weight"
        rows = sd[key].shape[0]
        m = _rng(seed, key + "/engineered-mix").standard_normal((rows, d.shape[0]))
        w = (m[:, :, None] * d[None, :, :]).sum(axis=1)
        sd[key] = torch.from_numpy(w.astype(np.float32)).to(dtype)
    return sd


def synth_pixel_values(seed: int, video_idx: int, n_tiles: int, image_size: int,
                       dtype=torch.bfloat16) -> torch.Tensor:
    """One synthetic video, already in normalised-pixel space: ``[n_tiles, 3, S, S]``.

    Each video gets its own brightness/contrast and a coarse spatial pattern that drifts from
    frame to frame (so different videos produce visibly different features and score margins are
    not pure noise), plus unit white noise as in SURVEY.md §8(d).
    """
    g = _rng(seed, f"video{video_idx}")
    S = image_size
    gain = np.float32(0.5 + 1.5 * g.random())
    bias = np.float32(g.normal(0.0, 0.7))
    coarse = g.standard_normal((n_tiles, 3, 8, 8), dtype=np.float32)
    drift = np.cumsum(coarse, axis=0) / np.sqrt(np.arange(1, n_tiles + 1, dtype=np.float32))[:, None, None, None]
    rep = -(-S // 8)
    field = np.repeat(np.repeat(drift, rep, axis=2), rep, axis=3)[:, :, :S, :S]
    noise = g.standard_normal((n_tiles, 3, S, S), dtype=np.float32)
    px = gain * (0.8 * field + 0.6 * noise) + bias
    return torch.from_numpy(np.ascontiguousarray(px)).to(dtype)


def synth_input_ids(n_image_tokens: int, caption_seed: int, n_caption: int = 32,
                    interleave_frames: Optional[int] = None, tokens: Optional[TokenProfile] = None) -> torch.Tensor:
    """Token ids with the structure of a tokenised MJ-VIDEO prompt: ``[1, n_image_tokens + 138]``.

    ``BOS, <|im_start|> + 61 system ids + <|im_end|>, <|im_start|> user\\n "Frame1: " <img>,
    <IMG_CONTEXT> * n, </img> \\n, 7 * ("FrameK: <image>\\n" as 4 literal ids), caption ids,
    <|im_end|><|im_start|>assistant\\n`` - the contiguous image run is the reference's default
    behaviour.  With ``interleave_frames=F`` the image run is split into F runs,
    one per "FrameK: <img>...</img>\\n" (what ``num_patches_list=[1]*F`` produces).
    """
    tk = INTERNLM2_TOKENS if tokens is None else tokens     # (``tokens``: another tokenizer family's special ids, same structure)
    fixed = _rng(0, "prompt-fixed")
    sys_ids = fixed.integers(1000, tk.text_hi, size=61).tolist()
    user_ids = fixed.integers(1000, tk.text_hi, size=2).tolist()
    frame_ids = fixed.integers(1000, tk.text_hi, size=(8, 3)).tolist()
    nl_id = tk.nl
    IMG_START_ID, IMG_END_ID, IMG_CONTEXT_ID = tk.img_start, tk.img_end, tk.img_context
    cap = _rng(caption_seed, "caption").integers(1000, tk.text_hi, size=n_caption).tolist()
    ids = [tk.bos, tk.im_start] + sys_ids + [tk.im_end, tk.im_start] + user_ids
    if interleave_frames:
        F = interleave_frames
        assert n_image_tokens % F == 0 and F <= 8
        per = n_image_tokens // F
        body: List[int] = []
        for k in range(F):
            body += frame_ids[k] + [IMG_START_ID] + [IMG_CONTEXT_ID] * per + [IMG_END_ID, nl_id]
        # keep the total length identical to the contiguous layout: 7 literal frames <-> 7*(3+1+2-2)
        filler = 7 * 4 + 3 + 1 + 2 - F * 6
        ids += body + (fixed.integers(1000, tk.text_hi, size=max(filler, 0)).tolist())
    else:
        ids += frame_ids[0] + [IMG_START_ID] + [IMG_CONTEXT_ID] * n_image_tokens + [IMG_END_ID, nl_id]
        for k in range(1, 8):
            ids += frame_ids[k] + [nl_id]
    ids += cap + list(tk.gating_pattern)
    t = torch.tensor(ids, dtype=torch.long).unsqueeze(0)
    if not interleave_frames:
        assert t.shape[1] == n_image_tokens + N_TEXT_TOKENS - 32 + n_caption - (5 - len(tk.gating_pattern)), t.shape
    return t


def pad_batch(ids_list: List[torch.Tensor], pad_id: int = PAD_ID, length: Optional[int] = None):
    """Right-pad ``[1, N_i]`` id rows to ``[B, N]`` as the reference collator does (dataset.py:445-469)."""
    n = max(int(t.shape[-1]) for t in ids_list) if length is None else length
    ids = torch.full((len(ids_list), n), pad_id, dtype=torch.long)
    mask = torch.zeros((len(ids_list), n), dtype=torch.long)
    for i, t in enumerate(ids_list):
        k = int(t.shape[-1])
        ids[i, :k] = t.reshape(-1)
        mask[i, :k] = 1
    return ids, mask


def remask(ids: torch.Tensor, mask: torch.Tensor, mode: Optional[str], pad_id: int = PAD_ID):
    """Re-arranges a right-padded batch for the mask cases of the fixtures: "left" = every shorter row LEFT-padded instead (valid
    tokens at the end of the row); "holes" = three caption tokens of every row masked out in place (ids untouched); None = as is."""
    if mode is None:
        return ids, mask
    ids, mask = ids.clone(), mask.clone()
    if mode == "left":
        N = ids.shape[1]
        for b in range(ids.shape[0]):
            L = int(mask[b].sum())
            row = ids[b, :L].clone()
            ids[b] = pad_id
            ids[b, N - L:] = row
            mask[b] = 0
            mask[b, N - L:] = 1
    elif mode == "holes":
        for b in range(ids.shape[0]):
            L = int(mask[b].sum())
            for c in (L - 30, L - 21, L - 20):      # inside the 32 caption tokens that precede the 5-token gating pattern
                mask[b, c] = 0
    else:
        raise ValueError(mode)
    return ids, mask


def stress_tensors(tensors: Dict[str, torch.Tensor], config, massive_frac: float = 0.01, massive_gain: float = 20.0,
                   logit_sigma: float = 10.0, fc1_sigma: float = 2.0, seed: int = 0) -> Dict[str, float]:
    """Re-scales random-init weights IN PLACE (``tensors``: checkpoint key -> tensor, e.g. ``dict(model.named_parameters())`` or a
    state dict) so that the forward sees the statistics of a TRAINED transformer instead of the benign ones N(0, 0.02) weights
    give (tools/stress_stats.py, tests/test_e2e_gpu.py::test_stressed_statistics_*):
      * massive activations: ``massive_frac`` of the hidden channels carry ``massive_gain`` x the others' magnitude in the
        residual stream's producers (rows of proj / fc2 and their biases in the vision tower, of wo / w2 in the language tower);
      * attention logits of standard deviation ``logit_sigma`` (+-3 sigma = +-30): the q and k rows of qkv / wqkv are scaled by the
        square root of logit_sigma / (the benign logit sigma: 0.41 in the vision tower, 0.82 in the language tower - SURVEY.md §8);
      * fc1 pre-activations of standard deviation ``fc1_sigma`` (benign: 0.64): fc1 weight and bias scaled.
    Returns the factors applied."""
    vc, lc = config.vision_config, config.llm_config
    g = torch.Generator().manual_seed(1000 + seed)

    def rows(n):
        k = max(1, int(round(massive_frac * n)))
        return torch.randperm(n, generator=g)[:k]

    vit_rows, llm_rows = rows(vc.hidden_size), rows(lc.hidden_size)
    hd_v = vc.hidden_size // vc.num_attention_heads
    benign_v = (vc.hidden_size * 0.02 ** 2) * hd_v ** 0.5 * hd_v ** -0.5      # sigma of q.k / sqrt(d) under N(0, 0.02) weights, unit inputs
    hd_l = lc.hidden_size // lc.num_attention_heads
    benign_l = (lc.hidden_size * 0.02 ** 2)
    qk_v, qk_l = (logit_sigma / benign_v) ** 0.5, (logit_sigma / benign_l) ** 0.5
    fc1 = fc1_sigma / (vc.hidden_size ** 0.5 * 0.02)
    G = lc.num_attention_heads // lc.num_key_value_heads
    for key, t in tensors.items():
        with torch.no_grad():
            if ".attn.proj." in key or ".mlp.fc2." in key:
                t[vit_rows.to(t.device)] *= massive_gain
            elif key.endswith("attention.wo.weight") or key.endswith("feed_forward.w2.weight"):
                t[llm_rows.to(t.device)] *= massive_gain
            elif ".attn.qkv." in key:
                t[: 2 * vc.hidden_size] *= qk_v                       # q rows, then k rows (modeling_intern_vit.py:212-216)
            elif key.endswith("attention.wqkv.weight"):
                v = t.view(lc.num_key_value_heads, G + 2, hd_l, t.shape[1])
                v[:, : G + 1] *= qk_l                                  # q heads of the group and its k head (modeling_internlm2.py:361-371)
            elif ".mlp.fc1." in key:
                t *= fc1
    return dict(massive_rows_vit=int(vit_rows.numel()), massive_rows_llm=int(llm_rows.numel()), massive_gain=massive_gain,
                qk_scale_vit=qk_v, qk_scale_llm=qk_l, fc1_scale=fc1)
