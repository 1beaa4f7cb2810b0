"""ORACLE - test infrastructure, never the product path.

CPU restatement (plain PyTorch eager ops, any dtype; bf16 is the reference's dtype) of the
reference's reward-scoring forward, written from the reference's modules with the SAME op order
and the SAME rounding points, so that on one machine it reproduces the imported reference
bit for bit (proven by tests/golden/make_golden.py, which runs both side by side in the build
container and commits the outputs as golden vectors - the reference's Python never travels).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product (``mj-video_amd/``) must never import it.

Parity status: PINNED - against outputs of the reference itself, imported from
/root/reference/scripts/model (moe_reward.py, internvl2/*.py) with the shim in
oracle/reference_shim.py; fixtures under tests/golden/.

Each function cites the reference lines it restates (paths relative to /root/reference/scripts/model).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

GATING_PATTERN = [92542, 92543, 525, 11353, 364]  # moe_reward.py:48

# The five FFN Linears (ViT fc1 / fc2, LLM w1 / w3 / w2) go through this name: F.linear itself (the reference's bf16 math,
# bit for bit) unless oracle/ref_fp8.py swaps in its MXFP8 operand rounding for the fp8 weight path.
_ffn_linear = F.linear


# --------------------------------------------------------------------------- vision tower
def _pos_embed(pos: torch.Tensor, grid: int, H: int, W: int) -> torch.Tensor:
    """internvl2/modeling_intern_vit.py:154-160 - fp32 bicubic resample of the patch pos-emb."""
    target = pos.dtype
    p = pos.float().reshape(1, grid, grid, -1).permute(0, 3, 1, 2)
    p = F.interpolate(p, size=(H, W), mode="bicubic", align_corners=False)
    return p.reshape(1, -1, H * W).permute(0, 2, 1).to(target)


def vit_embeddings(sd: Dict[str, torch.Tensor], cfg, pixel_values: torch.Tensor) -> torch.Tensor:
    """internvl2/modeling_intern_vit.py:162-174."""
    v = cfg.vision_config
    p = "model.vision_model.embeddings."
    w, b = sd[p + "patch_embedding.weight"], sd[p + "patch_embedding.bias"]
    dt = w.dtype
    x = F.conv2d(pixel_values, w, b, stride=v.patch_size)
    B, _, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "class_embedding"].expand(B, 1, -1).to(dt)
    x = torch.cat([cls, x], dim=1)
    pos = sd[p + "position_embedding"]
    pos = torch.cat([pos[:, :1, :], _pos_embed(pos[:, 1:, :], v.image_size // v.patch_size, H, W)], dim=1)
    return x + pos.to(dt)


def vit_attention(sd, prefix: str, heads: int, x: torch.Tensor) -> torch.Tensor:
    """internvl2/modeling_intern_vit.py:210-227 (_naive_attn; qk_normalization False for the 2B tower)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    scale = (C // heads) ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def vit_layer(sd, cfg, i: int, x: torch.Tensor) -> torch.Tensor:
    """internvl2/modeling_intern_vit.py:283-295."""
    v = cfg.vision_config
    p = f"model.vision_model.encoder.layers.{i}."
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], v.layer_norm_eps).to(x.dtype)
    x = x + vit_attention(sd, p + "attn.", v.num_attention_heads, h) * sd[p + "ls1"]
    h = F.layer_norm(x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], v.layer_norm_eps).to(x.dtype)
    h = _ffn_linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)
    h = _ffn_linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h * sd[p + "ls2"]


def pixel_shuffle(x: torch.Tensor, scale: float, ps_version: str) -> torch.Tensor:
    """internvl2/modeling_internvl_chat.py:228-242."""
    n, w, h, c = x.size()
    x = x.view(n, w, int(h * scale), int(c / scale))
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, int(h * scale), int(w * scale), int(c / (scale * scale)))
    if ps_version != "v1":
        x = x.permute(0, 2, 1, 3).contiguous()
    return x


def extract_feature(sd, cfg, pixel_values: torch.Tensor, probes: Optional[dict] = None) -> torch.Tensor:
    """internvl2/modeling_internvl_chat.py:244-262 (select_layer == -1)."""
    assert cfg.select_layer == -1, "MJ-VIDEO-2B uses the last ViT layer"
    x = vit_embeddings(sd, cfg, pixel_values)
    if probes is not None:
        probes["vit_embed"] = x
    for i in range(cfg.vision_config.num_hidden_layers):
        x = vit_layer(sd, cfg, i, x)
        if probes is not None:
            probes[f"vit_layer{i}"] = x
    x = x[:, 1:, :]
    g = int(x.shape[1] ** 0.5)
    x = x.reshape(x.shape[0], g, g, -1)
    x = pixel_shuffle(x, cfg.downsample_ratio, cfg.ps_version)
    x = x.reshape(x.shape[0], -1, x.shape[-1])
    x = F.layer_norm(x, (x.shape[-1],), sd["model.mlp1.0.weight"], sd["model.mlp1.0.bias"], 1e-5)
    x = F.linear(x, sd["model.mlp1.1.weight"], sd["model.mlp1.1.bias"])
    x = F.gelu(x)
    x = F.linear(x, sd["model.mlp1.3.weight"], sd["model.mlp1.3.bias"])
    if probes is not None:
        probes["vit_embeds"] = x
    return x


# --------------------------------------------------------------------------- language tower
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """internvl2/modeling_internlm2.py:138-143 - normalise in fp32, cast, THEN multiply by the gain."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_tables(cfg, seq_len: int, dtype, state: dict = None) -> tuple:
    """internvl2/modeling_internlm2.py:147-180,204-229 - fp32 tables cast to the model dtype.

    The tables are built at construction for ``max_position_embeddings`` positions and become
    bf16 buffers under ``model.to(bfloat16)``; a forward whose (padded) sequence is longer regrows them, and the
    dynamic-NTK variant then also replaces its ``inv_freq`` FOR GOOD (:209-221).  ``state`` (a dict the caller keeps
    between forwards of one model) carries that memory - ``max_seq_len_cached`` and the base in force - so that a sequence
    of calls reproduces the reference's; without it every call starts from a freshly constructed model."""
    l = cfg.llm_config
    dim = l.hidden_size // l.num_attention_heads
    maxpos = l.max_position_embeddings
    rs = l.rope_scaling
    if state is None:
        state = {}
    state.setdefault("cached", maxpos)
    state.setdefault("base", float(l.rope_theta))
    if seq_len > state["cached"]:
        state["cached"] = seq_len
        if rs is not None and rs["type"] == "dynamic" and seq_len > maxpos:
            state["base"] = float(l.rope_theta) * ((rs["factor"] * seq_len / maxpos) - (rs["factor"] - 1)) ** (dim / (dim - 2))
    inv_freq = 1.0 / (state["base"] ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(state["cached"], dtype=inv_freq.dtype)
    if rs is not None and rs["type"] == "linear":
        t = t / rs["factor"]
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype)[:seq_len], emb.sin().to(dtype)[:seq_len]


def _rotate_half(x):
    """internvl2/modeling_internlm2.py:233-237."""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def causal_padding_mask(attention_mask: torch.Tensor, dtype) -> torch.Tensor:
    """internvl2/modeling_internlm2.py:96-125,830-851 - additive [B,1,N,N] mask of finfo.min."""
    B, N = attention_mask.shape
    m = torch.full((N, N), torch.finfo(dtype).min)
    cond = torch.arange(N)
    m.masked_fill_(cond < (cond + 1).view(N, 1), 0)
    causal = m.to(dtype)[None, None, :, :].expand(B, 1, N, N)
    expanded = attention_mask[:, None, None, :].expand(B, 1, N, N).to(dtype)
    inverted = 1.0 - expanded
    pad = inverted.masked_fill(inverted.to(torch.bool), torch.finfo(dtype).min)
    return pad + causal


def llm_attention(sd, cfg, prefix: str, x, mask, cos, sin) -> torch.Tensor:
    """internvl2/modeling_internlm2.py:341-426 (eager)."""
    l = cfg.llm_config
    B, N, _ = x.shape
    H, KV = l.num_attention_heads, l.num_key_value_heads
    D = l.hidden_size // H
    G = H // KV
    qkv = F.linear(x, sd[prefix + "wqkv.weight"])
    qkv = qkv.view(B, N, KV, G + 2, D)
    q = qkv[..., :G, :].reshape(B, N, H, D).transpose(1, 2)
    k = qkv[..., -2, :].transpose(1, 2)
    v = qkv[..., -1, :].transpose(1, 2)
    c, s = cos.unsqueeze(0).unsqueeze(1), sin.unsqueeze(0).unsqueeze(1)
    q = (q * c) + (_rotate_half(q) * s)
    k = (k * c) + (_rotate_half(k) * s)
    k = k[:, :, None, :, :].expand(B, KV, G, N, D).reshape(B, H, N, D)
    v = v[:, :, None, :, :].expand(B, KV, G, N, D).reshape(B, H, N, D)
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D)
    w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(B, N, H * D)
    return F.linear(o, sd[prefix + "wo.weight"])


def llm_layer(sd, cfg, i: int, x, mask, cos, sin) -> torch.Tensor:
    """internvl2/modeling_internlm2.py:621-681,261-264."""
    eps = cfg.llm_config.rms_norm_eps
    p = f"model.language_model.model.layers.{i}."
    h = rms_norm(x, sd[p + "attention_norm.weight"], eps)
    x = x + llm_attention(sd, cfg, p + "attention.", h, mask, cos, sin)
    h = rms_norm(x, sd[p + "ffn_norm.weight"], eps)
    h = _ffn_linear(F.silu(_ffn_linear(h, sd[p + "feed_forward.w1.weight"])) *
                    _ffn_linear(h, sd[p + "feed_forward.w3.weight"]), sd[p + "feed_forward.w2.weight"])
    return x + h


# --------------------------------------------------------------------------- heads
def gating_mlp(sd, net: str, n_layers: int, x: torch.Tensor) -> torch.Tensor:
    """moe_reward.py:29-32,38-42 - Linear+ReLU stack, last layer linear."""
    for j in range(n_layers):
        x = F.linear(x, sd[f"{net}.layers.{j}.weight"], sd[f"{net}.layers.{j}.bias"])
        if j < n_layers - 1:
            x = F.relu(x)
    return x


def find_token_for_gating(lst: Sequence[int]) -> int:
    """moe_reward.py:50-57 - last occurrence of the 5-token pattern."""
    n = len(GATING_PATTERN)
    for j in range(len(lst) - n, -1, -1):
        if list(lst[j:j + n]) == GATING_PATTERN:
            return j
    raise ValueError("Token pattern not found in the list.")


@torch.no_grad()
def reward_forward(sd: Dict[str, torch.Tensor], cfg, pixel_values: torch.Tensor, input_ids: torch.Tensor,
                   attention_mask: Optional[torch.Tensor], img_context_token_id: int,
                   pad_token_id: Optional[int], lm_head: bool = False,
                   probes: Optional[dict] = None, rope_state: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """moe_reward.py:183-297 on top of internvl2/modeling_internvl_chat.py:146-226.

    ``sd`` is the checkpoint-layout state dict (all tensors of one dtype).  Returns the
    ``CustomOutput`` fields as a dict.  ``lm_head=True`` also performs the reference's unused
    92553-wide logits GEMM (modeling_internlm2.py:1080-1082) so CPU timings match what the
    reference executes.
    """
    l = cfg.llm_config
    emb = sd["model.language_model.model.tok_embeddings.weight"]
    x = F.embedding(input_ids, emb).clone()
    vit = extract_feature(sd, cfg, pixel_values, probes)
    B, N, C = x.shape
    x = x.reshape(B * N, C)
    sel = input_ids.reshape(B * N) == img_context_token_id
    if int(sel.sum()) != vit.reshape(-1, C).shape[0]:
        raise ValueError(f"{int(sel.sum())} <IMG_CONTEXT> tokens but {vit.reshape(-1, C).shape[0]} image embeddings")
    x[sel] = x[sel] * 0.0 + vit.reshape(-1, C)
    x = x.reshape(B, N, C)
    if probes is not None:
        probes["llm_embed"] = x

    if attention_mask is None:
        attention_mask = torch.ones((B, N), dtype=torch.bool)
    mask = causal_padding_mask(attention_mask, x.dtype)
    cos, sin = rope_tables(cfg, N, x.dtype, rope_state)   # (rope_state: the rotary cache memory of ONE model across forwards)
    for i in range(l.num_hidden_layers):
        x = llm_layer(sd, cfg, i, x, mask, cos, sin)
        if probes is not None:
            probes[f"llm_layer{i}"] = x
    x = rms_norm(x, sd["model.language_model.model.norm.weight"], l.rms_norm_eps)
    if lm_head:
        F.linear(x, sd["model.language_model.output.weight"]).float()

    if pad_token_id is None and B != 1:
        raise ValueError("Cannot handle batch sizes > 1 if no padding token is defined.")
    if pad_token_id is None:
        seqlen = torch.full((B,), -1, dtype=torch.long)
    else:
        seqlen = torch.eq(input_ids, pad_token_id).int().argmax(-1) - 1
        seqlen = seqlen % N
    rows = torch.arange(B)
    h_r = x[rows, seqlen]
    gpos = [find_token_for_gating(ids.tolist()) for ids in input_ids]
    h_g = x[rows, gpos, :]
    return reward_heads(sd, cfg, h_r, h_g)


@torch.no_grad()
def reward_heads(sd: Dict[str, torch.Tensor], cfg, h_r: torch.Tensor, h_g: torch.Tensor) -> Dict[str, torch.Tensor]:
    """moe_reward.py:239-297 - everything downstream of the two hidden-state rows: ``h_r`` [B, hidden] is the
    post-norm state at the last non-pad token (:229,243), ``h_g`` [B, hidden] at the gating pattern (:242-245)."""
    B = h_r.shape[0]
    rewards = F.linear(h_r, sd["regression_layer.weight"])
    rewards = rewards @ sd["reward_transform_matrix"]

    T = cfg.gating_temperature
    nl = cfg.gating_n_hidden + 1
    a = gating_mlp(sd, "aspect_gating", nl, h_g)
    aspect_gating_output = F.softmax(a / T, dim=1) * sd["aspect_gating.logit_scale"][0]
    crit = gating_mlp(sd, "criteria_gating", nl, h_g)
    a2c = {int(k): list(v) for k, v in cfg.aspect2criteria.items()}
    weights = {}
    for aspect, idx in a2c.items():
        weights[aspect] = F.softmax(crit[:, idx] / T, dim=-1) * sd["criteria_gating.logit_scale"][0]
    aspect_scores = torch.zeros(B, len(a2c))
    weighted = None
    for i, (aspect, idx) in enumerate(a2c.items()):
        weighted = (rewards[:, idx] * weights[aspect]).sum(dim=-1)
        aspect_scores[:, i] = weighted
    score = (aspect_scores * aspect_gating_output).sum(dim=-1)
    return dict(rewards=rewards, hidden_state=h_r, prompt_embedding=h_g, criteria_gating_output=crit,
                aspect_gating_output=aspect_gating_output,
                aspect_weights=torch.cat([weights[a_] for a_ in a2c], dim=-1),
                score=score, weighted_scores=weighted, aspect_scores=aspect_scores)


OUTPUT_FIELDS = ("rewards", "hidden_state", "prompt_embedding", "criteria_gating_output",
                 "aspect_gating_output", "aspect_weights", "score", "weighted_scores", "aspect_scores")
