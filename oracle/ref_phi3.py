"""ORACLE - test infrastructure, never the product path.

CPU restatement of the language tower BASELINE.json configs[4] names ("InternVL2-4B backbone"): InternVL2-4B = the InternViT
vision tower and glue of the 2B model + a Phi-3-mini decoder.  The reference itself cannot build that model
(/root/reference/scripts/model/internvl2/modeling_internvl_chat.py:125-130 accepts Llama / InternLM2 only; README.md:18
"MJ-VIDEO-4B coming soon"), but its ``InternVLChatModel`` takes a ready ``language_model`` (:100,121-122), which is how the
upstream 4B checkpoint is wired.  The authoritative source of the decoder is therefore a THIRD-PARTY dependency:

    transformers 5.15.0, transformers/models/phi3/modeling_phi3.py   (pinned: the version installed in the build image)

restated here in plain PyTorch ops with the same op order and rounding points (eager attention: the reference's
``InternVLChatModel.__init__`` sets ``attn_implementation = 'eager'`` without flash-attn, modeling_internvl_chat.py:114).
Everything outside the decoder (vision tower, projector, splice, heads) is oracle/ref_cpu.py's restatement of the reference.

Parity status: PINNED - tests/golden/make_golden_phi3.py builds the REFERENCE's InternVLChatRewardModeling around
transformers' own ``Phi3ForCausalLM`` (imported in the build container), requires this file to reproduce its outputs bit for
bit, and commits the outputs as fixtures (tests/golden/phi3_*.npz); nothing of transformers travels to the GPU box.
What stays unpinned (no upstream files offline): the 4B checkpoint's real LongRoPE factor lists and tokenizer ids - the
fixtures use seed-defined factors and stand-in ids, the arithmetic is the same.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ref_cpu

# the three FFN-side Linear calls go through this name so that oracle/ref_fp8.py's operand rounding can be swapped in
_ffn_linear = F.linear


def rope_scaling_of(l) -> Optional[dict]:
    rs = getattr(l, "rope_scaling", None)
    return rs if rs else None


def attention_factor(l) -> float:
    """transformers/modeling_rope_utils.py:_compute_longrope_parameters - sqrt(1 + ln(factor) / ln(original_max)) with
    factor = max_position_embeddings / original_max_position_embeddings (Phi-3's convention); 1.0 without LongRoPE."""
    rs = rope_scaling_of(l)
    if rs is None:
        return 1.0
    if rs.get("attention_factor") is not None:
        return float(rs["attention_factor"])
    orig = l.original_max_position_embeddings
    factor = rs.get("factor")
    if factor is None:
        factor = l.max_position_embeddings / orig
    return 1.0 if factor <= 1.0 else math.sqrt(1 + math.log(factor) / math.log(orig))


def inv_freq(l, seq_len: int, dtype=torch.float32) -> torch.Tensor:
    """modeling_phi3.py:Phi3RotaryEmbedding.compute_default_rope_parameters / modeling_rope_utils.py:_compute_longrope_parameters +
    longrope_frequency_update: the LONG factors when the (padded) sequence is longer than original_max_position_embeddings,
    the SHORT ones otherwise - chosen per forward from ``max(position_ids) + 1``, no state kept between forwards.

    ``dtype`` = the model's dtype.  The SHORT (and the unscaled) frequencies live in the module's ``inv_freq`` /
    ``original_inv_freq`` BUFFERS, which ``model.to(torch.bfloat16)`` - the reference driver's set-up order,
    eval_genai_mjvideo.py:112-116 - casts like every floating buffer: they reach the forward rounded to bf16 (the same fate as
    InternLM2's cos / sin caches).  The LONG frequencies are recomputed in fp32 inside every forward that
    needs them (longrope_frequency_update) and are never rounded."""
    dim = int((l.hidden_size // l.num_attention_heads) * getattr(l, "partial_rotary_factor", 1.0))
    shape = torch.arange(0, dim, 2, dtype=torch.int64).float() / dim
    rs = rope_scaling_of(l)
    if rs is None:
        return (1.0 / (float(l.rope_theta) ** shape)).to(dtype).float()
    if seq_len > l.original_max_position_embeddings:
        return 1.0 / (torch.tensor(rs["long_factor"], dtype=torch.float32) * float(l.rope_theta) ** shape)
    return (1.0 / (torch.tensor(rs["short_factor"], dtype=torch.float32) * float(l.rope_theta) ** shape)).to(dtype).float()


def rope_tables(cfg, seq_len: int, dtype, buffer_dtype=None):
    """modeling_phi3.py:Phi3RotaryEmbedding.forward for position_ids = arange(seq_len) (the reference passes none,
    modeling_internvl_chat.py:190-199): fp32 outer product, cat(freqs, freqs), cos / sin times the attention factor, cast.
    ``buffer_dtype``: the dtype the rotary module's buffers were cast to (default: ``dtype``, i.e. ``model.to(dtype)``).  The
    fp32 runs that measure the bf16 run's NOISE pass bfloat16 here: with fp32 buffers the short-factor frequencies differ from
    the bf16 model's by up to 2^-9 relative - radians of phase at positions in the thousands, a different positional encoding,
    not rounding noise (one decoder layer at 4B dims: 6.8 % against 0.56 % with the same frequencies)."""
    l = cfg.llm_config
    f = inv_freq(l, seq_len, dtype if buffer_dtype is None else buffer_dtype)
    pos = torch.arange(seq_len, dtype=torch.float32)
    freqs = (f[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]
    emb = torch.cat((freqs, freqs), dim=-1)
    a = attention_factor(l)
    return (emb.cos() * a).to(dtype), (emb.sin() * a).to(dtype)


def causal_padding_mask(attention_mask: torch.Tensor, dtype) -> torch.Tensor:
    """transformers/masking_utils.py:create_causal_mask, eager interface: [B, 1, N, N] additive mask, 0 where key <= query and the
    key is not padding, finfo.min elsewhere (one value, not a sum: eager_mask builds it with torch.where)."""
    B, N = attention_mask.shape
    q = torch.arange(N)
    allowed = (q[None, :] <= q[:, None])[None, :, :] & attention_mask.bool()[:, None, :]
    m = torch.where(allowed, torch.tensor(0.0, dtype=dtype), torch.tensor(torch.finfo(dtype).min, dtype=dtype))
    return m[:, None, :, :]


def attention(sd, cfg, prefix: str, x, mask, cos, sin) -> torch.Tensor:
    """modeling_phi3.py:Phi3Attention.forward + eager_attention_forward + apply_rotary_pos_emb."""
    l = cfg.llm_config
    B, N, _ = x.shape
    H, KV = l.num_attention_heads, l.num_key_value_heads
    D = l.hidden_size // H
    qkv = F.linear(x, sd[prefix + "qkv_proj.weight"])
    qp = H * D
    q = qkv[..., :qp].view(B, N, -1, D).transpose(1, 2)
    k = qkv[..., qp:qp + KV * D].view(B, N, -1, D).transpose(1, 2)
    v = qkv[..., qp + KV * D:].view(B, N, -1, D).transpose(1, 2)
    c, s = cos.unsqueeze(0).unsqueeze(1), sin.unsqueeze(0).unsqueeze(1)
    rd = c.shape[-1]
    rot = ref_cpu._rotate_half
    q = torch.cat([(q[..., :rd] * c) + (rot(q[..., :rd]) * s), q[..., rd:]], dim=-1)
    k = torch.cat([(k[..., :rd] * c) + (rot(k[..., :rd]) * s), k[..., rd:]], dim=-1)
    G = H // KV
    if G > 1:
        k = k[:, :, None, :, :].expand(B, KV, G, N, D).reshape(B, H, N, D)
        v = v[:, :, None, :, :].expand(B, KV, G, N, D).reshape(B, H, N, D)
    w = torch.matmul(q, k.transpose(2, 3)) * (D ** -0.5)
    w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(B, N, -1).contiguous()
    return F.linear(o, sd[prefix + "o_proj.weight"])


def mlp(sd, prefix: str, h: torch.Tensor) -> torch.Tensor:
    """modeling_phi3.py:Phi3MLP.forward: gate, up = gate_up_proj(h).chunk(2); down_proj(up * silu(gate))."""
    up = _ffn_linear(h, sd[prefix + "gate_up_proj.weight"])
    gate, up = up.chunk(2, dim=-1)
    return _ffn_linear(up * F.silu(gate), sd[prefix + "down_proj.weight"])


def layer(sd, cfg, i: int, x, mask, cos, sin) -> torch.Tensor:
    """modeling_phi3.py:Phi3DecoderLayer.forward (dropouts are identities in eval)."""
    eps = cfg.llm_config.rms_norm_eps
    p = f"model.language_model.model.layers.{i}."
    h = ref_cpu.rms_norm(x, sd[p + "input_layernorm.weight"], eps)        # Phi3RMSNorm == InternLM2RMSNorm (cast before gain)
    x = x + attention(sd, cfg, p + "self_attn.", h, mask, cos, sin)
    h = ref_cpu.rms_norm(x, sd[p + "post_attention_layernorm.weight"], eps)
    return x + mlp(sd, p + "mlp.", h)


def find_token_for_gating(lst: Sequence[int], pattern: Sequence[int]) -> int:
    """moe_reward.py:50-57 with the pattern a parameter (``token_pattern`` there is the InternLM2 tokenizer's ids of
    ``<|im_end|><|im_start|>assistant\\n``; the phi3-chat template's counterpart is ``<|end|><|assistant|>\\n``)."""
    n = len(pattern)
    pattern = list(pattern)
    for j in range(len(lst) - n, -1, -1):
        if list(lst[j:j + n]) == pattern:
            return j
    raise ValueError("Token pattern not found in the list.")


@torch.no_grad()
def reward_forward(sd: Dict[str, torch.Tensor], cfg, pixel_values: torch.Tensor, input_ids: torch.Tensor,
                   attention_mask: Optional[torch.Tensor], img_context_token_id: int, pad_token_id: Optional[int],
                   gating_pattern: Sequence[int], lm_head: bool = False, probes: Optional[dict] = None,
                   rope_buffer_dtype=None) -> Dict[str, torch.Tensor]:
    """moe_reward.py:183-297 on modeling_internvl_chat.py:146-226 with a Phi-3 language model (see the module docstring).
    ``rope_buffer_dtype``: see ``rope_tables`` (fp32 noise-floor runs of a bf16 model pass torch.bfloat16)."""
    l = cfg.llm_config
    x = F.embedding(input_ids, sd["model.language_model.model.embed_tokens.weight"]).clone()
    vit = ref_cpu.extract_feature(sd, cfg, pixel_values, probes)
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
    cos, sin = rope_tables(cfg, N, x.dtype, rope_buffer_dtype)
    for i in range(l.num_hidden_layers):
        x = layer(sd, cfg, i, x, mask, cos, sin)
        if probes is not None:
            probes[f"llm_layer{i}"] = x
    x = ref_cpu.rms_norm(x, sd["model.language_model.model.norm.weight"], l.rms_norm_eps)
    if lm_head:
        F.linear(x, sd["model.language_model.lm_head.weight"])
    if pad_token_id is None and B != 1:
        raise ValueError("Cannot handle batch sizes > 1 if no padding token is defined.")
    if pad_token_id is None:
        seqlen = torch.full((B,), -1, dtype=torch.long)
    else:
        seqlen = torch.eq(input_ids, pad_token_id).int().argmax(-1) - 1
        seqlen = seqlen % N
    rows = torch.arange(B)
    h_r = x[rows, seqlen]
    gpos = [find_token_for_gating(ids.tolist(), gating_pattern) for ids in input_ids]
    h_g = x[rows, gpos, :]
    return ref_cpu.reward_heads(sd, cfg, h_r, h_g)
