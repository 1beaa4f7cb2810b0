"""Host-side mirror of the reference's reward model, running on the gfx950 kernels of libmjv_hip.so.

``InternVLChatRewardModeling`` keeps the reference's class API (scripts/model/moe_reward.py:137-297):
same constructor, same attributes callers touch (``.model.img_context_token_id``, ``.model.device``,
``.config.pad_token_id``, ``.regression_layer`` ...), same ``forward`` signature and ``CustomOutput``
fields, same checkpoint key layout (``load_state_dict(strict=True)`` of an MJ-VIDEO checkpoint works).
The nn.Modules below only HOLD parameters under the reference's names; ``forward`` never calls a
torch op on them - every computation is a C-ABI call (mj_video_amd.ops).  There is no CPU path:
without the HIP library or a GPU the forward raises.

Differences that are deliberate, all output-preserving:
  * right-padded batches are packed (pad rows are never computed; the reference proves
    padded-batch == per-sample results);
  * the LM head GEMM and the 25 retained hidden states are skipped (their results are unused);
  * attention never materialises N x N scores.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, fields
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_ROPE_QKV, EPI_SCALE_RES, EPI_SILU_MUL, HeadsDesc)
from .chat_input import get_conv_template
from .configuration import InternVLChatConfig, InternVLChatRewardModelingConfig, Phi3Config

BF16 = torch.bfloat16

# the Linears the "mxfp8" FFN format can cover, and named subsets of them (set_ffn_format)
FFN_LINEARS = ("fc1", "fc2", "w13", "w2")
# Measured on both engineered rank sets against the reference's bf16 scores (profiles/r06_a_fp8_ffn_subset_study.txt, all 15
# subsets): w1 | w3 alone costs more rank agreement (5.1 - 5.8 x the reference's bf16 noise, rho 0.9988 - 0.9992) than the other
# three Linears together.  "mxfp8-rank999" = the LARGEST subset that keeps north_star's bar on both sets - Spearman >= 0.999
# (0.99920 @224^2, 0.99912 @448^2) and 0 flips on the decisive pairs; "mxfp8-vit" = the vision tower's FFN only (0.99959 /
# 0.99962).  The full set ("mxfp8": 0.99863 / 0.99820) keeps its own stated, looser tolerance.
FP8_PRESETS: Dict[str, frozenset] = {
    "mxfp8-rank999": frozenset(("fc1", "fc2", "w2")),
    "mxfp8-vit": frozenset(("fc1", "fc2")),
}

# `<|im_end|><|im_start|>assistant\n` in InternLM2 token ids (moe_reward.py:45-48)
token_pattern = [92542, 92543, 525, 11353, 364]


def find_token_for_gating(lst, pattern=None) -> int:
    """Index of the LAST occurrence of ``token_pattern`` in ``lst`` (moe_reward.py:50-57).  ``pattern``: another tokenizer's ids
    of the same marker (``config.gating_token_pattern``: the Phi-3 backbone of BASELINE configs[4]); default = the reference's
    module constant."""
    a = np.asarray(lst)
    pattern = token_pattern if pattern is None else list(pattern)
    n = len(pattern)
    if a.shape[0] >= n:
        hit = np.ones(a.shape[0] - n + 1, dtype=bool)
        for j, t in enumerate(pattern):
            hit &= a[j:a.shape[0] - n + 1 + j] == t
        idx = np.flatnonzero(hit)
        if idx.size:
            return int(idx[-1])
    raise ValueError("Token pattern not found in the list.")


@dataclass
class CustomOutput:
    """Field-for-field mirror of moe_reward.py:60-89 (attribute, key and index access like HF ModelOutput)."""
    rewards: Optional[torch.Tensor] = None
    hidden_state: Optional[torch.Tensor] = None
    prompt_embedding: Optional[torch.Tensor] = None
    criteria_gating_output: Optional[torch.Tensor] = None
    aspect_gating_output: Optional[torch.Tensor] = None
    aspect_weights: Optional[torch.Tensor] = None
    score: Optional[torch.Tensor] = None
    weighted_scores: Optional[torch.Tensor] = None
    aspect_scores: Optional[torch.Tensor] = None

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in fields(self) if getattr(self, f.name) is not None)

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else self.to_tuple()[k]

    def keys(self):
        return [f.name for f in fields(self) if getattr(self, f.name) is not None]


# ------------------------------------------------------------------------------- parameter holders
class _Linear(nn.Module):
    def __init__(self, fin: int, fout: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(fout), requires_grad=False)
        else:
            self.register_parameter("bias", None)


class _Norm(nn.Module):
    def __init__(self, dim: int, bias: bool, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim), requires_grad=False)


class _Act(nn.Module):  # index placeholder so mlp1 keeps the reference's 0/1/3 numbering
    pass


class _PatchEmbedding(nn.Module):
    def __init__(self, dim: int, patch: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, 3, patch, patch), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(dim), requires_grad=False)


class _VisionEmbeddings(nn.Module):
    def __init__(self, vc):
        super().__init__()
        npos = (vc.image_size // vc.patch_size) ** 2 + 1
        self.class_embedding = nn.Parameter(torch.empty(1, 1, vc.hidden_size), requires_grad=False)
        self.position_embedding = nn.Parameter(torch.empty(1, npos, vc.hidden_size), requires_grad=False)
        self.patch_embedding = _PatchEmbedding(vc.hidden_size, vc.patch_size)


class _VisionAttention(nn.Module):
    def __init__(self, vc):
        super().__init__()
        self.qkv = _Linear(vc.hidden_size, 3 * vc.hidden_size, bias=vc.qkv_bias)
        self.proj = _Linear(vc.hidden_size, vc.hidden_size)


class _VisionMLP(nn.Module):
    def __init__(self, vc):
        super().__init__()
        self.fc1 = _Linear(vc.hidden_size, vc.intermediate_size)
        self.fc2 = _Linear(vc.intermediate_size, vc.hidden_size)


class _VisionLayer(nn.Module):
    def __init__(self, vc):
        super().__init__()
        d = vc.hidden_size
        self.ls1 = nn.Parameter(torch.ones(d), requires_grad=False)
        self.ls2 = nn.Parameter(torch.ones(d), requires_grad=False)
        self.attn = _VisionAttention(vc)
        self.mlp = _VisionMLP(vc)
        self.norm1 = _Norm(d, True, vc.layer_norm_eps)
        self.norm2 = _Norm(d, True, vc.layer_norm_eps)


class _VisionEncoder(nn.Module):
    def __init__(self, vc):
        super().__init__()
        self.layers = nn.ModuleList([_VisionLayer(vc) for _ in range(vc.num_hidden_layers)])


class InternVisionModel(nn.Module):
    """Parameter layout of internvl2/modeling_intern_vit.py:364-430."""

    def __init__(self, vc):
        super().__init__()
        if vc.norm_type != "layer_norm" or vc.qk_normalization:
            raise NotImplementedError("only the layer_norm / no-qk-norm InternViT (InternVL2-2B tower) is built")
        if vc.hidden_size // vc.num_attention_heads != 64:
            raise NotImplementedError("ViT attention kernel is specialised on head_dim 64")
        self.config = vc
        self.embeddings = _VisionEmbeddings(vc)
        self.encoder = _VisionEncoder(vc)


class _LMAttention(nn.Module):
    def __init__(self, lc):
        super().__init__()
        hd = lc.hidden_size // lc.num_attention_heads
        self.wqkv = _Linear(lc.hidden_size, (lc.num_attention_heads + 2 * lc.num_key_value_heads) * hd, bias=lc.bias)
        self.wo = _Linear(lc.num_attention_heads * hd, lc.hidden_size, bias=lc.bias)


class _LMFeedForward(nn.Module):
    def __init__(self, lc):
        super().__init__()
        self.w1 = _Linear(lc.hidden_size, lc.intermediate_size, bias=False)
        self.w3 = _Linear(lc.hidden_size, lc.intermediate_size, bias=False)
        self.w2 = _Linear(lc.intermediate_size, lc.hidden_size, bias=False)


class _LMLayer(nn.Module):
    def __init__(self, lc):
        super().__init__()
        self.attention = _LMAttention(lc)
        self.feed_forward = _LMFeedForward(lc)
        self.attention_norm = _Norm(lc.hidden_size, False, lc.rms_norm_eps)
        self.ffn_norm = _Norm(lc.hidden_size, False, lc.rms_norm_eps)


class _Embedding(nn.Module):
    def __init__(self, n: int, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d), requires_grad=False)


class InternLM2Model(nn.Module):
    def __init__(self, lc):
        super().__init__()
        self.tok_embeddings = _Embedding(lc.vocab_size, lc.hidden_size)
        self.layers = nn.ModuleList([_LMLayer(lc) for _ in range(lc.num_hidden_layers)])
        self.norm = _Norm(lc.hidden_size, False, lc.rms_norm_eps)


class InternLM2ForCausalLM(nn.Module):
    """Parameter layout of internvl2/modeling_internlm2.py:987-1000 (``output`` is loaded, never used)."""

    def __init__(self, lc):
        super().__init__()
        if lc.bias:
            raise NotImplementedError("InternLM2 with attention biases is not built (MJ-VIDEO-2B has bias=False)")
        if lc.hidden_size // lc.num_attention_heads != 128:
            raise NotImplementedError("LLM attention kernel is specialised on head_dim 128")
        self.config = lc
        self.model = InternLM2Model(lc)
        self.output = _Linear(lc.hidden_size, lc.vocab_size, bias=False)

    def get_input_embeddings(self):
        return self.model.tok_embeddings


class _Alias:
    """attribute bag (not a Module: the parameters stay registered under their checkpoint names only)"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Phi3Attention(nn.Module):
    def __init__(self, lc):
        super().__init__()
        hd = lc.hidden_size // lc.num_attention_heads
        self.o_proj = _Linear(lc.num_attention_heads * hd, lc.hidden_size, bias=False)
        self.qkv_proj = _Linear(lc.hidden_size, (lc.num_attention_heads + 2 * lc.num_key_value_heads) * hd, bias=False)


class _Phi3MLP(nn.Module):
    def __init__(self, lc):
        super().__init__()
        self.gate_up_proj = _Linear(lc.hidden_size, 2 * lc.intermediate_size, bias=False)
        self.down_proj = _Linear(lc.intermediate_size, lc.hidden_size, bias=False)


class _Phi3Layer(nn.Module):
    """Parameter layout of transformers/models/phi3/modeling_phi3.py:Phi3DecoderLayer; the properties give the decoder loop the
    InternLM2 names of the same roles (attention_norm / ffn_norm / attention.wqkv / attention.wo / feed_forward.w2)."""

    def __init__(self, lc):
        super().__init__()
        self.self_attn = _Phi3Attention(lc)
        self.mlp = _Phi3MLP(lc)
        self.input_layernorm = _Norm(lc.hidden_size, False, lc.rms_norm_eps)
        self.post_attention_layernorm = _Norm(lc.hidden_size, False, lc.rms_norm_eps)

    @property
    def attention_norm(self):
        return self.input_layernorm

    @property
    def ffn_norm(self):
        return self.post_attention_layernorm

    @property
    def attention(self):
        return _Alias(wqkv=self.self_attn.qkv_proj, wo=self.self_attn.o_proj)

    @property
    def feed_forward(self):
        return _Alias(w2=self.mlp.down_proj)


class Phi3Model(nn.Module):
    def __init__(self, lc):
        super().__init__()
        self.embed_tokens = _Embedding(lc.vocab_size, lc.hidden_size)
        self.layers = nn.ModuleList([_Phi3Layer(lc) for _ in range(lc.num_hidden_layers)])
        self.norm = _Norm(lc.hidden_size, False, lc.rms_norm_eps)

    @property
    def tok_embeddings(self):
        return self.embed_tokens


class Phi3ForCausalLM(nn.Module):
    """Parameter layout of transformers/models/phi3/modeling_phi3.py:Phi3ForCausalLM - the language model of the InternVL2-4B
    backbone BASELINE configs[4] names (``lm_head`` is loaded, never used).  The reference's dispatch has no such branch
    (modeling_internvl_chat.py:125-130); the upstream 4B checkpoint's code adds it at exactly that place."""

    def __init__(self, lc):
        super().__init__()
        hd = lc.hidden_size // lc.num_attention_heads
        if hd != 96:
            raise NotImplementedError(f"Phi-3 attention is built for head_dim 96 (Phi-3-mini), got {hd}")
        if lc.partial_rotary_factor != 1.0:
            raise NotImplementedError("Phi-3 with a partial rotary factor is not built (Phi-3-mini rotates the whole head)")
        if lc.sliding_window is not None and lc.sliding_window < lc.max_position_embeddings:
            raise NotImplementedError("a sliding window shorter than the context is not built (Phi-3-mini-128k: 262144)")
        self.config = lc
        self.model = Phi3Model(lc)
        self.lm_head = _Linear(lc.hidden_size, lc.vocab_size, bias=False)

    def get_input_embeddings(self):
        return self.model.embed_tokens


class InternVLChatModel(nn.Module):
    """Parameter layout + attributes of internvl2/modeling_internvl_chat.py:100-144."""

    def __init__(self, config: InternVLChatConfig):
        super().__init__()
        self.config = config
        image_size = config.force_image_size or config.vision_config.image_size
        self.patch_size = config.vision_config.patch_size
        self.select_layer = config.select_layer
        self.template = config.template
        self.downsample_ratio = config.downsample_ratio
        self.ps_version = config.ps_version
        self.num_image_token = int((image_size // self.patch_size) ** 2 * (config.downsample_ratio ** 2))
        if config.select_layer != -1:
            raise NotImplementedError("select_layer != -1 is not built (MJ-VIDEO-2B uses the last ViT layer)")
        if config.ps_version == "v1" or config.downsample_ratio != 0.5:
            raise NotImplementedError("only pixel-shuffle v2 with downsample_ratio 0.5 is built")
        self.vision_model = InternVisionModel(config.vision_config)
        # the reference's dispatch (modeling_internvl_chat.py:125-130) + the Phi-3 branch of the upstream 4B checkpoint
        if isinstance(config.llm_config, Phi3Config):
            self.language_model = Phi3ForCausalLM(config.llm_config)
        else:
            self.language_model = InternLM2ForCausalLM(config.llm_config)
        vit_hidden, llm_hidden = config.vision_config.hidden_size, config.llm_config.hidden_size
        c4 = vit_hidden * int(1 / self.downsample_ratio) ** 2
        self.mlp1 = nn.Sequential(_Norm(c4, True, 1e-5), _Linear(c4, llm_hidden), _Act(), _Linear(llm_hidden, llm_hidden))
        self.img_context_token_id = None
        self.conv_template = get_conv_template(self.template) if self.template else None
        self.system_message = self.conv_template.system_message if self.conv_template else None

    @property
    def device(self):
        return self.mlp1[1].weight.device

    @property
    def dtype(self):
        return self.mlp1[1].weight.dtype

    @classmethod
    def from_pretrained(cls, name_or_path, config: Optional[InternVLChatConfig] = None,
                        allow_uninitialized: bool = False, **kwargs):
        """Builds the model from ``config.json`` in a local directory and loads the ``*.safetensors`` next to it
        (no hub access exists here; the reference's HF call is modeling_internvl_chat.py / moe_reward.py:142).
        Like the reference, it either returns real base weights or fails: a path that is not a local directory, or
        one without ``*.safetensors``, raises FileNotFoundError unless ``allow_uninitialized=True`` (the caller then
        owns the ``torch.empty`` parameters and must ``load_state_dict`` a full checkpoint before scoring)."""
        if config is None:
            config = InternVLChatConfig.from_pretrained(name_or_path)
        model = cls(config)
        files = []
        if isinstance(name_or_path, str) and os.path.isdir(name_or_path):
            files = sorted(f for f in os.listdir(name_or_path) if f.endswith(".safetensors"))
        if files:
            from safetensors.torch import load_file
            sd = {}
            for f in files:
                sd.update(load_file(os.path.join(name_or_path, f)))
            model.load_state_dict(sd, strict=True)
        elif not allow_uninitialized:
            raise FileNotFoundError(f"no *.safetensors weights under {name_or_path!r}: the model would score with "
                                    "uninitialised parameters (pass allow_uninitialized=True to build the skeleton only)")
        return model


class GatingNetwork(nn.Module):
    """Parameter layout of moe_reward.py:16-42."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, temperature: float = 10,
                 logit_scale: float = 1., hidden_dim: int = 1024, n_hidden: int = 3):
        super().__init__()
        self.temperature = temperature
        self.logit_scale = nn.Parameter(torch.ones(1) * logit_scale, requires_grad=False)
        layers = []
        for _ in range(n_hidden):
            layers.append(_Linear(in_features, hidden_dim))
            in_features = hidden_dim
        layers.append(_Linear(in_features, out_features, bias=bias))
        self.layers = nn.ModuleList(layers)


# ------------------------------------------------------------------------------------ the reward model
class InternVLChatRewardModeling(nn.Module):
    def __init__(self, name: str, config, base_model: Optional[InternVLChatModel] = None, allow_uninitialized: bool = False):
        """``allow_uninitialized``: build the base model's skeleton even if ``name`` holds no ``*.safetensors`` - for callers
        that load a FULL checkpoint right after (``load_state_dict(strict=True)`` then leaves no parameter uninitialised:
        scripts/eval/eval_genai_mjvideo.py with ``--checkpoint_path``); the default refuses, like ``from_pretrained``."""
        super().__init__()
        self.num_labels = getattr(config, "num_labels", 2)
        self.model = base_model if base_model is not None else InternVLChatModel.from_pretrained(
            name, config=config, allow_uninitialized=allow_uninitialized)
        config_dict = config.to_dict()
        self.num_objectives = config_dict["num_objectives"]
        self.num_aspects = config_dict["num_aspects"]
        self.aspect2criteria = {k: list(v) for k, v in config_dict["aspect2criteria"].items()}
        # sanity checks of moe_reward.py:153-157
        assert len(self.aspect2criteria) == self.num_aspects
        assert sum(len(v) for v in self.aspect2criteria.values()) == self.num_objectives
        temp = []
        for k in self.aspect2criteria.values():
            temp += k
        assert sum(len(set(v)) for v in self.aspect2criteria.values()) == len(set(temp))

        hidden_size = config.llm_config.hidden_size
        self.regression_layer = _Linear(hidden_size, self.num_objectives, bias=False)
        self.reward_transform_matrix = nn.Parameter(torch.eye(self.num_objectives), requires_grad=False)
        self.aspect_gating = GatingNetwork(hidden_size, self.num_aspects, temperature=config_dict["gating_temperature"],
                                           hidden_dim=config_dict["gating_hidden_dim"],
                                           n_hidden=config_dict["gating_n_hidden"])
        self.criteria_gating = GatingNetwork(hidden_size, config.num_objectives,
                                             temperature=config_dict["gating_temperature"],
                                             hidden_dim=config_dict["gating_hidden_dim"],
                                             n_hidden=config_dict["gating_n_hidden"])
        self.config = config
        self._phi3 = isinstance(config.llm_config, Phi3Config)   # language tower: Phi-3 (configs[4]) instead of InternLM2
        self._derived: Dict[str, object] = {}
        self._derived_sig = None
        self._ws: Dict[str, torch.Tensor] = {}
        self._rope: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._rope_state = None   # the reference's rotary cache state (see _rope_tables)
        self.last_packed34: Optional[torch.Tensor] = None
        self._ws_tag = "g0"
        self._host_cache = None
        self._side_streams = {}   # device -> stream of the asynchronous ids copy (see _host_ids_begin)
        self.debug_probes: Optional[Dict[str, torch.Tensor]] = None  # tests set {} to capture per-layer states
        self.use_gemm_workspace = True   # hand the GEMMs of a forward a split-K scratch (False: no GEMM slices K; tests)
        # "bf16" (default: the reference's arithmetic) or "mxfp8": the five FFN Linears of both towers (fc1 / fc2, w1 | w3 / w2 -
        # two thirds of the flops) run on MXFP8 operands (include/mjv.h; SURVEY.md §8(f)4, BASELINE configs[4]); weights are
        # quantised once in _prepare, activations in the producing kernel (norm / GELU / SiLU-mul epilogue).  Not a drop-in for
        # the reference's bf16 numbers: held to oracle/ref_fp8.py and reported with its own tolerance.
        self.ffn_format = "bf16"
        # which of the FFN Linears the "mxfp8" format covers (set_ffn_format(..., linears=...)): "fc1" / "fc2" of the vision
        # tower, "w13" (w1 | w3) / "w2" of the language tower.  All four = BASELINE configs[4]'s full fp8 weight path; a subset
        # is a PRESET (FP8_PRESETS) that trades speed for rank agreement with the reference's bf16 scores (DESIGN §7.4:
        # profiles/r06_a_fp8_ffn_subset_study.txt).  A Linear outside the set runs the bf16 kernels on bf16 operands; at the
        # seam the bf16 activations are block-quantised by the producing epilogue (or by mjv_quantize_mxfp8: bit-identical).
        self.ffn_fp8_linears = frozenset(FFN_LINEARS)
        # MEASUREMENT ONLY (tools/fp8_attn_side_study.py; never set by the product): with ffn_format "mxfp8", also run the four
        # attention-side Linears (qkv / proj, wqkv / wo) on MXFP8 operands through UNFUSED launches (standalone quantiser, standalone
        # RoPE) - the numerics an all-Linear fp8 path would have, to decide whether its fused kernels are worth writing
        # (True = all four; or a set of names out of {"qkv", "proj", "wqkv", "wo"}: the per-Linear sensitivity study of round 5)
        self._exp_fp8_attn_side = False
        # The reference ships two attention numerics.  "flash" (default since round 4): fp32 softmax on UNROUNDED scores - its
        # flash-attention path (modeling_intern_vit.py:229-244, modeling_internlm2.py:437-561), the one it runs on a GPU;
        # score_round_mode 2 of the C ABI.  "eager": the scores rounded to bf16 before the softmax exactly where its eager path
        # rounds them (modeling_intern_vit.py:210-227: bf16(q scale) k^T; modeling_internlm2.py:383-411: bf16(bf16(q k^T) / sqrt(d))) -
        # the path its CPU run (and so the oracle) takes; modes 0 / 1.  The fixtures decided (DESIGN §4 "Attention, round 4",
        # profiles/r04_e_attention_numerics_gate.txt): against the reference's bf16 AND fp32 runs the two settings are equally
        # close on every gate (single layers at production shape, full_c1 / full_c2, the engineered rank sets), and the
        # unrounded form needs 6 instead of 8 (ViT) / 12 (LLM) vector instructions per score pair.
        self.attention_scores = "flash"
        # north_star "RMSNorm / LayerNorm -> GEMM fusion": the four norms whose only consumer is one Linear (ViT norm1 -> qkv,
        # norm2 -> fc1; LLM attention_norm -> wqkv, ffn_norm -> w1 | w3) folded into that Linear - gain rounded into the weight
        # once (W' = bf16(W * g)), a statistics kernel that writes 8 bytes per row instead of the normalised row, rstd (and the
        # mean term) applied in the GEMM's epilogue (include/mjv.h "row_scale").  Algebraically the same function, with
        # DIFFERENT rounding points than the reference's bf16(norm(x)) -> Linear (one rounding fewer: the normalised rows are
        # never rounded).  Judged by the fixtures (DESIGN "Norm fusion, round 4", profiles/r04_f_norm_fusion_gate.txt): single
        # layers at production shape 0.27 % / 0.48 % from the reference's bf16 run (unfused 0.18 % / 0.30 %; bound 0.60 % / 0.88 %)
        # and CLOSER to its fp32 run (0.299 % / 0.407 % against 0.301 % / 0.436 %), engineered rank set 0 flips, -0.94 ms per step:
        # every gate the round-3 review named holds.  It stays OFF by default all the same: the folded form is an independent
        # sample of the rounding noise, not the reference's own rounding points, so it sits 1.3 - 1.5 x further from the
        # reference's bf16 numbers (random-head set @448^2: |hip - ref| rms 0.103 -> 0.133, one of 512 scores past the 8-sigma
        # bound of tests/test_e2e_gpu.py::test_rank_agreement_c2) for 1 % of the step - parity with the reference comes first.
        # Both settings are tested (single layers, tiny cases, last-layer trimming).
        self.norm_fusion = False
        # Two pieces of work a SCORER can leave out of the language tower (include/mjv.h ABI 6), both
        # output-preserving up to the re-association of fp32 sums, both stated in bench.py's line:
        # trim_last_layer: the heads read two late rows per sample (moe_reward.py:226-243), so the last decoder layer needs
        #   keys / values for every row but QUERIES only from the first selected row on: its wqkv runs as a k | v projection on all
        #   rows plus the full projection on that tail, its attention on the tail's queries.  (wo / FFN of that layer were
        #   already evaluated on the two selected rows only.)
        self.trim_last_layer = True
        # prefix_cache: every prompt starts with the same tokens (system prompt + "Frame1: <img>": conversation.py:354-365,
        #   eval_genai_mjvideo.py:132-137); under the causal mask their hidden states - and so their keys / values in all 24
        #   layers - depend on the weights and those ids alone.  The first forward that meets a prefix runs the tower over its
        #   first 64 * k tokens ALONE (_build_prefix) and keeps every layer's K / V rows; every forward - that first one included:
        #   a cold and a warm cache run the same computation - leaves those rows out of every GEMM / norm of the tower
        #   and attends to the cached keys (mjv_attn_desc.prefix_k / prefix_v).  Cached vs uncached evaluation: equal up to fp32
        #   re-association (bit-identical when no GEMM slices K); which of the two a forward gets depends on the batch's
        #   common prefix and on the thrash guard's history (_forward_group) - prefix_cache = False is the history-free mode.  Invalidated by any parameter change
        #   (load_state_dict, .to()), another prefix, another rotary base (dynamic NTK), another numerics setting.  False =
        #   recompute them every forward, as the reference does.
        self.prefix_cache = True
        self._prefix = None          # dict(key, P, k [layers][P, kv*128], v [layers][P, ld of the layer's V rows], v_last)
        self._prefix_misses = 0      # consecutive forwards that found no cached prefix (see _forward_group: thrash guard)
        self._prefix_last_miss = None
        self.prefix_cache_hits = 0   # forwards served from the cache (tests / bench report)

    # -- construction helpers -------------------------------------------------------------------
    @classmethod
    def from_config(cls, config, dtype=torch.float32, device=None) -> "InternVLChatRewardModeling":
        """Uninitialised parameters of the right shapes in ``dtype`` on ``device`` (fill with ``load_state_dict``)."""
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with torch.device(device if device is not None else "cpu"):
                return cls("<config>", config, base_model=InternVLChatModel(config))
        finally:
            torch.set_default_dtype(prev)

    # -- derived (pre-arranged) weights ----------------------------------------------------------
    def set_ffn_format(self, fmt: str, linears=None) -> "InternVLChatRewardModeling":
        """"bf16", "mxfp8" or a preset name of ``FP8_PRESETS`` (see ``ffn_format`` / ``ffn_fp8_linears``); ``linears``: an explicit
        subset of ``FFN_LINEARS`` for "mxfp8" (default: all four).  Takes effect at the next forward (weights are re-prepared)."""
        if fmt in FP8_PRESETS:
            if linears is not None:
                raise ValueError("a preset names its own Linears")
            fmt, linears = "mxfp8", FP8_PRESETS[fmt]
        elif isinstance(fmt, str) and fmt.startswith("mxfp8:"):     # "mxfp8:fc1+w13" = linears ("fc1", "w13")
            if linears is not None:
                raise ValueError("give the Linears either in the format string or as ``linears``")
            fmt, linears = "mxfp8", tuple(fmt[len("mxfp8:"):].split("+"))
        if fmt not in ("bf16", "mxfp8"):
            raise ValueError(f"ffn_format {fmt!r} not in ('bf16', 'mxfp8') or a preset {sorted(FP8_PRESETS)}")
        linears = frozenset(FFN_LINEARS if linears is None else linears)
        if not linears or linears - frozenset(FFN_LINEARS):
            raise ValueError(f"linears {sorted(linears)} must be a non-empty subset of {FFN_LINEARS}")
        self.ffn_fp8_linears = linears
        if fmt == "mxfp8":
            vc, lc = self.config.vision_config, self.config.llm_config
            for nm, k in (("vision hidden_size", vc.hidden_size), ("vision intermediate_size", vc.intermediate_size),
                          ("llm hidden_size", lc.hidden_size), ("llm intermediate_size", lc.intermediate_size)):
                if k % 128:
                    raise ValueError(f"mxfp8 FFN path: {nm} = {k} must be a multiple of 128 (the MFMA's K)")
        self.ffn_format = fmt
        return self

    def _exp8_set(self) -> frozenset:
        """the attention-side Linears the measurement flag ``_exp_fp8_attn_side`` puts on MXFP8 operands (empty in the product)"""
        f = self._exp_fp8_attn_side
        if not f or self.ffn_format != "mxfp8" or self.norm_fusion:
            return frozenset()
        return frozenset(("qkv", "proj", "wqkv", "wo")) if f is True else frozenset(f)

    def _signature(self):
        ps = list(self.parameters())
        return tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in ps) + (
            self.ffn_format, self.ffn_fp8_linears, bool(self.norm_fusion), self._exp8_set())

    def _prepare(self, device):
        """One-time weight layout conversion (redone if any parameter storage/version changed):
        K-padded patch-embedding matrix, 16-row interleaved w1|w3, head index tables."""
        sig = self._signature()
        if self._derived_sig == sig:
            return self._derived
        for p in self.parameters():
            if p.dtype != BF16:
                raise TypeError("the HIP path computes in bf16: call model.to(torch.bfloat16) first "
                                f"(found a {p.dtype} parameter)")
            if p.device.type != "cuda":
                raise RuntimeError("the reward model runs on the MI355X only: call model.cuda() first "
                                   "(there is no CPU path)")
        d: Dict[str, object] = {}
        vc, lc = self.config.vision_config, self.config.llm_config
        emb = self.model.vision_model.embeddings
        P = vc.patch_size
        kreal = 3 * P * P
        kpad = (kreal + 63) // 64 * 64
        wp = torch.zeros(vc.hidden_size, kpad, dtype=BF16, device=device)
        wp[:, :kreal] = emb.patch_embedding.weight.reshape(vc.hidden_size, kreal)
        d["patch_w"], d["patch_k"] = wp, kpad
        w13 = []
        for layer in self.model.language_model.model.layers:
            if self._phi3:   # gate, up = gate_up_proj(h).chunk(2): gate plays w1's role, up w3's (modeling_phi3.py:Phi3MLP.forward)
                gu = layer.mlp.gate_up_proj.weight
                w1, w3 = gu[: gu.shape[0] // 2], gu[gu.shape[0] // 2:]
            else:
                w1, w3 = layer.feed_forward.w1.weight, layer.feed_forward.w3.weight
            ff, h = w1.shape
            if ff % 16:
                raise NotImplementedError("intermediate_size must be a multiple of 16")
            w13.append(torch.stack([w1.reshape(ff // 16, 16, h), w3.reshape(ff // 16, 16, h)], dim=1).reshape(2 * ff, h).contiguous())
        d["w13"] = w13
        # k | v rows of the LAST decoder layer's wqkv (trim_last_layer): rows are (kv head, [q_0 .. q_{G-1}, k, v], 128)
        lc0 = self.config.llm_config
        KV0, G0 = lc0.num_key_value_heads, lc0.num_attention_heads // lc0.num_key_value_heads
        wl = self.model.language_model.model.layers[-1].attention.wqkv.weight
        if self._phi3:   # qkv_proj rows are [q heads | k heads | v heads]: the two projections of the trimmed last layer are row slices
            hd0 = lc0.hidden_size // lc0.num_attention_heads
            d["wq_last"], d["wkv_last"] = wl[:lc0.num_attention_heads * hd0], wl[lc0.num_attention_heads * hd0:]
        else:
            d["wkv_last"] = wl.view(KV0, G0 + 2, 128, wl.shape[1])[:, G0:].reshape(KV0 * 2 * 128, wl.shape[1]).contiguous()
        if self.norm_fusion and self._phi3:
            raise NotImplementedError("norm_fusion is built for the InternLM2 tower's GEMM epilogues only")
        if self.norm_fusion:
            def pad(v: torch.Tensor) -> torch.Tensor:      # column vectors are fetched in whole 256-tiles
                out = torch.zeros(ops.padded_rows(v.numel()), dtype=torch.float32, device=device)
                out[:v.numel()] = v
                return out

            def fold_ln(lin, norm):
                """Linear(LayerNorm(x)) = rstd (x W'^T - mean colsum(W')) + (W beta + b):  W' = bf16(W * gamma)"""
                wf = (lin.weight.float() * norm.weight.float()[None, :]).to(BF16)
                bias = lin.weight.float() @ norm.bias.float() + (lin.bias.float() if lin.bias is not None else 0.0)
                return wf.contiguous(), pad(wf.float().sum(dim=1)), pad(bias)

            d["vit_fold"] = [dict(qkv=fold_ln(l.attn.qkv, l.norm1), fc1=fold_ln(l.mlp.fc1, l.norm2))
                             for l in self.model.vision_model.encoder.layers]
            d["llm_fold"] = [dict(wqkv=(l.attention.wqkv.weight.float() * l.attention_norm.weight.float()[None, :]).to(BF16).contiguous(),
                                  w13=(w13[i].float() * l.ffn_norm.weight.float()[None, :]).to(BF16).contiguous())
                             for i, l in enumerate(self.model.language_model.model.layers)]
        if self.ffn_format == "mxfp8":   # the FFN weights as MXFP8 (elements + block scales), quantised once
            S8 = self.ffn_fp8_linears
            if "fc1" in S8:
                d["fc1_8"] = [ops.quantize_mxfp8(l.mlp.fc1.weight) for l in self.model.vision_model.encoder.layers]
            if "fc2" in S8:
                d["fc2_8"] = [ops.quantize_mxfp8(l.mlp.fc2.weight) for l in self.model.vision_model.encoder.layers]
            if "w13" in S8:
                d["w13_8"] = [ops.quantize_mxfp8(w) for w in w13]
            if "w2" in S8:
                d["w2_8"] = [ops.quantize_mxfp8(l.feed_forward.w2.weight) for l in self.model.language_model.model.layers]
            if self._exp8_set():
                d["qkv_8"] = [ops.quantize_mxfp8(l.attn.qkv.weight) for l in self.model.vision_model.encoder.layers]
                d["proj_8"] = [ops.quantize_mxfp8(l.attn.proj.weight) for l in self.model.vision_model.encoder.layers]
                d["wqkv_8"] = [ops.quantize_mxfp8(l.attention.wqkv.weight) for l in self.model.language_model.model.layers]
                d["wo_8"] = [ops.quantize_mxfp8(l.attention.wo.weight) for l in self.model.language_model.model.layers]
        offs, idx = [0], []
        for _, crit in self.aspect2criteria.items():
            idx += list(crit)
            offs.append(len(idx))
        d["group_offsets"] = torch.tensor(offs, dtype=torch.int32, device=device)
        d["group_index"] = torch.tensor(idx, dtype=torch.int32, device=device)
        d["pos"] = {}
        self._derived, self._derived_sig = d, sig
        return d

    def _pos_table(self, d, grid: int, device) -> torch.Tensor:
        """[1 + grid*grid, dim] position table; the patch part goes through the fp32 bicubic resample of
        modeling_intern_vit.py:154-160 when the tile grid differs from the checkpoint's (one-time weight prep;
        an exact identity at the native grid)."""
        if grid in d["pos"]:
            return d["pos"][grid]
        vc = self.config.vision_config
        pos = self.model.vision_model.embeddings.position_embedding
        g0 = vc.image_size // vc.patch_size
        patch = pos[:, 1:, :]
        if grid != g0:
            p = patch.float().reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
            p = torch.nn.functional.interpolate(p.cpu(), size=(grid, grid), mode="bicubic", align_corners=False)
            patch = p.reshape(1, -1, grid * grid).permute(0, 2, 1).to(BF16).to(device)
        table = torch.cat([pos[:, :1, :], patch], dim=1)[0].contiguous()
        d["pos"][grid] = table
        return table

    def _rope_tables(self, seq_len: int, device):
        """bf16 cos/sin tables of modeling_internlm2.py:147-229 (fp32 math, cast once), INCLUDING the reference's state:
        its rotary modules cache ``max_seq_len_cached`` positions (``max_position_embeddings`` at construction) and regrow
        the cache whenever a forward brings a longer (padded) sequence - and the dynamic-NTK variant then also replaces its
        ``inv_freq`` for good (:204-229), so every LATER forward, short ones included, rotates with the rescaled base.
        ``seq_len`` is the padded width of ``input_ids`` (``kv_seq_len`` of :367-372), as the reference passes it."""
        lc = self.config.llm_config
        dim = lc.hidden_size // lc.num_attention_heads
        rs = lc.rope_scaling
        st = self._rope_state = self._rope_next_state(seq_len)
        n = max(st["cached"], 1)
        key = (n, st["base"], str(device))
        if key in self._rope:
            return self._rope[key]
        inv_freq = 1.0 / (st["base"] ** (torch.arange(0, dim, 2).float() / dim))
        t = torch.arange(n, dtype=inv_freq.dtype)
        if rs is not None and rs["type"] == "linear":
            t = t / rs["factor"]
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        tabs = (emb.cos().to(BF16).to(device).contiguous(), emb.sin().to(BF16).to(device).contiguous())
        self._rope = {key: tabs}
        return tabs

    def _rope_next_state(self, seq_len: int) -> dict:
        """the rotary cache state (cached length, base) a forward of padded width ``seq_len`` leaves behind - computed WITHOUT
        touching ``self._rope_state``: ``_forward_group`` needs the base for its prefix-cache key before the batch is validated,
        and a forward that is then rejected (bad ids, <IMG_CONTEXT> count mismatch ...) must not advance the state - the
        reference's rotary modules only see sequences that reach the language tower."""
        lc = self.config.llm_config
        dim = lc.hidden_size // lc.num_attention_heads
        maxpos = lc.max_position_embeddings
        rs = lc.rope_scaling
        st = dict(self._rope_state) if self._rope_state is not None else dict(cached=maxpos, base=float(lc.rope_theta))
        if seq_len > st["cached"]:
            st["cached"] = seq_len
            if rs is not None and rs["type"] == "dynamic" and seq_len > maxpos:
                st["base"] = float(lc.rope_theta) * ((rs["factor"] * seq_len / maxpos) - (rs["factor"] - 1)) ** (dim / (dim - 2))
        return st

    def _rope_tables_phi3(self, seq_len: int, device):
        """bf16 cos / sin tables [seq_len rounded up to 1024, rotary_dim] of transformers/models/phi3/modeling_phi3.py:
        Phi3RotaryEmbedding.forward for position_ids = arange(seq_len) (the reference passes none, modeling_internvl_chat.py:190-199):
        fp32 pos x inv_freq, cat(freqs, freqs), cos / sin times the LongRoPE attention factor, cast.  inv_freq takes the LONG
        factor list when the padded width ``seq_len`` exceeds original_max_position_embeddings, the SHORT one otherwise
        (modeling_rope_utils.py: longrope_frequency_update - decided per forward, nothing kept between forwards).  The short and
        the unscaled frequencies are ROUNDED TO BF16 first: they live in module buffers that the reference driver's
        ``model.to(torch.bfloat16)`` casts (eval_genai_mjvideo.py:112-116; this model only runs in bf16); the long ones are
        recomputed in fp32 inside the forward and are not."""
        lc = self.config.llm_config
        dim = int((lc.hidden_size // lc.num_attention_heads) * lc.partial_rotary_factor)
        rs = lc.rope_scaling
        use_long = bool(rs) and seq_len > lc.original_max_position_embeddings
        n = (max(seq_len, 1) + 1023) // 1024 * 1024
        key = ("phi3", n, use_long, str(device))
        if key in self._rope:
            return self._rope[key]
        shape = torch.arange(0, dim, 2, dtype=torch.int64).float() / dim
        if rs:
            ext = torch.tensor(rs["long_factor"] if use_long else rs["short_factor"], dtype=torch.float32)
            inv_freq = 1.0 / (ext * float(lc.rope_theta) ** shape)
            if not use_long:
                inv_freq = inv_freq.to(BF16).float()
            factor = rs.get("factor")
            if factor is None:
                factor = lc.max_position_embeddings / lc.original_max_position_embeddings
            att = rs.get("attention_factor")
            if att is None:
                att = 1.0 if factor <= 1.0 else math.sqrt(1 + math.log(factor) / math.log(lc.original_max_position_embeddings))
        else:
            inv_freq, att = (1.0 / (float(lc.rope_theta) ** shape)).to(BF16).float(), 1.0
        pos = torch.arange(n, dtype=torch.float32)
        freqs = (inv_freq[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]
        emb = torch.cat((freqs, freqs), dim=-1)
        tabs = ((emb.cos() * att).to(BF16).to(device).contiguous(), (emb.sin() * att).to(BF16).to(device).contiguous())
        self._rope = {key: tabs}
        return tabs

    def _buf(self, name: str, rows: int, cols: int, device, dtype=BF16) -> torch.Tensor:
        name = f"{self._ws_tag}:{name}"
        t = self._ws.get(name)
        need = rows * cols
        if t is None or t.numel() < need or t.device != device or t.dtype != dtype:
            t = torch.empty(need, dtype=dtype, device=device)
            self._ws[name] = t
        return t[:need].view(rows, cols)

    def _buf8(self, name: str, rows: int, cols: int, device) -> "ops.MX8":
        """an MXFP8 scratch matrix of exactly ``rows`` rows (its scale records are laid out for that row count)"""
        data = self._buf(name, rows, cols, device, dtype=torch.uint8)
        scales = self._buf(name + ":s", 1, ops.mxfp8_scale_bytes(rows, cols), device, dtype=torch.uint8).view(-1)
        return ops.MX8(data, scales)

    # -- host-side analysis of the token ids (the reference does this with ids.tolist(), moe_reward.py:242)
    def _host_ids(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]):
        """Token ids / mask as host numpy arrays: ONE device->host copy per forward (none for CPU tensors, none when the
        same unmodified tensors are passed again).  The reference does the same round trip with ``ids.tolist()``
        (moe_reward.py:242); everything the kernels need from the ids (row maps, lengths) is derived on the host."""
        return self._host_ids_begin(input_ids, attention_mask)()

    def _host_ids_begin(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]):
        """Starts the device->host copy of the ids / mask and returns a function that waits for it and yields the numpy
        arrays.  The copy runs on a SIDE stream behind an event recorded on the current stream now: it waits for whatever
        the caller enqueued before the forward (the ids' producer among it) but not for what the forward enqueues next - the
        vision tower needs no ids, so its launches go out first and the host's wait for the ids (which in a stream of
        back-to-back batches is a wait for the previous batch's GPU work), its analysis of them and the uploads all hide
        under vision-tower kernels (tools/step_bubble.py: a blocking ``ids.cpu()`` at the top of the forward left the GPU idle
        for 0.7-0.8 ms of a 93 ms step)."""
        c = self._host_cache
        if (c is not None and c[0] is input_ids and c[1] == input_ids._version and c[2] is attention_mask
                and (attention_mask is None or c[3] == attention_mask._version)):
            return lambda: (c[4], c[5])

        def remember(ids, am):
            # the cache holds the tensors themselves: an address-based key could match a NEW tensor that the allocator
            # placed where a freed one used to live
            self._host_cache = (input_ids, input_ids._version, attention_mask,
                                None if attention_mask is None else attention_mask._version, ids, am)
            return ids, am

        if not input_ids.is_cuda or (attention_mask is not None and not attention_mask.is_cuda):
            return lambda: remember(input_ids.detach().to("cpu").numpy(),
                                    None if attention_mask is None else attention_mask.detach().to("cpu").numpy().astype(bool))
        dev = input_ids.device
        with torch.cuda.device(dev):
            side = self._side_streams.get(dev)
            if side is None:
                side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            pin_ids = torch.empty(input_ids.shape, dtype=input_ids.dtype, pin_memory=True)
            pin_am = None if attention_mask is None else torch.empty(attention_mask.shape, dtype=attention_mask.dtype, pin_memory=True)
            done = torch.cuda.Event()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                pin_ids.copy_(input_ids.detach(), non_blocking=True)
                if pin_am is not None:
                    pin_am.copy_(attention_mask.detach(), non_blocking=True)
                done.record(side)

        def wait():
            done.synchronize()
            return remember(pin_ids.numpy(), None if pin_am is None else pin_am.numpy().astype(bool))

        return wait

    def _analyse_ids(self, ids: np.ndarray, am: Optional[np.ndarray], n_tiles: int, prefix_lookup=None):
        """Validates the batch and derives everything the kernels need from the token ids (packed ids, positions, row maps).
        ``prefix_lookup`` (``forward`` with ``prefix_cache`` on): called with the ids of the batch's common prompt prefix - the
        longest run of leading tokens shared by every sequence that ends before any <IMG_CONTEXT> / selected row, cut to a
        multiple of 64 - and returns True when the language tower's keys / values of exactly those tokens are cached: the
        packed rows then leave the first ``skip`` tokens of every sequence out (positions keep counting from 0)."""
        if ids.ndim != 2:
            raise ValueError(f"input_ids must be [batch, seq], got {ids.shape}")
        B, N = ids.shape
        pad_id = self.config.pad_token_id
        if pad_id is None and B != 1:
            raise ValueError("Cannot handle batch sizes > 1 if no padding token is defined.")
        ctx_id = self.model.img_context_token_id
        if ctx_id is None:
            raise ValueError("model.model.img_context_token_id is not set (eval_genai_mjvideo.py:115)")
        if am is not None:
            if am.shape != ids.shape:
                raise ValueError(f"attention_mask shape {am.shape} != input_ids shape {ids.shape}")
            am = am.astype(bool)
        else:
            am = np.ones((B, N), dtype=bool)
        # ANY mask (round 6; rounds 1-5 took right-padded ones only).  The reference's eager path adds a [B, 1, N, N] mask of
        # {causal, key is valid} (modeling_internlm2.py:96-125,907-913) and numbers positions 0 .. N - 1 WITHOUT looking at the mask
        # (:893-898): a valid query sees exactly the valid keys at or before its column, rotated by their COLUMN indices.  Packing the
        # valid tokens of a row in column order, with those columns as positions, is the same computation - right padding, left
        # padding or holes alike; rows at masked columns are garbage in the reference and are never read (the selected rows must be
        # valid: checked below).
        lens = am.sum(axis=1)
        if (lens <= 0).any():
            raise ValueError("empty sequence in the batch")
        vocab = self.model.language_model.model.tok_embeddings.weight.shape[0]
        valid = ids[am]
        if valid.size and (int(valid.min()) < 0 or int(valid.max()) >= vocab):   # nn.Embedding raises here too
            raise IndexError(f"token id out of range [0, {vocab}): min {int(valid.min())}, max {int(valid.max())} "
                             "(tokenizer / checkpoint mismatch?)")
        rs, gs, ctx, cols, loc = [], [], [], [], []
        for b in range(B):
            row = ids[b]
            if pad_id is None:
                r = N - 1
            else:
                r = (int(np.argmax(row == pad_id)) - 1) % N
            g = find_token_for_gating(row, getattr(self.config, "gating_token_pattern", None))
            if not am[b, r] or not am[b, g]:
                raise ValueError(f"sample {b}: reward row {r} / gating row {g} lies in the masked part of the sequence "
                                 f"({int(lens[b])} valid tokens)")
            c = np.flatnonzero(row == ctx_id)
            if not am[b, c].all():
                raise ValueError(f"sample {b}: <IMG_CONTEXT> tokens in the masked part of the sequence")
            rs.append(r)
            gs.append(g)
            ctx.append(c)
            cols.append(np.flatnonzero(am[b]))                   # valid columns, in order = the packed rows of this sample
            loc.append(np.cumsum(am[b]) - 1)                     # column -> index among the valid ones
        n_ctx = sum(c.size for c in ctx)
        if n_ctx != n_tiles * self.model.num_image_token:
            raise ValueError(f"{n_ctx} <IMG_CONTEXT> tokens in input_ids but pixel_values holds {n_tiles} tiles x "
                             f"{self.model.num_image_token} image tokens (the reference would silently truncate, "
                             f"modeling_internvl_chat.py:178-186)")
        # the batch's common prompt prefix (system prompt + "Frame1: <img>": conversation.py:354-365, eval_genai_mjvideo.py:132-137)
        skip, prefix_ids = 0, None
        if prefix_lookup is not None:
            lim = min(min(int(c[0]) if c.size else N for c in ctx), min(rs), min(gs))
            masked = np.flatnonzero(~am[:, :lim].all(axis=0))     # (a prefix column must be a valid token of every sample)
            if masked.size:
                lim = int(masked[0])
            if B > 1 and lim > 0:
                differ = np.flatnonzero((ids[1:, :lim] != ids[0, :lim]).any(axis=0))
                if differ.size:
                    lim = int(differ[0])
            cand = (lim // 64) * 64
            if cand > 0:
                prefix_ids = np.ascontiguousarray(ids[0, :cand])
                if prefix_lookup(prefix_ids):
                    skip = cand
        reward_rows, gating_rows, packed, positions, img_rows = [], [], [], [], []
        tail_rows, tail_pos, sel_in_tail_r, sel_in_tail_g = [], [], [], []
        cu, cu_tail = [0], [0]
        for b in range(B):
            row, L, r, g = ids[b], int(lens[b]), rs[b], gs[b]
            base = cu[-1] - skip           # packed row of the token at column j of this sample = base + loc[j]  (j >= skip)
            lb = loc[b]
            reward_rows.append(base + lb[r])
            gating_rows.append(base + lb[g])
            packed.append(row[cols[b][skip:]])
            positions.append(cols[b][skip:])
            img_rows.append(base + lb[ctx[b]])
            cu.append(base + L)
            # the rows from the first selected one on: all the last decoder layer needs QUERIES for (moe_reward.py:226-243)
            t0 = int(lb[min(r, g)])          # index among the valid tokens
            tb = cu_tail[-1] - t0
            tail_rows.append(base + np.arange(t0, L))
            tail_pos.append(cols[b][t0:])
            sel_in_tail_r.append(tb + lb[r])
            sel_in_tail_g.append(tb + lb[g])
            cu_tail.append(tb + L)
        return dict(B=B, N=N, total=cu[-1], max_len=int(lens.max()) - skip, skip=skip, prefix_ids=prefix_ids,
                    ids=np.concatenate(packed).astype(np.int32), positions=np.concatenate(positions).astype(np.int32),
                    cu=np.asarray(cu, dtype=np.int32), img_rows=np.concatenate(img_rows).astype(np.int32),
                    sel_rows=np.asarray(reward_rows + gating_rows, dtype=np.int32),
                    tail_rows=np.concatenate(tail_rows).astype(np.int32), tail_pos=np.concatenate(tail_pos).astype(np.int32),
                    cu_tail=np.asarray(cu_tail, dtype=np.int32), max_tail=int(np.diff(cu_tail).max()),
                    sel_in_tail=np.asarray(sel_in_tail_r + sel_in_tail_g, dtype=np.int32))

    # -- towers ----------------------------------------------------------------------------------
    def _vision_tower(self, d, pixel_values: torch.Tensor, hidden: torch.Tensor, img_rows: torch.Tensor):
        """patchify -> 24 x ViT layer -> pixel-shuffle + mlp1, scattered into the <IMG_CONTEXT> rows of ``hidden``."""
        self._vision_tower_splice(self._vision_tower_launch(d, pixel_values), hidden, img_rows)

    def _vision_tower_launch(self, d, pixel_values: torch.Tensor):
        """Everything of the vision tower that needs no token ids: patchify -> 24 x ViT layer -> pixel-shuffle + LayerNorm +
        first projector layer.  Returns what ``_vision_tower_splice`` needs.  ``vit_chunk_tiles`` (measurement switch, default
        None = all tiles in one pass): the tower over that many tiles at a time, every chunk through the SAME scratch buffers -
        tiles are independent of one another, so only the launch sizes change."""
        dev = pixel_values.device
        vc = self.config.vision_config
        tiles, _, S, _ = pixel_values.shape
        P = vc.patch_size
        if S % P or (S // P) % 2:
            raise ValueError(f"image size {S} must be an even multiple of the patch size {P}")
        G = S // P
        per = (G // 2) ** 2
        mlp1 = self.model.mlp1
        ph = self._buf("proj_h", tiles * per, mlp1[1].out_features, dev)
        chunk = getattr(self, "vit_chunk_tiles", None) or tiles
        if self.debug_probes is not None:
            chunk = tiles            # (the per-layer probes hold every tile)
        for c0 in range(0, tiles, chunk):
            c1 = min(tiles, c0 + chunk)
            self._vision_tower_chunk(d, pixel_values[c0:c1], ph[c0 * per:c1 * per])
        return ph, tiles

    def _vision_tower_chunk(self, d, pixel_values: torch.Tensor, ph: torch.Tensor):
        dev = pixel_values.device
        vc = self.config.vision_config
        vm = self.model.vision_model
        tiles, _, S, _ = pixel_values.shape
        P, dim, ff, H = vc.patch_size, vc.hidden_size, vc.intermediate_size, vc.num_attention_heads
        G = S // P
        npatch, T = G * G, G * G + 1
        rows = tiles * T
        pos = self._pos_table(d, G, dev)
        patches = self._buf("patches", tiles * npatch, d["patch_k"], dev)
        ops.patchify(pixel_values, patches, P)
        x = self._buf("vit_x", rows, dim, dev)
        ops.cls_rows(x, vm.embeddings.class_embedding.view(-1), pos[0], tiles, T)
        ops.gemm(patches, d["patch_w"], x, EPI_SCALE_RES, bias=vm.embeddings.patch_embedding.bias, res=pos, res_mod=npatch,
                 res_off=1, out_group=npatch, out_pad=1)
        probes = self.debug_probes
        if probes is not None:
            probes["vit_embed"] = x.clone().view(tiles, T, dim)
        h = self._buf("vit_h", rows, dim, dev)
        qkv = self._buf("vit_qkv", rows, 3 * dim, dev)
        f = self._buf("vit_ff", rows, ff, dev)
        key = (self._ws_tag, "vit_cu", tiles, T)
        cu = self._ws.get(key)
        if cu is None or cu.device != dev:
            cu = torch.arange(0, (tiles + 1) * T, T, dtype=torch.int32, device=dev)
            self._ws[key] = cu
        for li, layer in enumerate(vm.encoder.layers):
            self._vit_layer(layer, x, h, qkv, f, cu, T, li)
            if probes is not None:
                probes[f"vit_layer{li}"] = x.clone().view(tiles, T, dim)
        mlp1 = self.model.mlp1
        ntok = tiles * (G // 2) ** 2
        pl = self._buf("proj_ln", ntok, 4 * dim, dev)
        ops.layernorm(x, mlp1[0].weight, mlp1[0].bias, pl, mlp1[0].eps, rows=ntok, gather_grid=G)
        ops.gemm(pl, mlp1[1].weight, ph, EPI_BIAS_GELU, bias=mlp1[1].bias)

    def _vision_tower_splice(self, vit, hidden: torch.Tensor, img_rows: torch.Tensor):
        """second projector layer, its rows written straight into the <IMG_CONTEXT> rows of ``hidden``
        (modeling_internvl_chat.py:176-179)"""
        ph, tiles = vit
        mlp1 = self.model.mlp1
        ops.gemm(ph, mlp1[3].weight, hidden, EPI_BIAS, bias=mlp1[3].bias, out_rows=img_rows)
        if self.debug_probes is not None:
            self.debug_probes["vit_embeds"] = hidden[img_rows.long()].clone().view(tiles, -1, hidden.shape[1])

    def _vit_layer(self, layer, x, h, qkv, f, cu, T, li: int):
        """One InternVisionEncoderLayer (modeling_intern_vit.py:283-295) in place on the packed rows ``x`` [tiles * T, dim];
        ``h`` / ``qkv`` / ``f`` are scratch buffers of [rows, dim] / [rows, 3 dim] / [rows, intermediate].  ``li``: the layer's
        index (the mxfp8 FFN path looks its quantised weights up by it)."""
        vc = self.config.vision_config
        dim, H = vc.hidden_size, vc.num_attention_heads
        scale = (dim // H) ** -0.5
        fold = self._derived["vit_fold"][li] if self.norm_fusion else None
        e8 = self._exp8_set()
        if fold is not None:
            rows, dev = x.shape[0], x.device
            rstd = self._buf("vit_rstd", 1, ops.padded_rows(rows), dev, dtype=torch.float32).view(-1)
            mrs = self._buf("vit_mrs", 1, ops.padded_rows(rows), dev, dtype=torch.float32).view(-1)
            ops.row_stats(x, rstd, mrs, vc.layer_norm_eps)
            wq, cq, bq = fold["qkv"]
            ops.gemm(x, wq, qkv, EPI_BIAS, folded_norm=(rstd, mrs, cq, bq))
        elif "qkv" in e8:
            h8a = self._buf8("vit_h8a", x.shape[0], dim, x.device)
            ops.layernorm_mxfp8(x, layer.norm1.weight, layer.norm1.bias, h8a, vc.layer_norm_eps)
            ops.gemm(h8a, self._derived["qkv_8"][li], qkv, EPI_BIAS, bias=layer.attn.qkv.bias)
        else:
            ops.layernorm(x, layer.norm1.weight, layer.norm1.bias, h, vc.layer_norm_eps)
            ops.gemm(h, layer.attn.qkv.weight, qkv, EPI_BIAS, bias=layer.attn.qkv.bias)
        ops.attention(qkv[:, :dim], qkv[:, dim:2 * dim], qkv[:, 2 * dim:], h, cu, T, H, 1, 64, False, scale,
                      2 if self.attention_scores == "flash" else 0)
        if "proj" in e8:
            a8 = ops.quantize_mxfp8(h, out=self._buf8("vit_a8", x.shape[0], dim, x.device))
            ops.gemm(a8, self._derived["proj_8"][li], x, EPI_SCALE_RES, bias=layer.attn.proj.bias, scale=layer.ls1, res=x)
        else:
            ops.gemm(h, layer.attn.proj.weight, x, EPI_SCALE_RES, bias=layer.attn.proj.bias, scale=layer.ls1, res=x)
        S8 = self.ffn_fp8_linears if self.ffn_format == "mxfp8" else frozenset()
        if "fc1" in S8 or "fc2" in S8:
            # norm2 -> fc1 (+GELU) -> fc2 on MXFP8 operands: the norm and the GELU epilogue write e4m3 + block scales.  A preset
            # that keeps one of the two Linears on bf16 operands switches formats at the seam between them.
            d = self._derived
            rows, dev = x.shape[0], x.device
            if "fc1" in S8:
                h8 = self._buf8("vit_h8", rows, dim, dev)
                ops.layernorm_mxfp8(x, layer.norm2.weight, layer.norm2.bias, h8, vc.layer_norm_eps)
                mid = self._buf8("vit_f8", rows, vc.intermediate_size, dev) if "fc2" in S8 else f
                ops.gemm(h8, d["fc1_8"][li], mid, EPI_BIAS_GELU, bias=layer.mlp.fc1.bias)
            else:
                ops.layernorm(x, layer.norm2.weight, layer.norm2.bias, h, vc.layer_norm_eps)
                ops.gemm(h, layer.mlp.fc1.weight, f, EPI_BIAS_GELU, bias=layer.mlp.fc1.bias)
                mid = ops.quantize_mxfp8(f, out=self._buf8("vit_f8", rows, vc.intermediate_size, dev))
            if "fc2" in S8:
                ops.gemm(mid, d["fc2_8"][li], x, EPI_SCALE_RES, bias=layer.mlp.fc2.bias, scale=layer.ls2, res=x)
            else:
                ops.gemm(mid, layer.mlp.fc2.weight, x, EPI_SCALE_RES, bias=layer.mlp.fc2.bias, scale=layer.ls2, res=x)
            return
        if fold is not None:
            ops.row_stats(x, rstd, mrs, vc.layer_norm_eps)
            w1f, c1f, b1f = fold["fc1"]
            ops.gemm(x, w1f, f, EPI_BIAS_GELU, folded_norm=(rstd, mrs, c1f, b1f))
        else:
            ops.layernorm(x, layer.norm2.weight, layer.norm2.bias, h, vc.layer_norm_eps)
            ops.gemm(h, layer.mlp.fc1.weight, f, EPI_BIAS_GELU, bias=layer.mlp.fc1.bias)
        ops.gemm(f, layer.mlp.fc2.weight, x, EPI_SCALE_RES, bias=layer.mlp.fc2.bias, scale=layer.ls2, res=x)

    @torch.no_grad()
    def run_vit_layer(self, li: int, x: torch.Tensor) -> torch.Tensor:
        """Test entry (not part of the reference's API): encoder layer ``li`` of the vision tower on given hidden states
        ``x`` [tiles, tokens, dim] - the same call sequence ``forward`` runs per layer, nothing before or after it."""
        dev = self.model.device
        vc = self.config.vision_config
        tiles, T, dim = x.shape
        self._prepare(dev)
        with torch.cuda.device(dev):
            ops.set_gemm_workspace(self._buf("gemm_ws", 1, ops.gemm_workspace_bytes(), dev, dtype=torch.uint8)
                                   if self.use_gemm_workspace else None)
            try:
                xs = x.to(dev, BF16).reshape(tiles * T, dim).clone()
                h = torch.empty_like(xs)
                qkv = torch.empty(tiles * T, 3 * dim, dtype=BF16, device=dev)
                f = torch.empty(tiles * T, vc.intermediate_size, dtype=BF16, device=dev)
                cu = torch.arange(0, (tiles + 1) * T, T, dtype=torch.int32, device=dev)
                self._vit_layer(self.model.vision_model.encoder.layers[li], xs, h, qkv, f, cu, T, li)
            finally:
                ops.set_gemm_workspace(None)
        return xs.view(tiles, T, dim)

    @torch.no_grad()
    def run_llm_layer(self, li: int, x: torch.Tensor) -> torch.Tensor:
        """Test entry: decoder layer ``li`` of the language tower on given hidden states ``x`` [batch, seq, hidden] (every
        sequence full length, positions 0 .. seq - 1): the call sequence ``forward`` runs for a layer that is not the
        trimmed last one."""
        dev = self.model.device
        d = self._prepare(dev)
        B, N, hd = x.shape
        with torch.cuda.device(dev):
            ops.set_gemm_workspace(self._buf("gemm_ws", 1, ops.gemm_workspace_bytes(), dev, dtype=torch.uint8)
                                   if self.use_gemm_workspace else None)
            try:
                xs = x.to(dev, BF16).reshape(B * N, hd).clone()
                cu = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device=dev)
                pos = torch.arange(N, dtype=torch.int32, device=dev).repeat(B)
                self._language_tower(d, xs, cu, pos, N, None, padded_len=N, only_layer=li)
            finally:
                ops.set_gemm_workspace(None)
        return xs.view(B, N, hd)

    def _language_tower(self, d, x: torch.Tensor, cu: torch.Tensor, positions: torch.Tensor, max_len: int,
                        sel_rows: Optional[torch.Tensor] = None, padded_len: Optional[int] = None,
                        only_layer: Optional[int] = None, tail=None, prefix=None, snapshot=None):
        """24 decoder layers in place on the packed rows ``x``.  With ``sel_rows`` (the 2 rows per sample the heads
        read) the LAST layer runs wo / FFN only on those rows - every other row of its output is never used
        (moe_reward.py:211,229,243) - and the [len(sel_rows), hidden] result is returned instead of ``x``.
        ``tail`` (with ``sel_rows``; ``trim_last_layer``) = (tail_rows, tail_pos, cu_tail, max_tail, sel_in_tail): the last
        layer computes queries only for those rows.  ``prefix`` = the cached keys / values of the prompt prefix the packed rows
        leave out (``prefix_cache``); ``snapshot`` (the prefix-only pass of ``_build_prefix``) = (P, store): keep rows [0, P) of
        every layer's K / V in ``store``."""
        dev = x.device
        lc = self.config.llm_config
        lm = self.model.language_model.model
        H, KV = lc.num_attention_heads, lc.num_key_value_heads
        G = H // KV
        hd = lc.hidden_size // H
        n, hdim = x.shape
        ff = lc.intermediate_size
        phi3 = self._phi3
        if phi3:
            cos, sin = self._rope_tables_phi3(max_len if padded_len is None else padded_len, dev)
        else:
            cos, sin = self._rope_tables(max_len if padded_len is None else padded_len, dev)
        hn = self._buf("llm_hn", n, hdim, dev)
        qkv = self._buf("llm_qkv", n, (H + 2 * KV) * hd, dev)
        if not phi3:
            q = self._buf("llm_q", n, H * hd, dev)
            k = self._buf("llm_k", n, KV * hd, dev)
        act = self._buf("llm_act", n, ff, dev)
        # InternLM2: q k^T / sqrt(d) with d = 128: an exact power of two.  Phi-3: q k^T * d^-0.5 with the scaling rounded to fp32
        # as torch rounds a Python scalar for a bf16 tensor op (modeling_phi3.py: eager_attention_forward)
        scale = float(np.float32(hd ** -0.5)) if phi3 else 1.0 / math.sqrt(hd)
        v_view = qkv[:, (H + KV) * hd:] if phi3 else qkv[:, (G + 1) * hd:]
        last = len(lm.layers) - 1
        mode = 2 if self.attention_scores == "flash" else 1
        for li, layer in enumerate(lm.layers):
            if only_layer is not None and li != only_layer:   # (run_llm_layer: one layer on given rows)
                continue
            lfold = d["llm_fold"][li] if self.norm_fusion else None
            e8 = self._exp8_set()
            exp8 = bool(e8)
            trim_here = li == last and sel_rows is not None and self.debug_probes is None and only_layer is None
            pk = prefix["k"][li] if prefix is not None else None
            if trim_here and tail is not None and lfold is None and not exp8:
                # last layer, queries only where the heads will read (trim_last_layer): k | v projection (+ rotary on k) on every
                # row, the full wqkv on the tail rows, attention of the tail's queries over all keys
                tail_rows, tail_pos, cu_tail, max_tail, sel_in_tail = tail
                nt, ns = tail_rows.numel(), sel_rows.numel()
                kv = self._buf("llm_kv_last", n, 2 * KV * hd, dev)
                ops.rmsnorm(x, layer.attention_norm.weight, hn, lc.rms_norm_eps)
                if phi3:
                    # [k heads | v heads] of every row (the k heads rotated in place), q heads of the tail rows only
                    ops.gemm(hn, d["wkv_last"], kv, EPI_BIAS)
                    ops.rope_heads(kv, KV, hd, cos, sin, positions)
                    hn_t = self._buf("llm_hn_tail", nt, hdim, dev)
                    q_t = self._buf("llm_q_tail", nt, H * hd, dev)
                    att_t = self._buf("llm_att_tail", nt, H * hd, dev)
                    ops.embed_gather(tail_rows, hn, hn_t, -1)
                    ops.gemm(hn_t, d["wq_last"], q_t, EPI_BIAS)
                    ops.rope_heads(q_t, H, hd, cos, sin, tail_pos)
                    pl = prefix["kv_last"] if prefix is not None else None
                    ops.attention(q_t, kv[:, :KV * hd], kv[:, KV * hd:], att_t, cu, max_len, H, G, hd, True, scale, mode,
                                  cu_seqlens_q=cu_tail, max_seqlen_q=max_tail,
                                  prefix_k=(pl[:, :KV * hd] if pl is not None else None), prefix_v=(pl[:, KV * hd:] if pl is not None else None))
                    att_s = self._buf("llm_att_sel", ns, hdim, dev)
                    x_s = self._buf("llm_x_sel", ns, hdim, dev)
                    hn_s = self._buf("llm_hn_sel", ns, hdim, dev)
                    act_s = self._buf("llm_act_sel", ns, ff, dev)
                    ops.embed_gather(sel_in_tail, att_t, att_s, -1)
                    ops.embed_gather(sel_rows, x, x_s, -1)
                    ops.gemm(att_s, layer.attention.wo.weight, x_s, EPI_SCALE_RES, res=x_s)
                    self._llm_ffn(d, li, layer, x_s, hn_s, act_s, "sel")
                    return x_s
                ops.gemm(hn, d["wkv_last"], kv, EPI_ROPE_QKV, rope=(cos, sin, positions, None, k, 0))
                hn_t = self._buf("llm_hn_tail", nt, hdim, dev)
                qkv_t = self._buf("llm_qkv_tail", nt, (H + 2 * KV) * hd, dev)
                q_t = self._buf("llm_q_tail", nt, H * hd, dev)
                k_t = self._buf("llm_k_tail", nt, KV * hd, dev)
                att_t = self._buf("llm_att_tail", nt, H * hd, dev)
                ops.embed_gather(tail_rows, hn, hn_t, -1)
                ops.gemm(hn_t, layer.attention.wqkv.weight, qkv_t, EPI_ROPE_QKV, rope=(cos, sin, tail_pos, q_t, k_t, G))
                ops.attention(q_t, k, kv[:, hd:], att_t, cu, max_len, H, G, hd, True, scale, mode, v_head_stride=2 * hd,
                              cu_seqlens_q=cu_tail, max_seqlen_q=max_tail,
                              prefix_k=pk, prefix_v=(prefix["v_last"][:, hd:] if prefix is not None else None))
                att_s = self._buf("llm_att_sel", ns, hdim, dev)
                x_s = self._buf("llm_x_sel", ns, hdim, dev)
                hn_s = self._buf("llm_hn_sel", ns, hdim, dev)
                act_s = self._buf("llm_act_sel", ns, ff, dev)
                ops.embed_gather(sel_in_tail, att_t, att_s, -1)
                ops.embed_gather(sel_rows, x, x_s, -1)
                ops.gemm(att_s, layer.attention.wo.weight, x_s, EPI_SCALE_RES, res=x_s)
                self._llm_ffn(d, li, layer, x_s, hn_s, act_s, "sel")
                return x_s
            if phi3:
                # qkv_proj -> [q heads | k heads | v heads] (modeling_phi3.py:Phi3Attention.forward); the rotary embedding in place on
                # the q and k heads (one launch: they are consecutive); attention reads the three column ranges where they lie
                ops.rmsnorm(x, layer.attention_norm.weight, hn, lc.rms_norm_eps)
                ops.gemm(hn, layer.attention.wqkv.weight, qkv, EPI_BIAS)
                ops.rope_heads(qkv, H + KV, hd, cos, sin, positions)
                if snapshot is not None:   # the prompt prefix's projection rows of this layer (rotated k and v in their columns)
                    snapshot[1]["k"].append(qkv[:snapshot[0]].clone())
                ops.attention(qkv[:, :H * hd], qkv[:, H * hd:(H + KV) * hd], v_view, hn, cu, max_len, H, G, hd, True, scale, mode,
                              prefix_k=(pk[:, H * hd:(H + KV) * hd] if pk is not None else None),
                              prefix_v=(pk[:, (H + KV) * hd:] if pk is not None else None))
                if self.debug_probes is not None and li == 0:
                    self.debug_probes["llm_attn0"] = dict(q=qkv[:, :H * hd].clone(), k=qkv[:, H * hd:(H + KV) * hd].clone(),
                                                          v=v_view.clone(), out=hn.clone(), kv_heads=KV)
            # wqkv with the rotary embedding + GQA de-interleave in its epilogue: q / k go (rotated) to their own buffers,
            # v stays in its columns of qkv (modeling_internlm2.py:359-381)
            elif lfold is not None:
                rstd = self._buf("llm_rstd", 1, ops.padded_rows(n), dev, dtype=torch.float32).view(-1)
                ops.row_stats(x, rstd, None, lc.rms_norm_eps)
                ops.gemm(x, lfold["wqkv"], qkv, EPI_ROPE_QKV, rope=(cos, sin, positions, q, k, G), folded_norm=(rstd,))
            elif "wqkv" in e8:
                h8a = self._buf8("llm_h8a", n, hdim, dev)
                ops.rmsnorm_mxfp8(x, layer.attention_norm.weight, h8a, lc.rms_norm_eps)
                ops.gemm(h8a, d["wqkv_8"][li], qkv, EPI_BIAS)
                ops.rope_split(qkv, q, k, cos, sin, positions, KV, G)
            else:
                ops.rmsnorm(x, layer.attention_norm.weight, hn, lc.rms_norm_eps)
                ops.gemm(hn, layer.attention.wqkv.weight, qkv, EPI_ROPE_QKV, rope=(cos, sin, positions, q, k, G))
            if snapshot is not None and not phi3:   # the prompt prefix's keys (rotated) and values of this layer
                snapshot[1]["k"].append(k[:snapshot[0]].clone())
                snapshot[1]["v"].append(qkv[:snapshot[0]].clone())
            if not phi3:
                ops.attention(q, k, v_view, hn, cu, max_len, H, G, hd, True, scale, mode, v_head_stride=(G + 2) * hd,
                              prefix_k=pk, prefix_v=(prefix["v"][li][:, (G + 1) * hd:] if prefix is not None else None))
                if self.debug_probes is not None and li == 0:   # operands / result of the first causal attention (parity tests)
                    self.debug_probes["llm_attn0"] = dict(q=q.clone(), k=k.clone(), v=v_view.clone(), out=hn.clone(), kv_heads=KV)
            if trim_here:
                ns = sel_rows.numel()
                att_s = self._buf("llm_att_sel", ns, hdim, dev)
                x_s = self._buf("llm_x_sel", ns, hdim, dev)
                hn_s = self._buf("llm_hn_sel", ns, hdim, dev)
                act_s = self._buf("llm_act_sel", ns, ff, dev)
                ops.embed_gather(sel_rows, hn, att_s, -1)     # row gathers (table = activation rows)
                ops.embed_gather(sel_rows, x, x_s, -1)
                if "wo" in e8:
                    ops.gemm(ops.quantize_mxfp8(att_s), d["wo_8"][li], x_s, EPI_SCALE_RES, res=x_s)
                else:
                    ops.gemm(att_s, layer.attention.wo.weight, x_s, EPI_SCALE_RES, res=x_s)
                self._llm_ffn(d, li, layer, x_s, hn_s, act_s, "sel")
                return x_s
            if "wo" in e8:
                ops.gemm(ops.quantize_mxfp8(hn, out=self._buf8("llm_a8a", n, hdim, dev)), d["wo_8"][li], x, EPI_SCALE_RES, res=x)
            else:
                ops.gemm(hn, layer.attention.wo.weight, x, EPI_SCALE_RES, res=x)
            self._llm_ffn(d, li, layer, x, hn, act, "all")
            if self.debug_probes is not None:
                self.debug_probes[f"llm_layer{li}"] = x.clone()
        return x

    def _llm_ffn(self, d, li: int, layer, x, hn, act, tag: str):
        """x += w2(silu(w1 h) * w3 h), h = ffn_norm(x)  (modeling_internlm2.py:261-264,669-679) in place on the rows ``x``;
        ``hn`` / ``act`` bf16 scratch of [rows, hidden] / [rows, intermediate].  mxfp8: the norm and the SiLU-mul epilogue write
        e4m3 + block scales, both GEMMs run on MXFP8 operands."""
        lc = self.config.llm_config
        S8 = self.ffn_fp8_linears if self.ffn_format == "mxfp8" else frozenset()
        if "w13" in S8 or "w2" in S8:
            rows, dev = x.shape[0], x.device
            if "w13" in S8:
                h8 = self._buf8(f"llm_h8_{tag}", rows, x.shape[1], dev)
                ops.rmsnorm_mxfp8(x, layer.ffn_norm.weight, h8, lc.rms_norm_eps)
                mid = self._buf8(f"llm_a8_{tag}", rows, lc.intermediate_size, dev) if "w2" in S8 else act
                ops.gemm(h8, d["w13_8"][li], mid, EPI_SILU_MUL)
            else:
                ops.rmsnorm(x, layer.ffn_norm.weight, hn, lc.rms_norm_eps)
                ops.gemm(hn, d["w13"][li], act, EPI_SILU_MUL)
                mid = ops.quantize_mxfp8(act, out=self._buf8(f"llm_a8_{tag}", rows, lc.intermediate_size, dev))
            if "w2" in S8:
                ops.gemm(mid, d["w2_8"][li], x, EPI_SCALE_RES, res=x)
            else:
                ops.gemm(mid, layer.feed_forward.w2.weight, x, EPI_SCALE_RES, res=x)
            return
        if self.norm_fusion:
            rstd = self._buf("llm_rstd", 1, ops.padded_rows(x.shape[0]), x.device, dtype=torch.float32).view(-1)
            ops.row_stats(x, rstd, None, lc.rms_norm_eps)
            ops.gemm(x, d["llm_fold"][li]["w13"], act, EPI_SILU_MUL, folded_norm=(rstd,))
        else:
            ops.rmsnorm(x, layer.ffn_norm.weight, hn, lc.rms_norm_eps)
            ops.gemm(hn, d["w13"][li], act, EPI_SILU_MUL)
        ops.gemm(act, layer.feed_forward.w2.weight, x, EPI_SCALE_RES, res=x)

    def _build_prefix(self, d, prefix_ids: np.ndarray, settings, padded_len: int, dev) -> None:
        """Fills ``self._prefix``: the language tower over the ``P`` prefix tokens alone (one short sequence, positions 0 .. P - 1;
        P is a multiple of 64), keeping every layer's rotated keys and values.  Under the causal mask these rows are exactly
        what any sequence that starts with those tokens computes for them - bit for bit when no GEMM slices K (a row's sums do
        not depend on the rows around it), up to fp32 re-association otherwise."""
        lc = self.config.llm_config
        P = int(prefix_ids.shape[0])
        H, KV, hd = lc.num_attention_heads, lc.num_key_value_heads, lc.hidden_size // lc.num_attention_heads
        G = H // KV
        ids = torch.from_numpy(prefix_ids.astype(np.int32)).to(dev)
        x = self._buf("llm_x", P, lc.hidden_size, dev)
        ops.embed_gather(ids, self.model.language_model.model.tok_embeddings.weight, x, self.model.img_context_token_id)
        cu = torch.tensor([0, P], dtype=torch.int32, device=dev)
        pos = torch.arange(P, dtype=torch.int32, device=dev)
        store = dict(k=[], v=[])
        self._language_tower(d, x, cu, pos, P, None, padded_len=padded_len, snapshot=(P, store))
        # the last layer's values once more in the [k | v] column layout of the trimmed last layer (wkv_last: kv head h at
        # columns 256 h + 128 ...), copied from their wqkv columns ((G + 2) 128 h + (G + 1) 128 ...): same sums, same values
        if self._phi3:
            # Phi-3 layout: store["k"][li] = the prefix rows of the layer's whole [q | k | v] projection (the live views' row stride);
            # the trimmed last layer reads [k | v] rows of their own: the same values, copied out
            kv_last = store["k"][-1][:, H * hd:].contiguous()
            v_last = kv_last
            extra = dict(kv_last=kv_last)
        else:
            v_last = torch.zeros(P, 2 * KV * hd, dtype=BF16, device=dev)
            v_last.view(P, KV, 2, hd)[:, :, 1] = store["v"][-1].view(P, KV, G + 2, hd)[:, :, G + 1]
            extra = {}
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))   # a later forward on ANOTHER stream waits for the rows to exist
        self._prefix = dict(settings=settings, ids=prefix_ids.copy(), P=P, k=store["k"], v=store["v"], v_last=v_last, ready=ready,
                            stream=torch.cuda.current_stream(dev).cuda_stream, **extra)

    # -- forward ---------------------------------------------------------------------------------
    def _forward_group(self, d, tag: str, pixel_values, host_ids, outs, lo: int, probes_ok: bool):
        """Scores the batch on the CURRENT stream of the model's device (outputs go to rows [lo, lo+B) of ``outs``).
        ``host_ids``: function returning the (ids, mask) numpy arrays (``_host_ids_begin``) - called only after the vision
        tower's launches are out."""
        dev = pixel_values.device
        self._ws_tag = tag
        # split-K scratch of this forward's GEMMs (private to this model instance and to the stream the forward runs on;
        # ops keeps it per THREAD: another thread scoring with another model instance has its own)
        ops.set_gemm_workspace(self._buf("gemm_ws", 1, ops.gemm_workspace_bytes(), dev, dtype=torch.uint8)
                               if self.use_gemm_workspace else None)
        vit = self._vision_tower_launch(d, pixel_values)     # (needs no ids: enqueued before the host waits for them)
        input_ids, attention_mask = host_ids()
        trimmed = self.debug_probes is None
        lc = self.config.llm_config
        hdim = lc.hidden_size
        # what the cached prefix rows depend on besides their ids: the weights (as _prepare tracks them), the rotary base
        # (dynamic NTK may have replaced it), every numerics setting, and which buffer the last layer's values live in
        tail_form = bool(trimmed and self.trim_last_layer and not self.norm_fusion and not self._exp8_set())
        if self._phi3:   # (LongRoPE keeps no state: the tables depend on whether the padded width exceeds the original window)
            settings = (self._derived_sig, int(input_ids.shape[1]) > lc.original_max_position_embeddings, self.attention_scores,
                        self.ffn_format, bool(self.norm_fusion), self._exp8_set(), bool(self.use_gemm_workspace), tail_form, str(dev))
        else:
            # (the rotary base this forward WILL rotate with - the state itself advances in the tower, after the batch is validated)
            settings = (self._derived_sig, self._rope_next_state(int(input_ids.shape[1]))["base"], self.attention_scores, self.ffn_format, bool(self.norm_fusion),
                        self._exp8_set(), bool(self.use_gemm_workspace), tail_form, str(dev))
        use_prefix = bool(self.prefix_cache and trimmed)
        hit = []

        def lookup(prefix_ids: np.ndarray) -> bool:
            c = self._prefix
            hit.append(c is not None and c["settings"] == settings and c["ids"].shape == prefix_ids.shape
                       and bool((c["ids"] == prefix_ids).all()))
            if hit[0]:
                self._prefix_misses = 0
                return True
            # a prefix this model holds no keys / values for (first forward, new weights, another prompt, another setting):
            # computed once, by a pass over the prefix tokens ALONE, before this forward's tower - a cold and a warm cache run
            # the same cached computation.  A caller whose prompts share NO constant prefix (every forward another one) would
            # pay that pass - about 3 ms - for nothing each time: after three misses in a row the prefix is only rebuilt when a
            # candidate REPEATS, and forwards in between run uncached (every row, as prefix_cache = False does).
            # What that makes a result depend on: the cached and the uncached evaluation of one batch are the same
            # function up to the re-association of fp32 sums - bit-identical when no GEMM slices K, a few per cent of the hidden
            # rows' norm otherwise (tests/test_e2e_gpu.py::test_prefix_cache_is_invisible_and_invalidates) - and WHICH of the two
            # runs depends on the call history (this guard) and on the batch (the prefix is what all its samples share).  A
            # caller that needs history-independent bits sets prefix_cache = False (scripts/eval: --no-prefix-cache).
            key = (settings, prefix_ids.tobytes())
            self._prefix_misses += 1
            repeat = self._prefix_last_miss == key
            self._prefix_last_miss = key
            if self._prefix_misses > 3 and not repeat:
                return False
            if repeat:
                self._prefix_misses = 0      # a stable prefix again: the next change of prompt gets its three tries back
            self._build_prefix(d, prefix_ids, settings, int(input_ids.shape[1]), dev)
            return True

        info = self._analyse_ids(input_ids, attention_mask, pixel_values.shape[0], lookup if use_prefix else None)  # host arrays
        B, total = info["B"], info["total"]
        prefix = self._prefix if info["skip"] else None
        if prefix is not None and hit[0]:
            self.prefix_cache_hits += 1
            cur = torch.cuda.current_stream(dev)
            if prefix["stream"] != cur.cuda_stream:   # built on another stream: wait for the rows, and tell the allocator who reads them
                cur.wait_event(prefix["ready"])
                for t in prefix["k"] + prefix["v"] + [prefix["v_last"]]:
                    t.record_stream(cur)

        def up(a):
            return torch.from_numpy(a).to(dev, non_blocking=True)

        ids, positions, cu = up(info["ids"]), up(info["positions"]), up(info["cu"])
        img_rows, sel_rows = up(info["img_rows"]), up(info["sel_rows"])
        tail = None
        if tail_form:
            tail = (up(info["tail_rows"]), up(info["tail_pos"]), up(info["cu_tail"]), info["max_tail"], up(info["sel_in_tail"]))
        hidden = self._buf("llm_x", total, hdim, dev)
        ops.embed_gather(ids, self.model.language_model.model.tok_embeddings.weight, hidden, self.model.img_context_token_id)
        self._vision_tower_splice(vit, hidden, img_rows)
        if self.debug_probes is not None and probes_ok:
            self.debug_probes["llm_embed"] = hidden.clone()
        last_x = self._language_tower(d, hidden, cu, positions, info["max_len"], sel_rows if trimmed else None, padded_len=info["N"],
                                      tail=tail, prefix=prefix)

        # final RMSNorm only on the 2 rows per sample the heads read (hidden_states[-1] is post-norm, moe_reward.py:211)
        h_r, h_g = outs["hidden_state"][lo:lo + B], outs["prompt_embedding"][lo:lo + B]
        norm_w = self.model.language_model.model.norm.weight
        if trimmed:   # last_x holds exactly the selected rows: [0, B) reward rows, [B, 2B) gating rows
            ops.rmsnorm(last_x[:B], norm_w, h_r, lc.rms_norm_eps)
            ops.rmsnorm(last_x[B:], norm_w, h_g, lc.rms_norm_eps)
        else:
            ops.rmsnorm(hidden, norm_w, h_r, lc.rms_norm_eps, row_index=sel_rows[:B])
            ops.rmsnorm(hidden, norm_w, h_g, lc.rms_norm_eps, row_index=sel_rows[B:])
        self._run_heads(d, h_r, h_g, outs, lo, B, dev)

    def _run_heads(self, d, h_r: torch.Tensor, h_g: torch.Tensor, outs, lo: int, B: int, dev):
        """moe_reward.py:239-297 on the two (post-norm) hidden-state rows per sample: the gating MLPs' hidden layers as
        batch-sized GEMMs, everything else (regression matvec, last gating layers, the softmaxes, the weighted sums) in
        one ``reward_heads`` launch.  Writes rows [lo, lo + B) of ``outs``."""
        hdim = self.config.llm_config.hidden_size
        gh = self.aspect_gating.layers[0].out_features

        def gating_hidden(net: GatingNetwork, name: str) -> torch.Tensor:
            cur = h_g
            for j, layer in enumerate(net.layers[:-1]):
                out = self._buf(f"gate_{name}{j & 1}", B, layer.out_features, dev)
                ops.gemm(cur, layer.weight, out, EPI_BIAS_RELU, bias=layer.bias)
                cur = out
            return cur

        ga = gating_hidden(self.aspect_gating, "a")
        gc = gating_hidden(self.criteria_gating, "c")
        nobj, nasp = self.num_objectives, self.num_aspects
        hd = HeadsDesc()
        hd.hr, hd.hg, hd.ldh, hd.hidden = h_r.data_ptr(), h_g.data_ptr(), h_r.stride(0), hdim
        hd.ga, hd.gc, hd.ldg, hd.gate_hidden = ga.data_ptr(), gc.data_ptr(), ga.stride(0), gh
        hd.w_reg = self.regression_layer.weight.data_ptr()
        hd.w_transform = self.reward_transform_matrix.data_ptr()
        la, lcg = self.aspect_gating.layers[-1], self.criteria_gating.layers[-1]
        hd.wa, hd.ba, hd.wc, hd.bc = la.weight.data_ptr(), la.bias.data_ptr(), lcg.weight.data_ptr(), lcg.bias.data_ptr()
        hd.ls_a, hd.ls_c = self.aspect_gating.logit_scale.data_ptr(), self.criteria_gating.logit_scale.data_ptr()
        hd.temperature = float(self.criteria_gating.temperature)
        hd.batch, hd.n_obj, hd.n_asp = B, nobj, nasp
        hd.group_offsets, hd.group_index = d["group_offsets"].data_ptr(), d["group_index"].data_ptr()
        hd.rewards = outs["rewards"][lo:].data_ptr()
        hd.criteria_gating = outs["criteria_gating_output"][lo:].data_ptr()
        hd.aspect_gating = outs["aspect_gating_output"][lo:].data_ptr()
        hd.aspect_weights = outs["aspect_weights"][lo:].data_ptr()
        hd.weighted_last = outs["weighted_scores"][lo:].data_ptr()
        hd.aspect_scores = outs["aspect_scores"][lo:].data_ptr()
        hd.score = outs["score"][lo:].data_ptr()
        hd.packed34 = outs["packed34"][lo:].data_ptr()
        ops.reward_heads(hd, dev)

    def _alloc_outputs(self, B: int, dev) -> Dict[str, torch.Tensor]:
        hdim = self.config.llm_config.hidden_size
        nobj, nasp = self.num_objectives, self.num_aspects
        return dict(
            rewards=torch.empty(B, nobj, dtype=BF16, device=dev),
            hidden_state=torch.empty(B, hdim, dtype=BF16, device=dev),
            prompt_embedding=torch.empty(B, hdim, dtype=BF16, device=dev),
            criteria_gating_output=torch.empty(B, nobj, dtype=BF16, device=dev),
            aspect_gating_output=torch.empty(B, nasp, dtype=BF16, device=dev),
            aspect_weights=torch.empty(B, nobj, dtype=BF16, device=dev),
            weighted_scores=torch.empty(B, dtype=BF16, device=dev),
            aspect_scores=torch.empty(B, nasp, dtype=torch.float32, device=dev),
            score=torch.empty(B, dtype=torch.float32, device=dev),
            packed34=torch.empty(B, 1 + nasp + nobj, dtype=torch.float32, device=dev))

    @torch.no_grad()
    def heads_forward(self, hidden_state: torch.Tensor, prompt_embedding: torch.Tensor) -> CustomOutput:
        """Everything downstream of the backbone (moe_reward.py:239-297) on given post-norm rows: ``hidden_state``
        [B, hidden] = state at the last non-pad token, ``prompt_embedding`` [B, hidden] = state at the gating pattern.
        Not part of the reference's API: the entry the isolated head-kernel tests and the engineered rank sets use
        (same code path ``forward`` ends with)."""
        dev = self.model.device
        if dev.type != "cuda":
            raise RuntimeError("heads_forward runs on the MI355X only")
        d = self._prepare(dev)
        h_r = hidden_state.to(dev, BF16).contiguous()
        h_g = prompt_embedding.to(dev, BF16).contiguous()
        B = h_r.shape[0]
        outs = self._alloc_outputs(B, dev)
        outs["hidden_state"].copy_(h_r)
        outs["prompt_embedding"].copy_(h_g)
        self._ws_tag = "g0"
        with torch.cuda.device(dev):
            self._run_heads(d, outs["hidden_state"], outs["prompt_embedding"], outs, 0, B, dev)
        self.last_packed34 = outs.pop("packed34")
        return CustomOutput(**outs)

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, input_ids: torch.Tensor = None,
                attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                image_flags: Optional[torch.Tensor] = None, past_key_values=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None) -> CustomOutput:
        if input_ids is None:
            raise ValueError("input_ids is required (the reward heads locate their rows from the token ids)")
        if position_ids is not None or past_key_values is not None or labels is not None:
            raise NotImplementedError("position_ids / past_key_values / labels are not part of the scoring path")
        if pixel_values.dim() != 4:
            raise ValueError(f"wrong pixel_values size: {tuple(pixel_values.shape)}")
        dev = self.model.device
        if dev.type != "cuda":
            raise RuntimeError("InternVLChatRewardModeling.forward runs on the MI355X only: move the model with .cuda()")
        d = self._prepare(dev)
        if pixel_values.dtype != BF16:
            raise TypeError(f"pixel_values must be bfloat16 like the model (got {pixel_values.dtype}); "
                            "callers cast with .to(torch.bfloat16) (eval_genai_mjvideo.py:131)")
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [batch, seq], got {tuple(input_ids.shape)}")
        if self.model.img_context_token_id is None:
            raise ValueError("model.model.img_context_token_id is not set (eval_genai_mjvideo.py:115)")
        pixel_values = pixel_values.to(dev).contiguous()
        B = input_ids.shape[0]
        if self.config.pad_token_id is None and B != 1:   # moe_reward.py:218-219
            raise ValueError("Cannot handle batch sizes > 1 if no padding token is defined.")
        outs = self._alloc_outputs(B, dev)
        try:
            # the model's device becomes the current device for the whole forward (allocations, events, the library's
            # per-device kernel attributes), whatever the caller's current device is: model.cuda(1) works like the reference
            with torch.cuda.device(dev):
                host_ids = self._host_ids_begin(input_ids, attention_mask)
                self._forward_group(d, "g0", pixel_values, host_ids, outs, 0, True)
        finally:   # the split-K scratch is this thread's default in ops: do not leave it behind for other gemm callers
            ops.set_gemm_workspace(None)
        self.last_packed34 = outs.pop("packed34")
        return CustomOutput(**outs)
