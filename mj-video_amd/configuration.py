"""Configuration objects of the MJ-VIDEO reward-scoring path.

Mirrors the reference's configuration surface (attribute names, kwargs override
rules, ``to_dict`` layout) without depending on ``transformers``:

* ``InternVisionConfig``        <- scripts/model/internvl2/configuration_intern_vit.py:16-120
* ``InternLM2Config``           <- scripts/model/internvl2/configuration_internlm2.py:27-150
* ``Phi3Config``                <- transformers/models/phi3/configuration_phi3.py (transformers 5.15; the language model of
                                   BASELINE configs[4]'s InternVL2-4B backbone - the reference has no Phi-3 code)
* ``InternVLChatConfig``        <- scripts/model/internvl2/configuration_internvl_chat.py:19-96
* ``InternVLChatRewardModelingConfig`` <- scripts/model/moe_reward.py:92-133

``from_pretrained(path_or_dir, **kwargs)`` reads ``config.json`` from a local
directory (there is no hub access) and then re-applies the reward-head kwargs
exactly as moe_reward.py:109-121 does.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, Optional


class _ConfigBase:
    """Minimal stand-in for the parts of HF ``PretrainedConfig`` the path uses."""

    model_type = ""

    def _init_common(self, kwargs: Dict[str, Any]) -> None:
        # attributes callers read on the reference's PretrainedConfig objects
        self.use_return_dict = kwargs.pop("use_return_dict", True)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.torch_dtype = kwargs.pop("torch_dtype", None)
        self.architectures = kwargs.pop("architectures", None)
        self.num_labels = kwargs.pop("num_labels", 2)
        kwargs.pop("model_type", None)
        kwargs.pop("transformers_version", None)
        self._extra = {}
        for k, v in kwargs.items():
            self._extra[k] = v
            setattr(self, k, v)

    def to_dict(self) -> Dict[str, Any]:
        out = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if not k.startswith("_")}
        out["model_type"] = self.model_type
        return out

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True, default=str)

    def save_pretrained(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as f:
            f.write(self.to_json_string())

    @classmethod
    def from_dict(cls, d: Dict[str, Any], **kwargs):
        d = copy.deepcopy(d)
        d.update(kwargs)
        return cls(**d)

    @staticmethod
    def _read_json(name_or_path: str) -> Dict[str, Any]:
        path = name_or_path
        if os.path.isdir(path):
            path = os.path.join(path, "config.json")
        if not os.path.isfile(path):
            raise FileNotFoundError(
                f"config for '{name_or_path}' not found: this build has no hub access, "
                f"pass a local directory holding config.json")
        with open(path) as f:
            return json.load(f)

    def __repr__(self) -> str:
        return f"{type(self).__name__} {self.to_json_string()}"


class InternVisionConfig(_ConfigBase):
    model_type = "intern_vit_6b"

    def __init__(self, num_channels=3, patch_size=14, image_size=224, qkv_bias=False, hidden_size=3200,
                 num_attention_heads=25, intermediate_size=12800, qk_normalization=True, num_hidden_layers=48,
                 use_flash_attn=True, hidden_act="gelu", norm_type="rms_norm", layer_norm_eps=1e-6, dropout=0.0,
                 drop_path_rate=0.0, attention_dropout=0.0, initializer_range=0.02, initializer_factor=0.1,
                 **kwargs):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.dropout = dropout
        self.drop_path_rate = drop_path_rate
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_channels = num_channels
        self.patch_size = patch_size
        self.image_size = image_size
        self.initializer_range = initializer_range
        self.initializer_factor = initializer_factor
        self.attention_dropout = attention_dropout
        self.layer_norm_eps = layer_norm_eps
        self.hidden_act = hidden_act
        self.norm_type = norm_type
        self.qkv_bias = qkv_bias
        self.qk_normalization = qk_normalization
        self.use_flash_attn = use_flash_attn
        self._init_common(kwargs)

    @classmethod
    def from_pretrained(cls, name_or_path, **kwargs):
        d = cls._read_json(name_or_path)
        if "vision_config" in d:
            d = d["vision_config"]
        return cls.from_dict(d, **kwargs)


class InternLM2Config(_ConfigBase):
    model_type = "internlm2"

    def __init__(self, vocab_size=103168, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, hidden_act="silu",
                 max_position_embeddings=2048, initializer_range=0.02, rms_norm_eps=1e-6, use_cache=True,
                 pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False, bias=True,
                 rope_theta=10000, rope_scaling=None, attn_implementation="eager", **kwargs):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.bias = bias
        self.num_key_value_heads = num_attention_heads if num_key_value_heads is None else num_key_value_heads
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        self.rope_theta = rope_theta
        self.rope_scaling = copy.deepcopy(rope_scaling)
        self._rope_scaling_validation()
        self.attn_implementation = attn_implementation or "eager"
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.tie_word_embeddings = tie_word_embeddings
        self._init_common(kwargs)

    def _rope_scaling_validation(self):
        # same acceptance rule as configuration_internlm2.py:128-150
        rs = self.rope_scaling
        if rs is None:
            return
        if not isinstance(rs, dict) or len(rs) != 2:
            raise ValueError("`rope_scaling` must be a dictionary with with two fields, `type` and `factor`, "
                             f"got {rs}")
        t, f = rs.get("type"), rs.get("factor")
        if t not in ("linear", "dynamic"):
            raise ValueError(f"`rope_scaling`'s type field must be one of ['linear', 'dynamic'], got {t}")
        if f is None or not isinstance(f, float) or f < 1.0:
            raise ValueError(f"`rope_scaling`'s factor field must be a float >= 1, got {f}")


class Phi3Config(_ConfigBase):
    """The fields of transformers' ``Phi3Config`` the scoring path reads (transformers/models/phi3/configuration_phi3.py,
    transformers 5.15).  ``rope_scaling`` = None or the LongRoPE dict {"type": "longrope" | "su", "short_factor": [...],
    "long_factor": [...]} (+ optional "factor" / "attention_factor"), lists of head_dim * partial_rotary_factor / 2 floats; the
    newer ``rope_parameters`` spelling is accepted and folded into the same fields."""
    model_type = "phi3"

    def __init__(self, vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, resid_pdrop=0.0, embd_pdrop=0.0, attention_dropout=0.0,
                 hidden_act="silu", max_position_embeddings=4096, original_max_position_embeddings=4096,
                 initializer_range=0.02, rms_norm_eps=1e-5, use_cache=True, tie_word_embeddings=False, rope_theta=10000.0,
                 rope_scaling=None, partial_rotary_factor=1.0, bos_token_id=1, eos_token_id=32000, pad_token_id=32000,
                 sliding_window=None, attn_implementation="eager", **kwargs):
        rp = kwargs.pop("rope_parameters", None)
        if rp:   # transformers >= 5 spelling
            rp = copy.deepcopy(rp)
            rope_theta = rp.pop("rope_theta", rope_theta)
            partial_rotary_factor = rp.pop("partial_rotary_factor", partial_rotary_factor)
            original_max_position_embeddings = rp.pop("original_max_position_embeddings", original_max_position_embeddings)
            kind = rp.pop("rope_type", rp.pop("type", "default"))
            if kind != "default":
                rope_scaling = dict(rp, type=kind)
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_attention_heads if num_key_value_heads is None else num_key_value_heads
        self.resid_pdrop = resid_pdrop
        self.embd_pdrop = embd_pdrop
        self.attention_dropout = attention_dropout
        self.hidden_act = hidden_act
        self.max_position_embeddings = max_position_embeddings
        self.original_max_position_embeddings = original_max_position_embeddings
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        self.rope_theta = rope_theta
        self.rope_scaling = copy.deepcopy(rope_scaling)
        self.partial_rotary_factor = partial_rotary_factor
        self._rope_scaling_validation()
        self.sliding_window = sliding_window
        self.attn_implementation = attn_implementation or "eager"
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.tie_word_embeddings = tie_word_embeddings
        self._init_common(kwargs)

    def _rope_scaling_validation(self):
        """configuration_phi3.py: the two factor lists must hold rotary_dim / 2 numbers each"""
        rs = self.rope_scaling
        if rs is None:
            return
        if not isinstance(rs, dict) or rs.get("type", rs.get("rope_type")) not in ("longrope", "su", "yarn"):
            raise ValueError(f"`rope_scaling` must be a LongRoPE dictionary (type 'longrope' / 'su'), got {rs}")
        if rs.get("type", rs.get("rope_type")) == "yarn":
            raise NotImplementedError("Phi-3 with yarn rope scaling is not built (InternVL2-4B's Phi-3-mini uses LongRoPE)")
        rs["type"] = "longrope"
        rs.pop("rope_type", None)
        n = int(self.hidden_size // self.num_attention_heads * self.partial_rotary_factor) // 2
        for key in ("short_factor", "long_factor"):
            f = rs.get(key)
            if not isinstance(f, (list, tuple)) or not all(isinstance(v, (int, float)) for v in f):
                raise ValueError(f"`rope_scaling`'s {key} field must be a list of numbers, got {f}")
            if len(f) != n:
                raise ValueError(f"`rope_scaling`'s {key} field must have length {n}, got {len(f)}")


class InternVLChatConfig(_ConfigBase):
    model_type = "internvl_chat"
    is_composition = True

    def __init__(self, vision_config=None, llm_config=None, use_backbone_lora=0, use_llm_lora=0, select_layer=-1,
                 force_image_size=None, downsample_ratio=0.5, template=None, dynamic_image_size=False,
                 use_thumbnail=False, ps_version="v1", min_dynamic_patch=1, max_dynamic_patch=6, **kwargs):
        if vision_config is None:
            vision_config = {}
        if llm_config is None:
            llm_config = {}
        if isinstance(vision_config, InternVisionConfig):
            vision_config = vision_config.to_dict()
        if isinstance(llm_config, (InternLM2Config, Phi3Config)):
            llm_config = llm_config.to_dict()
        self.vision_config = InternVisionConfig(**copy.deepcopy(vision_config))
        arch = (llm_config.get("architectures") or [None])[0]
        if arch == "InternLM2ForCausalLM":
            self.llm_config = InternLM2Config(**copy.deepcopy(llm_config))
        elif arch == "Phi3ForCausalLM":
            # not in the reference's dispatch (configuration_internvl_chat.py:50-55: Llama / InternLM2): the upstream InternVL2-4B
            # config adds exactly this branch; BASELINE configs[4] names that backbone
            self.llm_config = Phi3Config(**copy.deepcopy(llm_config))
        else:
            # configuration_internvl_chat.py:50-55 also admits LlamaForCausalLM; MJ-VIDEO-2B is InternLM2 only.
            raise ValueError("Unsupported architecture: {}".format(arch))
        self.use_backbone_lora = use_backbone_lora
        self.use_llm_lora = use_llm_lora
        self.select_layer = select_layer
        self.force_image_size = force_image_size
        self.downsample_ratio = downsample_ratio
        self.template = template
        self.dynamic_image_size = dynamic_image_size
        self.use_thumbnail = use_thumbnail
        self.ps_version = ps_version
        self.min_dynamic_patch = min_dynamic_patch
        self.max_dynamic_patch = max_dynamic_patch
        self.pad_token_id = kwargs.pop("pad_token_id", None)
        self._init_common(kwargs)

    def to_dict(self):
        out = super().to_dict()
        out["vision_config"] = self.vision_config.to_dict()
        out["llm_config"] = self.llm_config.to_dict()
        return out

    @classmethod
    def from_pretrained(cls, name_or_path, **kwargs):
        d = cls._read_json(name_or_path)
        return cls.from_dict(d, **kwargs)


class InternVLChatRewardModelingConfig(InternVLChatConfig):
    """Chat config + the MoE reward-head fields (moe_reward.py:92-133)."""

    _HEAD_FIELDS = ("num_objectives", "num_aspects", "aspect2criteria", "gating_temperature",
                    "gating_hidden_dim", "gating_n_hidden")
    # `<|im_end|><|im_start|>assistant\n` in InternLM2 token ids: the module constant of moe_reward.py:45-48
    INTERNLM2_GATING_PATTERN = (92542, 92543, 525, 11353, 364)

    def __init__(self, internVLChatConfigName_or_path=None, **kwargs):
        head = {k: kwargs.pop(k) for k in self._HEAD_FIELDS if k in kwargs}
        super().__init__(**kwargs)
        self.num_objectives = head.get("num_objectives", 0)
        self.num_aspects = head.get("num_aspects", 0)
        self.aspect2criteria = head.get("aspect2criteria", {})
        self.gating_temperature = head.get("gating_temperature", 1.0)
        self.gating_hidden_dim = head.get("gating_hidden_dim", 1024)
        self.gating_n_hidden = head.get("gating_n_hidden", 3)
        # the token ids whose LAST occurrence marks the gating row (end of the user turn + the assistant header).  The
        # reference hard-codes the InternLM2 tokenizer's ids (moe_reward.py:45-48); a config field here (not one of the
        # reference's: it rides in kwargs / config.json) because the Phi-3 backbone of configs[4] has another tokenizer.
        if not hasattr(self, "gating_token_pattern") or self.gating_token_pattern is None:
            self.gating_token_pattern = list(self.INTERNLM2_GATING_PATTERN)
        self.gating_token_pattern = [int(t) for t in self.gating_token_pattern]

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        head = {k: kwargs.pop(k) for k in cls._HEAD_FIELDS if k in kwargs}
        d = cls._read_json(pretrained_model_name_or_path)
        d.update(kwargs)
        config = cls(**d)
        # kwargs win over what config.json held (moe_reward.py:114-120)
        for k, v in head.items():
            setattr(config, k, v)
        # json turns int keys into strings; restore the caller-visible int keys
        config.aspect2criteria = {int(k): list(v) for k, v in dict(config.aspect2criteria).items()}
        return config


def mjvideo_2b_config_dict(image_size: int = 448) -> Dict[str, Any]:
    """The InternVL2-2B architecture MJ-VIDEO-2B is fine-tuned from (SURVEY.md §8 notation).

    The checkpoint's config.json is not available offline; these are its published dimensions
    (2 205 754 368 parameters, matching SURVEY.md §6).
    """
    return dict(
        vision_config=dict(
            architectures=["InternVisionModel"], num_channels=3, patch_size=14, image_size=448, qkv_bias=True,
            hidden_size=1024, num_attention_heads=16, intermediate_size=4096, qk_normalization=False,
            num_hidden_layers=24, use_flash_attn=True, hidden_act="gelu", norm_type="layer_norm",
            layer_norm_eps=1e-6, dropout=0.0, drop_path_rate=0.0, attention_dropout=0.0,
            initializer_range=0.02, initializer_factor=1.0),
        llm_config=dict(
            architectures=["InternLM2ForCausalLM"], vocab_size=92553, hidden_size=2048, intermediate_size=8192,
            num_hidden_layers=24, num_attention_heads=16, num_key_value_heads=8, hidden_act="silu",
            max_position_embeddings=32768, initializer_range=0.02, rms_norm_eps=1e-5, use_cache=True,
            pad_token_id=2, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False, bias=False,
            rope_theta=1000000, rope_scaling={"type": "dynamic", "factor": 2.0}, attn_implementation="eager"),
        select_layer=-1, force_image_size=image_size, downsample_ratio=0.5, template="internlm2-chat",
        dynamic_image_size=True, use_thumbnail=True, ps_version="v2", min_dynamic_patch=1, max_dynamic_patch=12,
    )


# stand-in special-token ids of the InternVL2-4B (Phi-3) tokenizer [recalled, unpinned: no tokenizer offline]: `<|end|>` 32007,
# `<|assistant|>` 32001, "\n" 13 - the phi3-chat template's counterpart of `<|im_end|><|im_start|>assistant\n`
# (conversation.py:368-379: roles ('<|user|>\n', '<|assistant|>\n'), sep '<|end|>')
PHI3_GATING_PATTERN = (32007, 32001, 13)


def longrope_factors(n: int, seed: int = 0):
    """Seed-defined stand-ins for a checkpoint's LongRoPE factor lists (the real lists of Phi-3-mini-128k are in its config.json,
    not available offline): ``n`` short factors rising from 1.0 to about 2.8 and long factors rising from 1.0 to about 64, the
    shape of the published ones (low frequencies are stretched more)."""
    import numpy as np
    g = np.random.Generator(np.random.Philox(key=[seed, 0x10A6]))
    t = np.linspace(0.0, 1.0, n)
    short = 1.0 + 1.8 * t ** 2 + 0.05 * g.random(n)
    long = np.exp(np.log(64.0) * t ** 1.5) + 0.05 * g.random(n)
    return [float(np.float32(v)) for v in short], [float(np.float32(v)) for v in long]


def internvl2_4b_config_dict(image_size: int = 448) -> Dict[str, Any]:
    """The InternVL2-4B architecture of BASELINE configs[4]: the vision tower of the 2B model (InternViT-300M-448px) + a
    Phi-3-mini-128k decoder (hidden 3072, 32 heads x 96, MHA, ff 8192, 32 layers, LongRoPE 4096 -> 131072) [recalled: published
    dimensions; the checkpoint's config.json, factor lists and tokenizer are not available offline]."""
    d = mjvideo_2b_config_dict(image_size)
    short, long = longrope_factors(48)
    d["llm_config"] = dict(
        architectures=["Phi3ForCausalLM"], vocab_size=32020, hidden_size=3072, intermediate_size=8192, num_hidden_layers=32,
        num_attention_heads=32, num_key_value_heads=32, hidden_act="silu", max_position_embeddings=131072,
        original_max_position_embeddings=4096, initializer_range=0.02, rms_norm_eps=1e-5, use_cache=True,
        rope_theta=10000.0, rope_scaling={"type": "longrope", "short_factor": short, "long_factor": long},
        bos_token_id=1, eos_token_id=32000, pad_token_id=32000, sliding_window=262144, tie_word_embeddings=False,
        attn_implementation="eager")
    d["template"] = "phi3-chat"
    d["gating_token_pattern"] = list(PHI3_GATING_PATTERN)
    return d


def tiny_phi3_config_dict(image_size: int = 56) -> Dict[str, Any]:
    """``tiny_config_dict`` with a two-layer Phi-3 decoder (2 heads x 96, MHA; LongRoPE with a 64-position original window so that
    BOTH factor lists are exercised by sequences of a few hundred tokens)."""
    d = internvl2_4b_config_dict(image_size)
    d["vision_config"].update(hidden_size=128, num_attention_heads=2, intermediate_size=512, num_hidden_layers=2,
                              image_size=image_size)
    d["llm_config"].update(hidden_size=192, num_attention_heads=2, num_key_value_heads=2, intermediate_size=512,
                           num_hidden_layers=2, max_position_embeddings=4096, original_max_position_embeddings=128)
    d["force_image_size"] = image_size
    return d


DEFAULT_ASPECT2CRITERIA = {
    0: [0, 1, 2, 3, 4],
    1: [5, 6, 7, 8, 9, 10],
    2: [11, 12, 13, 14, 15],
    3: [16, 17, 18, 19, 20, 21, 22],
    4: [23, 24, 25, 26, 27],
}


def mjvideo_head_kwargs() -> Dict[str, Any]:
    """Reward-head defaults of the eval driver (eval_genai_mjvideo.py:36-47)."""
    return dict(num_objectives=28, num_aspects=5, aspect2criteria=copy.deepcopy(DEFAULT_ASPECT2CRITERIA),
                gating_temperature=1.0, gating_hidden_dim=1024, gating_n_hidden=3)


def tiny_config_dict(image_size: int = 56) -> Dict[str, Any]:
    """A few-million-parameter config with the same structure, for full-intermediate fixtures.

    Head dims are kept at the production values (ViT 64, LLM 128) because the attention kernels are
    specialised on them; the vocabulary keeps its size because the gating token pattern
    (moe_reward.py:48) uses ids up to 92543.
    """
    d = mjvideo_2b_config_dict(image_size)
    d["vision_config"].update(hidden_size=128, num_attention_heads=2, intermediate_size=512,
                              num_hidden_layers=2, image_size=image_size)
    d["llm_config"].update(hidden_size=256, num_attention_heads=2, num_key_value_heads=1,
                           intermediate_size=512, num_hidden_layers=2)
    d["force_image_size"] = image_size
    return d
