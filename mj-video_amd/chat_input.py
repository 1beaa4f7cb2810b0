"""Prompt construction for the reward model: ``prepare_chat_input`` and the chat template.

Host-side string logic only.  Behaviour follows
scripts/model/internvl2/modeling_internvl_chat.py:36-89 (``prepare_chat_input``) and the
``internlm2-chat`` / ``phi3-chat`` templates of scripts/model/internvl2/conversation.py:240-249,354-379
(MPT separator style: ``system + sep + (role + message + sep)* + role``).

Quirk kept on purpose: callers never pass ``num_patches_list``, so it defaults
to ``[pixel_values.shape[0]]`` and only the FIRST ``<image>`` placeholder is expanded - into one
contiguous run of ``num_image_token * tiles`` context tokens; the other ``<image>`` strings stay
in the prompt as literal text.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

_SYSTEM_ZH = ("你是由上海人工智能实验室联合商汤科技开发的书生多模态大模型，英文名叫InternVL, "
              "是一个有用无害的人工智能助手。")


@dataclass
class ChatTemplate:
    """One MPT-style (ChatML-like) template; only what the reward path needs."""
    name: str
    system_template: str
    system_message: str
    roles: Tuple[str, str]
    sep: str
    messages: List[Tuple[str, Optional[str]]] = field(default_factory=list)

    def copy(self) -> "ChatTemplate":
        return ChatTemplate(self.name, self.system_template, self.system_message, self.roles, self.sep,
                            [tuple(m) for m in self.messages])

    def append_message(self, role: str, message: Optional[str]) -> None:
        self.messages.append((role, message))

    def get_prompt(self) -> str:
        out = self.system_template.format(system_message=self.system_message) + self.sep
        for role, message in self.messages:
            if message:
                if isinstance(message, tuple):
                    message = message[0]
                out += role + message + self.sep
            else:
                out += role
        return out


_TEMPLATES = {
    "internlm2-chat": ChatTemplate("internlm2-chat", "<|im_start|>system\n{system_message}", _SYSTEM_ZH,
                                   ("<|im_start|>user\n", "<|im_start|>assistant\n"), "<|im_end|>"),
    "Hermes-2": ChatTemplate("Hermes-2", "<|im_start|>system\n{system_message}", _SYSTEM_ZH,
                             ("<|im_start|>user\n", "<|im_start|>assistant\n"), "<|im_end|>"),
    # conversation.py:368-379 - the template of the Phi-3 based InternVL2-4B (BASELINE configs[4])
    "phi3-chat": ChatTemplate("phi3-chat", "<|system|>\n{system_message}", _SYSTEM_ZH,
                              ("<|user|>\n", "<|assistant|>\n"), "<|end|>"),
}


def get_conv_template(name: str) -> ChatTemplate:
    if name not in _TEMPLATES:
        raise KeyError(f"unknown conversation template '{name}' (available: {sorted(_TEMPLATES)})")
    return _TEMPLATES[name].copy()


def num_image_tokens_per_tile(config) -> int:
    """(image_size // patch)^2 * downsample^2  (modeling_internvl_chat.py:70-72)."""
    image_size = config.force_image_size or config.vision_config.image_size
    patch = config.vision_config.patch_size
    return int((image_size // patch) ** 2 * (config.downsample_ratio ** 2))


def build_query(config, question: str, n_tiles: Optional[int], history=None,
                num_patches_list: Optional[Sequence[int]] = None, IMG_START_TOKEN="<img>",
                IMG_END_TOKEN="</img>", IMG_CONTEXT_TOKEN="<IMG_CONTEXT>") -> str:
    """The prompt string that gets tokenised (everything of prepare_chat_input but the tokenizer)."""
    has_pixels = n_tiles is not None
    if history is None and has_pixels and "<image>" not in question:
        question = "<image>\n" + question
    if num_patches_list is None:
        num_patches_list = [n_tiles] if has_pixels else []
    assert not has_pixels or n_tiles == sum(num_patches_list)

    template = get_conv_template(config.template)
    for old_q, old_a in (history or []):
        template.append_message(template.roles[0], old_q)
        template.append_message(template.roles[1], old_a)
    template.append_message(template.roles[0], question)
    template.append_message(template.roles[1], None)
    query = template.get_prompt()

    per_tile = num_image_tokens_per_tile(config)
    for n in num_patches_list:
        image_tokens = IMG_START_TOKEN + IMG_CONTEXT_TOKEN * per_tile * n + IMG_END_TOKEN
        query = query.replace("<image>", image_tokens, 1)
    return query


def prepare_chat_input(config, tokenizer, pixel_values, question, generation_config, history=None,
                       return_history=False, num_patches_list=None, IMG_START_TOKEN="<img>",
                       IMG_END_TOKEN="</img>", IMG_CONTEXT_TOKEN="<IMG_CONTEXT>", verbose=False, device="cpu"):
    """Drop-in for the reference's ``prepare_chat_input`` (same signature, same side effect).

    ``tokenizer`` is duck-typed: ``convert_tokens_to_ids(str)`` and
    ``__call__(str, return_tensors='pt') -> {'input_ids', 'attention_mask'}``.
    Returns ``(input_ids[1, N], attention_mask[1, N])`` on ``device`` and sets
    ``generation_config['eos_token_id']`` to the id of the template separator.
    """
    n_tiles = None if pixel_values is None else int(pixel_values.shape[0])
    sep = get_conv_template(config.template).sep
    eos_token_id = tokenizer.convert_tokens_to_ids(sep.strip())
    query = build_query(config, question, n_tiles, history=history, num_patches_list=num_patches_list,
                        IMG_START_TOKEN=IMG_START_TOKEN, IMG_END_TOKEN=IMG_END_TOKEN,
                        IMG_CONTEXT_TOKEN=IMG_CONTEXT_TOKEN)
    model_inputs = tokenizer(query, return_tensors="pt")
    input_ids = model_inputs["input_ids"].to(device)
    attention_mask = model_inputs["attention_mask"].to(device)
    generation_config["eos_token_id"] = eos_token_id
    return input_ids, attention_mask


def video_prefix(num_frames: int) -> str:
    """``Frame1: <image>\\n ... FrameF: <image>\\n`` (eval_genai_mjvideo.py:132,135)."""
    return "".join(f"Frame{i + 1}: <image>\n" for i in range(num_frames))
