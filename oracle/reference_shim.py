"""ORACLE SUPPORT - build-container only; never shipped to the GPU box, never imported by the product.

Imports the REAL reference (``/root/reference/scripts/model``) unmodified, with the five
compatibility patches SURVEY.md §8(c) lists for transformers 5 / missing timm, so that
tests/golden/make_golden.py can run the reference's own ``InternVLChatRewardModeling.forward``
on CPU and (a) prove ``oracle/ref_cpu.py`` reproduces it, (b) emit golden vectors.
"""
from __future__ import annotations

import copy
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MJV_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "scripts", "model", "moe_reward.py"))


_loaded = {}


def load_reference():
    """Returns the reference's ``moe_reward`` module (cached)."""
    if "mod" in _loaded:
        return _loaded["mod"]
    if not reference_available():
        raise RuntimeError(f"reference not present under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # the reference tree is read-only
    import torch
    import torch.distributed as dist
    import transformers  # must come before the timm stub (its find_spec('timm') probe)
    from torch import nn

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        layers.DropPath = nn.Identity
        timm.models, models.layers = models, layers
        sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    import transformers.models.llama.modeling_llama as ml
    if not hasattr(ml, "LLAMA_INPUTS_DOCSTRING"):
        ml.LLAMA_INPUTS_DOCSTRING = ""
    for p in (os.path.join(REFERENCE_ROOT, "scripts", "model"), os.path.join(REFERENCE_ROOT, "scripts")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if not dist.is_initialized():  # modeling_internvl_chat.py:172 calls get_rank()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        dist.init_process_group("gloo", world_size=1, rank=0)
    import moe_reward  # noqa: E402  (the reference's module)
    from internvl2 import InternVLChatConfig
    InternVLChatConfig.has_no_defaults_at_init = True
    _loaded["mod"] = moe_reward
    return moe_reward


def build_reference_model(config_dict: dict, head_kwargs: dict, state_dict: dict, dtype, img_context_token_id: int,
                          pad_token_id):
    """Constructs the reference's InternVLChatRewardModeling from a config dict + checkpoint-layout
    state dict, following the eval driver's set-up order (eval_genai_mjvideo.py:73-116)."""
    import torch
    mr = load_reference()
    from internvl2 import InternVLChatModel

    cfg = mr.InternVLChatRewardModelingConfig(**copy.deepcopy(config_dict), **copy.deepcopy(head_kwargs))
    base_cfg_holder = {}

    def _from_pretrained(name, *a, **k):  # patch 5: config -> module, no hub, no post_init
        from internvl2 import InternVLChatConfig
        c = InternVLChatConfig(**copy.deepcopy(config_dict))
        base_cfg_holder["cfg"] = c
        return InternVLChatModel(c)

    orig = InternVLChatModel.from_pretrained
    InternVLChatModel.from_pretrained = staticmethod(_from_pretrained)
    try:
        model = mr.InternVLChatRewardModeling("synthetic", cfg)
    finally:
        InternVLChatModel.from_pretrained = orig
    missing = model.load_state_dict(state_dict, strict=True)
    model.config.pad_token_id = pad_token_id
    model = model.to(dtype)
    model.model.img_context_token_id = img_context_token_id
    model.eval()
    return model
