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


def build_reference_heads(config_dict: dict, head_kwargs: dict, head_state_dict: dict, dtype, pad_token_id):
    """The reference's InternVLChatRewardModeling with its OWN head code (moe_reward.py:213-297) and a backbone
    replaced by a replay stub: ``model.replay(h_r, h_g)`` runs the reference's ``forward`` on hidden states whose
    reward row (last non-pad token, :218-229) and gating row (the ``<|im_end|><|im_start|>assistant\\n`` pattern, :242-243)
    are the given ``[B, hidden]`` tensors.  Used to score OTHER head weights on backbone outputs that an earlier (expensive)
    reference run stored: everything downstream of those two rows is a function of the head weights only.
    The backbone that ``__init__`` constructs is shrunk to one layer per tower (it is never executed)."""
    import torch
    mr = load_reference()
    from internvl2 import InternVLChatModel, InternVLChatConfig

    cd = copy.deepcopy(config_dict)
    cd["vision_config"]["num_hidden_layers"] = 1
    cd["llm_config"]["num_hidden_layers"] = 1
    cd["llm_config"]["vocab_size"] = 128
    cfg = mr.InternVLChatRewardModelingConfig(**copy.deepcopy(cd), **copy.deepcopy(head_kwargs))
    orig = InternVLChatModel.from_pretrained
    InternVLChatModel.from_pretrained = staticmethod(lambda name, *a, **k: InternVLChatModel(InternVLChatConfig(**copy.deepcopy(cd))))
    try:
        model = mr.InternVLChatRewardModeling("synthetic-heads", cfg)
    finally:
        InternVLChatModel.from_pretrained = orig
    head_keys = {k: v for k, v in head_state_dict.items() if not k.startswith("model.")}
    res = model.load_state_dict(head_keys, strict=False)
    assert not res.unexpected_keys and all(k.startswith("model.") for k in res.missing_keys), res
    model.config.pad_token_id = pad_token_id
    model = model.to(dtype).eval()
    pattern = list(mr.token_pattern)
    ids = torch.tensor([[7] + pattern], dtype=torch.long)   # reward row = last index, gating row = 1; no pad id inside
    assert pad_token_id not in ids[0].tolist()

    class _Out:
        pass

    state = {}

    def _backbone(*a, **k):
        o = _Out()
        o.hidden_states = (state["h"],)
        return o

    model.model.forward = _backbone

    def replay(h_r, h_g):
        B, H = h_r.shape
        h = torch.zeros(B, ids.shape[1], H, dtype=h_r.dtype)
        h[:, -1] = h_r
        h[:, 1] = h_g
        state["h"] = h
        with torch.no_grad():
            return model.forward(None, ids.expand(B, -1), None)

    model.replay = replay
    return model


def build_reference_model_phi3(config_dict: dict, head_kwargs: dict, state_dict: dict, dtype, img_context_token_id: int,
                               pad_token_id, gating_pattern):
    """BASELINE configs[4]'s backbone as the REFERENCE would run it: the reference's own ``InternVLChatRewardModeling`` /
    ``InternVLChatModel`` code (vision tower, projector, splice, heads) around transformers' own ``Phi3ForCausalLM`` handed in
    through the ``language_model`` argument (modeling_internvl_chat.py:100,121-122) - the wiring of the upstream InternVL2-4B
    checkpoint, whose Phi-3 branch the reference's dispatch (:125-130) and config class (configuration_internvl_chat.py:50-55)
    lack.  Three shims, all oracle-side: (1) the reference's config class only admits Llama / InternLM2 ``llm_config``s, and
    reads nothing of it but ``hidden_size``: it gets a Llama-typed stand-in of the same hidden size; (2) ``Phi3ForCausalLM`` is
    built from transformers' ``Phi3Config`` with ``attn_implementation='eager'`` (what modeling_internvl_chat.py:114 sets
    without flash-attn); (3) ``moe_reward.token_pattern`` (the InternLM2 tokenizer's ids, moe_reward.py:45-48) is replaced by
    ``gating_pattern`` while this model runs - restore with ``model.restore_token_pattern()``."""
    import torch
    import transformers
    from transformers import Phi3Config, Phi3ForCausalLM
    mr = load_reference()
    from internvl2 import InternVLChatModel, InternVLChatConfig

    cd = copy.deepcopy(config_dict)
    lc = cd["llm_config"]
    assert lc["architectures"][0] == "Phi3ForCausalLM"
    stand_in = dict(architectures=["LlamaForCausalLM"], hidden_size=lc["hidden_size"], intermediate_size=lc["intermediate_size"],
                    num_hidden_layers=1, num_attention_heads=lc["num_attention_heads"], vocab_size=lc["vocab_size"])
    cd_ref = dict(cd, llm_config=stand_in)
    cd_ref.pop("gating_token_pattern", None)
    hf_kwargs = {k: v for k, v in lc.items() if k not in ("architectures", "attn_implementation")}
    phi_cfg = Phi3Config(**copy.deepcopy(hf_kwargs))
    phi_cfg._attn_implementation = "eager"

    def _from_pretrained(name, *a, **k):
        c = InternVLChatConfig(**copy.deepcopy(cd_ref))
        lm = Phi3ForCausalLM(phi_cfg)
        return InternVLChatModel(c, language_model=lm)

    cfg = mr.InternVLChatRewardModelingConfig(**copy.deepcopy(cd_ref), **copy.deepcopy(head_kwargs))
    orig = InternVLChatModel.from_pretrained
    InternVLChatModel.from_pretrained = staticmethod(_from_pretrained)
    try:
        model = mr.InternVLChatRewardModeling("synthetic-phi3", cfg)
    finally:
        InternVLChatModel.from_pretrained = orig
    model.load_state_dict(state_dict, strict=True)
    model.config.pad_token_id = pad_token_id
    model = model.to(dtype)
    model.model.img_context_token_id = img_context_token_id
    model.eval()
    assert model.model.language_model.config._attn_implementation == "eager"
    old = list(mr.token_pattern)
    mr.token_pattern[:] = list(gating_pattern)

    def restore():
        mr.token_pattern[:] = old

    model.restore_token_pattern = restore
    model.transformers_version = transformers.__version__
    return model
