"""MI355X-native reward scoring for MJ-VIDEO (package directory ``mj-video_amd``; import as ``mj_video_amd``).

Public surface = the reference's (scripts/model/__init__.py:1-2, scripts/data_processor/__init__.py:2):
``InternVLChatRewardModeling``, ``InternVLChatRewardModelingConfig``, ``prepare_chat_input``, ``load_video``.
Heavy sub-modules are imported lazily so that configuration / prompt logic works without the HIP library.
"""
from .configuration import (InternLM2Config, InternVisionConfig, InternVLChatConfig,  # noqa: F401
                            InternVLChatRewardModelingConfig)
from .chat_input import get_conv_template, prepare_chat_input  # noqa: F401

_LAZY = {
    "InternVLChatRewardModeling": ("modeling", "InternVLChatRewardModeling"),
    "InternVLChatModel": ("modeling", "InternVLChatModel"),
    "CustomOutput": ("modeling", "CustomOutput"),
    "load_video": ("video", "load_video"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
