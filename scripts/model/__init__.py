"""Drop-in for the reference's ``scripts/model`` package (its __init__.py:1-2 exports these three names):
put ``<repo>/scripts`` on sys.path and ``from model import ...`` resolves to the MI355X implementation."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from mj_video_amd import InternVLChatRewardModelingConfig, prepare_chat_input  # noqa: E402,F401


def __getattr__(name):
    if name in ("InternVLChatRewardModeling", "CustomOutput", "InternVLChatModel"):
        import mj_video_amd.modeling as m
        return getattr(m, name)
    raise AttributeError(name)
