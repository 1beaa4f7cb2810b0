"""Drop-in for the reference's ``scripts/data_processor`` package (``from data_processor import load_video``,
README.md:73).  VideoDataset / VideoDataCollator are training-side plumbing and out of scope."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from mj_video_amd.video import load_frames, load_video  # noqa: E402,F401
