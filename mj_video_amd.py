"""Import alias for the package directory ``mj-video_amd/``.

The package directory name is fixed by the project layout and is not a valid
Python identifier, so ``import mj_video_amd`` resolves to this one-file module,
which loads ``mj-video_amd/__init__.py`` as the package ``mj_video_amd`` and
replaces itself in ``sys.modules`` (sub-modules then import normally, e.g.
``from mj_video_amd.modeling import InternVLChatRewardModeling``).
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mj-video_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
