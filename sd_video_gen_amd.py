"""Import shim: the package directory is named ``sd-video-gen_amd`` (not a legal
Python identifier), so ``import sd_video_gen_amd`` resolves through this loader."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sd-video-gen_amd")
_spec = importlib.util.spec_from_file_location(
    "sd_video_gen_amd", os.path.join(_DIR, "__init__.py"),
    submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sd_video_gen_amd"] = _mod
_spec.loader.exec_module(_mod)
