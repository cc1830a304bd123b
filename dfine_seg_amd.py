"""Import shim: the package lives in the directory ``d-fine-seg_amd/`` (the name the project
layout prescribes), which is not a valid Python identifier. ``import dfine_seg_amd`` loads
that directory as a regular package under this name."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "d-fine-seg_amd")
_spec = importlib.util.spec_from_file_location(
    "dfine_seg_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
)
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["dfine_seg_amd"] = _pkg
_spec.loader.exec_module(_pkg)
