"""Import alias: ``import mscs_amd`` loads the package that lives in
``eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd/`` (a directory name that
is not a valid Python identifier)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    "mscs_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mscs_amd"] = _mod
_spec.loader.exec_module(_mod)
