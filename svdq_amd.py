"""
Import alias: ``import svdq_amd`` loads the package that lives in the directory
``svd-quantization-task-merging_amd/`` (that name is not a valid Python identifier, so it
cannot be imported directly).  After import, ``sys.modules["svdq_amd"]`` IS that package:
``from svdq_amd.rtvq import RTVQQuantizer`` etc. work as usual.
"""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "svd-quantization-task-merging_amd")
_spec = importlib.util.spec_from_file_location("svdq_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["svdq_amd"] = _mod
_spec.loader.exec_module(_mod)
