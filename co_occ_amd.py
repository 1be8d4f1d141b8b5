"""Import shim: the package lives in ``co-occ_amd/`` (a directory name Python cannot import
directly); ``import co_occ_amd`` loads it from there under this importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "co-occ_amd")
_spec = importlib.util.spec_from_file_location("co_occ_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["co_occ_amd"] = _mod
_spec.loader.exec_module(_mod)
