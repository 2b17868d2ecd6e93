"""
Import shim: the package directory is named `uni-slam_amd/` (not a Python identifier), so this module loads it
under the importable name `unislam_amd` and replaces itself in sys.modules.

    import unislam_amd as us
    from unislam_amd.decoders import Decoders
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "uni-slam_amd")
_spec = importlib.util.spec_from_file_location("unislam_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["unislam_amd"] = _mod
_spec.loader.exec_module(_mod)
