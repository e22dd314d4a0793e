"""Import shim: ``import bk_amd`` -> the package in ``bifurcationkit.jl_amd/`` (a directory name with a dot
cannot be imported by name).  Put the repo root on sys.path and use ``from bk_amd import hip``."""
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_root, "bifurcationkit.jl_amd")
_spec = importlib.util.spec_from_file_location("bk_amd", os.path.join(_pkg, "__init__.py"),
                                               submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bk_amd"] = _mod
_spec.loader.exec_module(_mod)
