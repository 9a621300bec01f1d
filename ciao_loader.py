"""Loads the product package `ciaoalgorithms.jl_amd/` (a directory name with a dot) as `ciaoalgorithms_jl_amd`."""
import importlib.util
import os
import sys

ALIAS = "ciaoalgorithms_jl_amd"
_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "ciaoalgorithms.jl_amd")


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
