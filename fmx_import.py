"""Registers the package directory `fuzzy-match_amd/` under the importable name `fuzzy_match_amd`."""
import importlib.util
import os
import sys

_NAME = "fuzzy_match_amd"
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzzy-match_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_ROOT, "__init__.py"),
                                                  submodule_search_locations=[_ROOT])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
