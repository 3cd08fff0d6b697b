"""Import alias: the package directory is named ``co-snarks_amd`` (hyphen), which is not a Python
identifier; ``import cosnarks_amd`` loads that package under this name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location(
    "cosnarks_amd", os.path.join(_here, "co-snarks_amd", "__init__.py"),
    submodule_search_locations=[os.path.join(_here, "co-snarks_amd")])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cosnarks_amd"] = _mod
_spec.loader.exec_module(_mod)
