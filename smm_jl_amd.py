"""Import alias: the package directory is named `smm.jl_amd/` (not a valid dotted module
name), so `import smm_jl_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "smm.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "smm_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["smm_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
