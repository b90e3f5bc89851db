"""Loader: the product package lives in the directory `shiftedproximaloperators.jl_amd/` (a name Python
cannot import directly because of the dot); `import spx_amd` gives that package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shiftedproximaloperators.jl_amd")
_spec = importlib.util.spec_from_file_location("spx_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["spx_amd"] = _mod
_spec.loader.exec_module(_mod)
