"""Import shim: the package directory is named `halo2-lasso_amd/` (not a valid Python identifier);
`import halo2_lasso_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "halo2-lasso_amd")
_spec = importlib.util.spec_from_file_location(
    "halo2_lasso_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["halo2_lasso_amd"] = _mod
_spec.loader.exec_module(_mod)
