"""Import alias: the package directory is `genie-smem_amd/` (not a valid Python identifier),
so `import genie_smem_amd` loads that directory as the package `genie_smem_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genie-smem_amd")
_spec = importlib.util.spec_from_file_location(
    "genie_smem_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["genie_smem_amd"] = _mod
_spec.loader.exec_module(_mod)
