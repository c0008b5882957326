"""Import shim: the package directory is `ddsp-pytorch_amd/` (a hyphen is not a
valid module name), so `import ddsp_pytorch_amd` resolves here and this file
replaces itself in `sys.modules` with the real package."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "ddsp-pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "ddsp_pytorch_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ddsp_pytorch_amd"] = _mod
_spec.loader.exec_module(_mod)
