"""Import shim: the package directory is named `point-unet_amd/` (not an importable identifier), so
`import point_unet_amd` loads that directory as the package `point_unet_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "point-unet_amd")
_spec = importlib.util.spec_from_file_location("point_unet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["point_unet_amd"] = _mod
_spec.loader.exec_module(_mod)
