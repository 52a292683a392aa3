"""Import shim: the package directory is named ``biggan-tensorflow_amd`` (not a valid Python
identifier); importing ``biggan_tensorflow_amd`` loads it under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "biggan-tensorflow_amd")
_spec = importlib.util.spec_from_file_location(
    "biggan_tensorflow_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["biggan_tensorflow_amd"] = _mod
_spec.loader.exec_module(_mod)
