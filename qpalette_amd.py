"""Import shim: the package directory is named ``q-palette_amd`` (not a valid Python identifier), so
``import qpalette_amd`` loads it from there under this importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "q-palette_amd")
_spec = importlib.util.spec_from_file_location(
    "qpalette_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["qpalette_amd"] = _mod
_spec.loader.exec_module(_mod)
