"""Import shim: the package directory is named ``miphei-vit_amd`` (not a valid Python identifier),
so ``import miphei_vit_amd`` resolves here and loads that directory as the package of the same name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miphei-vit_amd")
_spec = importlib.util.spec_from_file_location(
    "miphei_vit_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["miphei_vit_amd"] = _mod
_spec.loader.exec_module(_mod)
