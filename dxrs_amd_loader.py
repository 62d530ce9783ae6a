"""Registers the package directory `directx-raytracing-spheres-demo_amd/` (not a valid Python identifier) as the
importable module ``dxrs_amd``.  Usage: ``import dxrs_amd_loader; import dxrs_amd``."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "directx-raytracing-spheres-demo_amd")

if "dxrs_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("dxrs_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["dxrs_amd"] = _mod
    _spec.loader.exec_module(_mod)

PACKAGE_DIR = _PKG_DIR
