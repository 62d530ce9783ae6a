"""MI355X-native sphere-scene path tracer: Python plumbing over the C-ABI (libpt_hip.so) and the C++ host mirror
(libpt_host.so).  The directory name is not an importable identifier; import it through ``dxrs_amd_loader`` at the
repo root, which registers this package as ``dxrs_amd``.

The product path is the HIP library: nothing here falls back to the CPU.  ``Renderer`` raises if libpt_hip.so is
missing or no GPU is present.
"""
import sys as _sys

from . import abi_types  # noqa: F401
# `dxrs_amd.types` is the public name of the ABI struct module; the file is called abi_types.py so that running Python
# from inside this directory does not shadow the standard library's `types`.
types = abi_types
_sys.modules[__name__ + ".types"] = abi_types
from .abi_types import (  # noqa: F401
    PtSphere, PtMaterial, PtCamera, PtSceneData, PtGraphicsSettings, PtRect, PtConfig, PtAccelInfo, PtStats, PtBvhNode,
    SPHERE_DTYPE, MATERIAL_DTYPE, BVH_NODE_DTYPE,
)
from .host import HostLib, load_host  # noqa: F401
from .binding import HipLib, Renderer, PtError, load_hip, hip_library_path  # noqa: F401
