"""MI355X-native sphere-scene path tracer: Python plumbing over the C-ABI (libpt_hip.so) and the C++ host mirror
(libpt_host.so).  The directory name is not an importable identifier; import it through ``dxrs_amd_loader`` at the
repo root, which registers this package as ``dxrs_amd``.

The product path is the HIP library: nothing here falls back to the CPU.  ``Renderer`` raises if libpt_hip.so is
missing or no GPU is present.
"""
from .types import (  # noqa: F401
    PtSphere, PtMaterial, PtCamera, PtSceneData, PtGraphicsSettings, PtRect, PtConfig, PtAccelInfo, PtStats, PtBvhNode,
    SPHERE_DTYPE, MATERIAL_DTYPE, BVH_NODE_DTYPE,
)
from .host import HostLib, load_host  # noqa: F401
from .binding import HipLib, Renderer, PtError, load_hip, hip_library_path  # noqa: F401
