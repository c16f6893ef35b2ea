"""MI355X-native voxelize -> CSG -> JFA hot path of bigmat18/cuda-mesh-voxelization.

Native code: csrc/*.hip -> libvphip.so (C ABI in include/vphip.h).
Python here is harness plumbing only: ctypes binding (capi), torch-backed device buffers
(pipeline), mesh input helpers (mesh), Z-slab multi-GPU driver (slab).
"""
from . import capi, mesh  # noqa: F401

__all__ = ["capi", "mesh"]
