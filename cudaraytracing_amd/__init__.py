"""cudaraytracing_amd -- MI355X-native Monte-Carlo path tracer with the
config.json / Scene / Render surface of guomc9/CudaRayTracing.

The package is a thin ctypes layer over lib/libcrt.so (C ABI: include/crt.h),
which holds the C++ host side (OBJ/MTL loader, median-split BVH, flat scene
export) and the hand-written HIP kernels for gfx950.
"""
from .api import (MultiRender, Render, Scene, Task, device_count, device_math, device_philox, device_rcp_check, fov_to_radians,
                  get_inverse_view_matrix, image_load, shard_slots)
from ._capi import (FLAG_BOUNDED_RADIANCE, FLAG_FORCE_EXACT, FLAG_STATS, FLAG_TILED_OUTPUT, FLAG_TRACE_ALL, TRAVERSAL_FAST, TRAVERSAL_REFERENCE, TRAVERSAL_EXACT, GATHER_AUTO, GATHER_RCCL,
                    GATHER_COPY, INTERSECT_RAW_DIRECTIONS, INTERSECT_FORCE_EXACT, INTERSECT_VISIBILITY, CrtError)

__all__ = ["MultiRender", "GATHER_AUTO", "GATHER_RCCL", "GATHER_COPY", "INTERSECT_RAW_DIRECTIONS", "INTERSECT_FORCE_EXACT", "INTERSECT_VISIBILITY", "Render", "Scene", "Task", "device_count", "device_math", "device_philox", "device_rcp_check", "fov_to_radians",
           "get_inverse_view_matrix", "image_load", "shard_slots", "FLAG_STATS", "FLAG_TILED_OUTPUT", "FLAG_FORCE_EXACT", "FLAG_TRACE_ALL", "FLAG_BOUNDED_RADIANCE", "TRAVERSAL_FAST",
           "TRAVERSAL_REFERENCE", "TRAVERSAL_EXACT", "CrtError"]
