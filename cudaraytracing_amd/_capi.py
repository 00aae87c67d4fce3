"""ctypes binding of lib/libcrt.so (include/crt.h).  No CPU fallback: if the
library is missing or a device call fails, the caller gets an exception."""
import ctypes as C
import os

import numpy as np

from . import build as _build

CRT_OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP = -1, -2, -3   # include/crt.h: crt_status
TRAVERSAL_EXACT = 0      # the default: provably the reference's frame
TRAVERSAL_REFERENCE = 1
TRAVERSAL_FAST = 2       # + distance pruning (measured rate of lost rays, include/crt.h)
FLAG_STATS = 1
FLAG_TILED_OUTPUT = 2
FLAG_FORCE_EXACT = 4
FLAG_TRACE_ALL = 8
FLAG_BOUNDED_RADIANCE = 16
GATHER_AUTO, GATHER_RCCL, GATHER_COPY = 0, 1, 2
INTERSECT_RAW_DIRECTIONS, INTERSECT_FORCE_EXACT, INTERSECT_VISIBILITY = 0x100, 0x200, 0x400
TILE = 8


class CrtError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__("%s failed: %s (%d)%s" % (where, _strerror(status), status, (": " + detail) if detail else ""))
        self.status = status


class BvhNode(C.Structure):
    _fields_ = [("lc", C.c_int32), ("rc", C.c_int32), ("n", C.c_uint32), ("it", C.c_int32),
                ("aa", C.c_float * 3), ("bb", C.c_float * 3)]


class Triangle(C.Structure):
    _fields_ = [("v1", C.c_float * 3), ("v2", C.c_float * 3), ("v3", C.c_float * 3), ("normal", C.c_float * 3),
                ("area", C.c_float), ("area_of_obj", C.c_float), ("material", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ke", C.c_float * 3), ("ns", C.c_float), ("mode", C.c_int32),
                ("has_emit", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("first_tri", C.c_uint32), ("count", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.POINTER(BvhNode)), ("n_nodes", C.c_uint32), ("root", C.c_int32),
                ("tris", C.POINTER(Triangle)), ("n_tris", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("light_tris", C.POINTER(Triangle)), ("n_light_tris", C.c_uint32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("inv_view", C.c_float * 9), ("fov_y", C.c_float)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("p_rr", C.c_float),
                ("light_sample_n", C.c_int32), ("seed", C.c_uint64), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("traversal", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("probe_rays", C.c_uint64),
                ("inner_pops", C.c_uint64), ("leaf_pops", C.c_uint64), ("tri_tests", C.c_uint64), ("hits", C.c_uint64),
                ("stack_sum", C.c_uint64), ("stack_max", C.c_uint64), ("phase_cycles", C.c_uint64 * 24),
                ("kernel_ms", C.c_float), ("logic_ms", C.c_float), ("total_ms", C.c_float),
                ("kernel_launches", C.c_uint32), ("rays_untraced", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "phase_cycles"}
        d["phase_cycles"] = [int(x) for x in self.phase_cycles]
        return d


class MultiInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("gather", C.c_uint32), ("rccl_ranks", C.c_uint32), ("rccl_version", C.c_int32),
                ("render_ms", C.c_float), ("gather_ms", C.c_float), ("frame_ms", C.c_float), ("max_kernel_ms", C.c_float),
                ("bytes_per_rank", C.c_uint64), ("rays", C.c_uint64), ("paths", C.c_uint64), ("rays_untraced", C.c_uint64),
                ("fallback_reason", C.c_char * 160)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["fallback_reason"] = d["fallback_reason"].decode("utf-8", "replace")
        return d


class BvhBuildInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_nodes", C.c_uint32), ("levels", C.c_uint32), ("host_ranges", C.c_uint32),
                ("host_triangles", C.c_uint32), ("host_sorts", C.c_uint32), ("host_sort_elements", C.c_uint64), ("device_ms", C.c_float), ("total_ms", C.c_float), ("host_build_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AccelInfo(C.Structure):
    _fields_ = [("n_leaves", C.c_uint32), ("n_nodes2", C.c_uint32), ("n_nodes4", C.c_uint32), ("depth2", C.c_uint32), ("depth4", C.c_uint32),
                ("sah_on_device", C.c_uint32), ("index_splits", C.c_uint32), ("sah_ms", C.c_float), ("sah_device_ms", C.c_float), ("runtime_init_ms", C.c_float), ("layout_caps", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Task(C.Structure):
    _fields_ = [("n_objs", C.c_uint32), ("obj_path", (C.c_char * 512) * 8), ("mtl_dir", (C.c_char * 512) * 8),
                ("lookat", C.c_float * 3), ("up", C.c_float * 3), ("eye_pos", C.c_float * 3), ("fov_y", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("bvh_thresh_n", C.c_uint32),
                ("light_sample_n", C.c_uint32), ("spp", C.c_uint32), ("p_rr", C.c_float)]


NODE_DTYPE = np.dtype([("lc", "<i4"), ("rc", "<i4"), ("n", "<u4"), ("it", "<i4"), ("aa", "<f4", 3), ("bb", "<f4", 3)])
TRI_DTYPE = np.dtype([("v1", "<f4", 3), ("v2", "<f4", 3), ("v3", "<f4", 3), ("normal", "<f4", 3), ("area", "<f4"),
                      ("area_of_obj", "<f4"), ("material", "<i4")])
MAT_DTYPE = np.dtype([("kd", "<f4", 3), ("ke", "<f4", 3), ("ns", "<f4"), ("mode", "<i4"), ("has_emit", "<i4")])
LIGHT_DTYPE = np.dtype([("first_tri", "<u4"), ("count", "<u4")])

ABI_VERSION = 5  # include/crt.h: CRT_ABI_VERSION

# every symbol include/crt.h declares
EXPORTS = ["crt_strerror", "crt_last_error", "crt_abi_version", "crt_device_count", "crt_scene_create",
           "crt_scene_accel_info", "crt_scene_destroy", "crt_task_obj", "crt_shard_slots", "crt_render", "crt_render_device", "crt_render_range", "crt_render_range_device", "crt_last_launch_ms", "crt_radiance_storage", "crt_preview", "crt_preview_device", "crt_multi_create", "crt_multi_destroy",
           "crt_multi_render", "crt_multi_frame_device", "crt_intersect",
           "crt_device_math", "crt_device_philox", "crt_device_rcp_check", "crt_host_scene_create", "crt_host_scene_destroy",
           "crt_host_scene_add_obj", "crt_host_scene_set_bvh", "crt_host_scene_set_bvh_device", "crt_host_scene_desc", "crt_host_scene_num_objects",
           "crt_host_scene_object", "crt_inverse_view", "crt_task_load", "crt_image_load", "crt_write_png"]

_lib = None


def lib_path():
    return _build.LIB


def lib():
    """Loads libcrt.so (building it first if the sources are newer).  Raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CRT_LIB_PATH") or _build.build_lib()  # (CRT_LIB_PATH: an A/B build of the library, tools/ab.sh)
    L = C.CDLL(path)
    L.crt_abi_version.restype = C.c_int
    if L.crt_abi_version() != ABI_VERSION:
        raise RuntimeError("%s speaks ABI version %d, these bindings version %d (include/crt.h: CRT_ABI_VERSION)" % (path, L.crt_abi_version(), ABI_VERSION))
    L.crt_strerror.restype = C.c_char_p
    L.crt_strerror.argtypes = [C.c_int]
    L.crt_last_error.restype = C.c_char_p
    L.crt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.crt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.crt_scene_accel_info.argtypes = [C.c_void_p, C.POINTER(AccelInfo)]
    L.crt_scene_destroy.argtypes = [C.c_void_p]
    L.crt_shard_slots.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.crt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.crt_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.POINTER(Stats)]
    L.crt_render_range.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.POINTER(Stats)]
    L.crt_render_range_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.crt_last_launch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.crt_radiance_storage.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.crt_preview.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.crt_preview_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.crt_multi_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.crt_multi_destroy.argtypes = [C.c_void_p]
    L.crt_multi_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats),
                                   C.POINTER(MultiInfo)]
    L.crt_multi_frame_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.crt_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.crt_device_math.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.crt_device_philox.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.crt_device_rcp_check.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.crt_host_scene_create.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.crt_host_scene_destroy.argtypes = [C.c_void_p]
    L.crt_host_scene_add_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.crt_host_scene_set_bvh.argtypes = [C.c_void_p, C.c_uint32]
    L.crt_host_scene_set_bvh_device.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(BvhBuildInfo)]
    L.crt_host_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.crt_host_scene_num_objects.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.crt_host_scene_object.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_uint32)]
    L.crt_inverse_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.crt_task_load.argtypes = [C.c_char_p, C.POINTER(Task)]
    L.crt_task_obj.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint32]
    L.crt_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_uint64]
    L.crt_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    _lib = L
    return L


def _strerror(status):
    try:
        return lib().crt_strerror(status).decode()
    except Exception:
        return "status"


def check(status, where):
    if status != CRT_OK:
        raise CrtError(status, where, lib().crt_last_error().decode())


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
