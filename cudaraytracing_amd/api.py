"""Python mirror of the reference's Scene / Render / Task interface, a thin
layer over the C ABI (include/crt.h).  All work happens in libcrt.so: OBJ/MTL
ingestion and the BVH build in its C++ host layer, rendering in its HIP
kernels.  There is no Python or CPU rendering path.

reference: src/main.cu:40-90 (Task/config), include/Scene.h, include/Camera.h,
include/Render.cuh:357-557.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _capi as capi


def get_inverse_view_matrix(eye_pos, lookat, up):
    """reference: include/Camera.h:9-36.  Returns 9 floats, column-major."""
    e = np.asarray(eye_pos, dtype=np.float32)
    l = np.asarray(lookat, dtype=np.float32)
    u = np.asarray(up, dtype=np.float32)
    out = np.zeros(9, dtype=np.float32)
    capi.check(capi.lib().crt_inverse_view(capi.ptr(e), capi.ptr(l), capi.ptr(u), capi.ptr(out)), "crt_inverse_view")
    return out


def fov_to_radians(fov_y_deg):
    """reference: src/main.cu:278  `task.fov_y * (float)M_PI / 180` in float."""
    return np.float32(np.float32(np.float32(fov_y_deg) * np.float32(math.pi)) / np.float32(180))


class Task:
    """reference: struct Task + config_task(), src/main.cu:40-90."""

    def __init__(self, config_path, base_dir=None):
        t = capi.Task()
        capi.check(capi.lib().crt_task_load(os.fsencode(config_path), C.byref(t)), "crt_task_load")
        base = base_dir if base_dir is not None else os.getcwd()

        def res(p):
            p = p.decode()
            return p if os.path.isabs(p) else os.path.join(base, p)

        if t.n_objs <= 8:
            self.OBJ_paths = [(res(t.obj_path[i].value), res(t.mtl_dir[i].value)) for i in range(t.n_objs)]
        else:   # (any number of OBJ files, as src/main.cu:74-78 loops over them: the POD holds eight, crt_task_obj returns each)
            self.OBJ_paths = []
            for i in range(t.n_objs):
                o, m = C.create_string_buffer(4096), C.create_string_buffer(4096)
                capi.check(capi.lib().crt_task_obj(os.fsencode(config_path), i, o, m, 4096), "crt_task_obj")
                self.OBJ_paths.append((res(o.value), res(m.value)))
        self.lookat = np.array(t.lookat[:], dtype=np.float32)
        self.up = np.array(t.up[:], dtype=np.float32)
        self.eye_pos = np.array(t.eye_pos[:], dtype=np.float32)
        self.fov_y = np.float32(t.fov_y)
        self.width, self.height = int(t.width), int(t.height)
        self.bvh_thresh_n = int(t.bvh_thresh_n)
        self.light_sample_n = int(t.light_sample_n)
        self.P_RR = np.float32(t.p_rr)
        self.spp = int(t.spp)


class Scene:
    """reference: include/Scene.h:16-102 fed by Loader/Object (src/main.cu:122-145)."""

    def __init__(self, width, height):
        self._h = C.c_void_p()
        capi.check(capi.lib().crt_host_scene_create(width, height, C.byref(self._h)), "crt_host_scene_create")
        self.width, self.height = int(width), int(height)
        self._desc = None

    def add_obj(self, obj_path, mtl_dir):
        capi.check(capi.lib().crt_host_scene_add_obj(self._h, os.fsencode(obj_path), os.fsencode(mtl_dir)),
                   "crt_host_scene_add_obj")
        self._desc = None

    def set_BVH(self, thresh_n, device=None):
        """Scene::set_BVH (Scene.h:50-54).  device=None: the host builder; device=k: the same tree built on GPU k
        (byte-identical arrays, csrc/crt_bvh_build.hip), self.bvh_build_info says how it went."""
        if device is None:
            capi.check(capi.lib().crt_host_scene_set_bvh(self._h, thresh_n), "crt_host_scene_set_bvh")
            self.bvh_build_info = None
        else:
            info = capi.BvhBuildInfo()
            capi.check(capi.lib().crt_host_scene_set_bvh_device(self._h, thresh_n, device, C.byref(info)), "crt_host_scene_set_bvh_device")
            self.bvh_build_info = info.as_dict()
        self._desc = None

    def desc(self):
        if self._desc is None:
            d = capi.SceneDesc()
            capi.check(capi.lib().crt_host_scene_desc(self._h, C.byref(d)), "crt_host_scene_desc")
            self._desc = d
        return self._desc

    def _arr(self, p, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * dtype.itemsize)).from_address(C.addressof(p.contents))
        return np.frombuffer(buf, dtype=dtype).copy()

    def nodes(self):
        d = self.desc()
        return self._arr(d.nodes, d.n_nodes, capi.NODE_DTYPE)

    def triangles(self):
        d = self.desc()
        return self._arr(d.tris, d.n_tris, capi.TRI_DTYPE)

    def light_triangles(self):
        d = self.desc()
        return self._arr(d.light_tris, d.n_light_tris, capi.TRI_DTYPE)

    def materials(self):
        d = self.desc()
        return self._arr(d.materials, d.n_materials, capi.MAT_DTYPE)

    def lights(self):
        d = self.desc()
        return self._arr(d.lights, d.n_lights, capi.LIGHT_DTYPE)

    @property
    def root(self):
        return int(self.desc().root)

    def objects(self):
        n = C.c_uint32()
        capi.check(capi.lib().crt_host_scene_num_objects(self._h, C.byref(n)), "crt_host_scene_num_objects")
        out = []
        for i in range(n.value):
            a, l, c = C.c_float(), C.c_int32(), C.c_uint32()
            capi.check(capi.lib().crt_host_scene_object(self._h, i, C.byref(a), C.byref(l), C.byref(c)),
                       "crt_host_scene_object")
            out.append((bool(l.value), float(a.value), int(c.value)))
        return out

    def free(self):
        if self._h:
            capi.lib().crt_host_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @classmethod
    def from_task(cls, task, width=None, height=None, bvh_device=None):
        """render_view()'s scene set-up (src/main.cu:119-145,276); bvh_device=k builds the BVH on GPU k (same arrays)."""
        s = cls(width or task.width, height or task.height)
        for obj, mtl in task.OBJ_paths:
            s.add_obj(obj, mtl)
        s.set_BVH(task.bvh_thresh_n, device=bvh_device)
        return s


class Render:
    """reference: class Render, include/Render.cuh:357-557.

    run_view renders on the GPU through crt_render (host buffers) or
    crt_render_device (device buffers, used by the multi-GPU path)."""

    def __init__(self, scene, spp=16, P_RR=0.8, light_sample_n=1, device=0):
        self.scene = scene
        self.spp, self.P_RR, self.light_sample_n = int(spp), np.float32(P_RR), int(light_sample_n)
        self.seed = 0
        self.traversal = capi.TRAVERSAL_EXACT
        self.extra_flags = 0  # e.g. FLAG_FORCE_EXACT (test hook)
        self.device = device
        self._h = C.c_void_p()
        capi.check(capi.lib().crt_scene_create(C.byref(scene.desc()), device, C.byref(self._h)), "crt_scene_create")
        self.frame_buffer = None
        self.mean_buffer = None
        self.stats = None

    def _handle(self, what):
        """The crt_scene* of this renderer; raises when there is none (freed, or a MultiRender, whose handle is a crt_multi*)."""
        if not self._h:
            raise RuntimeError("%s: no single-device scene handle (freed, or a MultiRender)" % what)
        return self._h

    def accel_info(self):
        """How the acceleration trees of the FAST traversal were built (crt_scene_accel_info)."""
        self._handle("accel_info")
        a = capi.AccelInfo()
        capi.check(capi.lib().crt_scene_accel_info(self._h, C.byref(a)), "crt_scene_accel_info")
        return a.as_dict()

    def set_spp(self, spp):
        self.spp = int(spp)

    def set_P_RR(self, p):
        self.P_RR = np.float32(p)

    def set_light_sample_n(self, n):
        self.light_sample_n = int(n)

    def _cam(self, eye_pos, inv_view_mat, fovY):
        cam = capi.Camera()
        cam.eye[:] = [float(v) for v in np.asarray(eye_pos, dtype=np.float32)]
        cam.inv_view[:] = [float(v) for v in np.asarray(inv_view_mat, dtype=np.float32).reshape(9)]
        cam.fov_y = float(np.float32(fovY))
        return cam

    def _params(self, rank=0, world=1, flags=0, width=None, height=None):
        return capi.Params(width or self.scene.width, height or self.scene.height, self.spp, float(self.P_RR),
                           self.light_sample_n, self.seed, rank, world, self.traversal, flags)

    def run_view(self, eye_pos, inv_view_mat, fovY, stats=False, want_mean=True, width=None, height=None):
        """Renders the whole frame; returns the RGB8 frame buffer (H, W, 3)."""
        if not self._h:
            raise RuntimeError("Render.run_view after free()")
        cam = self._cam(eye_pos, inv_view_mat, fovY)
        prm = self._params(flags=(capi.FLAG_STATS if stats else 0) | self.extra_flags, width=width, height=height)
        w, h = prm.width, prm.height
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        mean = np.zeros((h, w, 3), dtype=np.float32) if want_mean else None
        st = capi.Stats()
        capi.check(capi.lib().crt_render(self._h, C.byref(cam), C.byref(prm), capi.ptr(rgb), capi.ptr(mean),
                                         C.byref(st)), "crt_render")
        self.frame_buffer, self.mean_buffer, self.stats = rgb, mean, st.as_dict()
        return rgb

    def run_view_range(self, eye_pos, inv_view_mat, fovY, sample_begin, sample_count, want_mean=True, width=None, height=None):
        """Progressive rendering: adds samples [sample_begin, sample_begin + sample_count) of the spp samples per pixel to the
        accumulator of the device scene; ranges go in ascending order from 0.  Returns the RGB8 frame once the range that ends
        at spp has been rendered (bit-identical to run_view), None before."""
        if not self._h:
            raise RuntimeError("Render.run_view_range after free()")
        cam = self._cam(eye_pos, inv_view_mat, fovY)
        prm = self._params(flags=self.extra_flags, width=width, height=height)
        w, h = prm.width, prm.height
        last = sample_begin + sample_count == self.spp
        rgb = np.zeros((h, w, 3), dtype=np.uint8) if last else None
        mean = np.zeros((h, w, 3), dtype=np.float32) if (last and want_mean) else None
        st = capi.Stats()
        capi.check(capi.lib().crt_render_range(self._h, C.byref(cam), C.byref(prm), int(sample_begin), int(sample_count),
                                               capi.ptr(rgb) if last else None, capi.ptr(mean), C.byref(st)), "crt_render_range")
        self.stats = st.as_dict()
        if last:
            self.frame_buffer, self.mean_buffer = rgb, mean
        return rgb

    def last_launch_ms(self):
        """(device ms, launches) of the render kernel of the last frame submitted on this handle (crt_last_launch_ms); the frame's
        stream must have been synchronized."""
        ms, n = C.c_float(), C.c_uint32()
        capi.check(capi.lib().crt_last_launch_ms(self._handle("last_launch_ms"), C.byref(ms), C.byref(n)), "crt_last_launch_ms")
        return float(ms.value), int(n.value)

    def radiance_storage(self):
        """(bytes, ring samples) of the per-path radiance storage of the last render on this handle (crt_radiance_storage): one value per
        path of a chunk, or -- FLAG_BOUNDED_RADIANCE -- a ring of that many samples."""
        b, r = C.c_uint64(), C.c_uint32()
        capi.check(capi.lib().crt_radiance_storage(self._handle("radiance_storage"), C.byref(b), C.byref(r)), "crt_radiance_storage")
        return int(b.value), int(r.value)

    def preview(self, want_mean=False, width=None, height=None):
        """Displayable frame of the progressive render in flight (crt_preview): returns (rgb (H, W, 3), mean or None, samples done).
        Reads the accumulator only -- the final frame does not depend on previews."""
        if not self._h:
            raise RuntimeError("Render.preview after free()")
        w, h = width or self.scene.width, height or self.scene.height
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        mean = np.zeros((h, w, 3), dtype=np.float32) if want_mean else None
        done = C.c_uint32(0)
        capi.check(capi.lib().crt_preview(self._h, capi.ptr(rgb), capi.ptr(mean), C.byref(done)), "crt_preview")
        return rgb, mean, int(done.value)

    def run_view_device(self, eye_pos, inv_view_mat, fovY, d_rgb_ptr, d_mean_ptr=None, stream=None, rank=0, world=1,
                        tiled=False, want_stats=True, width=None, height=None):
        """Enqueues a render whose outputs stay in device memory (raw device pointers)."""
        self._handle("run_view_device")
        cam = self._cam(eye_pos, inv_view_mat, fovY)
        flags = (capi.FLAG_TILED_OUTPUT if (tiled or world > 1) else 0) | self.extra_flags
        prm = self._params(rank=rank, world=world, flags=flags, width=width, height=height)
        st = capi.Stats()
        capi.check(capi.lib().crt_render_device(self._h, C.byref(cam), C.byref(prm), C.c_void_p(d_rgb_ptr),
                                                C.c_void_p(d_mean_ptr) if d_mean_ptr else None,
                                                C.c_void_p(stream) if stream else None,
                                                C.byref(st) if want_stats else None), "crt_render_device")
        if want_stats:
            self.stats = st.as_dict()
        return self.stats

    def intersect(self, origins, dirs, traversal=None):
        self._handle("intersect")
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        n = o.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        capi.check(capi.lib().crt_intersect(self._h, n, capi.ptr(o), capi.ptr(d),
                                            self.traversal if traversal is None else traversal, capi.ptr(tri),
                                            capi.ptr(t)), "crt_intersect")
        return tri, t

    def blocked(self, origins, dirs, limits, traversal=None):
        """blocked() of Render.cuh:19-27 for n visibility rays with t_to_light = limits: (blocked, blocking triangle or -1)"""
        self._handle("blocked")
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        n = o.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        t = np.array(limits, dtype=np.float32).reshape(n).copy()
        capi.check(capi.lib().crt_intersect(self._h, n, capi.ptr(o), capi.ptr(d),
                                            (self.traversal if traversal is None else traversal) | capi.INTERSECT_VISIBILITY,
                                            capi.ptr(tri), capi.ptr(t)), "crt_intersect")
        return t != 0.0, tri

    def save_frame_buffer(self, save_path):
        if self.frame_buffer is None:
            raise RuntimeError("save_frame_buffer before run_view")
        h, w, _ = self.frame_buffer.shape
        capi.check(capi.lib().crt_write_png(os.fsencode(save_path), w, h, capi.ptr(self.frame_buffer)), "crt_write_png")

    def get_frame_buffer(self):
        return self.frame_buffer

    def free(self):
        if self._h:
            capi.lib().crt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MultiRender(Render):
    """Render over several devices of one node in ONE process (include/crt.h, crt_multi): one replica of the scene per entry
    of `devices`, interleaved pixel-tile shards, one RCCL all-gather per frame.  The reference stops at device 0
    (src/main.cu:92-105)."""

    def __init__(self, scene, spp=16, P_RR=0.8, light_sample_n=1, devices=(0,), gather=capi.GATHER_AUTO):
        self.scene = scene
        self.spp, self.P_RR, self.light_sample_n = int(spp), np.float32(P_RR), int(light_sample_n)
        self.seed = 0
        self.traversal = capi.TRAVERSAL_EXACT
        self.extra_flags = 0
        self.devices = [int(d) for d in devices]
        self.device = self.devices[0]
        self._h = C.c_void_p()   # stays null: every inherited method that takes a crt_scene* fails its own check
        self._mh = C.c_void_p()  # the crt_multi*
        devs = (C.c_int * len(self.devices))(*self.devices)
        capi.check(capi.lib().crt_multi_create(C.byref(scene.desc()), devs, len(self.devices), gather, C.byref(self._mh)),
                   "crt_multi_create")
        self.frame_buffer = None
        self.mean_buffer = None
        self.stats = None
        self.rank_stats = None
        self.info = None

    def run_view(self, eye_pos, inv_view_mat, fovY, stats=False, want_mean=True, width=None, height=None, to_host=True):
        if not self._mh:
            raise RuntimeError("MultiRender.run_view after free()")
        cam = self._cam(eye_pos, inv_view_mat, fovY)
        prm = self._params(flags=(capi.FLAG_STATS if stats else 0) | self.extra_flags, width=width, height=height)
        w, h = prm.width, prm.height
        rgb = np.zeros((h, w, 3), dtype=np.uint8) if to_host else None
        mean = np.zeros((h, w, 3), dtype=np.float32) if (want_mean and to_host) else None
        st = (capi.Stats * len(self.devices))()
        info = capi.MultiInfo()
        capi.check(capi.lib().crt_multi_render(self._mh, C.byref(cam), C.byref(prm), capi.ptr(rgb), capi.ptr(mean), st,
                                               C.byref(info)), "crt_multi_render")
        self.rank_stats = [s.as_dict() for s in st]
        self.info = info.as_dict()
        tot = dict(self.rank_stats[0])
        for o in self.rank_stats[1:]:
            for k in ("paths", "rays", "shadow_rays", "probe_rays", "rays_untraced", "inner_pops", "leaf_pops", "tri_tests", "hits"):
                tot[k] += o[k]
            for k in ("kernel_ms", "total_ms"):
                tot[k] = max(tot[k], o[k])
        self.stats = tot
        if to_host:
            self.frame_buffer, self.mean_buffer = rgb, mean
        return rgb

    def run_view_range(self, *a, **k):
        raise NotImplementedError("progressive ranges are a single-device interface (crt_render_range)")

    def preview(self, *a, **k):
        raise NotImplementedError("previews are a single-device interface (crt_preview)")

    def accel_info(self):
        raise NotImplementedError("crt_scene_accel_info is a single-device interface")

    def run_view_device(self, *a, **k):
        raise NotImplementedError("MultiRender owns its device buffers (crt_multi_frame_device)")

    def intersect(self, *a, **k):
        raise NotImplementedError("crt_intersect is a single-device interface")

    def blocked(self, *a, **k):
        raise NotImplementedError("crt_intersect is a single-device interface")

    def last_launch_ms(self):
        raise NotImplementedError("crt_last_launch_ms is a single-device interface (per-rank kernel times: rank_stats)")

    def free(self):
        if self._mh:
            capi.lib().crt_multi_destroy(self._mh)
            self._mh = C.c_void_p()


def image_load(path):
    """(x, y, comp, samples (y, x, comp) uint8) of a texture file as the reference's stbi_load(path, &x, &y, &comp, 0) returns them
    (Loader.h:58): PNG, BMP, TGA."""
    x, y, comp = C.c_int32(), C.c_int32(), C.c_int32()
    capi.check(capi.lib().crt_image_load(os.fsencode(path), C.byref(x), C.byref(y), C.byref(comp), None, 0), "crt_image_load")
    a = np.zeros((y.value, x.value, comp.value), dtype=np.uint8)
    capi.check(capi.lib().crt_image_load(os.fsencode(path), C.byref(x), C.byref(y), C.byref(comp), capi.ptr(a), a.size), "crt_image_load")
    return x.value, y.value, comp.value, a


def shard_slots(width, height, rank, world):
    n = C.c_uint64()
    capi.check(capi.lib().crt_shard_slots(width, height, rank, world, C.byref(n)), "crt_shard_slots")
    return int(n.value)


def device_count():
    n = C.c_int()
    capi.check(capi.lib().crt_device_count(C.byref(n)), "crt_device_count")
    return n.value


def device_math(fn, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float32)
    bb = np.ascontiguousarray(b, dtype=np.float32) if b is not None else None
    out = np.zeros_like(a)
    capi.check(capi.lib().crt_device_math(device, fn.encode(), a.size, capi.ptr(a), capi.ptr(bb), capi.ptr(out)),
               "crt_device_math")
    return out


def device_rcp_check(device=0):
    """(mismatches inside the guarded range, mismatches outside it) of the kernels' short reciprocal against
    the division 1.0f / x over all 2^32 inputs (crt_device_rcp_check)."""
    import ctypes as C
    a, b = C.c_uint64(0), C.c_uint64(0)
    capi.check(capi.lib().crt_device_rcp_check(device, C.byref(a), C.byref(b)), "crt_device_rcp_check")
    return int(a.value), int(b.value)


def device_philox(ctr, key, device=0):
    c = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(-1, 4)
    k = np.ascontiguousarray(key, dtype=np.uint32).reshape(-1, 2)
    out = np.zeros_like(c)
    capi.check(capi.lib().crt_device_philox(device, c.shape[0], capi.ptr(c), capi.ptr(k), capi.ptr(out)),
               "crt_device_philox")
    return out
