"""ctypes binding of oracle/liboracle.so -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class OrcNode(C.Structure):
    _fields_ = [("lc", C.c_int32), ("rc", C.c_int32), ("n", C.c_uint32), ("it", C.c_int32),
                ("aa", C.c_float * 3), ("bb", C.c_float * 3)]


class OrcTri(C.Structure):
    _fields_ = [("v1", C.c_float * 3), ("v2", C.c_float * 3), ("v3", C.c_float * 3), ("e1", C.c_float * 3),
                ("e2", C.c_float * 3), ("normal", C.c_float * 3), ("kd", C.c_float * 3), ("ke", C.c_float * 3),
                ("ns", C.c_float), ("has_emit", C.c_int32), ("mode", C.c_int32), ("area", C.c_float),
                ("area_of_obj", C.c_float)]


class OrcCamera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("inv_view", C.c_float * 9), ("fov_y", C.c_float)]


class OrcParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("p_rr", C.c_float),
                ("light_sample_n", C.c_int32), ("seed", C.c_uint64), ("x0", C.c_uint32), ("y0", C.c_uint32),
                ("cw", C.c_uint32), ("ch", C.c_uint32)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "rays", "inner_pops", "leaf_pops", "tri_tests", "hits",
                                          "vertices", "shadow_rays", "probe_rays")] + \
               [("max_bvh_stack", C.c_uint32), ("max_depth", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


NODE_DTYPE = np.dtype([("lc", "<i4"), ("rc", "<i4"), ("n", "<u4"), ("it", "<i4"), ("aa", "<f4", 3), ("bb", "<f4", 3)])
TRI_DTYPE = np.dtype([("v1", "<f4", 3), ("v2", "<f4", 3), ("v3", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3),
                      ("normal", "<f4", 3), ("kd", "<f4", 3), ("ke", "<f4", 3), ("ns", "<f4"), ("has_emit", "<i4"),
                      ("mode", "<i4"), ("area", "<f4"), ("area_of_obj", "<f4")])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src_newer = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
        for f in ("crt_oracle.cpp", "crt_oracle.h", "det_math.h", "philox.h"))
    if src_newer:
        build()
    L = C.CDLL(LIB_PATH)
    L.orc_scene_new.restype = C.c_void_p
    L.orc_scene_add_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.orc_register_texture.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_register_texture.restype = None
    L.orc_scene_build.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_scene_free.argtypes = [C.c_void_p]
    for f in ("orc_scene_num_tris", "orc_scene_num_nodes", "orc_scene_num_lights", "orc_scene_num_objects"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_uint32
    L.orc_scene_root.argtypes = [C.c_void_p]
    L.orc_scene_root.restype = C.c_int32
    L.orc_scene_light_size.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_scene_light_size.restype = C.c_uint32
    L.orc_scene_object_area.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_scene_object_area.restype = C.c_float
    L.orc_scene_object_is_light.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_scene_get_nodes.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_scene_get_tris.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_scene_get_light_tris.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_inverse_view.argtypes = [C.c_void_p] * 4
    L.orc_render.argtypes = [C.c_void_p, C.POINTER(OrcCamera), C.POINTER(OrcParams), C.c_void_p, C.c_void_p,
                             C.c_void_p, C.POINTER(OrcStats)]
    L.orc_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(OrcStats)]
    L.orc_math.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_philox.argtypes = [C.c_void_p] * 3
    L.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                               C.c_void_p]
    L.orc_vec_op.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    L.orc_sample_hemisphere.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    L.orc_sample_lobe.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.orc_tonemap.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def stb_like_decode(path):
    """Decodes an image with PIL into what stbi_load(path, &x, &y, &comp, 0) returns: (x, y, comp, uint8 samples).
    An independent decoder: the product has its own PNG reader (csrc/crt_png.h)."""
    from PIL import Image
    with open(path, "rb") as f:
        head = f.read(12)
    if head[:2] == b"\xff\xd8" or head[:4] in (b"GIF8", b"8BPS", b"\x53\x80\xf6\x34") or head[:2] in (b"P5", b"P6") or head[:2] == b"#?":
        # JPEG: the inverse DCT, the chroma upsampling and the colour arithmetic are a decoder's own choice, so no second decoder returns
        # the reference's samples; GIF / PSD / PIC / PNM / HDR: first frame and background, matte removal, byte order and tone mapping are
        # that decoder's conventions.  The oracle is fed with what the REFERENCE's stb_image returned for the file (tests/golden/
        # stb_samples.npz, written by tests/golden/make_texture_golden.py from oracle/_ref/stb_probe, keyed by the file's SHA-1).
        import hashlib
        key = hashlib.sha1(open(path, "rb").read()).hexdigest()
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stb_samples.npz"))
        if key not in z.files:
            raise RuntimeError("no golden stb_image samples for the file %s (tests/golden/make_texture_golden.py)" % path)
        a = np.ascontiguousarray(z[key], dtype=np.uint8)
        return a.shape[1], a.shape[0], a.shape[2], a
    img = Image.open(path)
    if img.mode == "P":
        img = img.convert("RGBA" if "transparency" in img.info else "RGB")
    elif img.mode in ("1",):
        img = img.convert("L")
    elif img.mode in ("I;16", "I;16B", "I"):
        a = np.asarray(img).astype(np.uint32)
        img = Image.fromarray((a >> 8).astype(np.uint8), "L")  # stb keeps the high byte
    elif img.mode == "L" and "transparency" in img.info:
        key = img.info["transparency"]
        a = np.asarray(img)
        img = Image.fromarray(np.dstack([a, np.where(a == key, 0, 255).astype(np.uint8)]), "LA")
    elif img.mode == "RGB" and "transparency" in img.info:
        key = np.array(img.info["transparency"], dtype=np.uint8)
        a = np.asarray(img)
        alpha = np.where(np.all(a == key, axis=2), 0, 255).astype(np.uint8)
        img = Image.fromarray(np.dstack([a, alpha]), "RGBA")
    a = np.ascontiguousarray(np.asarray(img), dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return a.shape[1], a.shape[0], a.shape[2], a


def register_textures(obj_path, mtl_dir):
    """Finds the map_Kd files of an OBJ's material library and hands their decoded samples to the oracle."""
    mtl = None
    with open(obj_path) as f:
        for line in f:
            t = line.split()
            if len(t) >= 2 and t[0] == "mtllib":
                mtl = mtl_dir + "/" + t[1]
    if not mtl or not os.path.exists(mtl):
        return
    with open(mtl) as f:
        for line in f:
            t = line.split()
            if len(t) >= 2 and t[0] == "map_Kd":
                path = mtl_dir + "/" + t[1]
                x, y, comp, a = stb_like_decode(path)
                lib().orc_register_texture(path.encode(), x, y, comp, a.ctypes.data_as(C.c_void_p))


class OracleScene:
    """Scene loaded and BVH-built by the oracle's own restatement of the reference loader."""

    def __init__(self, obj_paths, thresh_n):
        L = lib()
        self.h = L.orc_scene_new()
        for obj, mtl in obj_paths:
            register_textures(obj, mtl)
            rc = L.orc_scene_add_obj(self.h, obj.encode(), mtl.encode())
            if rc != 0:
                raise RuntimeError("oracle failed to load %s (%d)" % (obj, rc))
        if L.orc_scene_build(self.h, thresh_n) != 0:
            raise RuntimeError("oracle BVH build failed")

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def num_tris(self):
        return lib().orc_scene_num_tris(self.h)

    @property
    def num_nodes(self):
        return lib().orc_scene_num_nodes(self.h)

    @property
    def root(self):
        return lib().orc_scene_root(self.h)

    @property
    def num_lights(self):
        return lib().orc_scene_num_lights(self.h)

    def light_size(self, i):
        return lib().orc_scene_light_size(self.h, i)

    def objects(self):
        L = lib()
        return [(bool(L.orc_scene_object_is_light(self.h, i)), float(L.orc_scene_object_area(self.h, i)))
                for i in range(L.orc_scene_num_objects(self.h))]

    def nodes(self):
        a = np.zeros(self.num_nodes, dtype=NODE_DTYPE)
        lib().orc_scene_get_nodes(self.h, _p(a))
        return a

    def tris(self):
        a = np.zeros(self.num_tris, dtype=TRI_DTYPE)
        lib().orc_scene_get_tris(self.h, _p(a))
        return a

    def light_tris(self, i):
        a = np.zeros(self.light_size(i), dtype=TRI_DTYPE)
        lib().orc_scene_get_light_tris(self.h, i, _p(a))
        return a

    def render(self, eye, inv_view, fov_y_rad, width, height, spp, p_rr, light_sample_n, seed=0, crop=None,
               want_L=False):
        x0, y0, cw, ch = crop if crop else (0, 0, width, height)
        cam = OrcCamera()
        cam.eye[:] = [float(v) for v in eye]
        cam.inv_view[:] = [float(v) for v in np.asarray(inv_view, dtype=np.float32).reshape(9)]
        cam.fov_y = float(np.float32(fov_y_rad))
        prm = OrcParams(width, height, spp, float(np.float32(p_rr)), light_sample_n, seed, x0, y0, cw, ch)
        rgb = np.zeros((ch, cw, 3), dtype=np.uint8)
        mean = np.zeros((ch, cw, 3), dtype=np.float32)
        Lbuf = np.zeros((ch, cw, spp, 3), dtype=np.float32) if want_L else None
        st = OrcStats()
        rc = lib().orc_render(self.h, C.byref(cam), C.byref(prm), _p(rgb), _p(mean), _p(Lbuf), C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        return rgb, mean, Lbuf, st.as_dict()

    def intersect(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        n = o.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        st = OrcStats()
        lib().orc_intersect(self.h, n, _p(o), _p(d), _p(tri), _p(t), C.byref(st))
        return tri, t, st.as_dict()


def inverse_view(eye, lookat, up):
    e = np.asarray(eye, dtype=np.float32)
    l = np.asarray(lookat, dtype=np.float32)
    u = np.asarray(up, dtype=np.float32)
    out = np.zeros(9, dtype=np.float32)
    lib().orc_inverse_view(_p(e), _p(l), _p(u), _p(out))
    return out


def math_fn(name, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    bb = np.ascontiguousarray(b, dtype=np.float32) if b is not None else None
    out = np.zeros_like(a)
    lib().orc_math(name.encode(), a.size, _p(a), _p(bb), _p(out))
    return out


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(_p(c), _p(k), _p(out))
    return out


def rng_draw(seed, pixel, k, depth, purpose, idx):
    u = np.zeros(4, dtype=np.uint32)
    f = np.zeros(4, dtype=np.float32)
    lib().orc_rng_draw(seed, pixel, k, depth, purpose, idx, _p(u), _p(f))
    return u, f


def vec_op(op, in_bits):
    a = np.asarray(in_bits, dtype=np.uint32).view(np.float32).copy()
    if a.size < 12:
        a = np.concatenate([a, np.zeros(12 - a.size, dtype=np.float32)])
    out = np.zeros(9, dtype=np.float32)
    n = lib().orc_vec_op(op.encode(), _p(a), _p(out))
    if n < 0:
        raise KeyError(op)
    return out[:n].view(np.uint32)


def sample_hemisphere(n, x1, x2):
    nn = np.asarray(n, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_sample_hemisphere(_p(nn), float(x1), float(x2), _p(out))
    return out


def sample_lobe(o, dt, dp, u1, u2):
    oo = np.asarray(o, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_sample_lobe(_p(oo), float(dt), float(dp), float(u1), float(u2), _p(out))
    return out


def tonemap(c):
    c = np.ascontiguousarray(c, dtype=np.float32)
    out = np.zeros(c.size, dtype=np.uint8)
    lib().orc_tonemap(c.size, _p(c), _p(out))
    return out.reshape(c.shape)
