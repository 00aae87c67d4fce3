"""CPU tests of the product's host layer (C++ loader / BVH / flat export / config / PNG) through the
C ABI, compared with the oracle's independent restatement, plus the boundary checks that need no GPU."""
import ctypes as C
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import cudaraytracing_amd as crt
from cudaraytracing_amd import _capi as capi
import oracle_lib as O
import util


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_flat_scene_matches_oracle_bit_for_bit(name):
    s, o = util.host_scene(name), util.oracle_scene(name)
    assert s.nodes().tobytes() == o.nodes().tobytes()
    assert s.root == o.root
    t1, t2 = s.triangles(), o.tris()
    for k in ("v1", "v2", "v3", "normal", "area", "area_of_obj"):
        assert np.array_equal(util.bits(t1[k]), util.bits(t2[k])), k
    mats = s.materials()
    for k in ("kd", "ke", "ns"):
        assert np.array_equal(util.bits(mats[k][t1["material"]]), util.bits(t2[k])), k
    assert np.array_equal(mats["mode"][t1["material"]], t2["mode"])
    assert np.array_equal(mats["has_emit"][t1["material"]], t2["has_emit"])
    lt = s.light_triangles()
    cat = np.concatenate([o.light_tris(i) for i in range(o.num_lights)])
    for k in ("v1", "v2", "v3", "normal", "area_of_obj"):
        assert np.array_equal(util.bits(lt[k]), util.bits(cat[k])), k
    lights = s.lights()
    assert lights["count"].tolist() == [o.light_size(i) for i in range(o.num_lights)]
    assert lights["first_tri"].tolist() == np.concatenate([[0], np.cumsum(lights["count"])[:-1]]).tolist()
    assert [(l, a) for l, a, _ in s.objects()] == o.objects()


def test_material_table_is_deduplicated():
    assert len(util.host_scene("veach-mis").materials()) == 9      # 4 plates, 4 lights, floor+wall share one
    assert len(util.host_scene("cornell-box").materials()) == 6


def test_inverse_view_and_fov_match_oracle():
    rng = np.random.default_rng(2)
    for name in ("cornell-box", "veach-mis"):
        t = util.task(name)
        assert np.array_equal(crt.get_inverse_view_matrix(t.eye_pos, t.lookat, t.up), O.inverse_view(t.eye_pos, t.lookat, t.up))
    for _ in range(50):
        e, l, u = rng.normal(size=3) * 100, rng.normal(size=3) * 10, np.array([0.1, 1.0, -0.2])
        assert np.array_equal(crt.get_inverse_view_matrix(e, l, u), O.inverse_view(e, l, u))
    # src/main.cu:278: task.fov_y * (float)M_PI / 180 evaluated in float
    assert crt.fov_to_radians(39.3077) == np.float32(np.float32(np.float32(39.3077) * np.float32(np.pi)) / np.float32(180))


def test_task_parses_reference_config_schema(tmp_path):
    t = util.task("cornell-box")
    assert (t.width, t.height, t.bvh_thresh_n, t.spp, t.light_sample_n) == (800, 600, 2, 2, 2)
    assert t.P_RR == np.float32(0.6) and t.fov_y == np.float32(39.3077)
    assert t.eye_pos.tolist() == [278.0, 273.0, -800.0] and t.lookat.tolist() == [278.0, 273.0, -799.0]
    v = util.task("veach-mis")
    assert v.eye_pos[2] == np.float32(1.23612e-06) and v.light_sample_n == 1 and v.spp == 4
    bad = tmp_path / "bad.json"
    bad.write_text('{"OBJ_paths": [], "width": 1}')
    with pytest.raises(crt.CrtError) as e:
        crt.Task(str(bad))
    assert e.value.status == -6  # CRT_ERR_PARSE
    with pytest.raises(crt.CrtError) as e:
        crt.Task(str(tmp_path / "missing.json"))
    assert e.value.status == -5  # CRT_ERR_IO


def test_task_with_more_than_eight_obj_files(tmp_path):
    """src/main.cu:74-78 loops over any number of OBJ_paths entries; the crt_task POD has eight slots, crt_task_obj returns the rest, and
    the scene built from eleven files (the stand-in room split into its shapes' worth of copies of one small OBJ) loads them all."""
    import json
    src = os.path.join(util.ROOT, "scenes", "veach-mis")
    cfg = json.load(open(os.path.join(src, "config.json")))
    cfg["OBJ_paths"] = [{"OBJ_path": os.path.join(src, "veach-mis.obj"), "MTL_dir": src + "/"} for _ in range(11)]
    cfg["OBJ_paths"][9]["MTL_dir"] = src + "//"       # a marker to see that entry 9 is entry 9
    p = tmp_path / "eleven.json"
    p.write_text(json.dumps(cfg))
    t = crt.Task(str(p))
    assert len(t.OBJ_paths) == 11 and t.OBJ_paths[9][1].endswith("//") and t.OBJ_paths[10][0].endswith("veach-mis.obj")
    one = crt.Scene.from_task(util.task("veach-mis"))
    s = crt.Scene.from_task(t)
    assert len(s.objects()) == 11 * len(one.objects()) and len(s.triangles()) == 11 * len(one.triangles())
    lib = capi.lib()
    o, m = C.create_string_buffer(8), C.create_string_buffer(8)
    assert lib.crt_task_obj(os.fsencode(str(p)), 0, o, m, 8) == -1       # buffer too small
    o, m = C.create_string_buffer(4096), C.create_string_buffer(4096)
    assert lib.crt_task_obj(os.fsencode(str(p)), 11, o, m, 4096) == -1   # index out of range


def test_loader_error_behaviour(tmp_path):
    s = crt.Scene(8, 8)
    with pytest.raises(crt.CrtError) as e:
        s.add_obj(str(tmp_path / "nope.obj"), str(tmp_path))
    assert e.value.status == -5
    # one vn per v is required (reference Loader.h:70-72 indexes normals with the vertex index)
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.5 0.5 0.5\nNs 1\n")
    (tmp_path / "t.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl a\nf 1/1/1 2/2/2 3/3/3\n")
    with pytest.raises(crt.CrtError) as e:
        s.add_obj(str(tmp_path / "t.obj"), str(tmp_path))
    assert e.value.status == -6
    with pytest.raises(crt.CrtError) as e:
        crt.Scene(8, 8).set_BVH(2)  # empty scene
    assert e.value.status == -1
    ok = crt.Scene(8, 8)
    (tmp_path / "t2.obj").write_text("mtllib m.mtl\nv 0 0 0\nvn 0 0 1\nv 1 0 0\nvn 0 0 1\nv 0 1 0\nvn 0 0 1\n"
                                     "f 1/1/1 2/2/2 3/3/3\nusemtl a\nf 1/1/1 2/2/2 3/3/3\nf 1 3 2\n")
    ok.add_obj(str(tmp_path / "t2.obj"), str(tmp_path))
    with pytest.raises(crt.CrtError) as e:
        ok.set_BVH(0)  # recurses forever in the reference (BVH.h:57-81)
    assert e.value.status == -1
    ok.set_BVH(2)
    # the face before the first usemtl is dropped (OBJLoader.h:120-123); a one-leaf tree is a valid scene
    assert len(ok.triangles()) == 2 and len(ok.nodes()) == 1 and ok.root == 0
    o = O.OracleScene([(str(tmp_path / "t2.obj"), str(tmp_path))], 2)
    assert ok.nodes().tobytes() == o.nodes().tobytes()


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    capi.check(capi.lib().crt_write_png(path.encode(), 53, 37, capi.ptr(img)), "crt_write_png")
    assert np.array_equal(np.asarray(Image.open(path)), img)
    big = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)  # more than one 64 KiB stored block
    capi.check(capi.lib().crt_write_png(path.encode(), 400, 300, capi.ptr(big)), "crt_write_png")
    assert np.array_equal(np.asarray(Image.open(path)), big)


def test_library_exports_every_symbol_the_header_declares():
    header = open(os.path.join(util.ROOT, "include", "crt.h")).read()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(crt_[a-z_0-9]+)\s*\(", body))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = capi.lib()
    for name in declared:
        getattr(lib, name)
    assert lib.crt_abi_version() == capi.ABI_VERSION == 5
    assert lib.crt_strerror(-4).decode() == "unsupported"


def test_device_entry_points_validate_arguments_without_a_gpu():
    lib = capi.lib()
    n = C.c_uint64()
    capi.check(lib.crt_shard_slots(800, 600, 0, 1, C.byref(n)), "crt_shard_slots")
    assert n.value == 100 * 75 * 64
    capi.check(lib.crt_shard_slots(100, 75, 2, 3, C.byref(n)), "crt_shard_slots")
    assert n.value == ((13 * 10 + 2) // 3) * 64
    assert lib.crt_shard_slots(0, 75, 0, 1, C.byref(n)) == -1
    assert lib.crt_shard_slots(8, 8, 3, 3, C.byref(n)) == -1
    h = C.c_void_p()
    assert lib.crt_scene_create(None, 0, C.byref(h)) == -1  # invalid description, reported before any device call
    d = capi.SceneDesc()
    assert lib.crt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert lib.crt_render(None, None, None, None, None, None) == -1
    assert b"null" in lib.crt_last_error()


def test_cornell_generator_reproduces_committed_obj(tmp_path):
    out = str(tmp_path / "c.obj")
    subprocess.check_call([sys.executable, os.path.join(util.ROOT, "scenes", "gen_cornell_box.py"), out],
                          stdout=subprocess.DEVNULL)
    a = hashlib.sha256(open(out, "rb").read()).hexdigest()
    b = hashlib.sha256(open(os.path.join(util.ROOT, "scenes", "cornell-box", "cornell-box.obj"), "rb").read()).hexdigest()
    assert a == b


def test_multi_device_entry_validates_arguments_without_a_gpu():
    lib = capi.lib()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert lib.crt_multi_create(None, devs, 2, 0, C.byref(h)) == -1
    sc = util.host_scene("veach-mis")
    assert lib.crt_multi_create(C.byref(sc.desc()), devs, 0, 0, C.byref(h)) == -1
    assert lib.crt_multi_create(C.byref(sc.desc()), devs, 2, 7, C.byref(h)) == -1
    assert b"gather" in lib.crt_last_error()
    rc = lib.crt_multi_create(C.byref(sc.desc()), devs, 2, 0, C.byref(h))
    if rc == 0:  # (a box with two GPUs)
        lib.crt_multi_destroy(h)
    else:
        assert rc in (-1, -2, -4)  # no device here / fewer than two devices / RCCL not loadable
    assert lib.crt_multi_render(None, None, None, None, None, None, None) == -1
    assert lib.crt_multi_destroy(None) == 0


def test_bench_started_plainly_with_several_gpus_starts_its_ranks(tmp_path):
    """`python bench.py --gpus 2` without torch.distributed.run must start the two ranks itself (it used to exit).
    There is no GPU here, so each rank stops at its own "needs a GPU" check -- after the launcher-free start has happened."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=util.ROOT)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_multi_gpu.py")
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2, r.stderr[-2000:]
    assert "must be launched with" not in r.stderr


def test_bench_workload_aliases_are_the_baseline_configurations(monkeypatch):
    """bench.py --workload c2..c5 = BASELINE.json's configs[1..4] (scene, size, spp); single fields can be overridden, and only an exact
    BASELINE configuration carries its id into the line (the id selects hash-stamped counters and the share model)."""
    import json
    import bench
    base = json.load(open(os.path.join(util.ROOT, "BASELINE.json")))["configs"]
    for wl, cfg in zip(("c2", "c3", "c4", "c5"), base[1:]):
        scene, w, h, spp = bench.WORKLOADS[wl]
        assert cfg.startswith(scene + ".obj %dx%d spp=%d" % (w, h, spp)), (wl, cfg)
        monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", wl])
        a = bench.parse_args()
        assert (a.scene, a.width, a.height, a.spp, a.workload_id) == (scene, w, h, spp, wl)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.scene, a.width, a.height, a.spp, a.workload_id) == ("cornell-box", 800, 600, 512, "c2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "c4", "--spp", "8"])
    a = bench.parse_args()
    assert (a.scene, a.width, a.height, a.spp, a.workload_id) == ("cornell-box", 3840, 2160, 8, None)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--scene", "veach-mis", "--spp", "1024"])
    assert bench.parse_args().workload_id == "c3"


def test_header_constants_match_the_python_mirror():
    """include/crt.h is the contract: the enum values the Python mirror passes must be the header's, and the default traversal mode
    (a zeroed crt_params) must be the provably exact one."""
    import re
    from cudaraytracing_amd import _capi as capi
    hdr = open(os.path.join(util.ROOT, "include", "crt.h")).read()

    def value(name):
        m = re.search(r"\b%s\s*=\s*(0x[0-9a-fA-F]+|\d+)u?" % name, hdr) or re.search(r"#define\s+%s\s+(0x[0-9a-fA-F]+|\d+)u?" % name, hdr)
        assert m, name
        return int(m.group(1), 0)

    assert value("CRT_TRAVERSAL_EXACT") == capi.TRAVERSAL_EXACT == 0
    assert value("CRT_TRAVERSAL_REFERENCE") == capi.TRAVERSAL_REFERENCE
    assert value("CRT_TRAVERSAL_FAST") == capi.TRAVERSAL_FAST
    for h, p in (("CRT_FLAG_STATS", capi.FLAG_STATS), ("CRT_FLAG_TILED_OUTPUT", capi.FLAG_TILED_OUTPUT), ("CRT_FLAG_TRACE_ALL", capi.FLAG_TRACE_ALL),
                 ("CRT_FLAG_FORCE_EXACT", capi.FLAG_FORCE_EXACT), ("CRT_INTERSECT_RAW_DIRECTIONS", capi.INTERSECT_RAW_DIRECTIONS),
                 ("CRT_INTERSECT_FORCE_EXACT", capi.INTERSECT_FORCE_EXACT), ("CRT_INTERSECT_VISIBILITY", capi.INTERSECT_VISIBILITY)):
        assert value(h) == p, h
    import cudaraytracing_amd as crt
    sc = crt.Scene(8, 8)
    # (a Render needs a GPU; the mirror's default is checked on the class attribute path instead)
    import inspect
    src = inspect.getsource(crt.Render.__init__)
    assert "TRAVERSAL_EXACT" in src and "TRAVERSAL_FAST" not in src


def test_the_documented_reference_side_binding_compiles(tmp_path):
    """INTEGRATION.md section 2 shows the `Render` class a maintainer of the reference would drop in (talking to libcrt.so through
    include/crt.h).  The code block is extracted and type-checked here -- against include/crt.h, the reference's vendored Eigen and
    test-only declarations of the Scene / BVH / Triangle / Material / Object getters it calls (tests/shim_decls/Scene.h) -- with the
    calls src/main.cu makes (main.cu:282, 368-377), so that a change of a struct in crt.h cannot silently break the documented binding."""
    import shutil
    import subprocess
    eigen = "/root/reference/include"
    if not os.path.isdir(os.path.join(eigen, "Eigen")) or not shutil.which("g++"):
        pytest.skip("needs the reference's vendored Eigen (/root/reference) and g++")
    text = open(os.path.join(util.ROOT, "INTEGRATION.md")).read()
    sect = text[text.index("## 2. Reference-side shim"):]
    code = sect[sect.index("```cpp") + 6:]
    code = code[:code.index("```")]
    assert "class Render" in code and "crt_scene_create" in code and "crt_render" in code
    (tmp_path / "RenderCrt.h").write_text(code)
    (tmp_path / "main.cpp").write_text('''
#include "RenderCrt.h"
int use(Scene* scene, Eigen::Vector3f eye, Eigen::Matrix3f inv_view, float fov_y)
{
    Render render(scene, 16, 0.8f, 1);                 // src/main.cu:282
    render.set_spp(64); render.set_P_RR(0.6f); render.set_light_sample_n(2);
    render.run_view(eye, inv_view, fov_y, nullptr);    // src/main.cu:372
    unsigned char* fb = render.get_frame_buffer();
    render.save_frame_buffer("out.png");
    render.free();
    return fb ? 0 : 1;
}
''')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=narrowing", "-I", os.path.join(util.ROOT, "include"),
           "-I", os.path.join(util.ROOT, "tests", "shim_decls"), "-I", eigen, str(tmp_path / "main.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
