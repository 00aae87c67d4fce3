"""map_Kd textures (reference: include/Loader.h:55-105, include/OBJLoader.h:184-193): per-triangle kd = mean of the
texels under the three vertices, with the reference's swapped width / height (Loader.h:58).  The product decodes PNG, JPEG,
BMP and TGA with its own readers (csrc/crt_png.h, crt_jpeg.h, crt_image.h), pinned against the reference's stb_image
(tests/golden/stb_decode.json); the oracle gets the samples from an independent decoder (PIL), or -- JPEG -- from the golden
samples of the reference's decoder."""
import os

import numpy as np
import pytest

import cudaraytracing_amd as crt
import oracle_lib as O
import util

PIL = pytest.importorskip("PIL.Image")


def _texture(kind, w, h, seed):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    if kind == "rgb":
        return PIL.fromarray(rgb, "RGB"), {}
    if kind == "rgba":
        return PIL.fromarray(np.dstack([rgb, rng.integers(0, 256, size=(h, w), dtype=np.uint8)]), "RGBA"), {}
    if kind == "grey":
        return PIL.fromarray(rgb[:, :, 0], "L"), {}
    if kind == "grey_alpha":
        return PIL.fromarray(rgb[:, :, :2], "LA"), {}
    if kind == "palette":
        return PIL.fromarray(rgb, "RGB").quantize(37), {}
    if kind == "palette4":  # 4-bit indices
        return PIL.fromarray(rgb, "RGB").quantize(11), {"bits": 4}
    if kind == "rgb16":
        a = rng.integers(0, 65536, size=(h, w), dtype=np.uint16)
        return PIL.fromarray(a), {}
    if kind == "rgb_big_filtered":  # large enough for dynamic Huffman blocks and every row filter
        yy, xx = np.mgrid[0:h, 0:w]
        g = np.stack([(xx * 3 + yy) % 256, (xx ^ yy) % 256, (yy * 5) % 256], axis=2).astype(np.uint8)
        return PIL.fromarray(g, "RGB"), {"optimize": True}
    raise ValueError(kind)


def _write_scene(d, kind, w, h, seed=5, n=7):
    """A floor of n x n textured quads (uv across the whole texture and beyond, negative too), a light above."""
    img, opts = _texture(kind, w, h, seed)
    img.save(os.path.join(d, "tex.png"), **opts)
    v, vt, f = [], [], []
    rng = np.random.default_rng(seed + 1)
    for i in range(n):
        for j in range(n):
            b = len(v)
            for (dx, dz) in ((0, 0), (0, 1), (1, 1), (1, 0)):
                v.append((i + dx, 0.0, j + dz))
                vt.append(((i + dx) / n * 1.7 - 0.3 + rng.uniform(-0.01, 0.01), (j + dz) / n * 2.1 - 0.6))
            f.append(("tex", b + 1, b + 2, b + 3))
            f.append(("tex", b + 1, b + 3, b + 4))
    b = len(v)
    for p in ((2, 4, 2), (5, 4, 2), (5, 4, 5), (2, 4, 5)):
        v.append(p)
        vt.append((0.5, 0.5))
    f.append(("light", b + 1, b + 2, b + 3))
    f.append(("light", b + 1, b + 3, b + 4))
    with open(os.path.join(d, "t.mtl"), "w") as m:
        m.write("newmtl tex\nKd 0.3 0.3 0.3\nmap_Kd tex.png\nNs 1\nnewmtl light\nKe 20 20 20\nKd 0 0 0\nNs 1\n")
    with open(os.path.join(d, "t.obj"), "w") as o:
        o.write("mtllib t.mtl\n")
        for p, t in zip(v, vt):
            o.write("v %r %r %r\nvn 0 1 0\nvt %r %r\n" % (p[0], p[1], p[2], t[0], t[1]))
        cur = None
        for mtl, a, bb, c in f:
            if mtl != cur:
                o.write("usemtl %s\n" % mtl)
                cur = mtl
            o.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, bb, bb, bb, c, c, c))
    return os.path.join(d, "t.obj"), d


@pytest.mark.parametrize("kind,w,h", [("rgb", 16, 16), ("rgb", 23, 9), ("rgba", 12, 12), ("grey", 12, 12), ("grey_alpha", 8, 8),
                                      ("palette", 16, 16), ("palette4", 16, 16), ("rgb16", 10, 10), ("rgb_big_filtered", 200, 200)])
def test_textured_materials_match_oracle(tmp_path, kind, w, h):
    obj, mtl = _write_scene(str(tmp_path), kind, w, h)
    # the product's PNG reader against PIL, sample for sample
    x, y, comp, ref = O.stb_like_decode(os.path.join(mtl, "tex.png"))
    scene = crt.Scene(32, 24)
    try:
        scene.add_obj(obj, mtl)
    except crt.CrtError as e:
        # non-square textures: the reference's swapped width / height (Loader.h:58) can index outside the image;
        # the oracle must reject the same scene
        assert (w, h) == (23, 9), e
        with pytest.raises(RuntimeError):
            O.OracleScene([(obj, mtl)], 2)
        return
    scene.set_BVH(2)
    osc = O.OracleScene([(obj, mtl)], 2)
    t1, t2 = scene.triangles(), osc.tris()
    mats = scene.materials()
    for k in ("kd", "ke", "ns"):
        assert np.array_equal(util.bits(mats[k][t1["material"]]), util.bits(t2[k])), k
    kd = mats["kd"][t1["material"]]
    assert len(np.unique(kd.round(6), axis=0)) > 10          # the texture really varies over the floor
    assert (x, y) == (w, h) and comp >= 1


def test_a_texture_that_cannot_be_decoded_fails_the_scene(tmp_path):
    obj, mtl = _write_scene(str(tmp_path), "rgb", 8, 8)
    with open(os.path.join(mtl, "tex.png"), "wb") as f:   # a GIF signature and nothing behind it (the name of the file does not matter)
        f.write(b"GIF89a" + b"\0" * 64)
    s = crt.Scene(8, 8)
    with pytest.raises(crt.CrtError) as e:
        s.add_obj(obj, mtl)
    assert "GIF" in str(e.value) and "tex.png" in str(e.value)
    os.remove(os.path.join(mtl, "tex.png"))
    with pytest.raises(crt.CrtError):
        crt.Scene(8, 8).add_obj(obj, mtl)


@pytest.mark.parametrize("kind,w,h", [("rgb", 9, 7), ("palette4", 9, 7), ("rgb_big_filtered", 64, 48)])
def test_damaged_png_files_are_decoded_or_rejected(tmp_path, kind, w, h):
    """Random damage to a valid PNG (flipped bytes, truncation, inserted bytes, overwritten length / CRC words): the reader
    returns an image or an error, it never reads out of bounds (a crash would take the test process down)."""
    import random
    random.seed(7)
    d = str(tmp_path)
    obj, mtl = _write_scene(d, kind, w, h)[:2]
    png = os.path.join(d, "tex.png")
    orig = open(png, "rb").read()
    decoded = rejected = 0
    for it in range(160):
        b = bytearray(orig)
        mode = it % 4
        if mode == 0:
            for _ in range(random.randint(1, 4)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif mode == 1:
            b = b[:random.randrange(1, len(b))]
        elif mode == 2:
            i = random.randrange(8, len(b))
            b[i:i] = bytes(random.randrange(256) for _ in range(random.randint(1, 16)))
        else:
            i = random.randrange(8, len(b) - 4)
            b[i:i + 4] = random.getrandbits(32).to_bytes(4, "big")
        with open(png, "wb") as f:
            f.write(bytes(b))
        try:
            sc = crt.Scene(8, 8)
            sc.add_obj(obj, mtl)
            decoded += 1
        except crt.CrtError:
            rejected += 1
    assert decoded + rejected == 160 and rejected > 40


def _fnv1a64(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xffffffffffffffff
    return "%016x" % h


def test_decoders_match_the_references_own_stb_image():
    """tests/golden/stb_decode.json = what the reference's vendored stb_image.h returns for the fixture files (x, y, comp and
    every sample, via oracle/ref_probe/stb_probe.c): the product's decoders (PNG plain and Adam7-interlaced, JPEG baseline and
    progressive, BMP, TGA, GIF, PSD, PIC, PNM, HDR -- every format that decoder reads) must return exactly that, and so must the
    feed of the oracle (PIL for the PNG files: 16 -> 8-bit reduction, palette / tRNS expansion, channel counts; the golden samples
    themselves for the JPEG, GIF, PSD, PIC, PNM and HDR files)."""
    import json
    gold = json.load(open(os.path.join(util.ROOT, "tests", "golden", "stb_decode.json")))["files"]
    assert len(gold) >= 115
    kinds = set()
    n_jpg = n_adam7 = 0
    for name, g in sorted(gold.items()):
        path = os.path.join(util.ROOT, "tests", "golden", "textures", name)
        x, y, comp, a = crt.image_load(path)
        assert (x, y, comp) == (g["x"], g["y"], g["comp"]), name
        assert _fnv1a64(a) == g["fnv1a64"], name
        assert list(a.reshape(-1)[:24]) == g["head"], name
        kinds.add(name.split("_")[0])
        if not name.endswith((".bmp", ".tga")):
            px, py, pc, pa = O.stb_like_decode(path)
            assert (px, py, pc) == (g["x"], g["y"], g["comp"]) and _fnv1a64(pa) == g["fnv1a64"], name
        n_jpg += name.endswith(".jpg")
        n_adam7 += "adam7" in name
    assert kinds == {"png", "bmp", "tga", "jpg", "gif", "psd", "pic", "pnm", "hdr"} and n_jpg >= 30 and n_adam7 >= 8


def _use_fixture_texture(d, fixture):
    """Points the scene of _write_scene at a copy of a fixture file of tests/golden/textures (square ones: the reference swaps
    width and height, Loader.h:58)."""
    import shutil
    ext = fixture.rsplit(".", 1)[1]
    shutil.copy(os.path.join(util.ROOT, "tests", "golden", "textures", fixture), os.path.join(d, "tex." + ext))
    os.remove(os.path.join(d, "tex.png"))
    m = open(os.path.join(d, "t.mtl")).read().replace("tex.png", "tex." + ext)
    open(os.path.join(d, "t.mtl"), "w").write(m)
    return os.path.join(d, "tex." + ext)


@pytest.mark.parametrize("fixture", ["jpg_420_16x16.jpg", "jpg_420_8x8.jpg", "jpg_420_1x1.jpg"])
def test_jpeg_textures_feed_the_materials(tmp_path, fixture):
    """map_Kd pointing at a JPEG file: per-triangle kd as the oracle computes it from the samples the REFERENCE's stb_image returns
    for that file (golden data), bit for bit."""
    d = str(tmp_path)
    obj, mtl = _write_scene(d, "rgb", 16, 16)
    tex = _use_fixture_texture(d, fixture)
    x, y, comp, mine = crt.image_load(tex)
    px, py, pc, ref = O.stb_like_decode(tex)
    assert (x, y, comp) == (px, py, pc) and comp == 3 and np.array_equal(mine, ref)
    scene = crt.Scene(32, 24)
    scene.add_obj(obj, mtl)
    scene.set_BVH(2)
    osc = O.OracleScene([(obj, mtl)], 2)
    t1, t2 = scene.triangles(), osc.tris()
    mats = scene.materials()
    for k in ("kd", "ke", "ns"):
        assert np.array_equal(util.bits(mats[k][t1["material"]]), util.bits(t2[k])), k
    if x > 1:
        assert len(np.unique(mats["kd"][t1["material"]].round(6), axis=0)) > 10


@pytest.mark.parametrize("name", ["jpg_420_q75.jpg", "jpg_420_prog.jpg", "jpg_444_restart_blocks.jpg", "jpg_grey_q80.jpg", "jpg_cmyk.jpg", "png_adam7_rgb8.png", "png_adam7_grey4.png"])
def test_damaged_jpeg_and_interlaced_png_files_are_decoded_or_rejected(tmp_path, name):
    """Random damage to valid JPEG / Adam7 PNG files: an image or an error, never an out-of-bounds access or an endless loop."""
    import random
    random.seed(13)
    orig = open(os.path.join(util.ROOT, "tests", "golden", "textures", name), "rb").read()
    p = str(tmp_path / name)
    ok = bad = 0
    for it in range(240):
        b = bytearray(orig)
        mode = it % 4
        if mode == 0:
            for _ in range(random.randint(1, 4)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif mode == 1:
            b = b[:random.randrange(1, len(b))]
        elif mode == 2:
            i = random.randrange(2, len(b))
            b[i:i] = bytes(random.randrange(256) for _ in range(random.randint(1, 16)))
        else:   # marker / length / table bytes near the head of the file
            i = random.randrange(2, min(len(b) - 4, 400))
            b[i:i + 2] = random.getrandbits(16).to_bytes(2, "big")
        with open(p, "wb") as f:
            f.write(bytes(b))
        try:
            x, y, comp, a = crt.image_load(p)
            assert a.size == x * y * comp and 1 <= comp <= 4
            ok += 1
        except crt.CrtError:
            bad += 1
    assert ok + bad == 240 and ok > 20 and bad > 20


@pytest.mark.parametrize("ext,fmt", [("bmp", "BMP"), ("tga", "TGA")])
def test_bmp_and_tga_textures_feed_the_materials(tmp_path, ext, fmt):
    """map_Kd pointing at a 24-bit BMP / TGA: per-triangle kd as the oracle computes it from the same samples."""
    d = str(tmp_path)
    obj, mtl = _write_scene(d, "rgb", 16, 16)
    PIL.open(os.path.join(d, "tex.png")).save(os.path.join(d, "tex." + ext), format=fmt)
    os.remove(os.path.join(d, "tex.png"))
    m = open(os.path.join(d, "t.mtl")).read().replace("tex.png", "tex." + ext)
    open(os.path.join(d, "t.mtl"), "w").write(m)
    x, y, comp, mine = crt.image_load(os.path.join(d, "tex." + ext))
    px, py, pc, ref = O.stb_like_decode(os.path.join(d, "tex." + ext))
    assert (x, y, comp) == (px, py, pc) == (16, 16, 3) and np.array_equal(mine, ref)
    scene = crt.Scene(32, 24)
    scene.add_obj(obj, mtl)
    scene.set_BVH(2)
    osc = O.OracleScene([(obj, mtl)], 2)
    t1, t2 = scene.triangles(), osc.tris()
    mats = scene.materials()
    for k in ("kd", "ke", "ns"):
        assert np.array_equal(util.bits(mats[k][t1["material"]]), util.bits(t2[k])), k
    assert len(np.unique(mats["kd"][t1["material"]].round(6), axis=0)) > 10


def test_files_that_are_no_image_are_rejected(tmp_path):
    """What the reference's decoder calls "unknown image type" (and a GIF signature of a version that does not exist, which it passes on
    to its other format tests) is CRT_ERR_UNSUPPORTED here; a missing file is CRT_ERR_IO."""
    for content in (b"GIF88a" + b"\x01" * 64, b"hello, world\n" * 8, b"P7\n2 2\n255\n" + b"\x07" * 12, b"#?RADIANCE" + b"\xff" * 40):
        p = str(tmp_path / "t.bin")
        with open(p, "wb") as f:
            f.write(content)
        with pytest.raises(crt.CrtError) as e:
            crt.image_load(p)
        assert e.value.status == -4, content[:8]
    with pytest.raises(crt.CrtError) as e:
        crt.image_load(str(tmp_path / "missing.png"))
    assert e.value.status == -5


@pytest.mark.parametrize("fixture", ["gif_local_palette_bg.gif", "psd_rgba8_rle_matte.psd", "pic_rgba_mixed_rle.pic", "pnm_p6_16.ppm", "hdr_rle.hdr"])
def test_the_rarer_formats_feed_the_materials(tmp_path, fixture):
    """map_Kd pointing at a GIF / PSD / PIC / PNM / HDR file: per-triangle kd as the oracle computes it from the samples the REFERENCE's
    stb_image returns for that file (golden data), bit for bit."""
    d = str(tmp_path)
    obj, mtl = _write_scene(d, "rgb", 16, 16)
    tex = _use_fixture_texture(d, fixture)
    x, y, comp, mine = crt.image_load(tex)
    px, py, pc, ref = O.stb_like_decode(tex)
    assert (x, y, comp) == (px, py, pc) and np.array_equal(mine, ref)
    scene = crt.Scene(32, 24)
    scene.add_obj(obj, mtl)
    scene.set_BVH(2)
    osc = O.OracleScene([(obj, mtl)], 2)
    t1, t2 = scene.triangles(), osc.tris()
    mats = scene.materials()
    for k in ("kd", "ke", "ns"):
        assert np.array_equal(util.bits(mats[k][t1["material"]]), util.bits(t2[k])), k
    assert len(np.unique(mats["kd"][t1["material"]].round(6), axis=0)) > 4


@pytest.mark.parametrize("name", ["gif_interlaced.gif", "gif_local_palette_bg.gif", "gif_noise_table_resets.gif", "psd_rgba8_rle_matte.psd", "psd_rgb16_raw.psd",
                                  "pic_rgba_mixed_rle.pic", "pic_long_runs.pic", "pnm_p6_comments.ppm", "pnm_p5_16_max1000.pgm", "hdr_rle.hdr",
                                  "hdr_flat_after_rle_rows.hdr"])
def test_damaged_files_of_the_rarer_formats_are_decoded_or_rejected(tmp_path, name):
    """Random damage to valid GIF / PSD / PIC / PNM / HDR files: an image or an error, never an out-of-bounds access or an endless loop."""
    import random
    random.seed(17)
    orig = open(os.path.join(util.ROOT, "tests", "golden", "textures", name), "rb").read()
    p = str(tmp_path / name)
    ok = bad = 0
    for it in range(200):
        b = bytearray(orig)
        mode = it % 4
        if mode == 0:
            for _ in range(random.randint(1, 4)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif mode == 1:
            b = b[:random.randrange(8, len(b))]
        elif mode == 2:
            i = random.randrange(6, len(b))
            b[i:i] = bytes(random.randrange(256) for _ in range(random.randint(1, 16)))
        else:   # header bytes
            i = random.randrange(4, min(len(b) - 2, 120))
            b[i:i + 2] = random.getrandbits(16).to_bytes(2, "big")
        with open(p, "wb") as f:
            f.write(bytes(b))
        try:
            x, y, comp, a = crt.image_load(p)
            assert a.size == x * y * comp and 1 <= comp <= 4
            ok += 1
        except crt.CrtError:
            bad += 1
    assert ok + bad == 200 and ok > 10 and bad > 5, (ok, bad)


@pytest.mark.parametrize("name", ["bmp_8_palette.bmp", "bmp_16_565.bmp", "bmp_32_v5_alpha_mask.bmp", "tga_24_rle.tga", "tga_cmap24_rle.tga", "tga_cmap32_idx16.tga"])
def test_damaged_bmp_and_tga_files_are_decoded_or_rejected(tmp_path, name):
    """Random damage to valid BMP / TGA files: an image or an error, never an out-of-bounds access."""
    import random
    random.seed(11)
    orig = open(os.path.join(util.ROOT, "tests", "golden", "textures", name), "rb").read()
    p = str(tmp_path / name)
    ok = bad = 0
    for it in range(200):
        b = bytearray(orig)
        mode = it % 3
        if mode == 0:
            for _ in range(random.randint(1, 4)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif mode == 1:
            b = b[:random.randrange(1, len(b))]
        else:
            i = random.randrange(2, min(len(b) - 4, 60))
            b[i:i + 4] = random.getrandbits(32).to_bytes(4, "little")
        with open(p, "wb") as f:
            f.write(bytes(b))
        try:
            x, y, comp, a = crt.image_load(p)
            assert a.size == x * y * comp and 1 <= comp <= 4
            ok += 1
        except crt.CrtError:
            bad += 1
    assert ok + bad == 200 and ok > 20


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", [None, "jpg_420_16x16.jpg", "gif_local_palette_bg.gif", "psd_rgba8_rle_matte.psd", "hdr_rle.hdr", "pnm_p6_16.ppm",
                                     "pic_rgba_mixed_rle.pic"])
def test_textured_scene_renders_like_oracle(tmp_path, fixture):
    """A frame of a textured scene (PNG written here / a JPEG, GIF, PSD, HDR, PNM or PIC fixture whose samples the oracle takes from the
    reference's own decoder): the mean buffer equals the oracle's bit for bit."""
    obj, mtl = _write_scene(str(tmp_path), "rgb", 16, 16)
    if fixture:
        _use_fixture_texture(str(tmp_path), fixture)
    scene = crt.Scene(48, 36)
    scene.add_obj(obj, mtl)
    scene.set_BVH(2)
    osc = O.OracleScene([(obj, mtl)], 2)
    eye = np.array([3.5, 3.0, -2.0], dtype=np.float32)
    iv = crt.get_inverse_view_matrix(eye, [3.5, 0.0, 3.5], [0.0, 1.0, 0.0])
    fov = crt.fov_to_radians(60.0)
    r = crt.Render(scene, 4, 0.6, 2)
    try:
        rgb = r.run_view(eye, iv, fov)
        orgb, omean, _, st = osc.render(eye, iv, fov, 48, 36, 4, 0.6, 2)
        assert np.array_equal(util.bits(r.mean_buffer), util.bits(omean))
        assert np.array_equal(rgb, orgb) and r.stats["rays"] == st["rays"]
        assert len(np.unique(rgb.reshape(-1, 3), axis=0)) > (50 if fixture is None or fixture.endswith(".jpg") else 20)
    finally:
        r.free()
