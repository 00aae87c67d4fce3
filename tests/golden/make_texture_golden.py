#!/usr/bin/env python3
"""Writes the texture fixtures of tests/golden/textures/ and records in tests/golden/stb_decode.json what the REFERENCE's
own decoder returns for them (oracle/_ref/stb_probe = the vendored stb_image.h of /root/reference compiled where it lies,
oracle/Makefile target ref_probe).  Authoring container only; the fixtures and the JSON are committed data.

The files cover what a map_Kd may point at (reference: include/Loader.h:55-105): PNG colour types 0/2/3/4/6 at 8 and 16 bit,
1/2/4-bit palettes and greys, tRNS keys; BMP with 1/4/8-bit palettes, 16-bit 5-5-5 and 5-6-5 bit fields, 24 bit, 32 bit with and
without alpha, top-down rows, V4/V5 headers, OS/2 header; TGA types 1/2/3/9/10/11, 15/16/24/32 bit, both origins, colour maps
with 16/24/32-bit entries; Adam7-interlaced PNG at every depth; JPEG: baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey,
CMYK, restart intervals, own Huffman and quantisation tables, sizes around the MCU boundaries, RGB component ids, Adobe transform 0."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "textures")
PROBE = os.path.join(ROOT, "oracle", "_ref", "stb_probe")


def bmp(path, w, h, bpp, rows, palette=None, masks=None, hsz=40, top_down=False, alpha_mask=0):
    """rows: list of h byte strings (already packed, unpadded), in TOP-DOWN image order."""
    pal = b"".join(struct.pack("<BBB", b, g, r) + (b"" if hsz == 12 else b"\0") for (r, g, b) in (palette or []))
    compress = 3 if masks else 0
    if hsz == 12:
        hdr = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        hdr = struct.pack("<IiiHHIIiiII", hsz, w, -h if top_down else h, 1, bpp, compress, 0, 2835, 2835, len(palette or []), 0)
        if hsz == 40 and masks:
            hdr += struct.pack("<III", *masks)
        elif hsz in (108, 124):
            m = masks or (0, 0, 0)
            hdr += struct.pack("<IIII", m[0], m[1], m[2], alpha_mask) + struct.pack("<I", 0x73524742) + b"\0" * 48
            if hsz == 124:
                hdr += b"\0" * 16
    data = b""
    order = rows if top_down else rows[::-1]
    for r in order:
        data += r + b"\0" * ((-len(r)) & 3)
    off = 14 + len(hdr) + len(pal)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off) + hdr + pal + data)


def tga(path, w, h, typ, bpp, pixels, cmap=None, cmap_bits=0, top_down=False, rle=False, id_bytes=b""):
    """pixels: list of w*h byte strings (one per pixel, file byte order), in FILE row order."""
    hdr = struct.pack("<BBBHHBHHHHBB", len(id_bytes), 1 if cmap else 0, typ + (8 if rle else 0), 0, len(cmap or []), cmap_bits, 0, 0, w, h, bpp,
                      0x20 if top_down else 0)
    body = b"".join(cmap or [])
    if not rle:
        body += b"".join(pixels)
    else:
        i = 0
        while i < len(pixels):
            run = 1
            while i + run < len(pixels) and run < 128 and pixels[i + run] == pixels[i]:
                run += 1
            if run > 1:
                body += bytes([0x80 | (run - 1)]) + pixels[i]
                i += run
            else:
                raw = 1
                while i + raw < len(pixels) and raw < 128 and (i + raw + 1 >= len(pixels) or pixels[i + raw] != pixels[i + raw + 1]):
                    raw += 1
                body += bytes([raw - 1]) + b"".join(pixels[i:i + raw])
                i += raw
    with open(path, "wb") as f:
        f.write(hdr + id_bytes + body)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    w, h = 13, 7
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    alpha = rng.integers(0, 256, (h, w), dtype=np.uint8)
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    rgb[2, 3:9] = rgb[2, 3]          # runs for the RLE encoders
    grey[4, 1:8] = grey[4, 1]
    P = lambda n: os.path.join(OUT, n)
    # ---- PNG (PIL writes them) ----
    Image.fromarray(rgb, "RGB").save(P("png_rgb8.png"))
    Image.fromarray(np.dstack([rgb, alpha]), "RGBA").save(P("png_rgba8.png"))
    Image.fromarray(grey, "L").save(P("png_grey8.png"))
    Image.fromarray(np.dstack([grey, alpha]), "LA").save(P("png_grey_alpha8.png"))
    Image.fromarray(rng.integers(0, 65536, (h, w), dtype=np.uint16)).save(P("png_grey16.png"))
    Image.fromarray(rgb, "RGB").quantize(37).save(P("png_palette8.png"))
    Image.fromarray(rgb, "RGB").quantize(11).save(P("png_palette4.png"), bits=4)
    Image.fromarray(rgb, "RGB").quantize(3).save(P("png_palette2.png"), bits=2)
    Image.fromarray((grey > 127).astype(np.uint8) * 255, "L").convert("1").save(P("png_grey1.png"))
    q = Image.fromarray(rgb, "RGB").quantize(9)
    q.save(P("png_palette_trns.png"), transparency=bytes([255, 0, 128, 255, 7, 255, 255, 255, 64]))
    Image.fromarray(rgb, "RGB").save(P("png_rgb8_trns_key.png"), transparency=tuple(int(v) for v in rgb[1, 1]))
    Image.fromarray(grey, "L").save(P("png_grey8_trns_key.png"), transparency=int(grey[0, 0]))
    yy, xx = np.mgrid[0:48, 0:64]
    big = np.stack([(xx * 3 + yy) % 256, (xx ^ yy) % 256, (yy * 5) % 256], axis=2).astype(np.uint8)
    Image.fromarray(big, "RGB").save(P("png_rgb8_filtered.png"), optimize=True)
    # 16-bit RGB / RGBA: PIL cannot write them; a minimal encoder
    import zlib

    def png16(path, arr, color_type):
        hh, ww, ch = arr.shape
        raw = b"".join(b"\0" + arr[y].astype(">u2").tobytes() for y in range(hh))

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        with open(path, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", ww, hh, 16, color_type, 0, 0, 0)) +
                    chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    png16(P("png_rgb16.png"), rng.integers(0, 65536, (h, w, 3), dtype=np.uint16), 2)
    png16(P("png_rgba16.png"), rng.integers(0, 65536, (h, w, 4), dtype=np.uint16), 6)
    # ---- BMP ----
    rows24 = [b"".join(bytes([p[2], p[1], p[0]]) for p in rgb[y]) for y in range(h)]
    bmp(P("bmp_24.bmp"), w, h, 24, rows24)
    bmp(P("bmp_24_topdown.bmp"), w, h, 24, rows24, top_down=True)
    bmp(P("bmp_24_os2.bmp"), w, h, 24, rows24, hsz=12)
    rows32 = [b"".join(bytes([p[2], p[1], p[0], a]) for p, a in zip(rgb[y], alpha[y])) for y in range(h)]
    bmp(P("bmp_32_alpha.bmp"), w, h, 32, rows32)
    rows32z = [b"".join(bytes([p[2], p[1], p[0], 0]) for p in rgb[y]) for y in range(h)]
    bmp(P("bmp_32_alpha_all_zero.bmp"), w, h, 32, rows32z)
    bmp(P("bmp_32_bitfields_xbgr.bmp"), w, h, 32, [b"".join(struct.pack("<I", (int(p[0]) << 0) | (int(p[1]) << 8) | (int(p[2]) << 16)) for p in rgb[y])
                                                    for y in range(h)], masks=(0xff, 0xff00, 0xff0000))
    bmp(P("bmp_32_v5_alpha_mask.bmp"), w, h, 32, [b"".join(struct.pack("<I", (int(a) << 24) | (int(p[0]) << 16) | (int(p[1]) << 8) | int(p[2]))
                                                          for p, a in zip(rgb[y], alpha[y])) for y in range(h)],
        masks=(0xff0000, 0xff00, 0xff), hsz=124, alpha_mask=0xff000000)
    v16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    rows16 = [b"".join(struct.pack("<H", int(v)) for v in v16[y]) for y in range(h)]
    bmp(P("bmp_16_555.bmp"), w, h, 16, rows16)
    bmp(P("bmp_16_565.bmp"), w, h, 16, rows16, masks=(0xf800, 0x07e0, 0x001f))
    bmp(P("bmp_16_v4_4444.bmp"), w, h, 16, rows16, masks=(0x0f00, 0x00f0, 0x000f), hsz=108, alpha_mask=0xf000)
    pal = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(200)]
    idx8 = rng.integers(0, 200, (h, w), dtype=np.uint8)
    bmp(P("bmp_8_palette.bmp"), w, h, 8, [bytes(idx8[y]) for y in range(h)], palette=pal)
    # (stb_image sizes an OS/2 palette as (offset - 14 - 24) / 3, four entries short: indices beyond it read its uninitialised stack)
    bmp(P("bmp_8_palette_os2.bmp"), w, h, 8, [bytes(np.minimum(idx8[y], 189)) for y in range(h)], palette=pal, hsz=12)
    idx4 = rng.integers(0, 16, (h, w), dtype=np.uint8)
    bmp(P("bmp_4_palette.bmp"), w, h, 4, [bytes((int(idx4[y, i]) << 4) | (int(idx4[y, i + 1]) if i + 1 < w else 0) for i in range(0, w, 2)) for y in range(h)],
        palette=pal[:16])
    idx1 = rng.integers(0, 2, (h, w), dtype=np.uint8)
    bmp(P("bmp_1_palette.bmp"), w, h, 1, [bytes(sum(int(idx1[y, i + k]) << (7 - k) for k in range(8) if i + k < w) for i in range(0, w, 8)) for y in range(h)],
        palette=pal[:2])
    # ---- TGA ----
    flat = lambda f: [f(y, x) for y in range(h) for x in range(w)]
    bgr = lambda y, x: bytes([rgb[y, x, 2], rgb[y, x, 1], rgb[y, x, 0]])
    bgra = lambda y, x: bytes([rgb[y, x, 2], rgb[y, x, 1], rgb[y, x, 0], alpha[y, x]])
    tga(P("tga_24.tga"), w, h, 2, 24, flat(bgr))
    tga(P("tga_24_topdown.tga"), w, h, 2, 24, flat(bgr), top_down=True)
    tga(P("tga_24_rle.tga"), w, h, 2, 24, flat(bgr), rle=True)
    tga(P("tga_32.tga"), w, h, 2, 32, flat(bgra), id_bytes=b"fixture")
    tga(P("tga_32_rle_topdown.tga"), w, h, 2, 32, flat(bgra), rle=True, top_down=True)
    tga(P("tga_16.tga"), w, h, 2, 16, flat(lambda y, x: struct.pack("<H", int(v16[y, x]))))
    tga(P("tga_15_rle.tga"), w, h, 2, 15, flat(lambda y, x: struct.pack("<H", int(v16[y, x]) & 0x7fff)), rle=True)
    tga(P("tga_grey8.tga"), w, h, 3, 8, flat(lambda y, x: bytes([grey[y, x]])))
    tga(P("tga_grey8_rle.tga"), w, h, 3, 8, flat(lambda y, x: bytes([grey[y, x]])), rle=True)
    tga(P("tga_grey_alpha16.tga"), w, h, 3, 16, flat(lambda y, x: bytes([grey[y, x], alpha[y, x]])))
    cm24 = [bytes([b, g, r]) for (r, g, b) in pal[:64]]
    idxc = rng.integers(0, 64, (h, w), dtype=np.uint8)
    idxc[3, 2:10] = idxc[3, 2]
    tga(P("tga_cmap24.tga"), w, h, 1, 8, flat(lambda y, x: bytes([idxc[y, x]])), cmap=cm24, cmap_bits=24)
    tga(P("tga_cmap24_rle.tga"), w, h, 1, 8, flat(lambda y, x: bytes([idxc[y, x]])), cmap=cm24, cmap_bits=24, rle=True)
    tga(P("tga_cmap32_idx16.tga"), w, h, 1, 16, flat(lambda y, x: struct.pack("<H", int(idxc[y, x]))),
        cmap=[c + bytes([i * 4]) for i, c in enumerate(cm24)], cmap_bits=32)
    tga(P("tga_cmap16.tga"), w, h, 1, 8, flat(lambda y, x: bytes([idxc[y, x]])), cmap=[struct.pack("<H", int(v)) for v in v16.reshape(-1)[:64]], cmap_bits=16)
    # ---- interlaced PNG (Adam7; reference: stb_image.h:5116-5150) ----
    def png_adam7(path, arr, color_type, depth, palette=None, trns=None):
        """arr: (h, w, channels) integer samples of `depth` bits; written as an Adam7 file with a different filter on every line."""
        hh, ww, ch = arr.shape
        bpp = max(1, ch * depth // 8)
        raw = b""
        fl = 0
        for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = arr[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            prev = None
            for row in sub:
                if depth == 16:
                    line = row.astype(">u2").tobytes()
                elif depth == 8:
                    line = row.astype(np.uint8).tobytes()
                else:
                    bits = "".join(format(int(v), "0%db" % depth) for v in row.reshape(-1))
                    bits += "0" * ((-len(bits)) % 8)
                    line = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
                ft = fl % 5
                fl += 1
                pl = prev if prev is not None else bytes(len(line))
                out = bytearray()
                for i, v in enumerate(line):
                    a = line[i - bpp] if i >= bpp else 0
                    b = pl[i]
                    c = pl[i - bpp] if i >= bpp else 0
                    if ft == 0:
                        pr = 0
                    elif ft == 1:
                        pr = a
                    elif ft == 2:
                        pr = b
                    elif ft == 3:
                        pr = (a + b) >> 1
                    else:
                        pp = a + b - c
                        pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                        pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    out.append((v - pr) & 255)
                raw += bytes([ft]) + bytes(out)
                prev = line

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        body = chunk(b"IHDR", struct.pack(">IIBBBBB", ww, hh, depth, color_type, 0, 0, 1))
        if palette is not None:
            body += chunk(b"PLTE", bytes(palette))
        if trns is not None:
            body += chunk(b"tRNS", bytes(trns))
        with open(path, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n" + body + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    png_adam7(P("png_adam7_rgb8.png"), rgb, 2, 8)
    png_adam7(P("png_adam7_rgba8_3x2.png"), np.dstack([rgb, alpha])[:2, :3], 6, 8)          # smaller than a pass period: empty passes
    png_adam7(P("png_adam7_grey8_1x1.png"), grey[:1, :1, None], 0, 8)
    png_adam7(P("png_adam7_rgb16.png"), rng.integers(0, 65536, (9, 11, 3)), 2, 16)
    png_adam7(P("png_adam7_grey4.png"), rng.integers(0, 16, (h, w, 1)), 0, 4)
    png_adam7(P("png_adam7_grey1.png"), rng.integers(0, 2, (10, 19, 1)), 0, 1)
    png_adam7(P("png_adam7_palette2_trns.png"), rng.integers(0, 4, (h, w, 1)), 3, 2, palette=[255, 0, 0, 0, 255, 0, 0, 0, 255, 9, 99, 199], trns=[255, 128, 0])
    png_adam7(P("png_adam7_grey_alpha8.png"), np.dstack([grey, alpha]), 4, 8)
    Image.fromarray(big, "RGB").save(P("png_adam7_pil_rgb.png"), optimize=True)  # (PIL writes no interlaced files: a plain one of the same data as reference point)
    # ---- JPEG (PIL / libjpeg writes them; reference: stb_image.h:1960-4070) ----
    yy, xx = np.mgrid[0:37, 0:51]
    smooth = np.stack([128 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), (xx * 5 + yy * 3) % 256, 255 - (yy * 6) % 256], axis=2)
    photo = np.clip(smooth + rng.normal(0, 12, smooth.shape), 0, 255).astype(np.uint8)   # smooth + noise: every coefficient band in use
    J = lambda name, arr, mode="RGB", **kw: Image.fromarray(arr, mode).save(P(name), "JPEG", **kw)
    J("jpg_444_q90.jpg", photo, quality=90, subsampling=0)
    J("jpg_422_q85.jpg", photo, quality=85, subsampling=1)
    J("jpg_420_q75.jpg", photo, quality=75, subsampling=2)
    J("jpg_420_q30_opt.jpg", photo, quality=30, subsampling=2, optimize=True)           # own Huffman tables
    J("jpg_420_q100.jpg", photo, quality=100, subsampling=2)
    J("jpg_420_q3.jpg", photo, quality=3, subsampling=2)                               # coarse tables: values clamp
    J("jpg_grey_q80.jpg", photo[:, :, 0], "L", quality=80)
    J("jpg_grey_prog.jpg", photo[:, :, 1], "L", quality=70, progressive=True)
    J("jpg_444_prog.jpg", photo, quality=88, subsampling=0, progressive=True)
    J("jpg_420_prog.jpg", photo, quality=60, subsampling=2, progressive=True)            # DC / AC first and refinement scans, EOB runs
    J("jpg_422_prog_opt.jpg", photo, quality=92, subsampling=1, progressive=True, optimize=True)
    J("jpg_420_restart_rows.jpg", photo, quality=80, subsampling=2, restart_marker_rows=1)
    J("jpg_444_restart_blocks.jpg", photo, quality=80, subsampling=0, restart_marker_blocks=3)
    J("jpg_420_prog_restart.jpg", photo, quality=80, subsampling=2, progressive=True, restart_marker_blocks=2)
    for (ww, hh) in ((1, 1), (2, 3), (7, 5), (8, 8), (9, 17), (16, 16), (17, 16), (33, 31)):
        J("jpg_420_%dx%d.jpg" % (ww, hh), np.ascontiguousarray(photo[:hh, :ww]), quality=85, subsampling=2)
    J("jpg_422_1x9.jpg", np.ascontiguousarray(photo[:9, :1]), quality=85, subsampling=1)
    J("jpg_444_random.jpg", rng.integers(0, 256, (24, 40, 3), dtype=np.uint8), quality=95, subsampling=0)   # noise: long codes, large coefficients
    J("jpg_420_random_q50.jpg", rng.integers(0, 256, (24, 40, 3), dtype=np.uint8), quality=50, subsampling=2)
    Image.fromarray(photo, "RGB").convert("CMYK").save(P("jpg_cmyk.jpg"), "JPEG", quality=85)             # Adobe APP14, four components
    q16 = [min(255, 3 + 4 * i) for i in range(64)]
    J("jpg_444_custom_qtables.jpg", photo, qtables=[q16, [min(255, 5 + 3 * i) for i in range(64)]], subsampling=0)
    # hand-patched variants of jpg_444_q90.jpg: component ids 'R' 'G' 'B' (stb_image then copies the planes instead of converting),
    # and the same with the JFIF segment renamed plus an Adobe APP14 segment with transform 0
    base = open(P("jpg_444_q90.jpg"), "rb").read()

    def patch_ids(b, ids):
        b = bytearray(b)
        i = 2
        while i < len(b):
            assert b[i] == 0xFF
            m = b[i + 1]
            L = (b[i + 2] << 8) | b[i + 3]
            if m in (0xC0, 0xC1, 0xC2):
                for k in range(3):
                    b[i + 10 + 3 * k] = ids[k]
            if m == 0xDA:
                for k in range(3):
                    b[i + 5 + 2 * k] = ids[k]
                break
            i += 2 + L
        return bytes(b)
    open(P("jpg_444_rgb_ids.jpg"), "wb").write(patch_ids(base, b"RGB"))
    adobe = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
    nojfif = base.replace(b"JFIF\x00", b"JFXX\x00", 1)
    open(P("jpg_444_adobe_rgb.jpg"), "wb").write(nojfif[:2] + adobe + nojfif[2:])
    # 4:1:1-like ratios that take the pixel-replication path (h = 4): libjpeg writes them with custom sampling factors
    try:
        J("jpg_411.jpg", photo, quality=85, subsampling="4:1:1")
    except Exception as e:  # PIL versions without that name
        print("no 4:1:1 fixture:", e, file=sys.stderr)
    # ---- what the reference's decoder says ----
    files = sorted(os.listdir(OUT))
    import hashlib
    import tempfile
    with tempfile.TemporaryDirectory() as dump:
        out = subprocess.check_output([PROBE] + [os.path.join(OUT, f) for f in files], env=dict(os.environ, STB_PROBE_DUMP=dump))
        gold = json.loads(out)
        # the JPEG files in full: no second decoder returns stb_image's samples for them (inverse DCT, upsampling and colour
        # arithmetic are the decoder's own), so the oracle of the tests is fed from here -- keyed by the SHA-1 of the file
        full = {}
        for f in files:
            if f.endswith(".jpg") and "error" not in gold[f]:
                g = gold[f]
                a = np.fromfile(os.path.join(dump, f + ".raw"), dtype=np.uint8).reshape(g["y"], g["x"], g["comp"])
                full[hashlib.sha1(open(os.path.join(OUT, f), "rb").read()).hexdigest()] = a
        np.savez_compressed(os.path.join(HERE, "stb_jpeg_samples.npz"), **full)
    bad = {k: v for k, v in gold.items() if "error" in v}
    if bad:
        print("stb_image rejects:", bad, file=sys.stderr)
    with open(os.path.join(HERE, "stb_decode.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_texture_golden.py", "decoder": "reference include/stb_image.h via oracle/ref_probe/stb_probe.c: "
                   "stbi_load(path, &x, &y, &comp, 0) as include/Loader.h:58 calls it", "hash": "FNV-1a 64 over the x*y*comp samples", "files": gold},
                  f, indent=1, sort_keys=True)
    print("%d fixtures, %d decoded by stb_image" % (len(files), len(files) - len(bad)))


if __name__ == "__main__":
    main()
