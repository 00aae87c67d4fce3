#!/usr/bin/env python3
"""Writes the texture fixtures of tests/golden/textures/ and records in tests/golden/stb_decode.json what the REFERENCE's
own decoder returns for them (oracle/_ref/stb_probe = the vendored stb_image.h of /root/reference compiled where it lies,
oracle/Makefile target ref_probe).  Authoring container only; the fixtures and the JSON are committed data.

The files cover what a map_Kd may point at (reference: include/Loader.h:55-105): PNG colour types 0/2/3/4/6 at 8 and 16 bit,
1/2/4-bit palettes and greys, tRNS keys; BMP with 1/4/8-bit palettes, 16-bit 5-5-5 and 5-6-5 bit fields, 24 bit, 32 bit with and
without alpha, top-down rows, V4/V5 headers, OS/2 header; TGA types 1/2/3/9/10/11, 15/16/24/32 bit, both origins, colour maps
with 16/24/32-bit entries; Adam7-interlaced PNG at every depth; JPEG: baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey,
CMYK, restart intervals, own Huffman and quantisation tables, sizes around the MCU boundaries, RGB component ids, Adobe transform 0;
binary PNM (8 / 16 bit, header comments); GIF (87a / 89a, interlaced, transparency, local palette, background fill, table resets, a
second frame that is not read); PSD (raw / PackBits, 8 / 16 bit, 1 - 5 channels, white-matte removal); Softimage PIC (raw, pure and
mixed run-length packets); Radiance HDR (run-length and flat scanlines, both signatures)."""
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
    # ---- PNM (reference: stb_image.h:7491-7630) ----
    open(P("pnm_p6.ppm"), "wb").write(b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes())
    open(P("pnm_p5.pgm"), "wb").write(b"P5 %d %d 255\n" % (w, h) + grey.tobytes())
    open(P("pnm_p6_comments.ppm"), "wb").write(b"P6\r\n# made by hand\n# second comment\r\n%d\t%d # trailing\n\n255\n" % (w, h) + rgb.tobytes())
    s16 = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    open(P("pnm_p6_16.ppm"), "wb").write(b"P6\n%d %d\n65535\n" % (w, h) + s16.astype(">u2").tobytes())
    open(P("pnm_p5_16_max1000.pgm"), "wb").write(b"P5\n%d %d\n1000\n" % (w, h) + (s16[:, :, 0] % 1001).astype(">u2").tobytes())
    open(P("pnm_p5_max15.pgm"), "wb").write(b"P5\n%d %d\n15\n" % (w, h) + (grey & 15).tobytes())
    # ---- GIF (reference: stb_image.h:6577-7080; only the first frame is read) ----

    def gif_lzw(indices, min_code):
        """Variable-width LZW as GIF packs it (least significant bit first), sub-blocks of at most 255 bytes."""
        clear, eoi = 1 << min_code, (1 << min_code) + 1
        out, acc, nbits = bytearray(), 0, 0
        size = min_code + 1

        def put(code):
            nonlocal acc, nbits
            acc |= code << nbits
            nbits += size
            while nbits >= 8:
                out.append(acc & 255)
                acc >>= 8
                nbits -= 8
        table, nxt = {}, eoi + 1
        put(clear)
        cur = ()
        for v in indices:
            v = int(v)
            t = cur + (v,)
            if len(t) == 1 or t in table:
                cur = t
                continue
            put(cur[0] if len(cur) == 1 else table[cur])
            table[t] = nxt
            nxt += 1
            if nxt > (1 << size) and size < 12:
                size += 1
            if nxt == 4096:
                put(clear)
                table, nxt, size = {}, eoi + 1, min_code + 1
            cur = (v,)
        if cur:
            put(cur[0] if len(cur) == 1 else table[cur])
        put(eoi)
        if nbits:
            out.append(acc & 255)
        blocks = b"".join(bytes([len(out[i:i + 255])]) + bytes(out[i:i + 255]) for i in range(0, len(out), 255))
        return bytes([min_code]) + blocks + b"\0"

    def gif(path, sw, sh, frames, gpal=None, bg=0, version=b"89a", extra=b""):
        """frames: list of dicts x, y, idx (2-D index array), optional lpal, transparent, interlace, dispose."""
        def table(pal):
            n = 2
            while n < len(pal):
                n *= 2
            return n, b"".join(bytes(c) for c in pal) + b"\0\0\0" * (n - len(pal))
        b = b"GIF" + version
        if gpal:
            n, t = table(gpal)
            b += struct.pack("<HHBBB", sw, sh, 0x80 | (n.bit_length() - 2), bg, 0) + t
        else:
            b += struct.pack("<HHBBB", sw, sh, 0, bg, 0)
        b += extra
        for fr in frames:
            idx = np.asarray(fr["idx"])
            fh, fw = idx.shape
            if "transparent" in fr or "dispose" in fr:
                tr = fr.get("transparent")
                b += b"\x21\xf9\x04" + bytes([(fr.get("dispose", 0) << 2) | (1 if tr is not None else 0)]) + struct.pack("<H", 7) + bytes([tr or 0]) + b"\0"
            pal = fr.get("lpal")
            flags = 0x40 if fr.get("interlace") else 0
            lt = b""
            ncol = len(pal or gpal)
            if pal:
                n, lt = table(pal)
                flags |= 0x80 | (n.bit_length() - 2)
            b += b"\x2c" + struct.pack("<HHHHB", fr.get("x", 0), fr.get("y", 0), fw, fh, flags) + lt
            rows = list(range(fh))
            if fr.get("interlace"):
                rows = [r for (a, st) in ((0, 8), (4, 8), (2, 4), (1, 2)) for r in range(a, fh, st)]
            mc = max(2, (max(ncol, 2) - 1).bit_length())
            b += gif_lzw(idx[rows].reshape(-1), fr.get("min_code", mc))
        open(path, "wb").write(b + b"\x3b")
    pal64 = pal[:64]
    i64 = rng.integers(0, 64, (h, w))
    i64[2, 2:11] = i64[2, 2]
    gif(P("gif_global_palette.gif"), w, h, [dict(idx=i64)], gpal=pal64)
    gif(P("gif_87a.gif"), w, h, [dict(idx=i64)], gpal=pal64, version=b"87a")
    gif(P("gif_interlaced.gif"), w, 19, [dict(idx=rng.integers(0, 64, (19, w)), interlace=True)], gpal=pal64)
    gif(P("gif_transparent.gif"), w, h, [dict(idx=i64, transparent=int(i64[0, 0]))], gpal=pal64)
    # a frame smaller than the screen, its own palette, a transparent index and a background index: what the frame leaves
    # untouched gets the background entry
    gif(P("gif_local_palette_bg.gif"), w, h, [dict(idx=rng.integers(0, 8, (4, 6)), x=3, y=2, lpal=pal[100:108], transparent=5)], gpal=pal64, bg=9)
    gif(P("gif_bg_is_transparent.gif"), w, h, [dict(idx=rng.integers(0, 64, (3, 5)), x=1, y=1, transparent=9)], gpal=pal64, bg=9)
    gif(P("gif_two_frames.gif"), w, h, [dict(idx=i64, dispose=2), dict(idx=rng.integers(0, 64, (h, w)), dispose=1)], gpal=pal64,
        extra=b"\x21\xff\x0bNETSCAPE2.0\x03\x01\x00\x00\x00" + b"\x21\xfe\x05hello\x00")
    gif(P("gif_two_colours.gif"), 19, 10, [dict(idx=rng.integers(0, 2, (10, 19)))], gpal=[(0, 0, 0), (255, 255, 255)])
    gif(P("gif_noise_table_resets.gif"), 96, 96, [dict(idx=rng.integers(0, 256, (96, 96)))], gpal=[tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(256)])
    gif(P("gif_smooth_long_strings.gif"), 80, 60, [dict(idx=(np.mgrid[0:60, 0:80][1] // 9 + np.mgrid[0:60, 0:80][0] // 17) % 16, interlace=True)], gpal=pal[:16])
    # ---- PSD (reference: stb_image.h:6078-6330) ----

    def packbits(row):
        out, i = bytearray(), 0
        while i < len(row):
            run = 1
            while i + run < len(row) and run < 128 and row[i + run] == row[i]:
                run += 1
            if run >= 3:
                out += bytes([257 - run, row[i]])
                i += run
                continue
            j = i
            while j < len(row) and j - i < 128 and not (j + 2 < len(row) and row[j] == row[j + 1] == row[j + 2]):
                j += 1
            out += bytes([j - i - 1]) + bytes(row[i:j])
            i = j
        return bytes(out)

    def psd(path, planes, depth=8, rle=False, resources=b""):
        ch, hh, ww = planes.shape
        b = b"8BPS" + struct.pack(">H6xHIIHH", 1, ch, hh, ww, depth, 3) + struct.pack(">I", 0) + struct.pack(">I", len(resources)) + resources + struct.pack(">I", 0)
        if not rle:
            b += struct.pack(">H", 0) + (planes.astype(">u2").tobytes() if depth == 16 else planes.astype(np.uint8).tobytes())
        else:
            rows = [packbits(bytes(planes[c, y].astype(np.uint8))) for c in range(ch) for y in range(hh)]
            b += struct.pack(">H", 1) + b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows)
        open(path, "wb").write(b)
    planes = np.stack([rgb[:, :, 0], rgb[:, :, 1], rgb[:, :, 2]])
    a_mixed = alpha.copy()
    a_mixed[0, :4] = 0
    a_mixed[1, :4] = 255
    a_mixed[3, 2:9] = a_mixed[3, 2]
    psd(P("psd_rgb8_raw.psd"), planes, resources=b"8BIM\x03\xed\0\0\0\0\0\x04abcd")
    psd(P("psd_rgb8_rle.psd"), planes, rle=True)
    psd(P("psd_rgba8_raw_matte.psd"), np.concatenate([planes, a_mixed[None]]))
    psd(P("psd_rgba8_rle_matte.psd"), np.concatenate([planes, a_mixed[None]]), rle=True)
    psd(P("psd_rgb16_raw.psd"), np.moveaxis(s16, 2, 0), depth=16)
    psd(P("psd_rgba16_raw.psd"), np.concatenate([np.moveaxis(s16, 2, 0), (a_mixed.astype(np.uint16) * 257)[None]]), depth=16)
    psd(P("psd_one_channel.psd"), grey[None])
    psd(P("psd_five_channels_rle.psd"), np.concatenate([planes, a_mixed[None], grey[None]]), rle=True)
    # ---- Softimage PIC (reference: stb_image.h:6343-6545) ----

    def pic(path, img4, packets):
        """img4: (h, w, 4) RGBA; packets: list of (type, channel mask)."""
        hh, ww, _ = img4.shape
        b = b"\x53\x80\xf6\x34" + struct.pack(">f", 3.71) + b"fixture".ljust(80, b"\0") + b"PICT" + struct.pack(">HHfHH", ww, hh, 1.0, 3, 0)
        for k, (typ, mask) in enumerate(packets):
            b += bytes([1 if k + 1 < len(packets) else 0, 8, typ, mask])
        for y in range(hh):
            for typ, mask in packets:
                chans = [c for c in range(4) if mask & (0x80 >> c)]
                px = [bytes(int(img4[y, x, c]) for c in chans) for x in range(ww)]
                if typ == 0:
                    b += b"".join(px)
                    continue
                x = 0
                while x < ww:
                    run = 1
                    while x + run < ww and px[x + run] == px[x] and run < (255 if typ == 1 else 300):
                        run += 1
                    if typ == 1:
                        b += bytes([run]) + px[x]
                        x += run
                    elif run >= 2:
                        b += (bytes([127 + run]) if run <= 128 else b"\x80" + struct.pack(">H", run)) + px[x]
                        x += run
                    else:
                        j = x
                        while j < ww and j - x < 128 and not (j + 1 < ww and px[j] == px[j + 1]):
                            j += 1
                        b += bytes([j - x - 1]) + b"".join(px[x:j])
                        x = j
        open(path, "wb").write(b)
    rgba = np.dstack([rgb, a_mixed])
    wide = np.zeros((5, 300, 4), dtype=np.uint8)
    wide[:, :, :3] = (np.arange(300)[None, :, None] // 150 * 200 + np.arange(5)[:, None, None] * 7) % 256   # runs longer than 128
    wide[:, :, 3] = 255
    wide[2, 100:103, 1] = (9, 8, 7)
    pic(P("pic_rgb_raw.pic"), rgba, [(0, 0xE0)])
    pic(P("pic_rgb_pure_rle.pic"), rgba, [(1, 0xE0)])
    pic(P("pic_rgba_mixed_rle.pic"), rgba, [(2, 0xE0), (1, 0x10)])
    pic(P("pic_channels_in_separate_packets.pic"), rgba, [(2, 0x80), (0, 0x40), (1, 0x20), (2, 0x10)])
    pic(P("pic_long_runs.pic"), wide, [(2, 0xE0)])
    # ---- Radiance HDR (reference: stb_image.h:7084-7290, 1883-1910) ----

    def rgbe(f):
        """float RGB (h, w, 3) -> RGBE bytes (h, w, 4)"""
        m = f.max(axis=2)
        e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0)
        sc = np.where(m > 1e-32, 256.0 / np.exp2(e), 0)
        out = np.zeros(f.shape[:2] + (4,), dtype=np.uint8)
        out[:, :, :3] = np.clip(f * sc[:, :, None], 0, 255).astype(np.uint8)
        out[:, :, 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
        return out

    def hdr(path, px, rle=True, magic=b"#?RADIANCE", rle_rows=None):
        hh, ww, _ = px.shape
        b = magic + b"\n# fixture\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (hh, ww)
        for y in range(hh):
            if not rle or (rle_rows is not None and y not in rle_rows):
                b += px[y].tobytes()
                continue
            b += bytes([2, 2, ww >> 8, ww & 255])
            for k in range(4):
                row, x = px[y, :, k], 0
                while x < ww:
                    run = 1
                    while x + run < ww and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        b += bytes([128 + run, int(row[x])])
                        x += run
                    else:
                        j = x
                        while j < ww and j - x < 128 and not (j + 2 < ww and row[j] == row[j + 1] == row[j + 2]):
                            j += 1
                        b += bytes([j - x]) + bytes(row[x:j])
                        x = j
        open(path, "wb").write(b)
    yy, xx = np.mgrid[0:11, 0:23]
    lum = np.stack([np.exp2((xx - 11) / 2.0) * (1 + yy / 5.0), 0.02 + xx * yy / 40.0, np.where((xx + yy) % 5 == 0, 0.0, 3.0 / (1 + yy))], axis=2)
    lum[4, 3:12] = lum[4, 3]
    lum[7, :, :] = 0.0
    e = rgbe(lum)
    hdr(P("hdr_rle.hdr"), e)
    hdr(P("hdr_rgbe_magic.hdr"), e, magic=b"#?RGBE")
    hdr(P("hdr_flat_narrow.hdr"), e[:, :7], rle=False)                 # narrower than 8: never run-length coded
    hdr(P("hdr_flat_wide.hdr"), e, rle=False)                          # flat data where run-length scanlines may stand
    hdr(P("hdr_flat_after_rle_rows.hdr"), e, rle_rows={0, 1})          # the third scanline is flat: the decoder restarts at pixel 1
    noise = rng.integers(0, 256, (9, 16, 4), dtype=np.uint8)
    noise[:, :, 3] = rng.integers(120, 136, (9, 16))
    noise[0, 0] = (200, 10, 10, 128)
    hdr(P("hdr_noise_rle.hdr"), noise)
    # ---- what the reference's decoder says ----
    files = sorted(os.listdir(OUT))
    import hashlib
    import tempfile
    with tempfile.TemporaryDirectory() as dump:
        out = subprocess.check_output([PROBE] + [os.path.join(OUT, f) for f in files], env=dict(os.environ, STB_PROBE_DUMP=dump))
        gold = json.loads(out)
        # the JPEG files in full: no second decoder returns stb_image's samples for them (inverse DCT, upsampling and colour
        # arithmetic are the decoder's own), so the oracle of the tests is fed from here -- keyed by the SHA-1 of the file; the GIF,
        # PSD, PIC, PNM and HDR files too (first-frame / matte / byte-order / tone-mapping conventions of that decoder)
        full = {}
        for f in files:
            if f.endswith((".jpg", ".gif", ".psd", ".pic", ".ppm", ".pgm", ".hdr")) and "error" not in gold[f]:
                g = gold[f]
                a = np.fromfile(os.path.join(dump, f + ".raw"), dtype=np.uint8).reshape(g["y"], g["x"], g["comp"])
                full[hashlib.sha1(open(os.path.join(OUT, f), "rb").read()).hexdigest()] = a
        np.savez_compressed(os.path.join(HERE, "stb_samples.npz"), **full)
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
