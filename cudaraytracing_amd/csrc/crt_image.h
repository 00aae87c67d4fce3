// cudaraytracing_amd/csrc/crt_image.h -- texture decoding for map_Kd (host layer).
//
// The reference decodes textures with its vendored stb_image: stbi_load(path, &height, &width, &channel, 0)
// (include/Loader.h:58), i.e. 8-bit samples, top row first, the file's own channel count.  This header returns the same
// (x, y, comp, samples) for every format that decoder reads:
//   PNG  own inflate / unfilter, plain and Adam7-interlaced (crt_png.h);
//   JPEG baseline and progressive, stb_image's inverse DCT / upsampling / colour arithmetic (crt_jpeg.h);
//   BMP  uncompressed: 1 / 4 / 8-bit palettes, 16-bit and 32-bit with channel masks (BI_BITFIELDS or the defaults), 24-bit;
//        RLE and embedded PNG / JPEG are rejected as stb_image rejects them;
//   TGA  types 1 / 2 / 3 and their run-length forms 9 / 10 / 11: 8-bit grey, 16-bit grey + alpha, 15 / 16-bit RGB (5-5-5),
//        24 / 32-bit, colour-mapped with 8 or 16-bit indices; origin bit honoured;
//   GIF (first frame), PSD, Softimage PIC, binary PNM, Radiance HDR (crt_formats.h).
// Pinned against the reference's own decoder: oracle/ref_probe/stb_probe.c compiles the vendored stb_image.h where it lies
// and tests/golden/stb_decode.json holds what it returns for the fixture files of tests/golden/textures/.
#ifndef CRT_IMAGE_H
#define CRT_IMAGE_H

#include "crt_formats.h"
#include "crt_jpeg.h"
#include "crt_png.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace crtimg {

typedef crtpng::Image Image; // width, height, comp, px (8-bit samples, row 0 = top)

namespace detail {

// bytes of a file; reads past the end give 0 (as the reference's decoder does with a truncated file)
struct Bytes {
    std::vector<uint8_t> d;
    size_t pos = 0;
    int u8() { return pos < d.size() ? d[pos++] : (pos++, 0); }
    int u16() { int a = u8(); return a | (u8() << 8); }
    uint32_t u32() { uint32_t a = (uint32_t)u16(); return a | ((uint32_t)u16() << 16); }
    void skip(long n) { if (n > 0) pos += (size_t)n; }
};

const int kMaxDim = 1 << 24;

// an n-bit channel value -> 8 bits by repeating its bit pattern (what the BMP loader of stb_image computes with its
// multiply / shift tables)
inline int widen_bits(unsigned v, int n)
{
    if (n <= 0) return 0;
    unsigned r = 0;
    for (int s = 8 - n; s > -n; s -= n) r |= s >= 0 ? v << s : v >> -s;
    return (int)(r & 255u);
}
inline int top_bit(uint32_t m) { int n = -1; while (m) { n++; m >>= 1; } return n; }
inline int bit_count(uint32_t m) { int n = 0; while (m) { n += (int)(m & 1u); m >>= 1; } return n; }
// the channel of `v` under `mask`, brought to 8 bits
inline int masked_channel(uint32_t v, uint32_t mask)
{
    const int hi = top_bit(mask), n = bit_count(mask);
    uint32_t x = v & mask;
    const int sh = hi - 7; // aligns the mask's top bit with bit 7
    x = sh < 0 ? x << -sh : x >> sh;
    return widen_bits((x & 255u) >> (8 - n), n);
}

inline std::string load_bmp(Bytes& b, Image& img)
{
    b.pos = 2;
    b.u32(); b.u16(); b.u16();                 // file size, reserved
    const long data_off = (long)(int32_t)b.u32();
    const int hsz = (int)b.u32();
    if (data_off < 0) return "corrupt BMP";
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return "BMP header of an unknown size";
    long w, h;
    if (hsz == 12) { w = b.u16(); h = b.u16(); } else { w = (long)(int32_t)b.u32(); h = (long)(int32_t)b.u32(); }
    if (b.u16() != 1) return "corrupt BMP (planes)";
    const int bpp = b.u16();
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
    bool alpha_may_be_unused = false; // 32-bit with the DEFAULT masks: an all-zero alpha channel means "no alpha"
    long extra = 14;
    auto default_masks = [&]() {
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; alpha_may_be_unused = true; }
        else mr = mg = mb = ma = 0;
    };
    if (hsz != 12) {
        const int compress = (int)b.u32();
        if (compress == 1 || compress == 2) return "run-length encoded BMP (not decoded by the reference's stb_image either)";
        if (compress >= 4) return "BMP with embedded JPEG / PNG";
        if (compress == 3 && bpp != 16 && bpp != 32) return "corrupt BMP (bit fields)";
        b.skip(20);                            // image size, resolution, colours used / important
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) b.skip(16);
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) default_masks();
                else { mr = b.u32(); mg = b.u32(); mb = b.u32(); extra += 12; if (mr == mg && mg == mb) return "corrupt BMP (masks)"; }
            }
        } else {                               // V4 / V5 header
            mr = b.u32(); mg = b.u32(); mb = b.u32(); ma = b.u32();
            if (compress != 3) default_masks();
            b.skip(4 + 48);
            if (hsz == 124) b.skip(16);
        }
    }
    const bool bottom_up = h > 0;
    if (h < 0) h = -h;
    if (w > kMaxDim || h > kMaxDim || w <= 0 || h <= 0) return "BMP dimensions";
    long psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (data_off - extra - 24) / 3; }
    else if (bpp < 16) psize = (data_off - extra - hsz) >> 2;
    if (psize == 0) {
        const long so_far = (long)b.pos;
        if (so_far <= 0 || so_far > 1024) return "corrupt BMP (header)";
        if (data_off < so_far || data_off - so_far > 1024) return "corrupt BMP (data offset)";
        b.skip(data_off - so_far);
    }
    const int comp = (bpp == 24 && ma == 0xff000000u) ? 3 : (ma ? 4 : 3);
    if ((uint64_t)w * (uint64_t)h * (uint64_t)comp > (1ull << 31)) return "BMP too large";
    if ((uint64_t)w * (uint64_t)h > 128ull * b.d.size()) return "corrupt BMP (far more pixels than the file can hold)"; // (stb_image would decode zeros)
    std::vector<uint8_t> out((size_t)w * h * comp);
    size_t z = 0;
    unsigned alpha_or = alpha_may_be_unused ? 0u : 255u;
    if (bpp < 16) {
        if (psize <= 0 || psize > 256) return "corrupt BMP (palette)";
        uint8_t pal[256][3] = {};
        for (long i = 0; i < psize; i++) {
            pal[i][2] = (uint8_t)b.u8(); pal[i][1] = (uint8_t)b.u8(); pal[i][0] = (uint8_t)b.u8();
            if (hsz != 12) b.u8();
        }
        b.skip(data_off - extra - hsz - psize * (hsz == 12 ? 3 : 4));
        if (bpp != 1 && bpp != 4 && bpp != 8) return "corrupt BMP (bits per pixel)";
        const long row_bytes = (w * bpp + 7) / 8, pad = (-row_bytes) & 3;
        for (long j = 0; j < h; j++) {
            int cur = 0;
            for (long i = 0; i < w; i++) {
                const int in_byte = (int)((i * bpp) & 7);
                if (in_byte == 0) cur = b.u8();
                const int idx = bpp == 8 ? cur : (cur >> (8 - bpp - in_byte)) & ((1 << bpp) - 1);
                const uint8_t* c = pal[idx]; // (an index beyond the palette reads black here; stb_image reads uninitialised stack)
                out[z++] = c[0]; out[z++] = c[1]; out[z++] = c[2];
                if (comp == 4) out[z++] = 255;
            }
            b.skip(pad);
        }
    } else {
        b.skip(data_off - extra - hsz);
        const long row_bytes = bpp == 24 ? 3 * w : (bpp == 16 ? 2 * w : 0), pad = (-row_bytes) & 3;
        const bool direct = bpp == 24 || (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0xff0000u && ma == 0xff000000u);
        if (bpp != 16 && bpp != 24 && bpp != 32) return "corrupt BMP (bits per pixel)";
        if (!direct) {
            if (!mr || !mg || !mb) return "corrupt BMP (masks)";
            if (bit_count(mr) > 8 || bit_count(mg) > 8 || bit_count(mb) > 8 || bit_count(ma) > 8) return "corrupt BMP (masks)";
        }
        for (long j = 0; j < h; j++) {
            for (long i = 0; i < w; i++) {
                unsigned a;
                if (direct) {
                    const int bl = b.u8(), gr = b.u8(), rd = b.u8();
                    out[z++] = (uint8_t)rd; out[z++] = (uint8_t)gr; out[z++] = (uint8_t)bl;
                    a = bpp == 32 ? (unsigned)b.u8() : 255u;
                } else {
                    const uint32_t v = bpp == 16 ? (uint32_t)b.u16() : b.u32();
                    out[z++] = (uint8_t)masked_channel(v, mr); out[z++] = (uint8_t)masked_channel(v, mg); out[z++] = (uint8_t)masked_channel(v, mb);
                    a = ma ? (unsigned)masked_channel(v, ma) : 255u;
                }
                alpha_or |= a;
                if (comp == 4) out[z++] = (uint8_t)a;
            }
            b.skip(pad);
        }
    }
    if (comp == 4 && alpha_or == 0)
        for (size_t i = 3; i < out.size(); i += 4) out[i] = 255;
    img.width = (int)w; img.height = (int)h; img.comp = comp;
    img.px.resize(out.size());
    const size_t row = (size_t)w * comp;
    for (long j = 0; j < h; j++) // rows are stored bottom-up unless the height is negative
        std::copy(out.begin() + (size_t)j * row, out.begin() + (size_t)(j + 1) * row, img.px.begin() + (size_t)(bottom_up ? h - 1 - j : j) * row);
    return "";
}

// the sanity test that decides whether a file without a magic number is a TGA
inline bool looks_like_tga(Bytes b)
{
    b.pos = 0;
    b.u8();
    const int cmap = b.u8();
    if (cmap > 1) return false;
    int t = b.u8();
    if (cmap == 1) {
        if (t != 1 && t != 9) return false;
        b.skip(4);
        const int e = b.u8();
        if (e != 8 && e != 15 && e != 16 && e != 24 && e != 32) return false;
        b.skip(4);
    } else {
        if (t != 2 && t != 3 && t != 10 && t != 11) return false;
        b.skip(9);
    }
    if (b.u16() < 1 || b.u16() < 1) return false;
    const int bpp = b.u8();
    if (cmap == 1 && bpp != 8 && bpp != 16) return false;
    return bpp == 8 || bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32;
}

inline std::string load_tga(Bytes& b, Image& img)
{
    b.pos = 0;
    const int id_len = b.u8(), indexed = b.u8();
    int type = b.u8();
    const int pal_start = b.u16(), pal_len = b.u16(), pal_bits = b.u8();
    b.u16(); b.u16();                          // x / y origin
    const int w = b.u16(), h = b.u16(), bpp = b.u8(), desc = b.u8();
    const bool rle = type >= 8;
    if (rle) type -= 8;
    const bool bottom_up = ((desc >> 5) & 1) == 0;
    auto comp_of = [](int bits, bool grey, bool& rgb16) {
        rgb16 = false;
        if (bits == 8) return 1;
        if (bits == 16 && grey) return 2;
        if (bits == 15 || bits == 16) { rgb16 = true; return 3; }
        if (bits == 24 || bits == 32) return bits / 8;
        return 0;
    };
    bool rgb16 = false;
    const int comp = indexed ? comp_of(pal_bits, false, rgb16) : comp_of(bpp, type == 3, rgb16);
    if (!comp) return "TGA pixel format";
    if (w > kMaxDim || h > kMaxDim || w < 1 || h < 1) return "TGA dimensions";
    if ((uint64_t)w * (uint64_t)h > 128ull * b.d.size()) return "corrupt TGA (far more pixels than the file can hold)"; // (a run-length packet holds at most 128)
    std::vector<uint8_t> out((size_t)w * h * comp), pal;
    b.skip(id_len);
    auto rgb555 = [&](uint8_t* o) {
        const int px = b.u16();
        o[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31); o[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31); o[2] = (uint8_t)(((px & 31) * 255) / 31);
    };
    if (indexed) {
        if (pal_len == 0) return "corrupt TGA (palette)";
        b.skip(pal_start);
        pal.resize((size_t)pal_len * comp);
        if (rgb16) for (int i = 0; i < pal_len; i++) rgb555(&pal[(size_t)i * comp]);
        else for (size_t i = 0; i < pal.size(); i++) pal[i] = (uint8_t)b.u8();
    }
    uint8_t px[4] = {0, 0, 0, 0};
    int run = 0;
    bool repeat = false;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        bool read = true;
        if (rle) {
            if (run == 0) { const int c = b.u8(); run = 1 + (c & 127); repeat = (c >> 7) != 0; }
            else read = !repeat;
        }
        if (read) {
            if (indexed) {
                int idx = bpp == 8 ? b.u8() : b.u16();
                if (idx >= pal_len) idx = 0;
                for (int j = 0; j < comp; j++) px[j] = pal[(size_t)idx * comp + j];
            } else if (rgb16) rgb555(px);
            else for (int j = 0; j < comp; j++) px[j] = (uint8_t)b.u8();
        }
        for (int j = 0; j < comp; j++) out[i * comp + j] = px[j];
        run--;
    }
    if (comp >= 3 && !rgb16) // stored blue first
        for (size_t i = 0; i < (size_t)w * h; i++) std::swap(out[i * comp], out[i * comp + 2]);
    img.width = w; img.height = h; img.comp = comp;
    img.px.resize(out.size());
    const size_t row = (size_t)w * comp;
    for (int j = 0; j < h; j++)
        std::copy(out.begin() + (size_t)j * row, out.begin() + (size_t)(j + 1) * row, img.px.begin() + (size_t)(bottom_up ? h - 1 - j : j) * row);
    return "";
}

} // namespace detail

// Returns "" on success, else an error text ("cannot open ..." for a missing file).
inline std::string load(const std::string& path, Image& img)
{
    using namespace detail;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open texture " + path;
    Bytes b;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) b.d.insert(b.d.end(), buf, buf + n);
    std::fclose(f);
    const std::vector<uint8_t>& d = b.d;
    auto starts = [&](const char* m, size_t len) { return d.size() >= len && std::equal(m, m + len, (const char*)d.data()); };
    if (starts("\x89PNG\r\n\x1a\n", 8)) return crtpng::load(path, img);
    if (starts("BM", 2)) {
        const std::string e = load_bmp(b, img);
        return e.empty() ? e : e + ": " + path;
    }
    // (the order of the reference's decoder: PNG, BMP, GIF, PSD, PIC, JPEG, PNM, HDR and TGA last, stb_image.h:1136-1185)
    typedef std::string (*Loader)(const std::vector<uint8_t>&, Image&);
    Loader other = nullptr;
    if (starts("GIF87a", 6) || starts("GIF89a", 6)) other = crtfmt::load_gif;
    else if (starts("8BPS", 4)) other = crtfmt::load_psd;
    else if (crtfmt::looks_like_pic(d)) other = crtfmt::load_pic;
    else if (crtjpg::looks_like_jpeg(d)) other = crtjpg::decode;
    else if (d.size() > 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) other = crtfmt::load_pnm;
    else if (crtfmt::looks_like_hdr(d)) other = crtfmt::load_hdr;
    if (other) {
        const std::string e = other(d, img);
        return e.empty() ? e : e + ": " + path;
    }
    if (looks_like_tga(b)) {
        const std::string e = load_tga(b, img);
        return e.empty() ? e : e + ": " + path;
    }
    return "unknown texture format (PNG, JPEG, BMP, TGA, GIF, PSD, PIC, PNM and Radiance HDR are decoded): " + path;
}

} // namespace crtimg
#endif
