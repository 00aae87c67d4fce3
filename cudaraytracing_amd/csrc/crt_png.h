// cudaraytracing_amd/csrc/crt_png.h -- minimal PNG reader for map_Kd textures (host layer).
//
// The reference loads textures with stb_image (`stbi_load(path, &x, &y, &comp, 0)`, Loader.h:58), which is
// not part of this build.  This header decodes what the texture path needs: PNG (plain or Adam7-interlaced) of colour type
// 0 / 2 / 3 / 4 / 6, bit depth 8 (16: the high byte, as stb does; 1 / 2 / 4: palette and greyscale), with the
// component count stb would report for req_comp = 0 (grey 1, grey+alpha 2, RGB 3, RGBA 4; palette 3, or 4 with a
// tRNS chunk; a tRNS colour key adds an alpha component).  Other formats are reported as unsupported.
// Self-contained: own inflate (RFC 1951), no zlib.
#ifndef CRT_PNG_H
#define CRT_PNG_H

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace crtpng {

struct Image {
    int width = 0, height = 0, comp = 0; // as stbi_load's x, y, comp
    std::vector<uint8_t> px;             // height * width * comp, row-major, row 0 = top
};

namespace detail {

struct Bits {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t acc = 0;
    int cnt = 0;
    bool ok = true;
    uint32_t get(int k)
    {
        while (cnt < k) {
            if (pos >= n) { ok = false; return 0; }
            acc |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        uint32_t v = acc & ((k == 32) ? 0xffffffffu : ((1u << k) - 1u));
        acc = k == 32 ? 0 : acc >> k;
        cnt -= k;
        return v;
    }
    void align() { acc = 0; cnt = 0; }
};

struct Huff {
    uint16_t count[16];
    uint16_t symbol[288];
    bool build(const uint8_t* len, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(Bits& b) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)b.get(1);
            if (!b.ok) return -1;
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

inline bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    Bits b{src, n};
    int last;
    do {
        last = (int)b.get(1);
        int type = (int)b.get(2);
        if (!b.ok) return false;
        if (type == 0) {
            b.align();
            if (b.pos + 4 > n) return false;
            uint32_t len = src[b.pos] | (src[b.pos + 1] << 8), nlen = src[b.pos + 2] | (src[b.pos + 3] << 8);
            b.pos += 4;
            if ((len ^ 0xffffu) != nlen || b.pos + len > n) return false;
            out.insert(out.end(), src + b.pos, src + b.pos + len);
            b.pos += len;
        } else if (type == 1 || type == 2) {
            Huff hl, hd;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) lens[i] = 8;
                for (; i < 256; i++) lens[i] = 9;
                for (; i < 280; i++) lens[i] = 7;
                for (; i < 288; i++) lens[i] = 8;
                hl.build(lens, 288);
                for (i = 0; i < 30; i++) lens[i] = 5;
                hd.build(lens, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
                if (!b.ok || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19];
                std::memset(cl, 0, sizeof(cl));
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)b.get(3);
                Huff hc;
                if (!hc.build(cl, 19)) return false;
                int i = 0;
                while (i < nlen + ndist) {
                    int sym = hc.decode(b);
                    if (sym < 0) return false;
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int prev = 0, rep;
                        if (sym == 16) { if (i == 0) return false; prev = lens[i - 1]; rep = 3 + (int)b.get(2); }
                        else if (sym == 17) rep = 3 + (int)b.get(3);
                        else rep = 11 + (int)b.get(7);
                        if (!b.ok || i + rep > nlen + ndist) return false;
                        while (rep--) lens[i++] = (uint8_t)prev;
                    }
                }
                if (!hl.build(lens, nlen) || !hd.build(lens + nlen, ndist)) return false;
            }
            for (;;) {
                int sym = hl.decode(b);
                if (sym < 0) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    int len = lbase[sym] + (int)b.get(lext[sym]);
                    int ds = hd.decode(b);
                    if (ds < 0 || ds >= 30) return false;
                    size_t dist = dbase[ds] + b.get(dext[ds]);
                    if (!b.ok || dist > out.size()) return false;
                    size_t from = out.size() - dist;
                    for (int k = 0; k < len; k++) out.push_back(out[from + k]);
                }
            }
        } else return false;
    } while (!last);
    return true;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

} // namespace detail

// Returns an empty string on success, else what is wrong with the file.
inline std::string load(const std::string& path, Image& img)
{
    using namespace detail;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t r;
    while ((r = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + r);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return "not a PNG file (only PNG textures are supported): " + path;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return "truncated PNG chunk in " + path;
        const uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return "bad IHDR in " + path;
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) return "bad PNG header in " + path;
    if (interlace > 1) return "bad PNG interlace method in " + path;
    int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!chans) return "bad PNG colour type in " + path;
    const bool small = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || depth == 16 || (small && (ctype == 0 || ctype == 3)))) return "unsupported PNG bit depth in " + path;
    if (ctype == 3 && depth == 16) return "bad PNG palette depth in " + path;
    if (idat.size() < 6) return "PNG without image data: " + path;
    std::vector<uint8_t> raw;
    if (!inflate(idat.data() + 2, idat.size() - 2, raw)) return "corrupt PNG data stream in " + path; // 2-byte zlib header
    const size_t bpp_bits = (size_t)chans * depth, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    // raw sample values (up to 16 bits) of the whole image, samp[(y * w + x) * chans + c]: filled pass by pass.  A non-interlaced
    // file is one pass over every pixel; an Adam7 file (stb_image.h:5116-5150 reads them too) is seven reduced images, each with
    // its own filtered scan lines, whose pixels land at (x0 + i * dx, y0 + j * dy).
    static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    {   // the data stream must hold every scan line of every pass BEFORE the image is allocated (a damaged header may say 2^24 x 2^24)
        uint64_t need = 0;
        for (int pass = 0; pass < (interlace ? 7 : 1); pass++) {
            const uint32_t x0 = interlace ? ax0[pass] : 0, y0 = interlace ? ay0[pass] : 0, dx = interlace ? adx[pass] : 1, dy = interlace ? ady[pass] : 1;
            if (x0 >= w || y0 >= h) continue;
            const uint64_t pw = (w + dx - 1 - x0) / dx, ph = (h + dy - 1 - y0) / dy;
            need += ((pw * bpp_bits + 7) / 8 + 1) * ph;
        }
        if ((uint64_t)raw.size() < need) return "short PNG data stream in " + path;
        if ((uint64_t)w * h * (uint64_t)chans > 0x7fffffffull) return "PNG too large: " + path; // (the reference's decoder: sizes are ints)
    }
    std::vector<uint16_t> samp((size_t)w * h * chans);
    size_t rp = 0;
    for (int pass = 0; pass < (interlace ? 7 : 1); pass++) {
        const uint32_t x0 = interlace ? ax0[pass] : 0, y0 = interlace ? ay0[pass] : 0, dx = interlace ? adx[pass] : 1, dy = interlace ? ady[pass] : 1;
        const uint32_t pw = (w + dx - 1 - x0) / dx, ph = (h + dy - 1 - y0) / dy;
        if (x0 >= w || y0 >= h || pw == 0 || ph == 0) continue; // an empty pass has no bytes in the stream
        const size_t stride = (pw * bpp_bits + 7) / 8;
        if (raw.size() < rp + (stride + 1) * ph) return "short PNG data stream in " + path;
        std::vector<uint8_t> cur(stride), prev(stride, 0);
        for (uint32_t j = 0; j < ph; j++) {
            const uint8_t* s = &raw[rp + (stride + 1) * j];
            const int ft = s[0];
            if (ft > 4) return "bad PNG filter in " + path;
            for (size_t x = 0; x < stride; x++) {
                int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0, v = s[1 + x];
                switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
                }
                cur[x] = (uint8_t)v;
            }
            const uint8_t* line = cur.data();
            uint16_t* o = &samp[((size_t)(y0 + j * dy) * w + x0) * chans];
            for (uint32_t i = 0; i < pw; i++, o += (size_t)dx * chans)
                for (int c = 0; c < chans; c++) {
                    const size_t k = (size_t)i * chans + c;
                    if (depth == 8) o[c] = line[k];
                    else if (depth == 16) o[c] = (uint16_t)(line[2 * k] << 8 | line[2 * k + 1]);
                    else { const int per = 8 / depth; o[c] = (uint16_t)((line[k / per] >> ((per - 1 - (int)(k % per)) * depth)) & ((1 << depth) - 1)); }
                }
            prev = cur;
        }
        rp += (stride + 1) * ph;
    }
    // expand to 8 bits per component (16 -> 8: the high byte, as stb_image converts), then to the component count stb_image reports
    int comp = chans;
    if (ctype == 3) comp = trns.empty() ? 3 : 4;
    else if (!trns.empty() && (ctype == 0 || ctype == 2)) comp = chans + 1;
    img.width = (int)w; img.height = (int)h; img.comp = comp;
    img.px.assign((size_t)w * h * comp, 0);
    const int scale = (ctype == 0 && small) ? 255 / ((1 << depth) - 1) : 1;
    for (uint32_t y = 0; y < h; y++) {
        const uint16_t* line = &samp[(size_t)y * w * chans];
        uint8_t* o = &img.px[(size_t)y * w * comp];
        for (uint32_t x = 0; x < w; x++) {
            if (ctype == 3) {
                int idx = line[x];
                if ((size_t)idx * 3 + 2 >= plte.size()) return "PNG palette index out of range in " + path;
                o[x * comp + 0] = plte[idx * 3]; o[x * comp + 1] = plte[idx * 3 + 1]; o[x * comp + 2] = plte[idx * 3 + 2];
                if (comp == 4) o[x * comp + 3] = (size_t)idx < trns.size() ? trns[idx] : 255;
            } else {
                bool key = !trns.empty() && (ctype == 0 || ctype == 2);
                for (int c = 0; c < chans; c++) {
                    const int full = line[(size_t)x * chans + c];
                    const int v = depth == 16 ? full >> 8 : full;
                    o[x * comp + c] = (uint8_t)(v * scale);
                    if (!trns.empty() && (ctype == 0 || ctype == 2)) {
                        int kv = depth == 16 ? ((trns.size() >= (size_t)2 * c + 2) ? (trns[2 * c] << 8 | trns[2 * c + 1]) : -1)
                                             : ((trns.size() >= (size_t)2 * c + 2) ? trns[2 * c + 1] : -1);
                        if (kv != full) key = false;
                    }
                }
                if (comp == chans + 1 && (ctype == 0 || ctype == 2)) o[x * comp + chans] = key ? 0 : 255;
            }
        }
    }
    return std::string();
}

} // namespace crtpng
#endif
