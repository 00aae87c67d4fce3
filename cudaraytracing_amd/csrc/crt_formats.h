// cudaraytracing_amd/csrc/crt_formats.h -- the five less common formats a map_Kd may point at: GIF, PSD, Softimage PIC, binary
// PNM and Radiance HDR.
//
// The reference hands every texture to its vendored stb_image (stbi_load(path, &h, &w, &comp, 0), include/Loader.h:58), which
// reads these files too; what it returns for them is partly the formats' definition and partly that decoder's own decisions,
// and both are restated here so that the samples are the same (pinned by tests/golden/stb_decode.json, which holds what the
// reference's decoder returned for the fixture files):
//   GIF  the FIRST frame as RGBA: pixels of the transparent index stay (0,0,0,0); with a background index > 0 the pixels the
//        frame does not cover get that palette entry with its red and blue swapped and alpha 255 (stb_image.h:6896-6904 copies
//        the entry in the order it is stored); a stream must begin with a clear code; interlaced frames;
//   PSD  version 1, RGB mode, 8 or 16 bits (high byte kept), raw or PackBits planes, always 4 channels out, and the removal
//        of the white matte from pixels that are partly transparent, in float arithmetic (stb_image.h:6290-6300);
//   PIC  8-bit channel packets, uncompressed / pure / mixed run-length; 4 channels when a packet carries alpha, else 3;
//   PNM  P5 / P6 with '#' comments in the header; a 16-bit file comes back as the SECOND byte of every sample (the reference's
//        decoder reads the big-endian samples into host-order words and keeps the words' high bytes, stb_image.h:1200-1206);
//   HDR  32-bit_rle_rgbe, "-Y h +X w" only; run-length and flat scanlines, the flat-data restart of stb_image.h:7243-7256;
//        float -> 8 bit as pow(v, 1/2.2f) * 255 + 0.5 in the host C library's double-precision pow (stb_image.h:1894), so the
//        low bit of a sample is as portable as that function is.
// Reads past the end of a file give zeros, as in the reference's decoder, except for flat Radiance pixels, where its buffer keeps
// the previous pixel.
#ifndef CRT_FORMATS_H
#define CRT_FORMATS_H

#include "crt_png.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace crtfmt {

typedef crtpng::Image Image;

struct Reader {
    const std::vector<uint8_t>& d;
    size_t pos = 0;
    explicit Reader(const std::vector<uint8_t>& data) : d(data) {}
    bool eof() const { return pos >= d.size(); }
    int u8() { return pos < d.size() ? d[pos++] : (pos++, 0); }
    int u16le() { const int a = u8(); return a | (u8() << 8); }
    int u16be() { const int a = u8(); return (a << 8) | u8(); }
    uint32_t u32be() { const uint32_t a = (uint32_t)u16be(); return (a << 16) | (uint32_t)u16be(); }
    void skip(uint64_t n) // (never far beyond the end: reads there give zeros)
    {
        const uint64_t lim = (uint64_t)d.size() + 16, p = (uint64_t)pos + n;
        pos = (size_t)(p > lim || p < n ? lim : p);
    }
};

const int kMaxDim = 1 << 24; // (the reference's decoder: STBI_MAX_DIMENSIONS)
// width * height * 4 must be an int, as in the reference's decoder (its size checks are made in int arithmetic)
inline bool fits(uint64_t w, uint64_t h, uint64_t comp) { return w * h * comp <= 0x7fffffffull; }

// ------------------------------------------------------------------ PNM ----
inline bool pnm_space(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }
inline std::string load_pnm(const std::vector<uint8_t>& file, Image& img)
{
    Reader r(file);
    r.u8();
    const int comp = r.u8() == '6' ? 3 : 1;
    int c = r.u8();
    bool overflow = false;
    auto blanks = [&]() { // white space and comment lines
        for (;;) {
            while (!r.eof() && pnm_space(c)) c = r.u8();
            if (r.eof() || c != '#') break;
            while (!r.eof() && c != '\n' && c != '\r') c = r.u8();
        }
    };
    auto number = [&]() {
        int v = 0;
        while (!r.eof() && c >= '0' && c <= '9') {
            v = v * 10 + (c - '0');
            c = r.u8();
            if (v > 214748364 || (v == 214748364 && c > '7')) { overflow = true; return 0; }
        }
        return v;
    };
    blanks();
    const int w = number();
    if (w == 0 || overflow) return "PNM header (width)";
    blanks();
    const int h = number();
    if (h == 0 || overflow) return "PNM header (height)";
    blanks();
    const int maxv = number(); // (the one character after it has been consumed: the samples follow)
    if (overflow || maxv > 65535) return "PNM sample range";
    const int bytes = maxv > 255 ? 2 : 1;
    if (w > kMaxDim || h > kMaxDim || !fits((uint64_t)w, (uint64_t)h, (uint64_t)comp * bytes)) return "PNM dimensions";
    const size_t n = (size_t)w * h * comp;
    if (r.pos > file.size() || file.size() - r.pos < n * bytes) return "truncated PNM";
    img.width = w; img.height = h; img.comp = comp;
    img.px.resize(n);
    const uint8_t* s = file.data() + r.pos;
    if (bytes == 1) std::memcpy(img.px.data(), s, n);
    else for (size_t i = 0; i < n; i++) img.px[i] = s[2 * i + 1];
    return "";
}

// ------------------------------------------------------------------ PSD ----
inline std::string load_psd(const std::vector<uint8_t>& file, Image& img)
{
    Reader r(file);
    r.u32be();
    if (r.u16be() != 1) return "PSD version";
    r.skip(6);
    const int channels = r.u16be();
    if (channels > 16) return "PSD channel count";
    const uint32_t h = r.u32be(), w = r.u32be();
    if (h > (uint32_t)kMaxDim || w > (uint32_t)kMaxDim) return "PSD dimensions";
    const int depth = r.u16be();
    if (depth != 8 && depth != 16) return "PSD bit depth (8 and 16 are read)";
    if (r.u16be() != 3) return "PSD colour mode (RGB is read)";
    r.skip(r.u32be()); // mode data
    r.skip(r.u32be()); // image resources
    r.skip(r.u32be()); // layers and masks
    const int compression = r.u16be();
    if (compression > 1) return "PSD compression";
    if (!fits(w, h, 4) || w == 0 || h == 0) return "PSD dimensions";
    const size_t n = (size_t)w * h;
    if (n > 128ull * file.size() + 1024) return "corrupt PSD (far more pixels than the file can hold)"; // (PackBits: at most 128 samples from 2 bytes)
    std::vector<uint8_t> out(n * 4);
    if (compression) {
        r.skip((uint64_t)h * channels * 2); // the byte counts of the rows
        for (int ch = 0; ch < 4; ch++) {
            uint8_t* p = out.data() + ch;
            if (ch >= channels) { for (size_t i = 0; i < n; i++) p[i * 4] = ch == 3 ? 255 : 0; continue; }
            size_t count = 0; // (one PackBits stream over the whole plane, whatever the depth says)
            while (count < n) {
                int len = r.u8();
                if (len == 128) continue;
                if (len < 128) {
                    len++;
                    if ((size_t)len > n - count) return "corrupt PSD (run-length data)";
                    for (int k = 0; k < len; k++) p[(count + k) * 4] = (uint8_t)r.u8();
                } else {
                    len = 257 - len;
                    if ((size_t)len > n - count) return "corrupt PSD (run-length data)";
                    const uint8_t v = (uint8_t)r.u8();
                    for (int k = 0; k < len; k++) p[(count + k) * 4] = v;
                }
                count += (size_t)len;
            }
        }
    } else {
        for (int ch = 0; ch < 4; ch++) {
            uint8_t* p = out.data() + ch;
            if (ch >= channels) { for (size_t i = 0; i < n; i++) p[i * 4] = ch == 3 ? 255 : 0; continue; }
            if (depth == 16) for (size_t i = 0; i < n; i++) p[i * 4] = (uint8_t)(r.u16be() >> 8);
            else for (size_t i = 0; i < n; i++) p[i * 4] = (uint8_t)r.u8();
        }
    }
    if (channels >= 4) { // colours stored blended against white: taken back out where alpha is neither 0 nor 255
        for (size_t i = 0; i < n; i++) {
            uint8_t* px = &out[i * 4];
            if (px[3] != 0 && px[3] != 255) {
                const float a = px[3] / 255.0f;
                const float ra = 1.0f / a;
                const float inv_a = 255.0f * (1 - ra);
                for (int k = 0; k < 3; k++) px[k] = (uint8_t)(int32_t)(px[k] * ra + inv_a); // (the low byte of the truncated value)
            }
        }
    }
    img.width = (int)w; img.height = (int)h; img.comp = 4;
    img.px.swap(out);
    return "";
}

// ------------------------------------------------------------------ PIC ----
inline bool looks_like_pic(const std::vector<uint8_t>& d)
{
    return d.size() >= 92 && d[0] == 0x53 && d[1] == 0x80 && d[2] == 0xf6 && d[3] == 0x34 && std::memcmp(&d[88], "PICT", 4) == 0;
}
inline std::string load_pic(const std::vector<uint8_t>& file, Image& img)
{
    Reader r(file);
    r.skip(92);
    const int w = r.u16be(), h = r.u16be();
    if (r.eof()) return "truncated PIC";
    if (w == 0 || h == 0 || !fits((uint64_t)w, (uint64_t)h, 4)) return "PIC dimensions";
    if ((uint64_t)w * (uint64_t)h > 65535ull * file.size()) return "corrupt PIC (far more pixels than the file can hold)";
    r.u32be(); r.u16be(); r.u16be(); // ratio, fields, pad
    struct Packet { int type, channel; };
    std::vector<Packet> packets;
    int chained, all = 0;
    do {
        if (packets.size() == 10) return "corrupt PIC (too many packets)";
        chained = r.u8();
        const int size = r.u8();
        Packet p;
        p.type = r.u8(); p.channel = r.u8();
        all |= p.channel;
        if (r.eof()) return "truncated PIC";
        if (size != 8) return "PIC packet that is not 8 bits per channel";
        packets.push_back(p);
    } while (chained);
    const int comp = (all & 0x10) ? 4 : 3;
    std::vector<uint8_t> out((size_t)w * h * 4, 0xff);
    bool shortfile = false;
    auto readval = [&](int channel, uint8_t* dest) { // the channels of one pixel that the packet carries (mask bit 0x80 = red ...)
        for (int i = 0, mask = 0x80; i < 4; i++, mask >>= 1)
            if (channel & mask) {
                if (r.eof()) { shortfile = true; return; }
                dest[i] = (uint8_t)r.u8();
            }
    };
    auto copyval = [](int channel, uint8_t* dest, const uint8_t* src) {
        for (int i = 0, mask = 0x80; i < 4; i++, mask >>= 1) if (channel & mask) dest[i] = src[i];
    };
    for (int y = 0; y < h; y++) {
        for (const Packet& p : packets) {
            uint8_t* dest = &out[(size_t)y * w * 4];
            if (p.type == 0) {
                for (int x = 0; x < w; x++, dest += 4) { readval(p.channel, dest); if (shortfile) return "truncated PIC"; }
            } else if (p.type == 1) {
                int left = w;
                while (left > 0) {
                    int count = r.u8();
                    if (r.eof()) return "truncated PIC";
                    if (count > left) count = (uint8_t)left;
                    uint8_t v[4];
                    readval(p.channel, v);
                    if (shortfile) return "truncated PIC";
                    for (int i = 0; i < count; i++, dest += 4) copyval(p.channel, dest, v);
                    left -= count;
                    if (count == 0 && r.pos > file.size()) return "truncated PIC"; // (a zero count repeats nothing: bounded by the file)
                }
            } else if (p.type == 2) {
                int left = w;
                while (left > 0) {
                    int count = r.u8();
                    if (r.eof()) return "truncated PIC";
                    if (count >= 128) {
                        count = count == 128 ? r.u16be() : count - 127;
                        if (count > left) return "corrupt PIC (scanline overrun)";
                        uint8_t v[4];
                        readval(p.channel, v);
                        if (shortfile) return "truncated PIC";
                        for (int i = 0; i < count; i++, dest += 4) copyval(p.channel, dest, v);
                    } else {
                        count++;
                        if (count > left) return "corrupt PIC (scanline overrun)";
                        for (int i = 0; i < count; i++, dest += 4) { readval(p.channel, dest); if (shortfile) return "truncated PIC"; }
                    }
                    left -= count;
                    if (count == 0 && r.pos > file.size()) return "truncated PIC";
                }
            } else return "PIC packet compression";
        }
    }
    img.width = w; img.height = h; img.comp = comp;
    img.px.resize((size_t)w * h * comp);
    for (size_t i = 0; i < (size_t)w * h; i++)
        for (int k = 0; k < comp; k++) img.px[i * comp + k] = out[i * 4 + k];
    return "";
}

// ------------------------------------------------------------------ GIF ----
inline std::string load_gif(const std::vector<uint8_t>& file, Image& img)
{
    Reader r(file);
    r.skip(4);
    const int ver = r.u8();
    if ((ver != '7' && ver != '9') || r.u8() != 'a') return "corrupt GIF (signature)";
    const int W = r.u16le(), H = r.u16le(), flags = r.u8(), bgindex = r.u8();
    r.u8(); // aspect ratio
    if (W == 0 || H == 0 || (uint64_t)W * (uint64_t)H > (1ull << 26)) return "GIF dimensions"; // (the reference's decoder stops at 2^29 pixels)
    struct Entry { uint8_t r, g, b, a; };
    Entry gpal[256], lpal[256];
    std::memset(gpal, 0, sizeof(gpal)); std::memset(lpal, 0, sizeof(lpal));
    auto table = [&](Entry* pal, int n, int transparent) {
        for (int i = 0; i < n; i++) { pal[i].r = (uint8_t)r.u8(); pal[i].g = (uint8_t)r.u8(); pal[i].b = (uint8_t)r.u8(); pal[i].a = transparent == i ? 0 : 255; }
    };
    if (flags & 0x80) table(gpal, 2 << (flags & 7), -1);
    int transparent = -1, eflags = 0;
    for (;;) {
        const int tag = r.u8();
        if (tag == 0x21) { // extension
            const int ext = r.u8();
            int len;
            if (ext == 0xF9) { // graphic control
                len = r.u8();
                if (len == 4) {
                    eflags = r.u8();
                    r.u16le(); // delay
                    if (transparent >= 0) gpal[transparent].a = 255;
                    if (eflags & 1) { transparent = r.u8(); gpal[transparent].a = 0; }
                    else { r.skip(1); transparent = -1; }
                } else { r.skip((uint64_t)len); continue; }
            }
            while ((len = r.u8()) != 0) { r.skip((uint64_t)len); if (r.pos > file.size()) break; }
            continue;
        }
        if (tag != 0x2C) return tag == 0x3B ? "GIF without an image" : "corrupt GIF (block)";
        break;
    }
    // ---- the first image descriptor ----
    const int x0 = r.u16le(), y0 = r.u16le(), w = r.u16le(), h = r.u16le();
    if (x0 + w > W || y0 + h > H) return "corrupt GIF (image descriptor)";
    const int lflags = r.u8();
    const Entry* pal;
    if (lflags & 0x80) { table(lpal, 2 << (lflags & 7), (eflags & 1) ? transparent : -1); pal = lpal; }
    else if (flags & 0x80) pal = gpal;
    else return "corrupt GIF (no colour table)";
    std::vector<uint8_t> out((size_t)W * H * 4, 0), touched((size_t)W * H, 0);
    // rows of the frame in the order the stream delivers them
    int row = 0, pass_step = (lflags & 0x40) ? 8 : 1, pass = (lflags & 0x40) ? 3 : 0, col = 0;
    bool full = w == 0 || h == 0; // (nothing to draw into)
    auto emit = [&](int index) {
        if (full) return;
        const size_t at = (size_t)(y0 + row) * W + (size_t)(x0 + col);
        touched[at] = 1;
        const Entry& e = pal[index];
        if (e.a > 128) { out[at * 4] = e.r; out[at * 4 + 1] = e.g; out[at * 4 + 2] = e.b; out[at * 4 + 3] = e.a; }
        if (++col >= w) {
            col = 0;
            row += pass_step;
            while (row >= h && pass > 0) { pass_step = 1 << pass; row = pass_step >> 1; --pass; }
            if (row >= h) full = true;
        }
    };
    // ---- LZW ----
    const int lzw_cs = r.u8();
    if (lzw_cs > 12) return "corrupt GIF (code size)";
    struct Code { int16_t prefix; uint8_t first, suffix; };
    std::vector<Code> codes(8192);
    std::vector<uint8_t> chain(8192);
    const int clear = 1 << lzw_cs;
    for (int i = 0; i < clear && i < 8192; i++) { codes[i].prefix = -1; codes[i].first = (uint8_t)i; codes[i].suffix = (uint8_t)i; }
    int codesize = lzw_cs + 1, codemask = (1 << codesize) - 1, avail = clear + 2, oldcode = -1, len = 0, valid_bits = 0;
    int32_t bits = 0;
    bool first = true;
    for (;;) {
        if (valid_bits < codesize) {
            if (len == 0) {
                len = r.u8();
                if (len == 0) break; // end of the data (or of the file)
            }
            --len;
            bits |= (int32_t)((uint32_t)r.u8() << valid_bits);
            valid_bits += 8;
            continue;
        }
        const int code = bits & codemask;
        bits >>= codesize;
        valid_bits -= codesize;
        if (code == clear) {
            codesize = lzw_cs + 1; codemask = (1 << codesize) - 1; avail = clear + 2; oldcode = -1; first = false;
        } else if (code == clear + 1) {
            break;
        } else if (code <= avail) {
            if (first) return "corrupt GIF (no clear code)";
            if (oldcode >= 0) {
                if (avail + 1 > 8192) return "corrupt GIF (too many codes)";
                Code& p = codes[avail++];
                p.prefix = (int16_t)oldcode;
                p.first = codes[oldcode].first;
                p.suffix = code == avail ? p.first : codes[code].first;
            } else if (code == avail) return "corrupt GIF (code)";
            // the string of `code`, first symbol first
            int n = 0;
            for (int c = code; c >= 0 && n < 8192; c = codes[c].prefix) chain[n++] = codes[c].suffix;
            while (n > 0) emit(chain[--n]);
            if ((avail & codemask) == 0 && avail <= 0x0FFF) { codesize++; codemask = (1 << codesize) - 1; }
            oldcode = code;
        } else return "corrupt GIF (code)";
    }
    if (bgindex > 0) // what the frame did not cover: the background entry in its stored byte order (blue first), opaque
        for (size_t i = 0; i < (size_t)W * H; i++)
            if (!touched[i]) { out[i * 4] = gpal[bgindex].b; out[i * 4 + 1] = gpal[bgindex].g; out[i * 4 + 2] = gpal[bgindex].r; out[i * 4 + 3] = 255; }
    img.width = W; img.height = H; img.comp = 4;
    img.px.swap(out);
    return "";
}

// ------------------------------------------------------------------ HDR ----
inline bool looks_like_hdr(const std::vector<uint8_t>& d)
{
    return (d.size() >= 11 && std::memcmp(d.data(), "#?RADIANCE\n", 11) == 0) || (d.size() >= 7 && std::memcmp(d.data(), "#?RGBE\n", 7) == 0);
}
inline std::string load_hdr(const std::vector<uint8_t>& file, Image& img)
{
    Reader r(file);
    auto line = [&]() { // up to the next '\n' (lines longer than 1022 characters are cut there)
        std::string s;
        int c = r.u8();
        while (!r.eof() && c != '\n') {
            s.push_back((char)c);
            if (s.size() == 1023) { while (!r.eof() && r.u8() != '\n') {} break; }
            c = r.u8();
        }
        return s;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") return "corrupt HDR (signature)";
    bool valid = false;
    for (;;) {
        const std::string t = line();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") valid = true;
    }
    if (!valid) return "HDR format (32-bit_rle_rgbe is read)";
    const std::string dims = line();
    if (dims.compare(0, 3, "-Y ") != 0) return "HDR data layout (-Y h +X w is read)";
    const char* p = dims.c_str() + 3;
    char* end = nullptr;
    const long height = std::strtol(p, &end, 10);
    while (*end == ' ') ++end;
    if (std::strncmp(end, "+X ", 3) != 0) return "HDR data layout (-Y h +X w is read)";
    const long width = std::strtol(end + 3, nullptr, 10);
    if (height > kMaxDim || width > kMaxDim || width < 1 || height < 1 || !fits((uint64_t)width, (uint64_t)height, 12)) return "HDR dimensions";
    if ((uint64_t)width * (uint64_t)height > 128ull * file.size() + 1024) return "corrupt HDR (far more pixels than the file can hold)";
    const size_t n = (size_t)width * height;
    std::vector<float> f(n * 3);
    auto convert = [&](float* o, const uint8_t* in) {
        if (in[3] != 0) {
            const float f1 = (float)std::ldexp(1.0f, in[3] - (int)(128 + 8));
            o[0] = in[0] * f1; o[1] = in[1] * f1; o[2] = in[2] * f1;
        } else o[0] = o[1] = o[2] = 0.0f;
    };
    auto flat_from = [&](size_t first_pixel) {
        // (beyond the end of the file the reference's decoder converts its four-byte buffer again without having refilled it: the
        // last pixel repeats -- which is what a file gets whose flat scanlines follow run-length ones, see below)
        uint8_t rgbe[4] = {0, 0, 0, 0};
        for (size_t i = first_pixel; i < n; i++) {
            for (int k = 0; k < 4 && !r.eof(); k++) rgbe[k] = (uint8_t)r.u8();
            convert(&f[i * 3], rgbe);
        }
    };
    if (width < 8 || width >= 32768) flat_from(0);
    else {
        std::vector<uint8_t> scan((size_t)width * 4);
        for (long j = 0; j < height; j++) {
            const int c1 = r.u8(), c2 = r.u8();
            int len = r.u8();
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                // a scanline that is not run-length coded: these four bytes become pixel 0 and the REST OF THE FILE is read as flat
                // pixels from pixel 1 on, whichever scanline this is (the reference's decoder restarts its flat loop there)
                const uint8_t rgbe[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)r.u8()};
                convert(&f[0], rgbe);
                flat_from(1);
                break;
            }
            len = (len << 8) | r.u8();
            if (len != width) return "corrupt HDR (scanline length)";
            for (int k = 0; k < 4; k++) {
                long i = 0;
                while (i < width) {
                    int count = r.u8();
                    const long nleft = width - i;
                    if (count > 128) {
                        const uint8_t v = (uint8_t)r.u8();
                        count -= 128;
                        if (count > nleft) return "corrupt HDR (run-length data)";
                        for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
                    } else {
                        if (count == 0 || count > nleft) return "corrupt HDR (run-length data)";
                        for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = (uint8_t)r.u8();
                    }
                }
            }
            for (long i = 0; i < width; i++) convert(&f[((size_t)j * width + i) * 3], &scan[(size_t)i * 4]);
        }
    }
    img.width = (int)width; img.height = (int)height; img.comp = 3;
    img.px.resize(n * 3);
    const float gamma_i = 1.0f / 2.2f, scale_i = 1.0f;
    for (size_t i = 0; i < n * 3; i++) {
        float z = (float)std::pow((double)(f[i] * scale_i), (double)gamma_i) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        img.px[i] = (uint8_t)(int)z;
    }
    return "";
}

} // namespace crtfmt
#endif
