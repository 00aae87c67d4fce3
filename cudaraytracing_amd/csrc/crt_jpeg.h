// cudaraytracing_amd/csrc/crt_jpeg.h -- JPEG decoding for map_Kd textures (host layer).
//
// The reference reads textures with its vendored stb_image v2.29: stbi_load(path, &x, &y, &comp, 0) (include/Loader.h:58).
// Huffman decoding and dequantisation are fixed by the JPEG standard (ITU-T T.81); what is a decoder's own choice -- and what this
// file therefore takes from stb_image so that the SAMPLES are the same numbers -- is
//   * the inverse DCT: the integer "islow" transform at 12 fractional bits with stb_image's constants, its two rounding shifts
//     (>> 10 after the columns with 2 extra bits kept, >> 17 after the rows with the level shift folded in) and clamping
//     (stb_image.h:2430-2525);
//   * chroma upsampling: the centred 3:1 filters -- vertical (3 near + far + 2) >> 2, horizontal likewise with replicated ends,
//     2 x 2 as the vertical sums filtered horizontally at (.. + 8) >> 4 -- and pixel replication for every other ratio
//     (stb_image.h:3464-3527, 3645-3655);
//   * YCbCr -> RGB in 20-bit fixed point, including the masking of the Cb term of green to its upper 16 bits
//     (stb_image.h:3659-3683);
//   * which files count as RGB / CMYK / YCCK (component ids 'R','G','B'; Adobe APP14 transform without JFIF), what a
//     4-component file returns (3 samples per pixel, Blinn's 8 x 8 multiply), grey files return 1 sample per pixel
//     (stb_image.h:3864-4024).
// Baseline and extended sequential (SOF0 / SOF1, 8 bit), progressive (SOF2), interleaved and non-interleaved scans, restart
// intervals, 8- and 16-bit quantisation tables.  Arithmetic coding, lossless, hierarchical and 12-bit files are rejected, as
// stb_image rejects them.  Pinned sample for sample against the reference's own decoder: tests/golden/stb_decode.json
// (oracle/ref_probe/stb_probe.c).  A truncated or corrupt stream is an error here; stb_image returns what it has in some cases.
#ifndef CRT_JPEG_H
#define CRT_JPEG_H

#include "crt_png.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace crtjpg {

typedef crtpng::Image Image;

namespace detail {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Canonical Huffman table, decoded as T.81 F.2.2.3 describes: per code length the smallest code, the largest code and the index
// of its first symbol.
struct Huff {
    bool defined = false;
    int32_t mincode[17], maxcode[18], first[17];
    uint8_t vals[256];
    int n = 0;
    bool build(const int counts[16])
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            first[len] = k;
            mincode[len] = code;
            if ((unsigned)(code + counts[len - 1]) > (1u << len)) return false; // more codes of this length than there is room for
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        n = k;
        defined = true;
        return n <= 256;
    }
};

// Entropy-coded segment: bits most significant first, 0xFF 0x00 is a data byte 0xFF, any other 0xFF xx is a marker that ends
// the segment -- from then on the reader supplies zero bits (as the reference's decoder does) and remembers the marker.
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    int marker = 0; // 0 = none seen
    bool starved = false; // bits were asked for after the data had ended
    void reset() { acc = 0; nbits = 0; marker = 0; }
    void fill()
    {
        while (nbits <= 24) {
            uint32_t b = 0;
            if (!marker) {
                if (p >= end) { marker = 0xD9; } // (a file that just stops: as if the end-of-image marker stood here)
                else {
                    b = *p++;
                    if (b == 0xFF) {
                        int c = p < end ? *p++ : 0xD9;
                        while (c == 0xFF) c = p < end ? *p++ : 0xD9; // fill bytes
                        if (c != 0) { marker = c; b = 0; }
                    }
                }
            }
            acc |= b << (24 - nbits);
            nbits += 8;
        }
    }
    int bit()
    {
        if (nbits < 1) fill();
        const int b = (int)(acc >> 31);
        acc <<= 1;
        nbits--;
        return b;
    }
    int bits(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    // T.81 F.2.2.1 RECEIVE + EXTEND
    int receive_extend(int n)
    {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    // T.81 F.2.2.3 DECODE; -1 = no code of up to 16 bits matches
    int decode(const Huff& h)
    {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.first[len] + code - h.mincode[len]];
        }
        return -1;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int x = 0, y = 0;     // samples that belong to the image
    int w2 = 0, h2 = 0;   // allocated plane: whole MCUs
    int bw = 0, bh = 0;   // blocks per row / column of the plane
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff; // progressive: 64 coefficients per block, natural order
};

inline int idct_const(float x) { return (int)(x * 4096 + 0.5); } // (12 fractional bits; float product, double sum: stb_image.h:2425)

// One pass of the 8-point inverse DCT ("islow", after the Independent JPEG Group's jidctint) on s[0..7]: the even part in x[0..3],
// the odd part in t[0..3], both scaled by 4096; the caller combines them as x[k] +- t[3 - k].  32-bit wrap-around arithmetic.
inline void idct_1d(const int32_t s[8], int32_t x[4], int32_t t[4])
{
    static const int32_t c0541 = idct_const(0.5411961f), c1847 = idct_const(-1.847759065f), c0765 = idct_const(0.765366865f),
                         c1175 = idct_const(1.175875602f), c0298 = idct_const(0.298631336f), c2053 = idct_const(2.053119869f),
                         c3072 = idct_const(3.072711026f), c1501 = idct_const(1.501321110f), c0899 = idct_const(-0.899976223f),
                         c2562 = idct_const(-2.562915447f), c1961 = idct_const(-1.961570560f), c0390 = idct_const(-0.390180644f);
    typedef uint32_t u;
    const u s0 = (u)s[0], s1 = (u)s[1], s2 = (u)s[2], s3 = (u)s[3], s4 = (u)s[4], s5 = (u)s[5], s6 = (u)s[6], s7 = (u)s[7];
    // even part
    const u z1 = (s2 + s6) * (u)c0541;
    const u e2 = z1 + s6 * (u)c1847, e3 = z1 + s2 * (u)c0765;
    const u e0 = (s0 + s4) * 4096u, e1 = (s0 - s4) * 4096u;
    x[0] = (int32_t)(e0 + e3); x[3] = (int32_t)(e0 - e3); x[1] = (int32_t)(e1 + e2); x[2] = (int32_t)(e1 - e2);
    // odd part
    const u a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
    const u z5 = (a + b) * (u)c1175;
    const u pc = z5 + c * (u)c0899, pd = z5 + d * (u)c2562, pa = a * (u)c1961, pb = b * (u)c0390;
    t[3] = (int32_t)(s1 * (u)c1501 + pc + pb);
    t[2] = (int32_t)(s3 * (u)c3072 + pd + pa);
    t[1] = (int32_t)(s5 * (u)c2053 + pd + pb);
    t[0] = (int32_t)(s7 * (u)c0298 + pc + pa);
}
inline uint8_t clamp8(int32_t v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// 8 x 8 block of dequantised coefficients (natural order) -> samples
inline void idct_block(uint8_t* out, int stride, const int16_t d[64])
{
    int32_t mid[64];
    for (int c = 0; c < 8; c++) { // columns; result keeps 2 extra bits
        int32_t s[8], x[4], t[4];
        for (int r = 0; r < 8; r++) s[r] = d[r * 8 + c];
        idct_1d(s, x, t);
        for (int k = 0; k < 4; k++) {
            const uint32_t e = (uint32_t)x[k] + 512u;
            mid[k * 8 + c] = (int32_t)(e + (uint32_t)t[3 - k]) >> 10;
            mid[(7 - k) * 8 + c] = (int32_t)(e - (uint32_t)t[3 - k]) >> 10;
        }
    }
    for (int r = 0; r < 8; r++) { // rows: 12 + 2 + 3 bits to remove, rounded, with the level shift of 128 added before the shift
        int32_t x[4], t[4];
        idct_1d(mid + r * 8, x, t);
        uint8_t* o = out + (size_t)r * stride;
        for (int k = 0; k < 4; k++) {
            const uint32_t e = (uint32_t)x[k] + 65536u + (128u << 17);
            o[k] = clamp8((int32_t)(e + (uint32_t)t[3 - k]) >> 17);
            o[7 - k] = clamp8((int32_t)(e - (uint32_t)t[3 - k]) >> 17);
        }
    }
}

struct Decoder {
    const uint8_t* data = nullptr;
    size_t size = 0, pos = 0;
    std::string err;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    uint16_t quant[4][64];
    Huff hdc[4], hac[4];
    Component comp[4];
    // current scan
    int scan_n = 0, order[4], ss = 0, se = 63, ah = 0, al = 0;
    int eob_run = 0;
    BitReader br;

    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    int u8() { return pos < size ? data[pos++] : (pos++, 0); }
    int u16() { int a = u8(); return (a << 8) | u8(); }
    bool eof() const { return pos >= size; }

    // the next marker code; 0 if the next byte does not start one
    int next_marker()
    {
        if (br.marker) { int m = br.marker; br.marker = 0; return m; }
        int x = u8();
        if (x != 0xFF) return 0;
        while (x == 0xFF) x = u8();
        return x;
    }

    bool segment(int m)
    {
        if (m == 0) return fail("JPEG: expected a marker");
        if (m == 0xDD) { // restart interval
            if (u16() != 4) return fail("JPEG: bad DRI length");
            restart_interval = u16();
            return true;
        }
        if (m == 0xDB) { // quantisation tables
            int L = u16() - 2;
            while (L > 0) {
                const int q = u8(), prec = q >> 4, t = q & 15;
                if (prec > 1) return fail("JPEG: bad quantisation table precision");
                if (t > 3) return fail("JPEG: bad quantisation table index");
                for (int i = 0; i < 64; i++) quant[t][kZigzag[i]] = (uint16_t)(prec ? u16() : u8());
                L -= prec ? 129 : 65;
            }
            return L == 0 ? true : fail("JPEG: bad DQT length");
        }
        if (m == 0xC4) { // Huffman tables
            int L = u16() - 2;
            while (L > 0) {
                const int q = u8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return fail("JPEG: bad Huffman table header");
                int counts[16], n = 0;
                for (int i = 0; i < 16; i++) { counts[i] = u8(); n += counts[i]; }
                if (n > 256) return fail("JPEG: bad Huffman table header");
                Huff& h = tc ? hac[th] : hdc[th];
                if (!h.build(counts)) return fail("JPEG: bad Huffman code lengths");
                for (int i = 0; i < n; i++) h.vals[i] = (uint8_t)u8();
                L -= 17 + n;
            }
            return L == 0 ? true : fail("JPEG: bad DHT length");
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) { // application data, comment
            int L = u16();
            if (L < 2) return fail("JPEG: bad APP / COM length");
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; i++) ok = (u8() == tag[i]) && ok;
                L -= 5;
                if (ok) jfif = true;
            } else if (m == 0xEE && L >= 12) {
                static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; i++) ok = (u8() == tag[i]) && ok;
                L -= 6;
                if (ok) { u8(); u16(); u16(); adobe_transform = u8(); L -= 6; }
            }
            pos += (size_t)L;
            return true;
        }
        return fail("JPEG: unsupported marker (arithmetic coding, lossless and hierarchical files are not decoded)");
    }

    bool frame_header()
    {
        const int Lf = u16();
        if (Lf < 11) return fail("JPEG: bad SOF length");
        if (u8() != 8) return fail("JPEG: only 8-bit samples");
        height = u16(); width = u16();
        if (height == 0) return fail("JPEG: height 0 (defined by a later DNL segment) is not decoded");
        if (width == 0) return fail("JPEG: width 0");
        ncomp = u8();
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) return fail("JPEG: bad component count");
        if (Lf != 8 + 3 * ncomp) return fail("JPEG: bad SOF length");
        rgb_ids = 0;
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.id = u8();
            if (ncomp == 3 && c.id == "RGB"[i]) rgb_ids++;
            const int q = u8();
            c.h = q >> 4; c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("JPEG: bad sampling factor");
            c.tq = u8();
            if (c.tq > 3) return fail("JPEG: bad quantisation table selector");
        }
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; i++) { if (comp[i].h > h_max) h_max = comp[i].h; if (comp[i].v > v_max) v_max = comp[i].v; }
        for (int i = 0; i < ncomp; i++)
            if (h_max % comp[i].h || v_max % comp[i].v) return fail("JPEG: fractional sampling ratios are not decoded");
        mcu_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcu_y = (height + 8 * v_max - 1) / (8 * v_max);
        if ((uint64_t)width * (uint64_t)height * (uint64_t)ncomp > (1ull << 30)) return fail("JPEG: image too large");
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
            c.bw = c.w2 / 8; c.bh = c.h2 / 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
        }
        return true;
    }

    bool scan_header()
    {
        const int Ls = u16();
        scan_n = u8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp) return fail("JPEG: bad SOS component count");
        if (Ls != 6 + 2 * scan_n) return fail("JPEG: bad SOS length");
        for (int i = 0; i < scan_n; i++) {
            const int id = u8(), q = u8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) which++;
            if (which == ncomp) return fail("JPEG: scan names an unknown component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return fail("JPEG: bad Huffman table selector");
            order[i] = which;
        }
        ss = u8(); se = u8();
        const int a = u8();
        ah = a >> 4; al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return fail("JPEG: bad progressive scan parameters");
        } else {
            if (ss != 0 || ah != 0 || al != 0) return fail("JPEG: bad scan parameters");
            se = 63;
        }
        return true;
    }

    void restart_state()
    {
        br.reset();
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
        eob_run = 0;
    }

    // sequential: one block, dequantised, into natural order
    bool block_sequential(Component& c, int16_t d[64])
    {
        const Huff &hd = hdc[c.td], &ha = hac[c.ta];
        if (!hd.defined || !ha.defined) return fail("JPEG: scan uses an undefined Huffman table");
        const uint16_t* q = quant[c.tq];
        std::memset(d, 0, 64 * sizeof(int16_t));
        const int t = br.decode(hd);
        if (t < 0 || t > 15) return fail("JPEG: bad Huffman code");
        const int diff = br.receive_extend(t);
        c.dc_pred += diff;
        const int dc = c.dc_pred * (int)q[0];
        if (dc < -32768 || dc > 32767) return fail("JPEG: DC coefficient out of range");
        d[0] = (int16_t)dc;
        for (int k = 1; k < 64;) {
            const int rs = br.decode(ha);
            if (rs < 0) return fail("JPEG: bad Huffman code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break; // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63) return fail("JPEG: coefficient index past the end of a block");
                const int z = kZigzag[k++];
                d[z] = (int16_t)(br.receive_extend(s) * (int)q[z]);
            }
        }
        return true;
    }

    bool block_dc_progressive(Component& c, int16_t* d)
    {
        if (se != 0) return fail("JPEG: DC scan with AC coefficients");
        if (ah == 0) {
            const Huff& hd = hdc[c.td];
            if (!hd.defined) return fail("JPEG: scan uses an undefined Huffman table");
            std::memset(d, 0, 64 * sizeof(int16_t));
            const int t = br.decode(hd);
            if (t < 0 || t > 15) return fail("JPEG: bad Huffman code");
            c.dc_pred += br.receive_extend(t);
            const int dc = c.dc_pred * (1 << al);
            if (dc < -32768 || dc > 32767) return fail("JPEG: DC coefficient out of range");
            d[0] = (int16_t)dc;
        } else if (br.bit()) {
            d[0] = (int16_t)(d[0] + (int16_t)(1 << al));
        }
        return true;
    }

    // T.81 G.1.2.3: one refinement bit for a coefficient that is already non-zero
    void refine(int16_t& v, int bit)
    {
        if (br.bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    }

    bool block_ac_progressive(Component& c, int16_t* d)
    {
        if (ss == 0) return fail("JPEG: AC scan that starts at the DC coefficient");
        const Huff& ha = hac[c.ta];
        if (!ha.defined) return fail("JPEG: scan uses an undefined Huffman table");
        if (ah == 0) { // first pass over this band
            if (eob_run) { eob_run--; return true; }
            for (int k = ss; k <= se;) {
                const int rs = br.decode(ha);
                if (rs < 0) return fail("JPEG: bad Huffman code");
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { // end of band for 2^r + extra blocks, this one included
                        eob_run = (1 << r) - 1;
                        if (r) eob_run += br.bits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) return fail("JPEG: coefficient index past the end of a block");
                    d[kZigzag[k++]] = (int16_t)(br.receive_extend(s) * (1 << al));
                }
            }
            return true;
        }
        // refinement pass (T.81 G.1.2.3)
        const int bit = 1 << al;
        if (eob_run) {
            eob_run--;
            for (int k = ss; k <= se; k++) {
                int16_t& v = d[kZigzag[k]];
                if (v != 0) refine(v, bit);
            }
            return true;
        }
        for (int k = ss; k <= se;) {
            const int rs = br.decode(ha);
            if (rs < 0) return fail("JPEG: bad Huffman code");
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += br.bits(r);
                    r = 64; // the rest of the band holds refinements only
                }
                // (r == 15: sixteen zero coefficients are skipped, nothing new is placed)
            } else {
                if (s != 1) return fail("JPEG: bad Huffman code");
                s = br.bit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t& v = d[kZigzag[k++]];
                if (v != 0) refine(v, bit);
                else {
                    if (r == 0) { v = (int16_t)s; break; }
                    r--;
                }
            }
        }
        return true;
    }

    // after `restart_interval` MCUs: the rest of the byte is padding and an RSTn marker follows; any other marker ends the scan here
    bool at_restart(int& todo, bool& stop)
    {
        if (--todo > 0) return true;
        if (br.nbits < 24) br.fill();
        if (br.marker < 0xD0 || br.marker > 0xD7) { stop = true; return true; }
        restart_state();
        todo = restart_interval ? restart_interval : 0x7fffffff;
        return true;
    }

    bool scan_data()
    {
        br.p = data + (pos < size ? pos : size);
        br.end = data + size;
        restart_state();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        bool stop = false;
        int16_t blk[64];
        if (scan_n == 1) { // one component: its blocks in raster order, only those that hold image samples
            Component& c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h && !stop; j++)
                for (int i = 0; i < w && !stop; i++) {
                    if (!progressive) {
                        if (!block_sequential(c, blk)) return false;
                        idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, blk);
                    } else {
                        int16_t* d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.bw);
                        if (!(ss == 0 ? block_dc_progressive(c, d) : block_ac_progressive(c, d))) return false;
                    }
                    at_restart(todo, stop);
                }
        } else { // interleaved: MCU after MCU, h x v blocks of each component
            for (int j = 0; j < mcu_y && !stop; j++)
                for (int i = 0; i < mcu_x && !stop; i++) {
                    for (int k = 0; k < scan_n; k++) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                const int bx = i * c.h + x, by = j * c.v + y;
                                if (!progressive) {
                                    if (!block_sequential(c, blk)) return false;
                                    idct_block(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, blk);
                                } else {
                                    if (!block_dc_progressive(c, c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.bw))) return false;
                                }
                            }
                    }
                    at_restart(todo, stop);
                }
        }
        pos = (size_t)(br.p - data);
        return true;
    }

    // the next marker after a scan: whatever the bit reader ran into, else the first 0xFF xx with xx neither 0x00 nor 0xFF
    void find_marker_after_scan()
    {
        if (br.marker) return;
        while (pos < size) {
            int x = data[pos++];
            while (x == 0xFF) {
                if (pos >= size) return;
                x = data[pos++];
                if (x != 0x00 && x != 0xFF) { br.marker = x; return; }
            }
        }
    }

    bool decode_planes()
    {
        std::memset(quant, 0, sizeof(quant));
        if (next_marker() != 0xD8) return fail("JPEG: no start-of-image marker");
        int m = next_marker();
        while (m != 0xC0 && m != 0xC1 && m != 0xC2) {
            if (!segment(m)) return false;
            m = next_marker();
            while (m == 0) {
                if (eof()) return fail("JPEG: no frame header");
                m = next_marker();
            }
        }
        progressive = m == 0xC2;
        if (!frame_header()) return false;
        m = next_marker();
        while (m != 0xD9) {
            if (m == 0xDA) {
                if (!scan_header() || !scan_data()) return false;
                find_marker_after_scan();
                m = next_marker();
                if (m >= 0xD0 && m <= 0xD7) m = next_marker();
                if (m == 0 && eof()) return fail("JPEG: the file ends inside the image data");
            } else if (m == 0xDC) {
                if (u16() != 4) return fail("JPEG: bad DNL length");
                if (u16() != height) return fail("JPEG: DNL height differs from the frame header");
                m = next_marker();
            } else {
                if (!segment(m)) return false;
                m = next_marker();
            }
        }
        if (progressive)
            for (int n = 0; n < ncomp; n++) {
                Component& c = comp[n];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; j++)
                    for (int i = 0; i < w; i++) {
                        int16_t* d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.bw);
                        for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * (int)quant[c.tq][k]);
                        idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, d);
                    }
            }
        return true;
    }
};

// ---- upsampling of one row of a component to full width (stb_image.h:3464-3527, 3645-3655) ----
inline const uint8_t* up_v2(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w)
{
    for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
    return out;
}
inline const uint8_t* up_h2(uint8_t* out, const uint8_t* in, int w)
{
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < w - 1; i++) {
        const int n = 3 * in[i] + 2;
        out[2 * i] = (uint8_t)((n + in[i - 1]) >> 2);
        out[2 * i + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[2 * w - 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[2 * w - 1] = in[w - 1];
    return out;
}
inline const uint8_t* up_hv2(uint8_t* out, const uint8_t* near_, const uint8_t* far_, int w)
{
    int cur = 3 * near_[0] + far_[0]; // the vertical filter, unscaled
    if (w == 1) { out[0] = out[1] = (uint8_t)((cur + 2) >> 2); return out; }
    out[0] = (uint8_t)((cur + 2) >> 2);
    for (int i = 1; i < w; i++) {
        const int prev = cur;
        cur = 3 * near_[i] + far_[i];
        out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
        out[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
    }
    out[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
    return out;
}
inline const uint8_t* up_repeat(uint8_t* out, const uint8_t* in, int w, int hs)
{
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = in[i];
    return out;
}

inline uint8_t mul8(uint8_t x, uint8_t y) // rounded x * y / 255 (stb_image.h:3858)
{
    const unsigned t = (unsigned)x * y + 128u;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

// one row of YCbCr -> RGB, 20-bit fixed point (stb_image.h:3657-3683)
inline void ycc_row(uint8_t* out, const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int n)
{
    const int kr = (int)(1.40200f * 4096.0f + 0.5f) << 8, kgr = (int)(0.71414f * 4096.0f + 0.5f) << 8,
              kgb = (int)(0.34414f * 4096.0f + 0.5f) << 8, kb = (int)(1.77200f * 4096.0f + 0.5f) << 8;
    for (int i = 0; i < n; i++) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int r_ = cr[i] - 128, b_ = cb[i] - 128;
        int r = yf + r_ * kr;
        int g = yf + r_ * -kgr + (int)((uint32_t)(b_ * -kgb) & 0xffff0000u);
        int b = yf + b_ * kb;
        r >>= 20; g >>= 20; b >>= 20;
        out[3 * i] = clamp8(r); out[3 * i + 1] = clamp8(g); out[3 * i + 2] = clamp8(b);
    }
}

} // namespace detail

inline bool looks_like_jpeg(const std::vector<uint8_t>& d) { return d.size() >= 2 && d[0] == 0xFF && d[1] == 0xD8; }

// stbi_load(path, &x, &y, &comp, 0) for a JPEG file: comp = 1 (one component) or 3 (three or four components)
inline std::string decode(const std::vector<uint8_t>& file, Image& img)
{
    using namespace detail;
    Decoder z;
    z.data = file.data(); z.size = file.size();
    if (!z.decode_planes()) return z.err.empty() ? std::string("JPEG: corrupt file") : z.err;
    const int W = z.width, H = z.height, nc = z.ncomp;
    const int n = nc >= 3 ? 3 : 1;
    const bool is_rgb = nc == 3 && (z.rgb_ids == 3 || (z.adobe_transform == 0 && !z.jfif));
    struct Up { int hs, vs, w_lores, ystep, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } up[4];
    for (int k = 0; k < nc; k++) {
        Up& u = up[k];
        u.hs = z.h_max / z.comp[k].h; u.vs = z.v_max / z.comp[k].v;
        u.ystep = u.vs >> 1; u.w_lores = (W + u.hs - 1) / u.hs; u.ypos = 0;
        u.line0 = u.line1 = z.comp[k].plane.data();
        u.buf.assign((size_t)W + 8, 0);
    }
    img.width = W; img.height = H; img.comp = n;
    img.px.assign((size_t)W * H * n, 0);
    const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < H; j++) {
        uint8_t* out = img.px.data() + (size_t)n * W * j;
        for (int k = 0; k < nc; k++) {
            Up& u = up[k];
            const bool bottom = u.ystep >= (u.vs >> 1);
            const uint8_t* near_ = bottom ? u.line1 : u.line0;
            const uint8_t* far_ = bottom ? u.line0 : u.line1;
            if (u.hs == 1 && u.vs == 1) row[k] = near_;
            else if (u.hs == 1 && u.vs == 2) row[k] = up_v2(u.buf.data(), near_, far_, u.w_lores);
            else if (u.hs == 2 && u.vs == 1) row[k] = up_h2(u.buf.data(), near_, u.w_lores);
            else if (u.hs == 2 && u.vs == 2) row[k] = up_hv2(u.buf.data(), near_, far_, u.w_lores);
            else row[k] = up_repeat(u.buf.data(), near_, u.w_lores, u.hs);
            if (++u.ystep >= u.vs) {
                u.ystep = 0;
                u.line0 = u.line1;
                if (++u.ypos < z.comp[k].y) u.line1 += z.comp[k].w2;
            }
        }
        if (n == 1) { std::memcpy(out, row[0], (size_t)W); continue; }
        if (nc == 3) {
            if (is_rgb) for (int i = 0; i < W; i++) { out[3 * i] = row[0][i]; out[3 * i + 1] = row[1][i]; out[3 * i + 2] = row[2][i]; }
            else ycc_row(out, row[0], row[1], row[2], W);
        } else { // four components: CMYK, YCCK, or YCbCr with a fourth channel that is ignored
            if (z.adobe_transform == 0) {
                for (int i = 0; i < W; i++) { const uint8_t m = row[3][i]; out[3 * i] = mul8(row[0][i], m); out[3 * i + 1] = mul8(row[1][i], m); out[3 * i + 2] = mul8(row[2][i], m); }
            } else {
                ycc_row(out, row[0], row[1], row[2], W);
                if (z.adobe_transform == 2)
                    for (int i = 0; i < W; i++) { const uint8_t m = row[3][i]; for (int c = 0; c < 3; c++) out[3 * i + c] = mul8((uint8_t)(255 - out[3 * i + c]), m); }
            }
        }
    }
    return std::string();
}

} // namespace crtjpg

#endif
