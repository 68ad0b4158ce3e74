// tsar_jpeg.h — JPEG (ITU-T T.81 / JFIF) decoding for the C++ host tools: the image bytes the reference's matcher starts from.
//
// The reference reads its views with OpenCV, `imread(path, IMREAD_GRAYSCALE)` (main.cpp:1302; IMREAD_COLOR with
// -color_processing, :1304), and the scenes it is run on hold JPEGs (scripts/courtyard.sh:7,16).  OpenCV hands a JPEG to libjpeg
// with out_color_space = JCS_GRAYSCALE for the former: the result is the LUMINANCE COMPONENT AS DECODED — no colour conversion,
// no chroma — through libjpeg's default "islow" inverse DCT; for the latter JCS_RGB with "fancy" chroma upsampling.  OpenCV and
// libjpeg are not part of /root/reference, so this file restates the PUBLISHED algorithms (T.81 for the entropy coding; the
// Loeffler-Ligtenberg-Moschytz 13-bit fixed-point inverse DCT, the triangle-filter chroma upsampling and the 16-bit fixed-point
// YCbCr conversion that libjpeg documents) and is pinned bit for bit against the libjpeg-turbo inside Pillow, which is in this image:
// tests/test_jpeg_decode.py (baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, optimised tables, restart intervals, odd sizes, real
// photographs).  Parity with OpenCV 3.4.5 itself: unpinned (not in the image), argued from its use of the same library.
//
// Supported: 8-bit baseline / extended sequential / progressive Huffman JPEGs with one (gray) or three (YCbCr) components.
// Not supported (the caller is told, never a silently different image): arithmetic coding, 12-bit, lossless, CMYK / YCCK / RGB
// component JPEGs, DNL; for the colour path also chroma layouts other than 4:4:4 / 4:2:2 / 4:2:0.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

namespace tsar_jpeg {

enum Want { LUMA = 0, BLUE = 1 };   // BLUE: the B of libjpeg's RGB output (what the colour path of the tools takes, tsar_io.h)

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool set = false;
    uint8_t vals[256];
    uint16_t look[512];        // 9 bits of lookahead: (length << 8) | symbol, 0 = longer than 9 bits
    int32_t maxcode[18];       // largest code of each length, -1 if none
    int32_t valoff[17];        // index of the first symbol of a length minus its first code
    int32_t fast[1024];        // AC tables, 10 bits of lookahead: (coefficient << 8) | (run << 4) | bits taken, when the code AND its
                               // magnitude bits fit (size > 0); 0 otherwise
    bool build(const uint8_t counts[17], const uint8_t* symbols, int n) {
        memset(look, 0, sizeof look);
        memcpy(vals, symbols, (size_t)n);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valoff[l] = k - code;
            for (int i = 0; i < counts[l]; i++, code++, k++) {
                if (code >= (1 << l)) return false;            // over-subscribed
                if (l <= 9) {
                    const int first = code << (9 - l);
                    for (int j = 0; j < (1 << (9 - l)); j++) look[first + j] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = counts[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        for (int i = 0; i < 1024; i++) {
            fast[i] = 0;
            const uint16_t e = look[i >> 1];
            if (!e) continue;
            const int len = e >> 8, run = (e & 255) >> 4, size = e & 15;
            if (size == 0 || len + size > 10) continue;
            const int v = (i >> (10 - len - size)) & ((1 << size) - 1);
            const int coef = v < (1 << (size - 1)) ? v - (1 << size) + 1 : v;
            fast[i] = coef * 256 + ((run << 4) | (len + size));
        }
        set = true;
        return k == n;
    }
};

// entropy-coded segment reader: byte stuffing (FF 00) removed, zeros fed once a marker is reached (the marker stays unread)
struct BitReader {
    const uint8_t* p = nullptr;
    const uint8_t* end = nullptr;
    uint64_t acc = 0;
    int n = 0;
    bool at_marker = false;
    int starved = 0;           // bits taken after the data ran out (a truncated or corrupt scan)
    void fill() {
        if (!at_marker && end - p >= 8 && n <= 56) {            // eight bytes at once when none of them is FF (stuffing or a marker)
            uint64_t v;
            memcpy(&v, p, 8);
            const uint64_t x = ~v;
            if (!((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull)) {
                v = __builtin_bswap64(v);
                const int k = (64 - n) >> 3;
                acc = k == 8 ? v : (acc << (8 * k)) | (v >> (64 - 8 * k));
                n += 8 * k;
                p += k;
                return;
            }
        }
        while (n <= 56) {
            uint32_t b = 0;
            if (!at_marker) {
                if (p >= end) at_marker = true;
                else if (*p != 0xFF) b = *p++;
                else if (p + 1 < end && p[1] == 0x00) { b = 0xFF; p += 2; }
                else at_marker = true;
            }
            if (at_marker) starved += 8;
            acc = (acc << 8) | b;
            n += 8;
        }
    }
    inline uint32_t peek(int k) { if (n < k) fill(); return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1u); }
    inline void skip(int k) { n -= k; }
    inline uint32_t bits(int k) { if (k == 0) return 0; const uint32_t v = peek(k); n -= k; return v; }
    inline int bit() { return (int)bits(1); }
    // zeros fed after the data ran out sit at the low end of the accumulator: consumed ones mean the scan was short
    bool overran() const { return starved > n; }
    void reset_at(const uint8_t* q) { p = q; acc = 0; n = 0; at_marker = false; starved = 0; }
};

static inline int decode_symbol(BitReader& br, const HuffTable& t) {
    const uint32_t c = br.peek(16);
    const uint16_t e = t.look[c >> 7];
    if (e) { br.skip(e >> 8); return e & 255; }
    for (int l = 10; l <= 16; l++) {
        const int32_t code = (int32_t)(c >> (16 - l));
        if (code <= t.maxcode[l]) { br.skip(l); return t.vals[(t.valoff[l] + code) & 255]; }
    }
    br.skip(16);
    return -1;
}
static inline int extend(uint32_t v, int s) { return (int)v < (1 << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v; }

struct Component {
    int id = 0, hs = 1, vs = 1, tq = 0;
    int wc = 0, hc = 0;              // size in samples (ceil(W hs / hmax), ...)
    int bw = 0, bh = 0;              // blocks that carry image samples (non-interleaved scans code exactly these)
    int bw_alloc = 0, bh_alloc = 0;  // blocks padded to whole MCUs (interleaved scans code these)
    bool needed = false;
    std::vector<int16_t> coef;       // progressive: [bh_alloc][bw_alloc][64], natural order
    std::vector<uint8_t> plane;      // decoded samples, bw_alloc * 8 wide
    int pred = 0;
    int td = 0, ta = 0;              // tables of the current scan
    int8_t coef_bits[64];            // progressive: the point transform each coefficient (zigzag index) was last coded with, -1 = never
};

// clamp(v + 128) as libjpeg's post-IDCT table does it: the argument is taken modulo 1024
static inline uint8_t idct_limit(int64_t v) {
    const int x = (int)(v & 1023);
    if (x < 128) return (uint8_t)(x + 128);
    if (x < 512) return 255;
    if (x < 896) return 0;
    return (uint8_t)(x - 896);
}

// The "islow" inverse DCT: Loeffler-Ligtenberg-Moschytz, 13-bit constants, two extra bits kept between the passes; columns
// first, then rows; each pass rounds once.  coef in natural order, the quantised values; q the quantisation table.  64-bit
// intermediates like libjpeg's `long` (a 32-bit form with range checks was measured slower on x86-64).
static inline void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, size_t stride) {
    const int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                  F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int64_t ws[64];
    for (int c = 0; c < 8; c++) {
        const int16_t* in = coef + c;
        const uint16_t* qq = q + c;
        if (!(in[8] | in[16] | in[24] | in[32] | in[40] | in[48] | in[56])) {
            const int64_t dc = (int64_t)in[0] * qq[0] * 4;        // the general formula's value for a column without AC terms
            for (int r = 0; r < 8; r++) ws[r * 8 + c] = dc;
            continue;
        }
        int64_t z2 = (int64_t)in[16] * qq[16], z3 = (int64_t)in[48] * qq[48];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t t2 = z1 - z3 * F1_847, t3 = z1 + z2 * F0_765;
        z2 = (int64_t)in[0] * qq[0];
        z3 = (int64_t)in[32] * qq[32];
        int64_t t0 = (z2 + z3) * 8192, t1 = (z2 - z3) * 8192;
        const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = (int64_t)in[56] * qq[56];
        t1 = (int64_t)in[40] * qq[40];
        t2 = (int64_t)in[24] * qq[24];
        t3 = (int64_t)in[8] * qq[8];
        z1 = t0 + t3;
        z2 = t1 + t2;
        z3 = t0 + t2;
        int64_t z4 = t1 + t3;
        const int64_t z5 = (z3 + z4) * F1_175;
        t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        const int64_t rnd = 1 << 10;
        ws[0 * 8 + c] = (t10 + t3 + rnd) >> 11;
        ws[7 * 8 + c] = (t10 - t3 + rnd) >> 11;
        ws[1 * 8 + c] = (t11 + t2 + rnd) >> 11;
        ws[6 * 8 + c] = (t11 - t2 + rnd) >> 11;
        ws[2 * 8 + c] = (t12 + t1 + rnd) >> 11;
        ws[5 * 8 + c] = (t12 - t1 + rnd) >> 11;
        ws[3 * 8 + c] = (t13 + t0 + rnd) >> 11;
        ws[4 * 8 + c] = (t13 - t0 + rnd) >> 11;
    }
    for (int r = 0; r < 8; r++) {
        const int64_t* w = ws + r * 8;
        uint8_t* o = out + (size_t)r * stride;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t t2 = z1 - z3 * F1_847, t3 = z1 + z2 * F0_765;
        int64_t t0 = (w[0] + w[4]) * 8192, t1 = (w[0] - w[4]) * 8192;
        const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
        z1 = t0 + t3;
        z2 = t1 + t2;
        z3 = t0 + t2;
        int64_t z4 = t1 + t3;
        const int64_t z5 = (z3 + z4) * F1_175;
        t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        const int64_t rnd = 1 << 17;
        o[0] = idct_limit((t10 + t3 + rnd) >> 18);
        o[7] = idct_limit((t10 - t3 + rnd) >> 18);
        o[1] = idct_limit((t11 + t2 + rnd) >> 18);
        o[6] = idct_limit((t11 - t2 + rnd) >> 18);
        o[2] = idct_limit((t12 + t1 + rnd) >> 18);
        o[5] = idct_limit((t12 - t1 + rnd) >> 18);
        o[3] = idct_limit((t13 + t0 + rnd) >> 18);
        o[4] = idct_limit((t13 - t0 + rnd) >> 18);
    }
}


struct Decoder {
    std::vector<uint8_t> file;
    std::string err;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    int adobe_transform = -1;
    uint16_t qt[4][64];
    bool qt_set[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[4];
    int mcus_x = 0, mcus_y = 0;

    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }

    bool load(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return fail("cannot open " + path);
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 4) { fclose(f); return fail("not a JPEG: " + path); }
        file.resize((size_t)sz);
        const bool ok = fread(file.data(), 1, file.size(), f) == file.size();
        fclose(f);
        if (!ok) return fail("short read: " + path);
        if (file[0] != 0xFF || file[1] != 0xD8) return fail("not a JPEG (no SOI): " + path);
        return true;
    }

    static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

    bool parse_dqt(const uint8_t* p, int len) {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1) return fail("bad quantisation table header");
            const int need = 1 + 64 * (pq + 1);
            if (len < need) return fail("short DQT");
            for (int k = 0; k < 64; k++) qt[tq][kZigzag[k]] = (uint16_t)(pq ? be16(p + 1 + 2 * k) : p[1 + k]);
            qt_set[tq] = true;
            p += need; len -= need;
        }
        return true;
    }
    bool parse_dht(const uint8_t* p, int len) {
        while (len > 0) {
            if (len < 17) return fail("short DHT");
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) return fail("bad Huffman table header");
            uint8_t counts[17] = {0};
            int n = 0;
            for (int l = 1; l <= 16; l++) { counts[l] = p[l]; n += p[l]; }
            if (n > 256 || len < 17 + n) return fail("bad Huffman table");
            if (!(tc ? ac[th] : dc[th]).build(counts, p + 17, n)) return fail("inconsistent Huffman table");
            p += 17 + n; len -= 17 + n;
        }
        return true;
    }
    bool parse_sof(const uint8_t* p, int len, int marker) {
        if (have_frame) return fail("two frames in one file");
        if (marker != 0xC0 && marker != 0xC1 && marker != 0xC2)
            return fail(marker >= 0xC9 ? "arithmetic-coded JPEG (not supported)" : "lossless / hierarchical JPEG (not supported)");
        if (len < 6) return fail("short SOF");
        if (p[0] != 8) return fail("12-bit JPEG (not supported)");
        H = be16(p + 1); W = be16(p + 3); ncomp = p[5];
        if (H == 0) return fail("JPEG without a line count (DNL, not supported)");
        if (W == 0) return fail("empty JPEG");
        if (ncomp != 1 && ncomp != 3) return fail("JPEG with " + std::to_string(ncomp) + " components (CMYK / YCCK: not supported)");
        if (len < 6 + 3 * ncomp) return fail("short SOF");
        progressive = marker == 0xC2;
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.id = p[6 + 3 * i];
            c.hs = p[7 + 3 * i] >> 4; c.vs = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i];
            if (c.hs < 1 || c.hs > 4 || c.vs < 1 || c.vs > 4 || c.tq > 3) return fail("bad component header");
            hmax = c.hs > hmax ? c.hs : hmax;
            vmax = c.vs > vmax ? c.vs : vmax;
        }
        if (ncomp == 1) { comp[0].hs = comp[0].vs = 1; hmax = vmax = 1; }    // a single component is never subsampled (T.81 A.2.2)
        if (comp[0].hs != hmax || comp[0].vs != vmax) return fail("JPEG with a subsampled first component (not supported)");
        if ((uint64_t)W * H > (1ull << 30)) return fail("JPEG larger than 2^30 pixels (refused)");
        mcus_x = (W + 8 * hmax - 1) / (8 * hmax);
        mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            c.wc = (W * c.hs + hmax - 1) / hmax;
            c.hc = (H * c.vs + vmax - 1) / vmax;
            c.bw = (c.wc + 7) / 8; c.bh = (c.hc + 7) / 8;
            c.bw_alloc = mcus_x * c.hs; c.bh_alloc = mcus_y * c.vs;
            memset(c.coef_bits, -1, sizeof c.coef_bits);
        }
        have_frame = true;
        return true;
    }

    // one block of a sequential scan; store == nullptr: only the bits are consumed
    bool block_sequential(BitReader& br, Component& c, int16_t* store, bool& any_ac) {
        const int t = decode_symbol(br, dc[c.td]);
        if (t < 0 || t > 15) return fail("bad DC code");
        const int diff = t ? extend(br.bits(t), t) : 0;
        c.pred = (int)((uint32_t)c.pred + (uint32_t)diff);      // (a damaged file may run the predictor out of range: wrap, no UB)
        if (store) store[0] = (int16_t)c.pred;
        const HuffTable& at = ac[c.ta];
        for (int k = 1; k < 64;) {
            const int32_t f = at.fast[br.peek(16) >> 6];
            if (f) {                                             // code and magnitude in one look-up
                k += (f >> 4) & 15;
                if (k > 63) return fail("AC run past the block");
                br.skip(f & 15);
                if (store) { store[kZigzag[k]] = (int16_t)(f >> 8); any_ac = true; }
                k++;
                continue;
            }
            const int rs = decode_symbol(br, at);
            if (rs < 0) return fail("bad AC code");
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > 63) return fail("AC run past the block");
                const int v = extend(br.bits(s), s);
                if (store) { store[kZigzag[k]] = (int16_t)v; any_ac = true; }
                k++;
            } else if (r == 15) k += 16;
            else break;
        }
        return true;
    }

    // progressive scans of one block (T.81 G.1.2)
    bool block_dc_first(BitReader& br, Component& c, int16_t* store, int al) {
        const int t = decode_symbol(br, dc[c.td]);
        if (t < 0 || t > 15) return fail("bad DC code");
        c.pred = (int)((uint32_t)c.pred + (uint32_t)(t ? extend(br.bits(t), t) : 0));
        if (store) store[0] = (int16_t)((uint32_t)c.pred << al);
        return true;
    }
    bool block_ac_first(BitReader& br, const HuffTable& at, int16_t* store, int ss, int se, int al, uint32_t& eobrun) {
        if (eobrun > 0) { eobrun--; return true; }
        for (int k = ss; k <= se;) {
            const int rs = decode_symbol(br, at);
            if (rs < 0) return fail("bad AC code");
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > 63) return fail("AC run past the block");
                store[kZigzag[k]] = (int16_t)(extend(br.bits(s), s) * (1 << al));
                k++;
            } else if (r == 15) k += 16;
            else {
                eobrun = (1u << r) - 1u;
                if (r) eobrun += br.bits(r);
                break;
            }
        }
        return true;
    }
    bool block_ac_refine(BitReader& br, const HuffTable& at, int16_t* store, int ss, int se, int al, uint32_t& eobrun) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        auto refine = [&](int16_t& v) {                          // one correction bit for a coefficient that is already non-zero
            if (br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
        };
        if (eobrun == 0) {
            for (; k <= se; k++) {
                const int rs = decode_symbol(br, at);
                if (rs < 0) return fail("bad AC code");
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s) {
                    if (s != 1) return fail("bad refinement code");
                    value = br.bit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1u << r;
                    if (r) eobrun += br.bits(r);
                    break;
                }
                for (; k <= se; k++) {                            // skip r coefficients that are still zero, refining the others on the way
                    int16_t& v = store[kZigzag[k]];
                    if (v != 0) refine(v);
                    else if (--r < 0) break;
                }
                if (s) {
                    if (k > se) return fail("refinement past the band");
                    store[kZigzag[k]] = (int16_t)value;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t& v = store[kZigzag[k]];
                if (v != 0) refine(v);
            }
            eobrun--;
        }
        return true;
    }

    // the marker that ends an entropy-coded segment (RSTn excluded unless asked for)
    const uint8_t* next_marker(const uint8_t* p, bool stop_at_rst) const {
        const uint8_t* end = file.data() + file.size();
        while (p + 1 < end) {
            if (p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF && (stop_at_rst || p[1] < 0xD0 || p[1] > 0xD7)) return p;
            p++;
        }
        return end;
    }

    bool decode_scan(const uint8_t* hdr, int len, const uint8_t*& pos) {
        if (!have_frame) return fail("scan before the frame header");
        if (len < 1) return fail("short SOS");
        const int ns = hdr[0];
        if (ns < 1 || ns > ncomp || len < 1 + 2 * ns + 3) return fail("bad scan header");
        Component* sc[4];
        for (int i = 0; i < ns; i++) {
            sc[i] = nullptr;
            for (int j = 0; j < ncomp; j++)
                if (comp[j].id == hdr[1 + 2 * i]) sc[i] = &comp[j];
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->td = hdr[2 + 2 * i] >> 4; sc[i]->ta = hdr[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        }
        const int ss = hdr[1 + 2 * ns], se = hdr[2 + 2 * ns], ah = hdr[3 + 2 * ns] >> 4, al = hdr[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) return fail("bad sequential scan parameters");
        const bool dc_scan = ss == 0;
        for (int i = 0; i < ns; i++) {
            if ((!progressive || (dc_scan && ah == 0)) && !dc[sc[i]->td].set) return fail("scan uses an undefined DC table");
            if ((!progressive || !dc_scan) && !ac[sc[i]->ta].set) return fail("scan uses an undefined AC table");
            if (!qt_set[sc[i]->tq]) return fail("component uses an undefined quantisation table");
        }
        if (progressive)
            for (int i = 0; i < ns; i++)
                for (int k = ss; k <= se; k++) sc[i]->coef_bits[k] = (int8_t)al;
        // a progressive AC scan of a component nobody asked for: its entropy-coded bytes are stepped over, not decoded
        if (progressive && !dc_scan && !sc[0]->needed) { pos = next_marker(pos, false); return true; }
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            if (!c.needed) continue;
            if (progressive && c.coef.empty()) c.coef.assign((size_t)c.bw_alloc * c.bh_alloc * 64, 0);
            if (c.plane.empty()) c.plane.assign((size_t)c.bw_alloc * 8 * c.bh_alloc * 8, 0);
        }
        BitReader br;
        br.p = pos; br.end = file.data() + file.size();
        for (int i = 0; i < ns; i++) sc[i]->pred = 0;
        uint32_t eobrun = 0;
        const bool interleaved = ns > 1;
        const int nx = interleaved ? mcus_x : sc[0]->bw, ny = interleaved ? mcus_y : sc[0]->bh;
        int until_restart = restart_interval, expect_rst = 0;
        int16_t scratch[64];
        for (int my = 0; my < ny; my++)
            for (int mx = 0; mx < nx; mx++) {
                if (restart_interval && until_restart == 0) {
                    // byte-align, take the RSTn marker, start afresh
                    const uint8_t* q = br.at_marker ? br.p : next_marker(br.p, true);
                    if (q + 1 >= br.end || q[0] != 0xFF || q[1] != 0xD0 + expect_rst) return fail("restart marker missing or out of sequence");
                    br.reset_at(q + 2);
                    expect_rst = (expect_rst + 1) & 7;
                    until_restart = restart_interval;
                    for (int i = 0; i < ns; i++) sc[i]->pred = 0;
                    eobrun = 0;
                }
                for (int i = 0; i < ns; i++) {
                    Component& c = *sc[i];
                    const int nh = interleaved ? c.hs : 1, nv = interleaved ? c.vs : 1;
                    for (int v = 0; v < nv; v++)
                        for (int h = 0; h < nh; h++) {
                            const int bx = mx * nh + h, by = my * nv + v;
                            int16_t* store = nullptr;
                            if (c.needed) store = progressive ? &c.coef[((size_t)by * c.bw_alloc + bx) * 64] : scratch;
                            bool okb = true;
                            if (!progressive) {
                                bool any_ac = false;
                                if (store) memset(scratch, 0, sizeof scratch);
                                okb = block_sequential(br, c, store, any_ac);
                                if (okb && store) idct_islow(scratch, qt[c.tq], &c.plane[((size_t)by * 8) * ((size_t)c.bw_alloc * 8) + (size_t)bx * 8], (size_t)c.bw_alloc * 8);
                            } else if (dc_scan) {
                                if (ah == 0) okb = block_dc_first(br, c, store, al);
                                else if (br.bit() && store) store[0] = (int16_t)(store[0] | (1 << al));
                            } else if (ah == 0) okb = block_ac_first(br, ac[c.ta], store, ss, se, al, eobrun);
                            else okb = block_ac_refine(br, ac[c.ta], store, ss, se, al, eobrun);
                            if (!okb) return false;
                        }
                }
                if (restart_interval) until_restart--;
                if (br.overran()) return fail("entropy-coded data ends before the scan does (truncated file?)");
            }
        pos = br.at_marker ? br.p : next_marker(br.p, false);
        return true;
    }

    void finish_progressive() {
        for (int i = 0; i < ncomp; i++) {
            Component& c = comp[i];
            if (!c.needed || c.coef.empty()) continue;
            const size_t stride = (size_t)c.bw_alloc * 8;
            for (int by = 0; by < c.bh_alloc; by++)
                for (int bx = 0; bx < c.bw_alloc; bx++)
                    idct_islow(&c.coef[((size_t)by * c.bw_alloc + bx) * 64], qt[c.tq], &c.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
            std::vector<int16_t>().swap(c.coef);
        }
    }

    bool decode(Want want) {
        const uint8_t* p = file.data() + 2;
        const uint8_t* end = file.data() + file.size();
        bool seen_eoi = false, seen_scan = false;
        while (p + 4 <= end && !seen_eoi) {
            if (p[0] != 0xFF) return fail("marker expected");
            while (p + 1 < end && p[1] == 0xFF) p++;            // fill bytes
            const int m = p[1];
            p += 2;
            if (m == 0xD9) { seen_eoi = true; break; }
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; // stand-alone markers
            if (p + 2 > end) break;
            const int len = be16(p) - 2;
            const uint8_t* body = p + 2;
            if (len < 0 || body + len > end) return fail("segment runs past the end of the file");
            p = body + len;
            if (m == 0xDB) { if (!parse_dqt(body, len)) return false; }
            else if (m == 0xC4) { if (!parse_dht(body, len)) return false; }
            else if (m == 0xDD) { if (len < 2) return fail("short DRI"); restart_interval = be16(body); }
            else if (m == 0xEE) { if (len >= 12 && memcmp(body, "Adobe", 5) == 0) adobe_transform = body[11]; }
            else if (m == 0xDC) return fail("DNL (not supported)");
            else if (m >= 0xC0 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
                if (!parse_sof(body, len, m)) return false;
                if (ncomp == 3) {
                    const bool rgb_ids = comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B';
                    if (adobe_transform == 0 || (adobe_transform < 0 && rgb_ids)) return fail("RGB-component JPEG (not supported)");
                }
                comp[0].needed = true;
                if (want == BLUE && ncomp == 3) comp[1].needed = true;
            } else if (m == 0xDA) {
                if (!decode_scan(body, len, p)) return false;
                seen_scan = true;
            }
        }
        if (!have_frame || !seen_scan) return fail("no image data");
        if (progressive) {
            // libjpeg smooths the blocks of a progressive file whose scans stop before the low frequencies are exact (DC and the
            // first AC terms): an interrupted transfer, not a matcher input.  Refused rather than reproduced.
            for (int i = 0; i < ncomp; i++)
                for (int k = 0; k < 10 && comp[i].needed; k++)
                    if (comp[i].coef_bits[k] != 0) return fail("progressive JPEG whose scans do not complete the low frequencies (refused)");
            finish_progressive();
        }
        return true;
    }
};

// the chroma plane at full resolution: libjpeg's "fancy" upsampling (a 3:1 triangle filter along each halved axis, edges
// replicated; plain replication when the plane is one or two samples wide)
static inline bool upsample(const Component& c, int hmax, int vmax, int W, int H, std::vector<uint8_t>& out, std::string& err) {
    const size_t stride = (size_t)c.bw_alloc * 8;
    const int hx = hmax / c.hs, vx = vmax / c.vs;
    if (hmax % c.hs || vmax % c.vs || !((hx == 1 && vx == 1) || (hx == 2 && vx == 1) || (hx == 2 && vx == 2))) {
        err = "chroma layout " + std::to_string(hx) + "x" + std::to_string(vx) + " (colour path: only 4:4:4, 4:2:2, 4:2:0)";
        return false;
    }
    out.resize((size_t)W * H);
    const int wc = c.wc, hc = c.hc;
    const bool fancy = wc > 2;
    std::vector<int> col((size_t)wc + 1);
    for (int y = 0; y < H; y++) {
        uint8_t* o = out.data() + (size_t)y * W;
        if (hx == 1) { memcpy(o, &c.plane[(size_t)y * stride], (size_t)W); continue; }
        const int yc = vx == 2 ? y >> 1 : y;
        const uint8_t* r0 = &c.plane[(size_t)yc * stride];
        if (!fancy) { for (int x = 0; x < W; x++) o[x] = r0[x >> 1]; continue; }
        if (vx == 1) {
            for (int x = 0; x < W; x++) {
                const int i = x >> 1;
                if (x & 1) o[x] = i == wc - 1 ? r0[i] : (uint8_t)((3 * r0[i] + r0[i + 1] + 2) >> 2);
                else o[x] = i == 0 ? r0[0] : (uint8_t)((3 * r0[i] + r0[i - 1] + 1) >> 2);
            }
        } else {
            int yn = (y & 1) ? yc + 1 : yc - 1;                  // the nearer neighbouring chroma row
            yn = yn < 0 ? 0 : (yn > hc - 1 ? hc - 1 : yn);
            const uint8_t* r1 = &c.plane[(size_t)yn * stride];
            for (int i = 0; i < wc; i++) col[i] = 3 * r0[i] + r1[i];
            for (int x = 0; x < W; x++) {
                const int i = x >> 1;
                if (x & 1) o[x] = i == wc - 1 ? (uint8_t)((col[i] * 4 + 7) >> 4) : (uint8_t)((col[i] * 3 + col[i + 1] + 7) >> 4);
                else o[x] = i == 0 ? (uint8_t)((col[0] * 4 + 8) >> 4) : (uint8_t)((col[i] * 3 + col[i - 1] + 8) >> 4);
            }
        }
    }
    return true;
}

// width and height from the frame header alone
static inline bool size(const std::string& path, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t b[4];
    bool ok = fread(b, 1, 2, f) == 2 && b[0] == 0xFF && b[1] == 0xD8, found = false;
    while (ok && !found) {
        if (fread(b, 1, 2, f) != 2 || b[0] != 0xFF) { ok = false; break; }
        while (b[1] == 0xFF) { if (fread(b + 1, 1, 1, f) != 1) { ok = false; break; } }
        if (!ok) break;
        const int m = b[1];
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9 || m == 0xDA || fread(b + 2, 1, 2, f) != 2) { ok = false; break; }
        const int len = ((b[2] << 8) | b[3]) - 2;
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            uint8_t s[5];
            if (len < 5 || fread(s, 1, 5, f) != 5) { ok = false; break; }
            h = (s[1] << 8) | s[2]; w = (s[3] << 8) | s[4];
            found = true;
        } else if (len < 0 || fseek(f, len, SEEK_CUR) != 0) ok = false;
    }
    fclose(f);
    return ok && found && w > 0 && h > 0;
}

// The image as 8-bit samples, w * h, row-major: LUMA = libjpeg's JCS_GRAYSCALE output (the Y plane of a YCbCr file, the only plane
// of a gray one); BLUE = the B of its RGB output (Y + 1.772 (Cb - 128) in 16-bit fixed point, clamped; a gray file: Y).
static inline bool read(const std::string& path, Want want, std::vector<uint8_t>& img, int& w, int& h, std::string* why = nullptr) {
    Decoder d;
    std::vector<uint8_t> cb;
    bool ok = false;
    try {
        ok = d.load(path) && d.decode(want);
        if (ok && want == BLUE && d.ncomp == 3) ok = upsample(d.comp[1], d.hmax, d.vmax, d.W, d.H, cb, d.err);
    } catch (const std::bad_alloc&) { d.err = "out of memory"; ok = false; }
    if (!ok) { if (why) *why = d.err; return false; }
    w = d.W; h = d.H;
    img.resize((size_t)w * h);
    const Component& y = d.comp[0];
    const size_t stride = (size_t)y.bw_alloc * 8;
    for (int r = 0; r < h; r++) memcpy(&img[(size_t)r * w], &y.plane[(size_t)r * stride], (size_t)w);
    if (!cb.empty()) {
        int tab[256];
        for (int i = 0; i < 256; i++) tab[i] = (int)((116130LL * (i - 128) + 32768) >> 16);      // 1.772 * 65536 rounded, half added, floor
        for (size_t i = 0; i < img.size(); i++) {
            const int v = img[i] + tab[cb[i]];
            img[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    return true;
}

}  // namespace tsar_jpeg
