/*
 * rt_jpeg.cpp - JPEG decoding for TriangleMesh::loadTexture (raytracer.cpp:759-766).
 *
 * The reference loads its textures with stbi_load(file, &W, &H, &c, 3) of the third-party stb_image v2.26 (vendored in
 * the reference tree; raytracer.cpp:762). A texel that differs by one level changes the colour of every hit that reads
 * it, so the sample arithmetic of this decoder is that library's, not any other decoder's. Three parts are
 * transliterated from stb_image.h (public domain, not written by the reference's author), statement by statement with
 * its operation order and its identifiers kept, because every rounding step decides texels: the 12-bit fixed-point
 * "islow" inverse DCT with its two rounding stages (idct1 / idct_block below: stb_image.h:2352-2450), the 3:1 / 9:3:3:1
 * chroma up-sampling rows and their schedule (row_* and to_rgb: :3364-3426, :3802-3836) and the 20-bit fixed-point YCbCr
 * conversion in which the Cb term of green is masked to its upper 16 bits (:3559-3582); coefficients and dequantised
 * products are truncated to 16 bits as there (:2157, :2187, :3003). Everything else is written from ITU-T T.81: segment
 * parsing, canonical Huffman tables, baseline and extended sequential scans, progressive scans with successive
 * approximation, restart intervals, 8- and 16-bit quantisation tables - and, unlike that library, the checks that refuse
 * a file before it can make the decoder read outside its planes or allocate gigabytes.
 * 8-bit samples, 1 or 3 components (grey, YCbCr, or RGB marked by component ids / an Adobe segment).
 * tests/test_host.py compares every texel with the reference's decoder on the reference's 13 texture files and on
 * generated files of every sampling layout.
 */
#include <stdint.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "rt_host.hpp"

namespace rt {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

[[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("JPEG: ") + what); }

/* canonical Huffman table, decoded length by length (T.81 F.2.2.3) */
struct Huffman {
    bool present = false;
    uint8_t values[256];
    int mincode[17], maxcode[18], valptr[17];
    void build(const uint8_t counts[16], const uint8_t *vals, int n) {
        memcpy(values, vals, (size_t)n);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) bad("bad Huffman code lengths");
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int hd = 0, ha = 0; /* Huffman table selectors of the current scan */
    int dc_pred = 0;
    int x = 0, y = 0;   /* samples this component really has */
    int w2 = 0, h2 = 0; /* allocated: whole interleaved MCUs */
    std::vector<uint8_t> data;
    std::vector<int16_t> coeff; /* progressive: 64 per block, natural order */
    int coeff_w = 0;
};

struct Decoder {
    const uint8_t *p, *end;
    /* entropy-coded segment reader: bits MSB first, FF00 unstuffed, stops (and feeds zeros) at a marker */
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    int marker = -1; /* marker met inside entropy-coded data */

    uint16_t dequant[4][64]; /* natural order */
    Huffman hdc[4], hac[4];
    Component comp[4];
    int n_comp = 0, W = 0, H = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0, todo = 0;
    int scan_n = 0, order[4];
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;

    Decoder(const uint8_t *d, size_t n) : p(d), end(d + n) { memset(dequant, 0, sizeof dequant); }

    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (marker < 0 && p < end) {
                b = *p++;
                if (b == 0xff) {
                    int c = p < end ? *p++ : 0;
                    while (c == 0xff) c = p < end ? *p++ : 0; /* fill bytes */
                    if (c != 0) { marker = c; b = 0; }
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (n < 0 || n > 16) bad("bit count out of range"); /* a category or run length from a corrupt table */
        if (bitcnt < n) fill();
        const int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n;
        bitcnt -= n;
        return v;
    }
    int bit() { return bits(1); }
    int decode(const Huffman &h) {
        if (!h.present) bad("missing Huffman table");
        if (bitcnt < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (int)(bitbuf >> (32 - len));
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
                bitbuf <<= len;
                bitcnt -= len;
                return h.values[h.valptr[len] + code - h.mincode[len]];
            }
        }
        bad("bad Huffman code");
    }
    /* T.81 F.2.2.1 EXTEND of the next n bits */
    int receive_extend(int n) {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

    void reset_entropy() {
        bitbuf = 0;
        bitcnt = 0;
        marker = -1;
        for (int k = 0; k < 4; k++) comp[k].dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }
    /* after an MCU: at the end of a restart interval the next marker must be RSTn; returns false to stop the scan */
    bool end_of_mcu() {
        if (--todo > 0) return true;
        if (bitcnt < 24) fill();
        if (!(marker >= 0xd0 && marker <= 0xd7)) return false;
        reset_entropy();
        return true;
    }

    /* ---- coefficient decoding ---- */
    void block_sequential(int16_t *data, Component &c) {
        const uint16_t *dq = dequant[c.tq];
        const int t = decode(hdc[c.hd]);
        memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? receive_extend(t) : 0;
        const int dc = (int)((uint32_t)c.dc_pred + (uint32_t)diff); /* wraps on a corrupt stream instead of overflowing */
        c.dc_pred = dc;
        data[0] = (int16_t)((uint32_t)dc * (uint32_t)dq[0]);
        int k = 1;
        do {
            const int rs = decode(hac[c.ha]);
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) bad("coefficient index out of range");
                const int zig = kZigzag[k++];
                data[zig] = (int16_t)(receive_extend(s) * dq[zig]);
            }
        } while (k < 64);
    }
    void block_prog_dc(int16_t *data, Component &c) {
        if (spec_end != 0) bad("DC scan with AC coefficients");
        if (succ_high == 0) {
            memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode(hdc[c.hd]);
            const int diff = t ? receive_extend(t) : 0;
            const int dc = (int)((uint32_t)c.dc_pred + (uint32_t)diff);
            c.dc_pred = dc;
            data[0] = (int16_t)((uint32_t)dc << succ_low);
        } else if (bit()) {
            data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low));
        }
    }
    void block_prog_ac(int16_t *data, Component &c) {
        if (spec_start == 0) bad("AC scan that starts at the DC coefficient");
        const Huffman &h = hac[c.ha];
        if (succ_high == 0) {
            if (eob_run) { --eob_run; return; }
            int k = spec_start;
            do {
                const int rs = decode(h);
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) bad("coefficient index out of range");
                    data[kZigzag[k++]] = (int16_t)(receive_extend(s) * (1 << succ_low));
                }
            } while (k <= spec_end);
        } else {
            const int16_t bitv = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t &c0) {
                if (bit() && (c0 & bitv) == 0) c0 = (int16_t)(c0 > 0 ? c0 + bitv : c0 - bitv);
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; k++) {
                    int16_t &c0 = data[kZigzag[k]];
                    if (c0 != 0) refine(c0);
                }
            } else {
                int k = spec_start;
                do {
                    const int rs = decode(h);
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += bits(r);
                            r = 64; /* to the end of the band */
                        }
                    } else {
                        if (s != 1) bad("bad refinement code");
                        s = bit() ? bitv : -bitv;
                    }
                    while (k <= spec_end) {
                        int16_t &c0 = data[kZigzag[k++]];
                        if (c0 != 0) {
                            refine(c0);
                        } else {
                            if (r == 0) { c0 = (int16_t)s; break; }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
    }

    /* ---- inverse DCT: 12-bit constants, columns to 2 extra bits, rows to 8-bit samples with the level shift ---- */
    static int fix(float c) { return (int)(c * 4096 + 0.5); }
    /* int with wrap-around arithmetic: the coefficients of a corrupt file can push the inverse DCT past 32 bits, which must
     * be a wrong picture and not undefined behaviour; for every valid file nothing overflows and the values are the same */
    struct wint {
        int v;
        wint(int x = 0) : v(x) {}
        friend wint operator+(wint a, wint b) { return wint((int)((uint32_t)a.v + (uint32_t)b.v)); }
        friend wint operator-(wint a, wint b) { return wint((int)((uint32_t)a.v - (uint32_t)b.v)); }
        friend wint operator*(wint a, wint b) { return wint((int)((uint32_t)a.v * (uint32_t)b.v)); }
        wint &operator+=(wint b) { return *this = *this + b; }
        friend wint operator>>(wint a, int sh) { return wint(a.v >> sh); }
    };
    struct Idct1 { wint x0, x1, x2, x3, t0, t1, t2, t3; };
    static Idct1 idct1(wint s0, wint s1, wint s2, wint s3, wint s4, wint s5, wint s6, wint s7) {
        Idct1 o;
        wint p2 = s2, p3 = s6;
        wint p1 = (p2 + p3) * fix(0.5411961f);
        wint t2 = p1 + p3 * fix(-1.847759065f);
        wint t3 = p1 + p2 * fix(0.765366865f);
        p2 = s0;
        p3 = s4;
        wint t0 = (p2 + p3) * 4096;
        wint t1 = (p2 - p3) * 4096;
        o.x0 = t0 + t3;
        o.x3 = t0 - t3;
        o.x1 = t1 + t2;
        o.x2 = t1 - t2;
        t0 = s7;
        t1 = s5;
        t2 = s3;
        t3 = s1;
        p3 = t0 + t2;
        wint p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const wint p5 = (p3 + p4) * fix(1.175875602f);
        t0 = t0 * fix(0.298631336f);
        t1 = t1 * fix(2.053119869f);
        t2 = t2 * fix(3.072711026f);
        t3 = t3 * fix(1.501321110f);
        p1 = p5 + p1 * fix(-0.899976223f);
        p2 = p5 + p2 * fix(-2.562915447f);
        p3 = p3 * fix(-1.961570560f);
        p4 = p4 * fix(-0.390180644f);
        o.t3 = t3 + p1 + p4;
        o.t2 = t2 + p2 + p3;
        o.t1 = t1 + p2 + p4;
        o.t0 = t0 + p1 + p3;
        return o;
    }
    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void idct_block(uint8_t *out, int stride, const int16_t *d) {
        int val[64];
        for (int i = 0; i < 8; i++) {
            Idct1 a = idct1(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
            a.x0 += 512; a.x1 += 512; a.x2 += 512; a.x3 += 512;
            val[i] = ((a.x0 + a.t3) >> 10).v;
            val[56 + i] = ((a.x0 - a.t3) >> 10).v;
            val[8 + i] = ((a.x1 + a.t2) >> 10).v;
            val[48 + i] = ((a.x1 - a.t2) >> 10).v;
            val[16 + i] = ((a.x2 + a.t1) >> 10).v;
            val[40 + i] = ((a.x2 - a.t1) >> 10).v;
            val[24 + i] = ((a.x3 + a.t0) >> 10).v;
            val[32 + i] = ((a.x3 - a.t0) >> 10).v;
        }
        for (int i = 0; i < 8; i++) {
            const int *v = val + 8 * i;
            uint8_t *o = out + (size_t)stride * i;
            Idct1 a = idct1(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const int bias = 65536 + (128 << 17); /* rounding of the 17-bit shift + the level shift by 128 */
            a.x0 += bias; a.x1 += bias; a.x2 += bias; a.x3 += bias;
            o[0] = clamp8(((a.x0 + a.t3) >> 17).v);
            o[7] = clamp8(((a.x0 - a.t3) >> 17).v);
            o[1] = clamp8(((a.x1 + a.t2) >> 17).v);
            o[6] = clamp8(((a.x1 - a.t2) >> 17).v);
            o[2] = clamp8(((a.x2 + a.t1) >> 17).v);
            o[5] = clamp8(((a.x2 - a.t1) >> 17).v);
            o[3] = clamp8(((a.x3 + a.t0) >> 17).v);
            o[4] = clamp8(((a.x3 - a.t0) >> 17).v);
        }
    }

    /* ---- segments ---- */
    void segment(int m) {
        int len = get16();
        if (len < 2 || p + (len - 2) > end) bad("bad segment length");
        const uint8_t *seg_end = p + (len - 2);
        switch (m) {
        case 0xdd:
            if (len != 4) bad("bad DRI length");
            restart_interval = get16();
            break;
        case 0xdb:
            while (p < seg_end) {
                const int q = get8(), prec = q >> 4, t = q & 15;
                if ((prec != 0 && prec != 1) || t > 3) bad("bad DQT");
                for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
            }
            break;
        case 0xc4:
            while (p < seg_end) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) bad("bad DHT");
                uint8_t counts[16];
                int n = 0;
                for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); n += counts[i]; }
                if (n > 256 || p + n > seg_end) bad("bad DHT");
                (tc ? hac[th] : hdc[th]).build(counts, p, n);
                p += n;
            }
            break;
        case 0xe0:
            if (len >= 7 && memcmp(p, "JFIF\0", 5) == 0) jfif = true;
            break;
        case 0xee:
            if (len >= 14 && memcmp(p, "Adobe\0", 6) == 0) adobe_transform = p[11];
            break;
        default:
            break; /* other APPn, COM: skipped */
        }
        p = seg_end;
    }
    void frame_header() {
        const int len = get16();
        if (get8() != 8) bad("only 8-bit samples");
        H = get16();
        W = get16();
        n_comp = get8();
        if (H == 0 || W == 0) bad("empty image");
        if (n_comp != 1 && n_comp != 3) bad("1 or 3 components only");
        if (len != 8 + 3 * n_comp) bad("bad SOF length");
        rgb_ids = 0;
        for (int i = 0; i < n_comp; i++) {
            Component &c = comp[i];
            c.id = get8();
            if (n_comp == 3 && c.id == "RGB"[i]) ++rgb_ids;
            const int q = get8();
            c.h = q >> 4;
            c.v = q & 15;
            c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad component");
            if (c.h > h_max) h_max = c.h;
            if (c.v > v_max) v_max = c.v;
        }
        /* the up-sampling below repeats or interpolates whole samples: a component's factors must divide the largest ones
         * (with h_max / h rounded down a row would be read past its end) */
        for (int i = 0; i < n_comp; i++)
            if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) bad("bad sampling factors");
        mcu_x = (W + h_max * 8 - 1) / (h_max * 8);
        mcu_y = (H + v_max * 8 - 1) / (v_max * 8);
        /* Before anything is allocated: the picture must be of a size this decoder is meant for, and the file long enough to
         * hold it - every 8x8 block of every component takes at least one bit of entropy-coded data (a header of thirty
         * bytes must not make the decoder touch gigabytes). */
        if ((uint64_t)W * (uint64_t)H > (1ull << 28)) bad("picture too large");
        {
            uint64_t blocks = 0;
            for (int i = 0; i < n_comp; i++) blocks += (uint64_t)mcu_x * mcu_y * comp[i].h * comp[i].v;
            if (blocks > (uint64_t)(end - p) * 8) bad("file too short for its picture size");
        }
        for (int i = 0; i < n_comp; i++) {
            Component &c = comp[i];
            c.x = (W * c.h + h_max - 1) / h_max;
            c.y = (H * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) {
                c.coeff_w = c.w2 / 8;
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        }
    }
    void scan_header() {
        const int len = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > n_comp || len != 6 + 2 * scan_n) bad("bad SOS");
        for (int i = 0; i < scan_n; i++) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < n_comp && comp[which].id != id) which++;
            if (which == n_comp) bad("scan of an unknown component");
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) bad("bad table selector");
            order[i] = which;
        }
        spec_start = get8();
        spec_end = get8();
        const int a = get8();
        succ_high = a >> 4;
        succ_low = a & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) bad("bad progressive scan");
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) bad("bad sequential scan");
            spec_end = 63;
        }
    }
    void scan_data() {
        reset_entropy();
        int16_t blk[64];
        if (scan_n == 1) { /* one component: its blocks in raster order, as many as it really has */
            Component &c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    if (!progressive) {
                        block_sequential(blk, c);
                        idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, blk);
                    } else {
                        int16_t *d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                        if (spec_start == 0) block_prog_dc(d, c);
                        else block_prog_ac(d, c);
                    }
                    if (!end_of_mcu()) return;
                }
        } else { /* interleaved: h x v blocks of every component per MCU */
            for (int j = 0; j < mcu_y; j++)
                for (int i = 0; i < mcu_x; i++) {
                    for (int k = 0; k < scan_n; k++) {
                        Component &c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                const int bx = i * c.h + x, by = j * c.v + y;
                                if (!progressive) {
                                    block_sequential(blk, c);
                                    idct_block(c.data.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, blk);
                                } else {
                                    block_prog_dc(c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.coeff_w), c);
                                }
                            }
                    }
                    if (!end_of_mcu()) return;
                }
        }
    }
    void finish_progressive() {
        for (int n = 0; n < n_comp; n++) {
            Component &c = comp[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    int16_t *d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int q = 0; q < 64; q++) d[q] = (int16_t)(d[q] * dequant[c.tq][q]);
                    idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, d);
                }
        }
    }
    int next_marker() {
        if (marker >= 0) { const int m = marker; marker = -1; return m; }
        int x = get8();
        while (x != 0xff) { /* junk between segments */
            if (p >= end) return -1;
            x = get8();
        }
        while (x == 0xff) x = get8();
        return x;
    }
    void decode_image() {
        if (get8() != 0xff || get8() != 0xd8) bad("not a JPEG file");
        int m = next_marker();
        while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
            if (m < 0 || m == 0xd9 || m == 0xda) bad("no frame header");
            if ((m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) bad("unsupported coding process (lossless, hierarchical or arithmetic)");
            segment(m);
            m = next_marker();
        }
        progressive = m == 0xc2;
        frame_header();
        m = next_marker();
        while (m != 0xd9) {
            if (m < 0) bad("truncated file");
            if (m == 0xda) {
                scan_header();
                scan_data();
                if (marker < 0) { /* zeros or junk after the scan: on to the next marker */
                    while (p < end) {
                        if (*p++ == 0xff && p < end && *p != 0 && *p != 0xff) { marker = *p++; break; }
                    }
                    if (marker < 0) bad("truncated file");
                }
            } else if (m == 0xdc) {
                if (get16() != 4 || get16() != H) bad("bad DNL");
            } else {
                segment(m);
            }
            m = next_marker();
        }
        if (progressive) finish_progressive();
    }

    /* ---- up-sampling and colour conversion ---- */
    static const uint8_t *row_1(uint8_t *, const uint8_t *near_, const uint8_t *, int, int) { return near_; }
    static const uint8_t *row_v2(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w, int) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
        return out;
    }
    static const uint8_t *row_h2(uint8_t *out, const uint8_t *in, const uint8_t *, int w, int) {
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            const int n = 3 * in[i] + 2;
            out[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    static const uint8_t *row_hv2(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w, int) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return out; }
        int t1 = 3 * near_[0] + far_[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near_[i] + far_[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    static const uint8_t *row_generic(uint8_t *out, const uint8_t *near_, const uint8_t *, int w, int hs) {
        for (int i = 0; i < w; i++)
            for (int j = 0; j < hs; j++) out[i * hs + j] = near_[i];
        return out;
    }
    void to_rgb(std::vector<uint8_t> &rgb) {
        typedef const uint8_t *(*RowFn)(uint8_t *, const uint8_t *, const uint8_t *, int, int);
        struct Res { RowFn fn; const uint8_t *line0, *line1; int hs, vs, w_lores, ystep, ypos; std::vector<uint8_t> buf; };
        Res res[3];
        for (int k = 0; k < n_comp; k++) {
            Res &r = res[k];
            r.hs = h_max / comp[k].h;
            r.vs = v_max / comp[k].v;
            r.ystep = r.vs >> 1;
            r.w_lores = (W + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].data.data();
            r.buf.assign((size_t)W + 8, 0);
            r.fn = (r.hs == 1 && r.vs == 1) ? row_1 : (r.hs == 1 && r.vs == 2) ? row_v2 : (r.hs == 2 && r.vs == 1) ? row_h2 : (r.hs == 2 && r.vs == 2) ? row_hv2 : row_generic;
            if (r.fn == row_generic) r.buf.assign((size_t)r.w_lores * r.hs + 8, 0);
        }
        const bool is_rgb = n_comp == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));
        rgb.assign((size_t)W * H * 3, 0);
        const uint8_t *row[3] = {nullptr, nullptr, nullptr};
        const auto fixed = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
        const int cr_r = fixed(1.40200f), cr_g = -fixed(0.71414f), cb_g = -fixed(0.34414f), cb_b = fixed(1.77200f);
        for (int j = 0; j < H; j++) {
            for (int k = 0; k < n_comp; k++) {
                Res &r = res[k];
                const bool y_bot = r.ystep >= (r.vs >> 1);
                row[k] = r.fn(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            uint8_t *out = rgb.data() + (size_t)W * 3 * j;
            if (n_comp == 1) {
                for (int i = 0; i < W; i++) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = row[0][i];
            } else if (is_rgb) {
                for (int i = 0; i < W; i++) { out[3 * i] = row[0][i]; out[3 * i + 1] = row[1][i]; out[3 * i + 2] = row[2][i]; }
            } else {
                for (int i = 0; i < W; i++) {
                    const int y_fixed = (row[0][i] << 20) + (1 << 19);
                    const int cb = row[1][i] - 128, cr = row[2][i] - 128;
                    int r = y_fixed + cr * cr_r;
                    int g = y_fixed + cr * cr_g + (int)((unsigned)(cb * cb_g) & 0xffff0000u);
                    int b = y_fixed + cb * cb_b;
                    r >>= 20;
                    g >>= 20;
                    b >>= 20;
                    out[3 * i] = clamp8(r);
                    out[3 * i + 1] = clamp8(g);
                    out[3 * i + 2] = clamp8(b);
                }
            }
        }
    }
};

} // namespace

bool looksLikeJPEG(const uint8_t *data, size_t n) { return n >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff; }

void decodeJPEG(const uint8_t *data, size_t n, std::vector<uint8_t> &rgb, int &W, int &H) {
    Decoder d(data, n);
    d.decode_image();
    d.to_rgb(rgb);
    W = d.W;
    H = d.H;
}

} // namespace rt
