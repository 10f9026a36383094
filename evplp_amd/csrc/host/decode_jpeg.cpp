// JPEG (ITU-T T.81) decoder -- baseline, extended-sequential (Huffman, 8 bit) and progressive -- whose
// reconstruction arithmetic is that of stb_image v2.16, the decoder the reference links
// (reflectcuts/stb/stb_image.h, used at rt/rtcommon.h:144), so that decoded texels are identical:
//   * de-quantised coefficients are held in 16 bits;
//   * inverse DCT: the 12-bit fixed-point "islow" factorisation, column pass rounded to 2 fractional bits
//     (+512 >> 10), row pass +65536 + (128 << 17) >> 17, clamped to [0, 255];
//   * chroma upsampling: 3:1 triangle filters ((3 near + far + 2) >> 2; 2x2: (3 a + b + 8) >> 4 on the
//     vertically filtered sums), pixel replication for other ratios;
//   * YCbCr -> RGB in 20-bit fixed point with 12-bit coefficients, the Cb term of G masked to 16 bits;
//   * three components are RGB (no conversion) when their ids are 'R','G','B', or when an Adobe APP14
//     marker says transform 0 and there is no JFIF marker.
// CMYK / YCCK (4 components), arithmetic coding and 12-bit precision are rejected.
#include "decoders.hpp"

#include <cstring>
#include <stdexcept>

namespace evplp {
namespace {

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct HuffTable {
    // canonical code: for each length, the first code value and the index of its first symbol
    int mincode[17], maxcode[18], valptr[17];
    uint8_t values[256];
    bool defined = false;
    void build(const uint8_t counts[16], const uint8_t *vals, int nvals) {
        std::memcpy(values, vals, (size_t)nvals);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) throw std::runtime_error("jpeg: bad Huffman code lengths");
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0;         // size in samples
    int w2 = 0, h2 = 0;       // size padded to whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;   // progressive: raw coefficients of every block, [block][64] in natural order
    int coeff_w = 0;
};

struct Decoder {
    const uint8_t *p, *end, *base;
    int W = 0, H = 0, ncomp = 0; bool progressive = false;
    int h_max = 1, v_max = 1, mcu_w = 8, mcu_h = 8, mcus_x = 0, mcus_y = 0;
    Component comp[4];
    uint16_t dequant[4][64]; bool dq_defined[4] = { false, false, false, false };
    HuffTable hdc[4], hac[4];
    int restart_interval = 0;
    bool jfif = false; int adobe_transform = -1; int rgb_ids = 0;
    // entropy-coded segment reader
    uint32_t bitbuf = 0; int bitcnt = 0; int marker = -1; bool nomore = false;
    int eob_run = 0, todo = 0;
    int scan_n = 0, order[4] = { 0, 0, 0, 0 }, ss = 0, se = 63, ah = 0, al = 0;

    Decoder(const uint8_t *d, size_t n) : p(d), end(d + n), base(d) {}

    int byte() { return p < end ? *p++ : 0; }
    int be16() { int a = byte(); return (a << 8) | byte(); }

    // ---- bit reader: MSB first, FF 00 -> FF, any other FF xx is a marker and ends the segment (zeros follow)
    void fill() {
        while (bitcnt <= 24) {
            int b = nomore ? 0 : byte();
            if (b == 0xff) {
                int c = byte();
                while (c == 0xff) c = byte();
                if (c != 0) { marker = c; nomore = true; b = 0; }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) fill();
        int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n; bitcnt -= n;
        return v;
    }
    int getbit() { return getbits(1); }
    int decode_symbol(const HuffTable &t) {
        if (!t.defined) throw std::runtime_error("jpeg: scan uses an undefined Huffman table");
        if (bitcnt < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (int)(bitbuf >> (32 - len));
            if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) {
                bitbuf <<= len; bitcnt -= len;
                return t.values[t.valptr[len] + code - t.mincode[len]];
            }
        }
        throw std::runtime_error("jpeg: bad Huffman code");
    }
    // (two's-complement wrap-around, spelled out: corrupted files reach these with values the signed operators leave undefined)
    static int wrap_add(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
    static int wrap_mul(int a, int b) { return (int)((uint32_t)a * (uint32_t)b); }
    static int wrap_shl(int a, int n) { return (int)((uint32_t)a << n); }
    // T.81 F.2.2.1 EXTEND
    int receive_extend(int s) {
        if (s == 0) return 0;
        int v = getbits(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    void reset_entropy() {
        bitbuf = 0; bitcnt = 0; nomore = false; marker = -1;
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
        eob_run = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
    }

    // ---- block decoders
    void block_baseline(int16_t *d, Component &c) {
        std::memset(d, 0, 64 * sizeof(int16_t));
        const uint16_t *q = dequant[c.tq];
        int t = decode_symbol(hdc[c.td]);
        if (t > 15) throw std::runtime_error("jpeg: bad DC category");
        int diff = receive_extend(t);
        c.dc_pred = wrap_add(c.dc_pred, diff);
        d[0] = (int16_t)wrap_mul(c.dc_pred, q[0]);
        const HuffTable &ac = hac[c.ta];
        int k = 1;
        do {
            int rs = decode_symbol(ac), s = rs & 15, r = rs >> 4;
            if (s == 0) { if (rs != 0xf0) break; k += 16; }
            else {
                k += r;
                if (k > 63) throw std::runtime_error("jpeg: coefficient index out of range");
                int z = kZigzag[k++];
                d[z] = (int16_t)wrap_mul(receive_extend(s), q[z]);
            }
        } while (k < 64);
    }
    void block_prog_dc(int16_t *d, Component &c) {
        if (se != 0) throw std::runtime_error("jpeg: DC scan with AC coefficients");
        if (ah == 0) {
            std::memset(d, 0, 64 * sizeof(int16_t));
            int t = decode_symbol(hdc[c.td]);
            if (t > 15) throw std::runtime_error("jpeg: bad DC category");
            c.dc_pred = wrap_add(c.dc_pred, receive_extend(t));
            d[0] = (int16_t)wrap_shl(c.dc_pred, al);
        } else if (getbit()) d[0] = (int16_t)(d[0] + (int16_t)(1 << al));
    }
    void block_prog_ac(int16_t *d, Component &c) {
        if (ss == 0) throw std::runtime_error("jpeg: AC scan starting at the DC coefficient");
        const HuffTable &ac = hac[c.ta];
        if (ah == 0) {                                       // first pass over this band (T.81 G.1.2.2)
            if (eob_run) { --eob_run; return; }
            int k = ss;
            do {
                int rs = decode_symbol(ac), s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) { eob_run = (1 << r); if (r) eob_run += getbits(r); --eob_run; break; }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) throw std::runtime_error("jpeg: coefficient index out of range");
                    d[kZigzag[k++]] = (int16_t)wrap_shl(receive_extend(s), al);
                }
            } while (k <= se);
            return;
        }
        // refinement of this band (T.81 G.1.2.3)
        const int16_t bit = (int16_t)(1 << al);
        auto refine = [&](int16_t &c0) {
            if (getbit() && (c0 & bit) == 0) c0 = (int16_t)(c0 > 0 ? c0 + bit : c0 - bit);
        };
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; k++) { int16_t &c0 = d[kZigzag[k]]; if (c0 != 0) refine(c0); }
            return;
        }
        int k = ss;
        do {
            int rs = decode_symbol(ac), s = rs & 15, r = rs >> 4;
            int newval = 0;
            if (s == 0) {
                if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += getbits(r); r = 64; }   // rest of the band: refinements only
            } else {
                if (s != 1) throw std::runtime_error("jpeg: bad refinement code");
                newval = getbit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t &c0 = d[kZigzag[k++]];
                if (c0 != 0) refine(c0);
                else { if (r == 0) { c0 = (int16_t)newval; break; } --r; }
            }
        } while (k <= se);
    }

    // ---- inverse DCT into an 8x8 block of a plane
    // (the arithmetic wraps like the two's-complement machine code of the reference's decoder does: a corrupted file can drive the
    // 32-bit intermediates over the top, which as signed C++ arithmetic would be undefined behaviour -- found by tools/host_fuzz)
    struct wi {
        uint32_t u;
        wi() : u(0) {}
        wi(int x) : u((uint32_t)x) {}
        operator int() const { return (int)u; }
        friend wi operator+(wi a, wi b) { wi r; r.u = a.u + b.u; return r; }
        friend wi operator-(wi a, wi b) { wi r; r.u = a.u - b.u; return r; }
        friend wi operator*(wi a, wi b) { wi r; r.u = a.u * b.u; return r; }
        wi &operator+=(wi b) { u += b.u; return *this; }
        wi &operator*=(wi b) { u *= b.u; return *this; }
    };
    static constexpr int fx(float x) { return (int)((double)x * 4096 + 0.5); }
    struct Odd { wi t0, t1, t2, t3; };
    static inline void idct_1d(wi s0, wi s1, wi s2, wi s3, wi s4, wi s5, wi s6, wi s7, wi &x0, wi &x1, wi &x2, wi &x3, Odd &o) {
        wi p2 = s2, p3 = s6;
        wi p1 = (p2 + p3) * wi(fx(0.5411961f));
        wi t2 = p1 + p3 * wi(fx(-1.847759065f));
        wi t3 = p1 + p2 * wi(fx(0.765366865f));
        wi t0 = (s0 + s4) * wi(4096), t1 = (s0 - s4) * wi(4096);
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2; wi p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
        wi p5 = (p3 + p4) * wi(fx(1.175875602f));
        t0 *= wi(fx(0.298631336f)); t1 *= wi(fx(2.053119869f)); t2 *= wi(fx(3.072711026f)); t3 *= wi(fx(1.501321110f));
        p1 = p5 + p1 * wi(fx(-0.899976223f)); p2 = p5 + p2 * wi(fx(-2.562915447f));
        p3 *= wi(fx(-1.961570560f)); p4 *= wi(fx(-0.390180644f));
        o.t3 = t3 + p1 + p4; o.t2 = t2 + p2 + p3; o.t1 = t1 + p2 + p4; o.t0 = t0 + p1 + p3;
    }
    static inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }
    static void idct(uint8_t *out, int stride, const int16_t *d) {
        int v[64];
        for (int i = 0; i < 8; i++) {
            wi x0, x1, x2, x3; Odd o;
            idct_1d(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i], x0, x1, x2, x3, o);
            x0 += wi(512); x1 += wi(512); x2 += wi(512); x3 += wi(512);
            v[i] = (int)(x0 + o.t3) >> 10; v[56 + i] = (int)(x0 - o.t3) >> 10;
            v[8 + i] = (int)(x1 + o.t2) >> 10; v[48 + i] = (int)(x1 - o.t2) >> 10;
            v[16 + i] = (int)(x2 + o.t1) >> 10; v[40 + i] = (int)(x2 - o.t1) >> 10;
            v[24 + i] = (int)(x3 + o.t0) >> 10; v[32 + i] = (int)(x3 - o.t0) >> 10;
        }
        for (int i = 0; i < 8; i++) {
            const int *r = v + 8 * i; uint8_t *o8 = out + (size_t)i * stride;
            wi x0, x1, x2, x3; Odd o;
            idct_1d(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], x0, x1, x2, x3, o);
            const wi bias(65536 + (128 << 17));
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            o8[0] = clamp8((int)(x0 + o.t3) >> 17); o8[7] = clamp8((int)(x0 - o.t3) >> 17);
            o8[1] = clamp8((int)(x1 + o.t2) >> 17); o8[6] = clamp8((int)(x1 - o.t2) >> 17);
            o8[2] = clamp8((int)(x2 + o.t1) >> 17); o8[5] = clamp8((int)(x2 - o.t1) >> 17);
            o8[3] = clamp8((int)(x3 + o.t0) >> 17); o8[4] = clamp8((int)(x3 - o.t0) >> 17);
        }
    }

    // ---- restart handling: true = keep decoding, false = the scan ends here
    bool mcu_done() {
        if (--todo > 0) return true;
        if (bitcnt < 24) fill();
        if (marker < 0xd0 || marker > 0xd7) return false;
        reset_entropy();
        return true;
    }

    void decode_scan() {
        reset_entropy();
        int16_t blk[64];
        if (scan_n == 1) {               // non-interleaved: the component's own block grid
            Component &c = comp[order[0]];
            int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++) for (int i = 0; i < bw; i++) {
                if (!progressive) { block_baseline(blk, c); idct(&c.plane[(size_t)c.w2 * j * 8 + i * 8], c.w2, blk); }
                else {
                    int16_t *d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
                    if (ss == 0) block_prog_dc(d, c); else block_prog_ac(d, c);
                }
                if (!mcu_done()) return;
            }
            return;
        }
        for (int j = 0; j < mcus_y; j++) for (int i = 0; i < mcus_x; i++) {
            for (int k = 0; k < scan_n; k++) {
                Component &c = comp[order[k]];
                for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) {
                    int bx = i * c.h + x, by = j * c.v + y;
                    if (!progressive) { block_baseline(blk, c); idct(&c.plane[(size_t)c.w2 * by * 8 + bx * 8], c.w2, blk); }
                    else block_prog_dc(&c.coeff[64 * ((size_t)bx + (size_t)by * c.coeff_w)], c);
                }
            }
            if (!mcu_done()) return;
        }
    }

    void finish_progressive() {
        for (int n = 0; n < ncomp; n++) {
            Component &c = comp[n];
            int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            const uint16_t *q = dequant[c.tq];
            for (int j = 0; j < bh; j++) for (int i = 0; i < bw; i++) {
                int16_t *d = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
                for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * q[k]);
                idct(&c.plane[(size_t)c.w2 * j * 8 + i * 8], c.w2, d);
            }
        }
    }

    // ---- marker segments
    void read_dqt(int len) {
        len -= 2;
        while (len > 0) {
            int q = byte(), prec = q >> 4, t = q & 15;
            if (prec > 1 || t > 3) throw std::runtime_error("jpeg: bad quantisation table");
            for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = (uint16_t)(prec ? be16() : byte());
            dq_defined[t] = true;
            len -= prec ? 129 : 65;
        }
        if (len != 0) throw std::runtime_error("jpeg: bad DQT length");
    }
    void read_dht(int len) {
        len -= 2;
        while (len > 0) {
            int q = byte(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad Huffman table header");
            uint8_t counts[16], vals[256]; int n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)byte(); n += counts[i]; }
            if (n > 256) throw std::runtime_error("jpeg: too many Huffman symbols");
            for (int i = 0; i < n; i++) vals[i] = (uint8_t)byte();
            (tc == 0 ? hdc[th] : hac[th]).build(counts, vals, n);
            len -= 17 + n;
        }
        if (len != 0) throw std::runtime_error("jpeg: bad DHT length");
    }
    void read_sof(int len, bool prog) {
        progressive = prog;
        if (len < 11) throw std::runtime_error("jpeg: bad SOF length");
        if (byte() != 8) throw std::runtime_error("jpeg: only 8-bit precision is supported");
        H = be16(); W = be16(); ncomp = byte();
        if (H == 0 || W == 0) throw std::runtime_error("jpeg: zero image size");
        if (ncomp != 1 && ncomp != 3) throw std::runtime_error("jpeg: only grey and three-component images are supported");
        if (len != 8 + 3 * ncomp) throw std::runtime_error("jpeg: bad SOF length");
        rgb_ids = 0;
        static const char rgb[3] = { 'R', 'G', 'B' };
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.id = byte(); if (ncomp == 3 && c.id == rgb[i]) rgb_ids++;
            int q = byte(); c.h = q >> 4; c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) throw std::runtime_error("jpeg: bad sampling factors");
            c.tq = byte(); if (c.tq > 3) throw std::runtime_error("jpeg: bad quantisation table id");
        }
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; i++) { if (comp[i].h > h_max) h_max = comp[i].h; if (comp[i].v > v_max) v_max = comp[i].v; }
        // Two refusals the reference's stb_image v2.16 does not make (found by tools/host_fuzz; later stb versions make the first too):
        // sampling factors that do not divide the largest one -- its resamplers (and these) then read a component's rows as if they were
        // as wide as the image, past the end of the plane; and a frame that asks for more memory than any data behind so small a file
        // could fill (an all-zero block costs two bits; 64 samples per block, three components).
        for (int i = 0; i < ncomp; i++) if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) throw std::runtime_error("jpeg: sampling factors that are not integer ratios");
        if ((uint64_t)W * (uint64_t)H > (1ull << 28) || (uint64_t)W * (uint64_t)H > 4096ull * (uint64_t)(end - base) + (1ull << 20))
            throw std::runtime_error("jpeg: frame larger than its data can fill");
        mcu_w = 8 * h_max; mcu_h = 8 * v_max;
        mcus_x = (W + mcu_w - 1) / mcu_w; mcus_y = (H + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.x = (W * c.h + h_max - 1) / h_max; c.y = (H * c.v + v_max - 1) / v_max;
            c.w2 = mcus_x * c.h * 8; c.h2 = mcus_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) { c.coeff_w = c.w2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
        }
    }
    void read_sos(int len) {
        scan_n = byte();
        if (scan_n < 1 || scan_n > ncomp || len != 6 + 2 * scan_n) throw std::runtime_error("jpeg: bad scan header");
        for (int i = 0; i < scan_n; i++) {
            int id = byte(), q = byte(), which = -1;
            for (int k = 0; k < ncomp; k++) if (comp[k].id == id) which = k;
            if (which < 0) throw std::runtime_error("jpeg: scan names an unknown component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) throw std::runtime_error("jpeg: bad Huffman table id");
            order[i] = which;
        }
        ss = byte(); se = byte(); int a = byte(); ah = a >> 4; al = a & 15;
        if (progressive) { if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) throw std::runtime_error("jpeg: bad progressive scan parameters"); }
        else { if (ss != 0 || ah != 0 || al != 0) throw std::runtime_error("jpeg: bad sequential scan parameters"); se = 63; }
        for (int i = 0; i < scan_n; i++) if (!dq_defined[comp[order[i]].tq]) throw std::runtime_error("jpeg: scan uses an undefined quantisation table");
    }

    int next_marker() {
        if (marker >= 0) { int m = marker; marker = -1; return m; }
        // tolerate filler: look for FF followed by a marker code
        while (p < end) {
            int b = byte();
            if (b != 0xff) continue;
            int c = byte();
            while (c == 0xff) c = byte();
            if (c != 0) return c;
        }
        return -1;
    }

    void decode() {
        if (byte() != 0xff || byte() != 0xd8) throw std::runtime_error("jpeg: no SOI");
        bool have_frame = false;
        for (;;) {
            int m = next_marker();
            if (m < 0) throw std::runtime_error("jpeg: no EOI");
            if (m == 0xd9) break;
            if (m >= 0xd0 && m <= 0xd7) continue;      // stray RSTn
            if (m == 0x01) continue;
            int len = be16();
            if (len < 2 || (size_t)(len - 2) > (size_t)(end - p)) throw std::runtime_error("jpeg: bad segment length");
            const uint8_t *seg_end = p + len - 2;
            switch (m) {
                case 0xc0: case 0xc1: case 0xc2:
                    if (have_frame) throw std::runtime_error("jpeg: more than one frame");
                    read_sof(len, m == 0xc2); have_frame = true; break;
                case 0xc4: read_dht(len); break;
                case 0xdb: read_dqt(len); break;
                case 0xdd: if (len != 4) throw std::runtime_error("jpeg: bad DRI length"); restart_interval = be16(); break;
                case 0xe0: if (len >= 7 && std::memcmp(p, "JFIF\0", 5) == 0) jfif = true; break;
                case 0xee: if (len >= 14 && std::memcmp(p, "Adobe\0", 6) == 0) adobe_transform = p[11]; break;
                case 0xda:
                    if (!have_frame) throw std::runtime_error("jpeg: scan before the frame header");
                    read_sos(len);
                    decode_scan();
                    seg_end = nullptr;
                    break;
                default:
                    if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) throw std::runtime_error("jpeg: unsupported coding process (lossless / hierarchical / arithmetic)");
                    break;      // APPn, COM and unknown segments are skipped
            }
            if (seg_end) p = seg_end;
        }
        if (!have_frame) throw std::runtime_error("jpeg: no frame");
        if (progressive) finish_progressive();
    }
};

inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }

} // namespace

bool is_jpeg(const uint8_t *d, size_t n) { return n >= 3 && d[0] == 0xff && d[1] == 0xd8 && d[2] == 0xff; }

DecodedImage decode_jpeg(const uint8_t *data, size_t size) {
    Decoder z(data, size);
    z.decode();
    const int W = z.W, H = z.H;
    DecodedImage img; img.w = W; img.h = H; img.channels = z.ncomp >= 3 ? 3 : 1;
    img.rgb.assign((size_t)W * H * 3, 0);
    const bool is_rgb = z.ncomp == 3 && (z.rgb_ids == 3 || (z.adobe_transform == 0 && !z.jfif));
    std::vector<uint8_t> line[3];
    for (int k = 0; k < z.ncomp; k++) line[k].assign((size_t)W + 8, 0);
    const int fr = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, fg_cr = ((int)(0.71414f * 4096.0f + 0.5f)) << 8,
              fg_cb = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, fb = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
    for (int j = 0; j < H; j++) {
        const uint8_t *row[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < z.ncomp; k++) {
            const Component &c = z.comp[k];
            const int hs = z.h_max / c.h, vs = z.v_max / c.v;
            const int wl = (W + hs - 1) / hs;
            int near_r = j / vs, far_r = near_r;
            if (vs == 2) far_r = (j & 1) ? near_r + 1 : near_r - 1;
            if (far_r < 0) far_r = 0;
            if (far_r > c.y - 1) far_r = c.y - 1;
            if (near_r > c.y - 1) near_r = c.y - 1;
            const uint8_t *in_near = &c.plane[(size_t)near_r * c.w2], *in_far = &c.plane[(size_t)far_r * c.w2];
            uint8_t *out = line[k].data();
            if (hs == 1 && vs == 1) { row[k] = in_near; continue; }
            if (hs == 1 && vs == 2) { for (int i = 0; i < wl; i++) out[i] = div4(3 * in_near[i] + in_far[i] + 2); }
            else if (hs == 2 && vs == 1) {
                if (wl == 1) out[0] = out[1] = in_near[0];
                else {
                    out[0] = in_near[0]; out[1] = div4(in_near[0] * 3 + in_near[1] + 2);
                    int i;
                    for (i = 1; i < wl - 1; i++) { int n = 3 * in_near[i] + 2; out[2 * i] = div4(n + in_near[i - 1]); out[2 * i + 1] = div4(n + in_near[i + 1]); }
                    out[2 * i] = div4(in_near[wl - 2] * 3 + in_near[wl - 1] + 2); out[2 * i + 1] = in_near[wl - 1];
                }
            } else if (hs == 2 && vs == 2) {
                if (wl == 1) out[0] = out[1] = div4(3 * in_near[0] + in_far[0] + 2);
                else {
                    int t1 = 3 * in_near[0] + in_far[0];
                    out[0] = div4(t1 + 2);
                    for (int i = 1; i < wl; i++) { int t0 = t1; t1 = 3 * in_near[i] + in_far[i]; out[2 * i - 1] = div16(3 * t0 + t1 + 8); out[2 * i] = div16(3 * t1 + t0 + 8); }
                    out[2 * wl - 1] = div4(t1 + 2);
                }
            } else { for (int i = 0; i < W; i++) out[i] = in_near[i / hs]; }
            row[k] = out;
        }
        uint8_t *o = &img.rgb[(size_t)j * W * 3];
        if (z.ncomp == 1) { for (int i = 0; i < W; i++, o += 3) o[0] = o[1] = o[2] = row[0][i]; }
        else if (is_rgb) { for (int i = 0; i < W; i++, o += 3) { o[0] = row[0][i]; o[1] = row[1][i]; o[2] = row[2][i]; } }
        else for (int i = 0; i < W; i++, o += 3) {
            int y_fixed = (row[0][i] << 20) + (1 << 19);
            int cr = row[2][i] - 128, cb = row[1][i] - 128;
            int r = y_fixed + cr * fr;
            int g = y_fixed + (cr * -fg_cr) + (int)((uint32_t)(cb * -fg_cb) & 0xffff0000u);
            int b = y_fixed + cb * fb;
            r >>= 20; g >>= 20; b >>= 20;
            o[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); o[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g); o[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
        }
    }
    return img;
}

DecodedImage decode_image_file(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> bytes;
    uint8_t buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    try {
        if (is_jpeg(bytes.data(), bytes.size())) return decode_jpeg(bytes.data(), bytes.size());
        if (is_png(bytes.data(), bytes.size())) return decode_png(bytes.data(), bytes.size());
    } catch (const std::exception &e) { throw std::runtime_error(path + ": " + e.what()); }
    throw std::runtime_error(path + ": not a JPEG or PNG file");
}

} // namespace evplp
