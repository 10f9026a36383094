// PNG (ISO/IEC 15948) decoder producing what stbi_load(path, .., 3) of stb_image v2.16 returns:
// 8-bit RGB; 16-bit samples keep their high byte; 1/2/4-bit grey is scaled by 255/(2^depth - 1);
// palette entries expand to RGB; alpha (colour type 4/6, tRNS) is dropped, not multiplied.
// `channels` is the component count stb reports for the file: colour-type components (3 for a palette), plus
// one when a tRNS chunk supplies transparency.
#include "decoders.hpp"

#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace evplp {
namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// Reverses the per-scanline filters of one (sub-)image in place.  raw: h rows of (1 + stride) bytes.
void unfilter(uint8_t *raw, size_t stride, int h, int bpp /* bytes per complete pixel, >= 1 */) {
    const uint8_t *prev = nullptr;
    for (int y = 0; y < h; y++) {
        uint8_t *row = raw + (size_t)y * (stride + 1);
        int f = row[0]; uint8_t *cur = row + 1;
        if (f > 4) throw std::runtime_error("png: invalid filter type");
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
            int b = prev ? prev[i] : 0;
            int c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
            int pred = f == 0 ? 0 : f == 1 ? a : f == 2 ? b : f == 3 ? ((a + b) >> 1) : paeth(a, b, c);
            cur[i] = (uint8_t)(cur[i] + pred);
        }
        prev = cur;
    }
}

} // namespace

bool is_png(const uint8_t *d, size_t n) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    return n >= 8 && std::memcmp(d, sig, 8) == 0;
}

DecodedImage decode_png(const uint8_t *d, size_t n) {
    if (!is_png(d, n)) throw std::runtime_error("png: bad signature");
    size_t pos = 8;
    uint32_t W = 0, H = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat; uint8_t palette[256 * 3] = { 0 }; int pal_len = 0; bool has_trns = false, seen_ihdr = false, done = false;
    while (!done) {
        if (pos + 8 > n) throw std::runtime_error("png: truncated chunk header");
        uint32_t len = be32(d + pos); const uint8_t *type = d + pos + 4; const uint8_t *body = d + pos + 8;
        if ((size_t)len > n - pos - 8 || n - pos - 8 - len < 4) throw std::runtime_error("png: truncated chunk");
        auto is = [&](const char *t) { return std::memcmp(type, t, 4) == 0; };
        if (is("IHDR")) {
            if (len != 13 || seen_ihdr) throw std::runtime_error("png: bad IHDR");
            seen_ihdr = true;
            W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9];
            if (body[10] != 0 || body[11] != 0) throw std::runtime_error("png: unknown compression / filter method");
            interlace = body[12];
            if (interlace > 1) throw std::runtime_error("png: unknown interlace method");
            if (W == 0 || H == 0 || W > (1u << 24) || H > (1u << 24)) throw std::runtime_error("png: bad dimensions");
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) throw std::runtime_error("png: bad bit depth");
            if (ctype != 0 && ctype != 2 && ctype != 3 && ctype != 4 && ctype != 6) throw std::runtime_error("png: bad colour type");
            if (ctype == 3 && depth == 16) throw std::runtime_error("png: 16-bit palette");
            if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) throw std::runtime_error("png: bit depth not allowed for the colour type");
        } else if (!seen_ihdr) throw std::runtime_error("png: first chunk is not IHDR");
        else if (is("PLTE")) {
            if (len > 768 || len % 3) throw std::runtime_error("png: bad PLTE");
            pal_len = (int)len / 3; std::memcpy(palette, body, len);
        } else if (is("tRNS")) has_trns = true;
        else if (is("IDAT")) idat.insert(idat.end(), body, body + len);
        else if (is("IEND")) done = true;
        else if (!(type[0] & 32)) throw std::runtime_error("png: unknown critical chunk");
        pos += 12 + (size_t)len;
    }
    if (idat.empty()) throw std::runtime_error("png: no image data");
    if (ctype == 3 && pal_len == 0) throw std::runtime_error("png: palette image without PLTE");
    const int comps = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    const int bits_pp = comps * depth;
    const int bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    auto stride_of = [&](uint32_t w) { return ((size_t)w * bits_pp + 7) / 8; };

    size_t expect = 0;
    static const int xs[7] = { 0, 4, 0, 2, 0, 1, 0 }, ys[7] = { 0, 0, 4, 0, 2, 0, 1 }, dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
    if (!interlace) expect = (stride_of(W) + 1) * H;
    else for (int p = 0; p < 7; p++) {
        uint32_t pw = (W - xs[p] + dx[p] - 1) / dx[p], phh = (H - ys[p] + dy[p] - 1) / dy[p];
        if (pw && phh) expect += (stride_of(pw) + 1) * phh;
    }
    std::vector<uint8_t> raw = zlib_inflate(idat.data(), idat.size(), expect);
    if (raw.size() < expect) throw std::runtime_error("png: not enough pixel data");

    DecodedImage img; img.w = (int)W; img.h = (int)H;
    img.channels = ctype == 3 ? (has_trns ? 4 : 3) : comps + (has_trns ? 1 : 0);
    img.rgb.assign((size_t)W * H * 3, 0);
    // sample k of pixel x in an unfiltered row, reduced to 8 bits
    auto sample = [&](const uint8_t *row, uint32_t x, int k) -> int {
        if (depth == 8) return row[(size_t)x * comps + k];
        if (depth == 16) return row[((size_t)x * comps + k) * 2];                       // high byte
        size_t bit = (size_t)x * depth;                                                // comps == 1 below 8 bits
        int v = (row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1);
        if (ctype == 3) return v;
        return v * (depth == 1 ? 0xff : depth == 2 ? 0x55 : 0x11);
    };
    auto put = [&](const uint8_t *row, uint32_t sx, uint32_t ox, uint32_t oy) {
        uint8_t *o = &img.rgb[((size_t)oy * W + ox) * 3];
        if (ctype == 3) {
            int idx = sample(row, sx, 0);
            if (idx >= pal_len) { o[0] = o[1] = o[2] = 0; if (idx >= 256) throw std::runtime_error("png: palette index out of range"); }
            else { o[0] = palette[idx * 3]; o[1] = palette[idx * 3 + 1]; o[2] = palette[idx * 3 + 2]; }
        } else if (comps <= 2) { o[0] = o[1] = o[2] = (uint8_t)sample(row, sx, 0); }
        else { o[0] = (uint8_t)sample(row, sx, 0); o[1] = (uint8_t)sample(row, sx, 1); o[2] = (uint8_t)sample(row, sx, 2); }
    };
    if (!interlace) {
        size_t stride = stride_of(W);
        unfilter(raw.data(), stride, (int)H, bpp);
        for (uint32_t y = 0; y < H; y++) { const uint8_t *row = raw.data() + (size_t)y * (stride + 1) + 1; for (uint32_t x = 0; x < W; x++) put(row, x, x, y); }
    } else {
        size_t off = 0;
        for (int p = 0; p < 7; p++) {
            uint32_t pw = (W - xs[p] + dx[p] - 1) / dx[p], phh = (H - ys[p] + dy[p] - 1) / dy[p];
            if (!pw || !phh) continue;
            size_t stride = stride_of(pw);
            unfilter(raw.data() + off, stride, (int)phh, bpp);
            for (uint32_t y = 0; y < phh; y++) {
                const uint8_t *row = raw.data() + off + (size_t)y * (stride + 1) + 1;
                for (uint32_t x = 0; x < pw; x++) put(row, x, xs[p] + x * dx[p], ys[p] + y * dy[p]);
            }
            off += (stride + 1) * phh;
        }
    }
    return img;
}

} // namespace evplp
