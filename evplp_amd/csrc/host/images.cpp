#include "images.hpp"
#include "../../../include/evplp.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace evplp {

// floatimage.cpp:178-199: "PF\n<w> <h>\n-1\n", then the rows of the (top-down) image written
// last row first, RGB fp32 little-endian.
int save_pfm(const char *path, int32_t w, int32_t h, const float *rgb) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return EVPLP_ERR_IO;
    std::fprintf(f, "PF\n%d %d\n-1\n", w, h);
    for (int32_t i = 0; i < h; i++) {
        if (std::fwrite(rgb + (size_t)w * (h - i - 1) * 3, sizeof(float), (size_t)w * 3, f) != (size_t)w * 3) { std::fclose(f); return EVPLP_ERR_IO; }
    }
    std::fclose(f);
    return EVPLP_OK;
}

// floatimage.cpp:133-176 LoadPFM (colour, little-endian only)
int load_pfm(const char *path, int32_t *w, int32_t *h, float *rgb, size_t cap) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return EVPLP_ERR_IO;
    char magic[3] = { 0, 0, 0 }; int ww = 0, hh = 0; float scale = 0.f;
    if (std::fscanf(f, "%2s %d %d %f", magic, &ww, &hh, &scale) != 4 || std::strcmp(magic, "PF") != 0 || ww <= 0 || hh <= 0) { std::fclose(f); return EVPLP_ERR_PARSE; }
    std::fgetc(f);  // single whitespace after the scale
    *w = ww; *h = hh;
    if (!rgb) { std::fclose(f); return EVPLP_OK; }
    if (cap < (size_t)ww * hh * 3) { std::fclose(f); return EVPLP_ERR_INVALID; }
    for (int32_t i = 0; i < hh; i++) {
        if (std::fread(rgb + (size_t)ww * (hh - i - 1) * 3, sizeof(float), (size_t)ww * 3, f) != (size_t)ww * 3) { std::fclose(f); return EVPLP_ERR_IO; }
    }
    std::fclose(f);
    return EVPLP_OK;
}

namespace {
uint32_t crc_table[256]; bool crc_ready = false;
uint32_t crc32(const uint8_t *p, size_t n, uint32_t c = 0) {
    if (!crc_ready) { for (uint32_t i = 0; i < 256; i++) { uint32_t k = i; for (int j = 0; j < 8; j++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1; crc_table[i] = k; } crc_ready = true; }
    c = ~c;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}
void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    put32(out, crc32(td.data(), td.size()));
}
}

// floatimage.cpp:241-258: p = pow(c, 1/2.2) (fp32), min(p*255.99, 255.0) in double, truncated to a
// byte; 8-bit RGB PNG.  The container is written with stored (uncompressed) deflate blocks: the
// decoded pixels are identical to stb_image_write's output, the compressed bytes are not.
int save_png(const char *path, int32_t w, int32_t h, const float *rgb) {
    std::vector<uint8_t> raw; raw.reserve((size_t)h * (w * 3 + 1));
    for (int32_t y = 0; y < h; y++) {
        raw.push_back(0);  // filter: none
        for (int32_t x = 0; x < w * 3; x++) {
            float p = std::pow(rgb[(size_t)y * w * 3 + x], (float)(1 / 2.2));
            double q = (double)p * 255.99; if (q > 255.0) q = 255.0;
            p = (float)q;
            raw.push_back((uint8_t)(int)p);
        }
    }
    std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw.size(); i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    size_t pos = 0;
    do {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<uint8_t> ihdr; put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return EVPLP_ERR_IO;
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? EVPLP_OK : EVPLP_ERR_IO;
}

// Radiance RGBE (.hdr), FloatImage::SaveHDR (floatimage.cpp:223-239) -> RGBE_WriteHeader + RGBE_WritePixels_RLE
// (common/floatimage/rgbe.cpp:73-134, 231-336; Bruce Walter's public rgbe.c format): header
// "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y h +X w\n"; per scanline (widths 8..32767) the marker 2,2,w_hi,w_lo and
// the four channels (r, g, b, e) each run-length encoded with runs of >= 4 equal bytes; other widths are
// written flat.
namespace {
void float_to_rgbe(unsigned char out[4], float r, float g, float b) {
    float v = r; if (g > v) v = g; if (b > v) v = b;
    if (v < 1e-32) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    int e; v = (float)(std::frexp(v, &e) * 256.0 / v);
    out[0] = (unsigned char)(r * v); out[1] = (unsigned char)(g * v); out[2] = (unsigned char)(b * v); out[3] = (unsigned char)(e + 128);
}
void rle_channel(std::vector<uint8_t> &out, const unsigned char *data, int n) {
    const int kMinRun = 4;
    int cur = 0;
    while (cur < n) {
        int beg = cur, run = 0, old_run = 0;
        while (run < kMinRun && beg < n) {      // next run of at least 4 equal bytes
            beg += run; old_run = run; run = 1;
            while (beg + run < n && run < 127 && data[beg] == data[beg + run]) run++;
        }
        if (old_run > 1 && old_run == beg - cur) {   // the data before the long run is itself a short run
            out.push_back((uint8_t)(128 + old_run)); out.push_back(data[cur]);
            cur = beg;
        }
        while (cur < beg) {                          // literal bytes up to the run
            int cnt = std::min(beg - cur, 128);
            out.push_back((uint8_t)cnt); out.insert(out.end(), data + cur, data + cur + cnt);
            cur += cnt;
        }
        if (run >= kMinRun) { out.push_back((uint8_t)(128 + run)); out.push_back(data[beg]); cur += run; }
    }
}
}
int save_hdr(const char *path, int32_t w, int32_t h, const float *rgb) {
    std::vector<uint8_t> out;
    char hdr[128];
    int n = std::snprintf(hdr, sizeof hdr, "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    out.insert(out.end(), hdr, hdr + n);
    if (w < 8 || w > 0x7fff) {
        for (size_t i = 0; i < (size_t)w * h; i++) { unsigned char px[4]; float_to_rgbe(px, rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]); out.insert(out.end(), px, px + 4); }
    } else {
        std::vector<unsigned char> buf((size_t)4 * w);
        for (int32_t y = 0; y < h; y++) {
            out.push_back(2); out.push_back(2); out.push_back((uint8_t)(w >> 8)); out.push_back((uint8_t)(w & 0xff));
            for (int32_t x = 0; x < w; x++) {
                unsigned char px[4]; const float *p = rgb + 3 * ((size_t)y * w + x);
                float_to_rgbe(px, p[0], p[1], p[2]);
                buf[x] = px[0]; buf[x + w] = px[1]; buf[x + 2 * w] = px[2]; buf[x + 3 * w] = px[3];
            }
            for (int c = 0; c < 4; c++) rle_channel(out, buf.data() + (size_t)c * w, w);
        }
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) return EVPLP_ERR_IO;
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? EVPLP_OK : EVPLP_ERR_IO;
}

// FloatImage::LoadHDR (floatimage.cpp:201-221) = RGBE_ReadHeader + RGBE_ReadPixels_RLE (rgbe.cpp): Radiance picture,
// "-Y h +X w", flat or new-style run-length-encoded scanlines; rgb = mantissa * 2^(e - 136), (0,0,0) for e = 0.
int load_hdr(const char *path, int32_t *w, int32_t *h, float *rgb, size_t cap) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return EVPLP_ERR_IO;
    char line[256]; int ww = 0, hh = 0; bool have_size = false;
    if (!std::fgets(line, sizeof line, f) || std::strncmp(line, "#?", 2) != 0) { std::fclose(f); return EVPLP_ERR_PARSE; }
    while (std::fgets(line, sizeof line, f)) if (std::sscanf(line, "-Y %d +X %d", &hh, &ww) == 2) { have_size = true; break; }
    if (!have_size || ww <= 0 || hh <= 0) { std::fclose(f); return EVPLP_ERR_PARSE; }
    *w = ww; *h = hh;
    if (!rgb) { std::fclose(f); return EVPLP_OK; }
    if (cap < (size_t)ww * hh * 3) { std::fclose(f); return EVPLP_ERR_INVALID; }
    auto to_float = [](const unsigned char *px, float *out) {
        if (px[3]) { float s = std::ldexp(1.0f, (int)px[3] - (128 + 8)); out[0] = px[0] * s; out[1] = px[1] * s; out[2] = px[2] * s; }
        else out[0] = out[1] = out[2] = 0.0f;
    };
    std::vector<unsigned char> scan((size_t)4 * ww);
    for (int y = 0; y < hh; y++) {
        unsigned char px[4];
        if (std::fread(px, 1, 4, f) != 4) { std::fclose(f); return EVPLP_ERR_IO; }
        float *row = rgb + (size_t)y * ww * 3;
        if (ww < 8 || ww > 0x7fff || px[0] != 2 || px[1] != 2 || (px[2] & 0x80)) {
            // flat scanline (and, as in rgbe.cpp, the rest of the file is flat too)
            to_float(px, row);
            size_t rest = (size_t)ww * (hh - y) - 1;
            for (size_t i = 0; i < rest; i++) { if (std::fread(px, 1, 4, f) != 4) { std::fclose(f); return EVPLP_ERR_IO; } to_float(px, row + 3 * (i + 1)); }
            std::fclose(f);
            return EVPLP_OK;
        }
        if ((((int)px[2]) << 8 | px[3]) != ww) { std::fclose(f); return EVPLP_ERR_PARSE; }
        for (int c = 0; c < 4; c++) {
            unsigned char *dst = scan.data() + (size_t)c * ww; int x = 0;
            while (x < ww) {
                unsigned char b[2];
                if (std::fread(b, 1, 2, f) != 2) { std::fclose(f); return EVPLP_ERR_IO; }
                if (b[0] > 128) { int n = b[0] - 128; if (n == 0 || n > ww - x) { std::fclose(f); return EVPLP_ERR_PARSE; } while (n--) dst[x++] = b[1]; }
                else {
                    int n = b[0]; if (n == 0 || n > ww - x) { std::fclose(f); return EVPLP_ERR_PARSE; }
                    dst[x++] = b[1];
                    if (--n > 0) { if (std::fread(dst + x, 1, (size_t)n, f) != (size_t)n) { std::fclose(f); return EVPLP_ERR_IO; } x += n; }
                }
            }
        }
        for (int x = 0; x < ww; x++) { unsigned char q[4] = { scan[x], scan[x + ww], scan[x + 2 * ww], scan[x + 3 * ww] }; to_float(q, row + 3 * x); }
    }
    std::fclose(f);
    return EVPLP_OK;
}

// floatimage.cpp:260-273 Save: dispatch on the extension
int save_image(const char *path, int32_t w, int32_t h, const float *rgb) {
    std::string p(path);
    size_t i = p.find_last_of('.');
    if (i == std::string::npos || i + 1 >= p.size()) return EVPLP_ERR_INVALID;
    std::string ext = p.substr(i + 1);
    if (ext == "pfm") return save_pfm(path, w, h, rgb);
    if (ext == "hdr") return save_hdr(path, w, h, rgb);
    if (ext == "png") return save_png(path, w, h, rgb);
    return EVPLP_ERR_INVALID;  // "unsupported file format"
}

} // namespace evplp
