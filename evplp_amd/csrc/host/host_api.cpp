// Host-side pieces of the reference interface that need no GPU: progressive schedule, image
// output surface, error metrics.
#include "../evplp_types.h"
#include "images.hpp"
#include "decoders.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// rt/rtcomphoton/rtcomphoton.h:1033-1063 (runs after numIterations++; Float = float, reflectcuts.h:56)
extern "C" void evplp_progressive_step(int32_t n, float alpha, float clamp_start, uint32_t n_vpl, uint32_t n_light,
                                       float *radius, float *clamp, float *pdf_mc, int32_t force_vsl,
                                       float *vsl_radius, float *vsl_inv_pi_r2) {
    const float inv_pi = 0.318309886183790671537767526745028724068919291480912897495f;
    float ratio = ((float)n + alpha) / (float)(n + 1);                                   // :1037
    *radius *= std::sqrt(ratio);                                                         // :1038
    *clamp = clamp_start * std::pow((float)n, alpha);                                    // :1039
    *pdf_mc = (float)n_vpl / (float)n_light * inv_pi / (*radius * *radius);              // :1040
    if (force_vsl && vsl_radius && vsl_inv_pi_r2) {
        *vsl_radius *= std::sqrt(ratio);                                                 // :1049
        if (*vsl_radius <= 0.008f) *vsl_radius = std::max(*vsl_radius, 0.008f);          // :1050-1054
        *vsl_inv_pi_r2 = inv_pi / (*vsl_radius * *vsl_radius);                           // :1056
    }
}

extern "C" int evplp_save_image(const char *path, int32_t w, int32_t h, const float *rgb) {
    if (!path || !rgb || w <= 0 || h <= 0) return EVPLP_ERR_INVALID;
    return evplp::save_image(path, w, h, rgb);
}
extern "C" int evplp_decode_image(const char *path, int32_t *w, int32_t *h, int32_t *channels, uint8_t *rgb, size_t cap) {
    if (!path || !w || !h) return EVPLP_ERR_INVALID;
    try {
        evplp::DecodedImage img = evplp::decode_image_file(path);
        *w = img.w; *h = img.h; if (channels) *channels = img.channels;
        if (rgb) {
            if (cap < img.rgb.size()) return EVPLP_ERR_INVALID;
            std::memcpy(rgb, img.rgb.data(), img.rgb.size());
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "evplp_decode_image: %s\n", e.what());
        return std::strstr(e.what(), "cannot open") ? EVPLP_ERR_IO : EVPLP_ERR_PARSE;
    }
    return EVPLP_OK;
}
extern "C" int evplp_load_image(const char *path, int32_t *w, int32_t *h, float *rgb, size_t cap) {
    if (!path || !w || !h) return EVPLP_ERR_INVALID;
    std::string p(path);
    size_t i = p.find_last_of('.');
    std::string ext = i == std::string::npos ? "" : p.substr(i + 1);
    for (char &c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == "hdr") return evplp::load_hdr(path, w, h, rgb, cap);
    return evplp::load_pfm(path, w, h, rgb, cap);
}
extern "C" int evplp_load_pfm(const char *path, int32_t *w, int32_t *h, float *rgb, size_t cap) {
    if (!path || !w || !h) return EVPLP_ERR_INVALID;
    return evplp::load_pfm(path, w, h, rgb, cap);
}
// relMSE over the pixels a mask keeps (mask_rgb8: 3 bytes per pixel as evplp_decode_image returns them; a pixel is
// kept when any channel is non-zero).  The reference ships scene/conference/conference_mask.png -- white with black
// outlines around the emitters, whose edges it does not anti-alias (scene/conference/README.md) -- and leaves the
// masked metric to external scripts; this is ComputeRelMse (floatimage.cpp:86-112) restricted to the kept pixels.
extern "C" double evplp_image_rel_mse_masked(int32_t npix, const float *a, const float *ref, const uint8_t *mask_rgb8) {
    float result = 0; int64_t kept = 0;
    for (int32_t i = 0; i < npix; i++) {
        if (mask_rgb8 && !(mask_rgb8[3 * i] | mask_rgb8[3 * i + 1] | mask_rgb8[3 * i + 2])) continue;
        float rx = ref[3 * i], ry = ref[3 * i + 1], rz = ref[3 * i + 2];
        float dx = a[3 * i] - rx, dy = a[3 * i + 1] - ry, dz = a[3 * i + 2] - rz;
        result += (dx * dx + dy * dy + dz * dz) / (rx * rx + ry * ry + rz * rz + 0.001f);
        kept++;
    }
    return kept ? result / (float)kept : 0.0;
}
// math/color.h:18-46, 83-88: Heat(t) = Hsl2Rgb(((1 - t) * 240) / 360, 1, 0.5), with the reference's Hsl2Rgb as written
// (its l < 0.5 branch multiplies by the hue instead of the saturation; Heat always takes the other branch)
namespace {
float hue2rgb(float v1, float v2, float h) {
    if (h < 0.0f) h += 1.0f;
    if (h > 1.0f) h -= 1.0f;
    if ((6.0f * h) < 1.0f) return v1 + (v2 - v1) * 6.0f * h;
    if ((2.0f * h) < 1.0f) return v2;
    if ((3.0f * h) < 2.0f) return v1 + (v2 - v1) * ((2.0f / 3.0f) - h) * 6.0f;
    return v1;
}
void heat(float t, float *rgb) {
    const float h = ((1.0f - t) * 240.0f) / 360.0f, s = 1.0f, l = 0.5f;
    const float v2 = (l < 0.5f) ? (l * (1.0f + h)) : ((s + l) - (s * l));
    const float v1 = 2.0f * l - v2;
    rgb[0] = hue2rgb(v1, v2, h + (1.0f / 3.0f)); rgb[1] = hue2rgb(v1, v2, h); rgb[2] = hue2rgb(v1, v2, h - (1.0f / 3.0f));
}
} // namespace
// FloatImage::ComputeSquareErrorHeatImage / ComputeRelSquareErrorHeatImage (common/floatimage/floatimage.cpp:21-62)
extern "C" int evplp_image_error_heat(int32_t npix, const float *img, const float *ref, float max_error, int32_t relative, float *out_rgb) {
    if (npix < 0 || !img || !ref || !out_rgb) return EVPLP_ERR_INVALID;
    for (int32_t i = 0; i < npix; i++) {
        float rx = ref[3 * i], ry = ref[3 * i + 1], rz = ref[3 * i + 2];
        float dx = img[3 * i] - rx, dy = img[3 * i + 1] - ry, dz = img[3 * i + 2] - rz;
        float e = dx * dx + dy * dy + dz * dz;                       // glm::distance2 / dot(diff, diff)
        if (relative) e = e / (rx * rx + ry * ry + rz * rz + 0.001f);
        heat(std::min(e / max_error, 1.0f), out_rgb + 3 * (size_t)i);
    }
    return EVPLP_OK;
}
// common/floatimage/floatimage.cpp:64-84 (Float accumulator)
extern "C" double evplp_image_mse(int32_t npix, const float *a, const float *ref) {
    float result = 0;
    for (int32_t i = 0; i < npix; i++) {
        float dx = a[3 * i] - ref[3 * i], dy = a[3 * i + 1] - ref[3 * i + 1], dz = a[3 * i + 2] - ref[3 * i + 2];
        result += dx * dx + dy * dy + dz * dz;
    }
    return result / (float)npix;
}
// floatimage.cpp:86-112
extern "C" double evplp_image_rel_mse(int32_t npix, const float *a, const float *ref) {
    float result = 0;
    for (int32_t i = 0; i < npix; i++) {
        float rx = ref[3 * i], ry = ref[3 * i + 1], rz = ref[3 * i + 2];
        float dx = a[3 * i] - rx, dy = a[3 * i + 1] - ry, dz = a[3 * i + 2] - rz;
        float num = dx * dx + dy * dy + dz * dz;
        float den = rx * rx + ry * ry + rz * rz + 0.001f;
        result += num / den;
    }
    return result / (float)npix;
}

