// Output surface of the reference: FloatImage::Save by extension
// (common/floatimage/floatimage.cpp:260-273): PFM (:178-199), HDR (:223-239) and PNG (:241-258).
#pragma once
#include <cstddef>
#include <cstdint>

namespace evplp {
// rgb: top-down rows, 3 floats per pixel.  Returns EVPLP_OK / EVPLP_ERR_IO / EVPLP_ERR_INVALID.
int save_image(const char *path, int32_t w, int32_t h, const float *rgb);
int save_pfm(const char *path, int32_t w, int32_t h, const float *rgb);
int save_png(const char *path, int32_t w, int32_t h, const float *rgb);
int save_hdr(const char *path, int32_t w, int32_t h, const float *rgb);
int load_pfm(const char *path, int32_t *w, int32_t *h, float *rgb, size_t capacity_floats);
int load_hdr(const char *path, int32_t *w, int32_t *h, float *rgb, size_t capacity_floats);
}
