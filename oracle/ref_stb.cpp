// oracle/_ref/libref_pin.so, second translation unit: entry points INTO the image decoder the reference links
// (stb_image v2.16, vendored under reflectcuts/stb/ and compiled from there -- nothing is copied), called exactly
// as RtTexture does:  stbi_set_flip_vertically_on_load(..); stbi_load(path, &w, &h, &channel, 3)
// (rt/rtcommon.h:32,144).  Test infrastructure: pins evplp_decode_image (tests/test_oracle_pins.py) and generates
// tests/golden/textures.npz (tests/golden/make_golden.py).
#define STB_IMAGE_IMPLEMENTATION
#include "stb/stb_image.h"

extern "C" {
unsigned char *ref_stbi_load(const char *path, int *w, int *h, int *channels, int flip) {
    stbi_set_flip_vertically_on_load(flip);
    return stbi_load(path, w, h, channels, 3);
}
void ref_stbi_free(void *p) { stbi_image_free(p); }
const char *ref_stbi_failure(void) { return stbi_failure_reason(); }
}
