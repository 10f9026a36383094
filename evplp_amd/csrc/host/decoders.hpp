// 8-bit image decoders for textured materials (map_Kd / map_Ks / map_Ns of the scene's .mtl files).
// The reference decodes them with the stb_image v2.16 vendored in its tree,
//     stbi_load(filepath, &width, &height, &channel, 3)      (rt/rtcommon.h:144, flip on load :32)
// and the decoded bytes enter the BRDF directly (value / 255, gamma 1.0), so these decoders reproduce that
// decoder's arithmetic bit for bit: the integer "islow" IDCT with 2 extra bits between the passes, the
// 3:1 triangle chroma upsampling, the 12-bit fixed-point YCbCr conversion, high-byte 16 -> 8 bit PNG
// reduction.  Own implementation (no third-party code); checked byte-for-byte against the reference's
// decoder through oracle/_ref (tests/test_oracle_pins.py, fixtures in tests/golden/textures.npz).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace evplp {

struct DecodedImage {
    int w = 0, h = 0;
    int channels = 0;             // components in the FILE as stbi_load reports them (1, 2, 3 or 4)
    std::vector<uint8_t> rgb;     // always 3 interleaved components (req_comp = 3), rows top to bottom
};

// RFC 1950 / 1951.  Throws std::runtime_error on malformed input.
std::vector<uint8_t> zlib_inflate(const uint8_t *data, size_t size, size_t size_hint);

bool is_png(const uint8_t *data, size_t size);
bool is_jpeg(const uint8_t *data, size_t size);
DecodedImage decode_png(const uint8_t *data, size_t size);
DecodedImage decode_jpeg(const uint8_t *data, size_t size);
// by content, like stbi_load: JPEG, then PNG.  Throws std::runtime_error naming the path.
DecodedImage decode_image_file(const std::string &path);

} // namespace evplp
