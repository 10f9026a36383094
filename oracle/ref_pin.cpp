// oracle/_ref/libref_pin.so -- thin extern "C" entry points INTO the reference's own code, used
// only by tests/test_oracle_pins.py to pin the parts of the oracle that the reference can vouch
// for in this container:
//   * FloatImage::Save / FlipY / ComputeMse / ComputeRelMse  (common/floatimage/floatimage.cpp)
//   * glm::lookAt / glm::perspective / glm::translate exactly as RtStableCamera::computeVpMatrix
//     and the jitter matrix use them (rt/rtcommon.h:586-591, rt/rtcomphoton/rtcomphoton.h:943-952),
//     from the GLM 0.9.8 vendored under dependencies/include.
//   * Aabb::Union / DiagonalLength2 (math/aabb.h) feeding the photon radius (rtcommon.h:805-814).
// This file contains no reference code; it calls it.
#include "common/floatimage/floatimage.h"
#include "math/aabb.h"

#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/transform.hpp>

static FloatImage make_image(int w, int h, const float *rgb) {
    FloatImage img((size_t)w, (size_t)h);
    std::memcpy(img.getFloats(), rgb, sizeof(float) * 3 * (size_t)w * h);
    return img;
}

extern "C" {

int ref_save(const char *path, int w, int h, const float *rgb_top_down) {
    try { FloatImage::Save(make_image(w, h, rgb_top_down), path); } catch (...) { return -1; }
    return 0;
}
void ref_flip_y(int w, int h, const float *rgb, float *out) {
    FloatImage f = FloatImage::FlipY(make_image(w, h, rgb));
    std::memcpy(out, f.getFloats(), sizeof(float) * 3 * (size_t)w * h);
}
// FloatImage::Compute[Rel]SquareErrorHeatImage (floatimage.cpp:21-62) and FloatImage::Load (PFM / HDR)
void ref_error_heat(int w, int h, const float *a, const float *ref, float max_error, int relative, float *out) {
    FloatImage r = relative ? FloatImage::ComputeRelSquareErrorHeatImage(make_image(w, h, a), make_image(w, h, ref), max_error)
                            : FloatImage::ComputeSquareErrorHeatImage(make_image(w, h, a), make_image(w, h, ref), max_error);
    std::memcpy(out, r.getFloats(), sizeof(float) * 3 * (size_t)w * h);
}
int ref_load_hdr(const char *path, int w, int h, float *out) {
    FloatImage r = FloatImage::LoadHDR(path);
    if ((int)r.getSize().x != w || (int)r.getSize().y != h) return -1;
    std::memcpy(out, r.getFloats(), sizeof(float) * 3 * (size_t)w * h);
    return 0;
}
double ref_mse(int w, int h, const float *a, const float *ref) { return FloatImage::ComputeMse(make_image(w, h, a), make_image(w, h, ref)); }
double ref_rel_mse(int w, int h, const float *a, const float *ref) { return FloatImage::ComputeRelMse(make_image(w, h, a), make_image(w, h, ref)); }

// projection * view (rtcommon.h:586-591), optionally pre-multiplied by the jitter translation
// (rtcomphoton.h:949-951).  out: 16 floats, column-major like GLM.
void ref_view_projection(const float origin[3], const float lookat[3], const float up[3], float fovy, float aspect,
                         const float jitter[2], float out[16]) {
    glm::mat4 view = glm::lookAt(glm::vec3(origin[0], origin[1], origin[2]), glm::vec3(lookat[0], lookat[1], lookat[2]), glm::vec3(up[0], up[1], up[2]));
    glm::mat4 proj = glm::perspective(fovy, aspect, 0.1f, 100.0f);
    glm::mat4 m = proj * view;
    if (jitter) m = glm::translate(glm::vec3(jitter[0], jitter[1], 0)) * m;
    std::memcpy(out, &m[0][0], sizeof(float) * 16);
}
// fovx (degrees) -> fovy exactly as rtcommon.h:556-560
float ref_fovx_to_fovy(float fovx_degree, float aspect) { return 2.0f * std::atan2(std::tan(glm::radians(fovx_degree) * 0.5f), aspect); }

// RtScene::findBoundingSphereRadius (rtcommon.h:805-814) over a vertex list
float ref_bounding_sphere_radius(int nverts, const float *verts) {
    Aabb bbox;
    for (int i = 0; i < nverts; i++) bbox = Aabb::Union(bbox, Vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]));
    float diameter = std::sqrt(Aabb::DiagonalLength2(bbox));
    return diameter / 2.0f;
}

}

// ---- the jitter sampler and the triangle area feeding the light CDF (round 2 pins)
#include "sampler/independent.h"
extern "C" {
// IndependentSampler(seed).nextVec2() n times, exactly as RtComPhoton::run draws its jitter (rtcomphoton.h:887, 949):
// std::mt19937 through std::uniform_real_distribution<float> (common/rng.h:34) -- here libstdc++'s, the authors' was MSVC's --
// with the two draws of nextVec2 passed as constructor arguments (sampler/independent.h:37-40: evaluation order unspecified).
void ref_jitter_vec2(uint32_t seed, int n, float *out) {
    IndependentSampler s(seed);
    for (int i = 0; i < n; i++) { Vec2 v = s.nextVec2(); out[2 * i] = (float)v.x; out[2 * i + 1] = (float)v.y; }
}
// the NDC jitter itself: (2 u - 1) * invResolution (rtcomphoton.h:949)
void ref_jitter_ndc(uint32_t seed, int n, float res_x, float res_y, float *out) {
    IndependentSampler s(seed);
    glm::vec2 inv_res(1.0f / res_x, 1.0f / res_y);
    for (int i = 0; i < n; i++) { glm::vec2 j = (2.0f * glm::vec2(s.nextVec2()) - glm::vec2(1)) * inv_res; out[2 * i] = j.x; out[2 * i + 1] = j.y; }
}
// Triangle::ComputeArea (shapes/trianglemesh.cpp:13-19) cannot be compiled here (the file needs Assimp); this is its expression
// on the vendored GLM: glm::length(glm::cross(b - a, c - a)) / 2.0f
float ref_triangle_area(const float a[3], const float b[3], const float c[3]) {
    glm::vec3 A(a[0], a[1], a[2]), B(b[0], b[1], b[2]), Cc(c[0], c[1], c[2]);
    return glm::length(glm::cross(B - A, Cc - A)) / 2.0f;
}

}
