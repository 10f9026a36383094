// Polygons of an OBJ file through the library's reader (scene_io.cpp triangulate_polygon; tests/test_host_sanitizers.py builds this with ASan + UBSan):
// convex ones come out as a fan from their first corner, concave ones ear-clipped -- n - 2 triangles whose areas add up to the polygon's,
// every one wound like the polygon (a fan of an L-shaped or arrow-shaped face covers area outside it, or folds back over it).
#include "../../evplp_amd/csrc/host/scene_io.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#include <cstdlib>
// link stubs: scene_io.cpp's upload_scene talks to the device library through these; this harness never calls it
extern "C" {
int32_t evplp_add_texture(evplp_context *, int32_t, int32_t, const float *) { std::abort(); }
int32_t evplp_add_material(evplp_context *, const evplp_material *) { std::abort(); }
int32_t evplp_add_mesh(evplp_context *, const float *, const float *, int32_t, const int32_t *, int32_t, int32_t) { std::abort(); }
int32_t evplp_set_arealight(evplp_context *, int32_t, const float *) { std::abort(); }
int32_t evplp_set_camera(evplp_context *, const evplp_camera *) { std::abort(); }
int32_t evplp_build_accel(evplp_context *) { std::abort(); }
}
namespace evplp { void set_context_error(evplp_context *, const char *) {} }
struct V { double x, y, z; };
static V sub(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V cross(V a, V b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // polygons in the plane spanned by (ex, ey) around an origin: 2-D corners, counter-clockwise
    const std::vector<std::vector<std::pair<double, double>>> polys = {
        { { 0, 0 }, { 2, 0 }, { 2, 1 }, { 1, 1 }, { 1, 2 }, { 0, 2 } },                          // L
        { { 0, 0 }, { 1, 0.4 }, { 2, 0 }, { 1, 2 } },                                              // arrow: a quad with one reflex corner (at index 1)
        { { 0, 0 }, { 2, 0 }, { 2.5, 1.5 }, { 1, 2.5 }, { -0.5, 1.5 } },                           // convex pentagon
        { { 0, 0 }, { 3, 0 }, { 3, 3 }, { 2, 3 }, { 2, 1 }, { 1, 1 }, { 1, 3 }, { 0, 3 } },        // U
        { { 1, 0 }, { 1.3, 0.9 }, { 2.2, 0.9 }, { 1.45, 1.45 }, { 1.75, 2.3 }, { 1, 1.8 }, { 0.25, 2.3 }, { 0.55, 1.45 }, { -0.2, 0.9 }, { 0.7, 0.9 } },   // star
        { { 0, 0 }, { 1, 0 }, { 2, 0 }, { 2, 2 }, { 0, 2 } },                                      // convex with a flat corner
    };
    const V origin = { 0.3, -1.2, 2.0 }, ex = { 0.6, 0.0, 0.8 }, ey = { -0.48, 0.8, 0.36 };       // orthonormal
    int failures = 0;
    for (size_t p = 0; p < polys.size(); p++) for (int flip = 0; flip < 2; flip++) {
        std::vector<std::pair<double, double>> c = polys[p];
        if (flip) std::reverse(c.begin(), c.end());                                                // clockwise in the plane: the other orientation
        const std::string path = dir + "/poly_" + std::to_string(p) + "_" + std::to_string(flip) + ".obj";
        { std::ofstream f(path); f.precision(9);
          for (auto &q : c) f << "v " << origin.x + q.first * ex.x + q.second * ey.x << " " << origin.y + q.first * ex.y + q.second * ey.y << " " << origin.z + q.first * ex.z + q.second * ey.z << "\n";
          f << "f"; for (size_t k = 0; k < c.size(); k++) f << " " << k + 1; f << "\n"; }
        double area2 = 0; for (size_t k = 0; k < c.size(); k++) { const auto &a = c[k], &b = c[(k + 1) % c.size()]; area2 += a.first * b.second - b.first * a.second; }
        const V n = cross(ex, ey); const double want = 0.5 * area2;                                // signed: negative when flipped
        evplp::MeshData m = evplp::load_single_mesh_obj(path);
        const size_t ntri = m.idx.size() / 3;
        double got = 0; bool wound = true;
        for (size_t t = 0; t < ntri; t++) {
            V q[3]; for (int k = 0; k < 3; k++) { const int i = m.idx[3 * t + k]; q[k] = { m.verts[3 * i], m.verts[3 * i + 1], m.verts[3 * i + 2] }; }
            const double a = 0.5 * dot(cross(sub(q[1], q[0]), sub(q[2], q[0])), n);
            got += a; if (a * want < -1e-6 * want * want) wound = false;                                         // (a flat corner may give a zero-area triangle)
        }
        const bool ok = ntri == c.size() - 2 && std::fabs(got - want) < 1e-5 * std::fabs(want) && wound;
        std::printf("polygon %zu%s: %zu corners -> %zu triangles, area %.6f (polygon %.6f)%s\n", p, flip ? " (clockwise)" : "", c.size(), ntri, got, want, ok ? "" : "  FAIL");
        if (!ok) failures++;
    }
    std::printf(failures ? "FAILED\n" : "ok\n");
    return failures ? 1 : 0;
}
