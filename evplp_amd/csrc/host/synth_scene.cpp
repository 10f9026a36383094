// Procedural "conference-like" closed room, written as OBJ + MTL + light OBJ + scene JSON in the
// reference's schema.  Every mesh of the reference repository is a Git-LFS pointer stub
// (SURVEY section 0), so benchmark and parity inputs have to be synthesised; results on this
// scene are labelled "synthetic scene".  Layout follows SURVEY 8(d): a box room seen by the
// conference camera (scene/conference/conference_vpl.json:16-33), a table, chairs, a cabinet,
// eight ceiling quads merged into ONE light mesh (the reference allows exactly one,
// rt/rtcommon.h:794-795), Lambert rho_d in U(0.2,0.8) per object, every fifth object with a
// Phong lobe (rho_s = 0.2, e = 20), light intensity [17,12,4,0].  All surfaces are consistently
// wound (outward for objects, inward for the room): back-face hits end light paths
// (rt/lighttracing.cu:124).
#include "../../../include/evplp.h"
#include "json.hpp"

#include <cmath>
#include <cstdio>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

struct Face { float o[3], u[3], v[3]; int object; };
struct Rng32 { uint32_t s; float next() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffffff) / 16777216.0f; } };

void add_face(std::vector<Face> &faces, const float o[3], const float u[3], const float v[3], int object) {
    Face f; for (int k = 0; k < 3; k++) { f.o[k] = o[k]; f.u[k] = u[k]; f.v[k] = v[k]; } f.object = object; faces.push_back(f);
}
// axis-aligned box; outward = true for solid objects, false for the room shell (normals point inside)
void add_box(std::vector<Face> &faces, const float lo[3], const float hi[3], bool outward, int object, bool separate_objects = false) {
    float d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    for (int axis = 0; axis < 3; axis++) for (int side = 0; side < 2; side++) {
        int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
        float o[3] = { lo[0], lo[1], lo[2] }, u[3] = { 0, 0, 0 }, v[3] = { 0, 0, 0 };
        if (side) o[axis] = hi[axis];
        // u x v must point along +axis for (side=1, outward) and (side=0, inward)
        bool plus = (side == 1) == outward;
        if (plus) { u[a1] = d[a1]; v[a2] = d[a2]; }
        else { u[a2] = d[a2]; v[a1] = d[a1]; }
        add_face(faces, o, u, v, separate_objects ? object + axis * 2 + side : object);
    }
}
float len(const float a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// tessellate faces into a grid with a triangle budget proportional to area; append to an OBJ stream
size_t write_faces(FILE *f, const std::vector<Face> &faces, size_t target_tris, const char *mtl_prefix, bool use_mtl) {
    double total = 0; for (auto &fc : faces) total += (double)len(fc.u) * len(fc.v);
    size_t vbase = 1, tris = 0; int cur_obj = -1;
    for (auto &fc : faces) {
        double area = (double)len(fc.u) * len(fc.v);
        double cells = std::max(1.0, (double)target_tris * area / total / 2.0);
        double ratio = len(fc.u) / std::max(len(fc.v), 1e-6f);
        int nu = std::max(1, (int)std::lround(std::sqrt(cells * ratio)));
        int nv = std::max(1, (int)std::lround(cells / nu));
        if (use_mtl && fc.object != cur_obj) { std::fprintf(f, "usemtl %s%d\n", mtl_prefix, fc.object); cur_obj = fc.object; }
        for (int j = 0; j <= nv; j++) for (int i = 0; i <= nu; i++) {
            float s = (float)i / nu, t = (float)j / nv;
            std::fprintf(f, "v %.9g %.9g %.9g\n", fc.o[0] + fc.u[0] * s + fc.v[0] * t, fc.o[1] + fc.u[1] * s + fc.v[1] * t, fc.o[2] + fc.u[2] * s + fc.v[2] * t);
            std::fprintf(f, "vt %.9g %.9g\n", s, t);
        }
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
            size_t a = vbase + (size_t)j * (nu + 1) + i, b = a + 1, c = b + (nu + 1), d = a + (nu + 1);
            std::fprintf(f, "f %zu/%zu %zu/%zu %zu/%zu\nf %zu/%zu %zu/%zu %zu/%zu\n", a, a, b, b, c, c, a, a, c, c, d, d);
            tris += 2;
        }
        vbase += (size_t)(nu + 1) * (nv + 1);
    }
    return tris;
}

} // namespace

extern "C" int evplp_synth_scene(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed, int32_t res_x, int32_t res_y) {
    if (!out_dir || !name || target_triangles < 12 || res_x <= 0 || res_y <= 0) return EVPLP_ERR_INVALID;
    mkdir(out_dir, 0755);
    std::string base = std::string(out_dir) + "/" + name;
    std::vector<Face> faces; int nobj = 0;
    // room shell (six separately coloured surfaces), normals inward
    { float lo[3] = { -8.f, -7.f, 0.f }, hi[3] = { 18.f, 9.f, 6.5f }; add_box(faces, lo, hi, false, nobj, true); nobj += 6; }
    // table: top + two pedestals
    { float lo[3] = { 0.f, -1.f, 1.0f }, hi[3] = { 10.f, 3.f, 1.15f }; add_box(faces, lo, hi, true, nobj++); }
    { float lo[3] = { 1.5f, 0.5f, 0.f }, hi[3] = { 2.5f, 1.5f, 1.0f }; add_box(faces, lo, hi, true, nobj++); }
    { float lo[3] = { 7.5f, 0.5f, 0.f }, hi[3] = { 8.5f, 1.5f, 1.0f }; add_box(faces, lo, hi, true, nobj++); }
    // chairs: seat, back, leg
    auto chair = [&](float cx, float cy, float bx, float by) {
        { float lo[3] = { cx - 0.4f, cy - 0.4f, 0.55f }, hi[3] = { cx + 0.4f, cy + 0.4f, 0.65f }; add_box(faces, lo, hi, true, nobj); }
        { float lo[3] = { cx + bx * 0.4f - (bx != 0 ? 0.05f : 0.4f), cy + by * 0.4f - (by != 0 ? 0.05f : 0.4f), 0.65f },
                hi[3] = { cx + bx * 0.4f + (bx != 0 ? 0.05f : 0.4f), cy + by * 0.4f + (by != 0 ? 0.05f : 0.4f), 1.55f }; add_box(faces, lo, hi, true, nobj); }
        { float lo[3] = { cx - 0.075f, cy - 0.075f, 0.f }, hi[3] = { cx + 0.075f, cy + 0.075f, 0.55f }; add_box(faces, lo, hi, true, nobj); }
        nobj++;
    };
    for (int i = 0; i < 7; i++) { chair(0.9f + 1.37f * i, -2.1f, 0, -1); chair(0.9f + 1.37f * i, 4.1f, 0, 1); }
    chair(-1.1f, 1.0f, -1, 0); chair(11.1f, 1.0f, 1, 0);
    // cabinet on the far wall, a side board, a column
    { float lo[3] = { -7.99f, -4.f, 0.f }, hi[3] = { -7.f, 4.f, 2.2f }; add_box(faces, lo, hi, true, nobj++); }
    { float lo[3] = { 2.f, 8.2f, 0.f }, hi[3] = { 9.f, 8.99f, 0.9f }; add_box(faces, lo, hi, true, nobj++); }
    { float lo[3] = { 13.f, 5.f, 0.f }, hi[3] = { 13.8f, 5.8f, 6.49f }; add_box(faces, lo, hi, true, nobj++); }

    // light: 8 ceiling quads facing down (u x v = -z)
    std::vector<Face> lights;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) {
        float cx = -0.5f + 3.7f * i, cy = -1.5f + 5.0f * j;
        float o[3] = { cx - 1.0f, cy - 0.4f, 6.45f }, u[3] = { 0.f, 0.8f, 0.f }, v[3] = { 2.0f, 0.f, 0.f };
        add_face(lights, o, u, v, 0);
    }

    FILE *f = std::fopen((base + ".obj").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    std::fprintf(f, "# synthetic conference-like room (evplp_synth_scene seed %u)\nmtllib %s.mtl\n", seed, name);
    size_t ntri = write_faces(f, faces, (size_t)target_triangles, "obj", true);
    std::fclose(f);
    f = std::fopen((base + ".mtl").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    Rng32 rng{ seed * 2654435761u + 12345u };
    for (int i = 0; i < nobj; i++) {
        float kd[3] = { 0.2f + 0.6f * rng.next(), 0.2f + 0.6f * rng.next(), 0.2f + 0.6f * rng.next() };
        bool glossy = (i % 5) == 4;
        std::fprintf(f, "newmtl obj%d\nKd %.9g %.9g %.9g\nKs %.9g %.9g %.9g\nNs %.9g\n\n", i, kd[0], kd[1], kd[2],
                     glossy ? 0.2f : 0.f, glossy ? 0.2f : 0.f, glossy ? 0.2f : 0.f, glossy ? 20.f : 0.f);
    }
    std::fclose(f);
    f = std::fopen((base + "_lights.obj").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    std::fprintf(f, "# area light: 8 ceiling quads, one mesh\n");
    write_faces(f, lights, 128, "", false);
    std::fclose(f);

    using evplp::Json;
    Json root = Json::object();
    root.set("resX", Json::number(res_x)); root.set("resY", Json::number(res_y));
    Json sc; sc.type = Json::Array; sc.arr.push_back(Json::string(std::string(name) + ".obj")); root.set("scene", sc);
    Json al = Json::object(); al.set("obj", Json::string(std::string(name) + "_lights.obj"));
    Json in; in.type = Json::Array; for (double v : { 17.0, 12.0, 4.0, 0.0 }) in.arr.push_back(Json::number(v)); al.set("intensity", in);
    root.set("arealight", al);
    Json cam = Json::object();   // scene/conference/conference_vpl.json:16-33
    auto vec = [](double a, double b, double c) { Json j; j.type = Json::Array; j.arr = { Json::number(a), Json::number(b), Json::number(c) }; return j; };
    cam.set("origin", vec(15.56, -4.79, 4.37)); cam.set("direction", vec(1.15, 2.28, 1.76)); cam.set("up", vec(0, 0, 1)); cam.set("fovx", Json::number(70.0));
    root.set("camera", cam);
    Json pf = Json::object();    // BASELINE.md section 2, config #2 (Instant Radiosity 4096 VPL slots)
    pf.set("rngOffset", Json::number(0)); pf.set("numMaxIteration", Json::number(1)); pf.set("timeLimitMs", Json::number(1e9));
    pf.set("frameMode", Json::string("accumulate")); pf.set("renderMode", Json::string("vpl")); pf.set("misMode", Json::string("one"));
    pf.set("combinedFilename", Json::string(std::string(name) + "_combined.pfm"));
    pf.set("weightedPhotonFilename", Json::string(std::string(name) + "_weightedpm.pfm"));
    pf.set("weightedVplFilename", Json::string(std::string(name) + "_weightedvpl.pfm"));
    pf.set("statFilename", Json::string(std::string(name) + "_stat.json"));
    pf.set("useJitter", Json::boolean(true)); pf.set("useStat", Json::boolean(true));
    pf.set("numLightPaths", Json::number(1024)); pf.set("numVplLightPaths", Json::number(1024)); pf.set("numMaxBounces", Json::number(3));
    pf.set("radiusPercentage", Json::number(0.0));
    Json run = Json::object(); run.set("photonSplat", Json::boolean(false)); pf.set("run", run);
    root.set("photonfam", pf);
    f = std::fopen((base + ".json").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    std::string text = root.dump();
    std::fwrite(text.data(), 1, text.size(), f); std::fputc('\n', f);
    std::fclose(f);
    return (int)std::min<size_t>(ntri, 0x7fffffff);
}
