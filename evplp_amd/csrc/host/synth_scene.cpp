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
#include "images.hpp"
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

// scene JSON in the reference's schema: conference camera (scene/conference/conference_vpl.json:16-33) + a `photonfam`
// block shaped like BASELINE config #2
static int write_scene_json(const std::string &base, const char *name, int32_t res_x, int32_t res_y) {
    FILE *f;
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
    return EVPLP_OK;
}

static int synth_boxes(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed, int32_t res_x, int32_t res_y) {
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

    if (int rc = write_scene_json(base, name, res_x, res_y)) return rc;
    return (int)std::min<size_t>(ntri, 0x7fffffff);
}

// ------------------------------------------------------------------------------------------------------------
// style 1, "furnished": the same room, light and camera, but furniture made of curved and thin parts (ellipsoid
// cushions, cylinder legs / columns / arm rests, a round-edged table), arbitrarily rotated clutter on and under the
// table, and a few thousand small occluders (the leaves of two plants, the slats of a blind).  The real conference
// model is ~331 k triangles of curved furniture in a plain room; an any-hit packet walk sees far more partial
// occlusion and many more non-axis-aligned leaf boxes here than among tessellated boxes.  Closed, outward wound.
namespace {
struct Patch {
    int kind;                 // 0 plane o + ex s + ey t | 1 ellipsoid | 2 cylinder side (axis ez, t in [0,1]) | 3 disk (normal +ez)
    float o[3], ex[3], ey[3], ez[3];
    bool flip;                // swap the winding
    int min_u, min_v;         // tessellation floor
    float weight;             // triangle density relative to area
    int object;
};
struct Frame { float o[3], x[3], y[3], z[3]; };
void v_set(float *d, float a, float b, float c) { d[0] = a; d[1] = b; d[2] = c; }
Frame frame_at(float x, float y, float z, float yaw, float tilt) {      // rotation about z, then about the rotated x
    Frame f; v_set(f.o, x, y, z);
    float cy = std::cos(yaw), sy = std::sin(yaw), ct = std::cos(tilt), st = std::sin(tilt);
    v_set(f.x, cy, sy, 0.f);
    v_set(f.y, -sy * ct, cy * ct, st);
    v_set(f.z, sy * st, -cy * st, ct);
    return f;
}
void to_world(const Frame &f, const float l[3], float w[3], bool point) {
    for (int k = 0; k < 3; k++) w[k] = (point ? f.o[k] : 0.f) + f.x[k] * l[0] + f.y[k] * l[1] + f.z[k] * l[2];
}
void eval_patch(const Patch &p, float s, float t, float out[3]) {
    const float two_pi = 6.283185307179586f, pi = 3.14159265358979f;
    float a = 0.f, b = 0.f, c = 0.f;
    if (p.kind == 0) { a = s; b = t; }
    else if (p.kind == 1) { float th = pi * t, ph = two_pi * s; a = std::sin(th) * std::cos(ph); b = std::sin(th) * std::sin(ph); c = std::cos(th); }
    else if (p.kind == 2) { float ph = two_pi * s; a = std::cos(ph); b = std::sin(ph); c = t; }
    else { float ph = two_pi * s; a = std::cos(ph) * t; b = std::sin(ph) * t; }
    for (int k = 0; k < 3; k++) out[k] = p.o[k] + p.ex[k] * a + p.ey[k] * b + p.ez[k] * c;
}
double patch_area(const Patch &p) {
    float lx = len(p.ex), ly = len(p.ey), lz = len(p.ez);
    const double pi = 3.14159265358979;
    if (p.kind == 0) return (double)lx * ly;
    if (p.kind == 1) { double a = lx, b = ly, c = lz; return 4.0 * pi * std::pow((std::pow(a * b, 1.6) + std::pow(a * c, 1.6) + std::pow(b * c, 1.6)) / 3.0, 1.0 / 1.6); }
    if (p.kind == 2) return 2.0 * pi * 0.5 * (lx + ly) * lz;
    return pi * lx * ly;
}
struct Builder {
    std::vector<Patch> patches;
    void plane(const Frame &f, const float lo[3], const float u[3], const float v[3], int object, float weight) {
        Patch p{}; p.kind = 0; p.flip = false; p.min_u = p.min_v = 1; p.weight = weight; p.object = object;
        to_world(f, lo, p.o, true); to_world(f, u, p.ex, false); to_world(f, v, p.ey, false); v_set(p.ez, 0, 0, 0);
        patches.push_back(p);
    }
    // oriented box [lo, hi] in the local frame, outward
    void box(const Frame &f, const float lo[3], const float hi[3], int object, float weight) {
        float d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
        for (int axis = 0; axis < 3; axis++) for (int side = 0; side < 2; side++) {
            int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            float o[3] = { lo[0], lo[1], lo[2] }, u[3] = { 0, 0, 0 }, v[3] = { 0, 0, 0 };
            if (side) o[axis] = hi[axis];
            if (side == 1) { u[a1] = d[a1]; v[a2] = d[a2]; } else { u[a2] = d[a2]; v[a1] = d[a1]; }
            plane(f, o, u, v, object, weight);
        }
    }
    void ellipsoid(const Frame &f, const float c[3], float rx, float ry, float rz, int object, float weight) {
        Patch p{}; p.kind = 1; p.flip = true; p.min_u = 10; p.min_v = 6; p.weight = weight; p.object = object;
        to_world(f, c, p.o, true);
        float ax[3] = { rx, 0, 0 }, ay[3] = { 0, ry, 0 }, az[3] = { 0, 0, rz };
        to_world(f, ax, p.ex, false); to_world(f, ay, p.ey, false); to_world(f, az, p.ez, false);
        patches.push_back(p);
    }
    // closed cylinder: base centre c, axis (0,0,1) of the local frame, radius r, height h
    void cylinder(const Frame &f, const float c[3], float r, float h, int object, float weight, int min_u = 10) {
        Patch p{}; p.kind = 2; p.flip = false; p.min_u = min_u; p.min_v = 1; p.weight = weight; p.object = object;
        to_world(f, c, p.o, true);
        float ax[3] = { r, 0, 0 }, ay[3] = { 0, r, 0 }, az[3] = { 0, 0, h };
        to_world(f, ax, p.ex, false); to_world(f, ay, p.ey, false); to_world(f, az, p.ez, false);
        patches.push_back(p);
        Patch top = p; top.kind = 3; top.min_v = 1; for (int k = 0; k < 3; k++) top.o[k] = p.o[k] + p.ez[k];
        patches.push_back(top);
        Patch bot = p; bot.kind = 3; bot.flip = true; bot.min_v = 1;
        patches.push_back(bot);
    }
};
size_t write_patches(FILE *f, const std::vector<Patch> &patches, size_t target_tris, const char *mtl_prefix) {
    double total = 0; for (auto &p : patches) total += patch_area(p) * p.weight;
    size_t vbase = 1, tris = 0; int cur_obj = -1;
    std::vector<float> pos;
    for (auto &p : patches) {
        double cells = std::max(1.0, (double)target_tris * patch_area(p) * p.weight / total / 2.0);
        double lu, lv;
        if (p.kind == 0) { lu = len(p.ex); lv = len(p.ey); }
        else if (p.kind == 1) { lu = 6.283 * 0.5 * (len(p.ex) + len(p.ey)); lv = 3.1416 * len(p.ez); }
        else if (p.kind == 2) { lu = 6.283 * 0.5 * (len(p.ex) + len(p.ey)); lv = len(p.ez); }
        else { lu = 6.283 * 0.5 * (len(p.ex) + len(p.ey)); lv = 0.5 * (len(p.ex) + len(p.ey)); }
        double ratio = lu / std::max(lv, 1e-6);
        int nu = std::max(p.min_u, (int)std::lround(std::sqrt(cells * ratio)));
        int nv = std::max(p.min_v, (int)std::lround(cells / std::max(nu, 1)));
        if (p.kind == 3) nv = std::max(1, std::min(nv, 3));
        if (p.object != cur_obj) { std::fprintf(f, "usemtl %s%d\n", mtl_prefix, p.object); cur_obj = p.object; }
        pos.resize((size_t)(nu + 1) * (nv + 1) * 3);
        for (int j = 0; j <= nv; j++) for (int i = 0; i <= nu; i++) {
            // closed parametrisations reuse the s = 0 column so the seam is watertight
            float s = (p.kind != 0 && i == nu) ? 0.0f : (float)i / nu, t = (float)j / nv;
            float *q = &pos[((size_t)j * (nu + 1) + i) * 3];
            eval_patch(p, s, t, q);
            std::fprintf(f, "v %.9g %.9g %.9g\nvt %.9g %.9g\n", q[0], q[1], q[2], (float)i / nu, t);
        }
        auto same = [&](size_t a, size_t b) { return pos[a * 3] == pos[b * 3] && pos[a * 3 + 1] == pos[b * 3 + 1] && pos[a * 3 + 2] == pos[b * 3 + 2]; };
        auto tri = [&](size_t a, size_t b, size_t c) {
            if (same(a, b) || same(b, c) || same(a, c)) return;          // pole / centre fans: no zero-area triangles
            if (p.flip) std::swap(b, c);
            std::fprintf(f, "f %zu/%zu %zu/%zu %zu/%zu\n", vbase + a, vbase + a, vbase + b, vbase + b, vbase + c, vbase + c);
            tris++;
        };
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
            size_t a = (size_t)j * (nu + 1) + i, b = a + 1, c = b + (nu + 1), d = a + (nu + 1);
            tri(a, b, c); tri(a, c, d);
        }
        vbase += (size_t)(nu + 1) * (nv + 1);
    }
    return tris;
}
} // namespace

static int synth_furnished(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed, int32_t res_x, int32_t res_y, bool textured) {
    std::string base = std::string(out_dir) + "/" + name;
    Builder B; int nobj = 0;
    Rng32 rng{ seed * 747796405u + 2891336453u };
    auto uni = [&](float a, float b) { return a + (b - a) * rng.next(); };
    const Frame world = frame_at(0, 0, 0, 0, 0);
    const float furn = 1.0f, shell = 0.02f;
    // room shell, normals inward: six separately coloured planes with a low triangle density (large plain polygons in the real model)
    {
        const float lo[3] = { -8.f, -7.f, 0.f }, hi[3] = { 18.f, 9.f, 6.5f }, d[3] = { 26.f, 16.f, 6.5f };
        for (int axis = 0; axis < 3; axis++) for (int side = 0; side < 2; side++) {
            int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            float o[3] = { lo[0], lo[1], lo[2] }, u[3] = { 0, 0, 0 }, v[3] = { 0, 0, 0 };
            if (side) o[axis] = hi[axis];
            if (side == 0) { u[a1] = d[a1]; v[a2] = d[a2]; } else { u[a2] = d[a2]; v[a1] = d[a1]; }
            B.plane(world, o, u, v, nobj++, shell);
        }
    }
    // table: a slab with half-cylinder long edges, two cylindrical pedestals with foot plates
    {
        int ob = nobj++;
        { float lo[3] = { 0.f, -0.85f, 1.0f }, hi[3] = { 10.f, 2.85f, 1.15f }; B.box(world, lo, hi, ob, furn); }
        Frame e1 = frame_at(0.f, -0.85f, 1.075f, 0.f, 0.f), e2 = frame_at(0.f, 2.85f, 1.075f, 0.f, 0.f);
        // cylinders along x: local z -> world x
        Frame ex1 = e1; v_set(ex1.x, 0, 1, 0); v_set(ex1.y, 0, 0, 1); v_set(ex1.z, 1, 0, 0);
        Frame ex2 = e2; v_set(ex2.x, 0, 1, 0); v_set(ex2.y, 0, 0, 1); v_set(ex2.z, 1, 0, 0);
        float c0[3] = { 0, 0, 0 };
        B.cylinder(ex1, c0, 0.075f, 10.f, ob, furn, 12); B.cylinder(ex2, c0, 0.075f, 10.f, ob, furn, 12);
        int ped = nobj++;
        for (float px : { 2.0f, 8.0f }) {
            float c[3] = { px, 1.0f, 0.05f }; B.cylinder(world, c, 0.35f, 0.95f, ped, furn, 24);
            float f0[3] = { px, 1.0f, 0.0f }; B.cylinder(world, f0, 0.8f, 0.05f, ped, furn, 32);
        }
    }
    // chairs: ellipsoid seat + back cushions, a gas-spring column, five-star base of tilted cylinders with caster spheres, two arm rests
    auto chair = [&](float cx, float cy, float yaw) {
        int ob = nobj++;
        Frame f = frame_at(cx, cy, 0.f, yaw + uni(-0.35f, 0.35f), 0.f);
        { float c[3] = { 0, 0, 0.60f }; B.ellipsoid(f, c, 0.42f, 0.42f, 0.07f, ob, furn); }
        { Frame fb = f; float l[3] = { 0, -0.40f, 1.10f }, w[3]; to_world(f, l, w, true); fb = frame_at(w[0], w[1], w[2], yaw, uni(-0.25f, -0.05f));
          float c[3] = { 0, 0, 0 }; B.ellipsoid(fb, c, 0.40f, 0.06f, 0.42f, ob, furn); }
        { float c[3] = { 0, -0.36f, 0.60f }; B.cylinder(f, c, 0.025f, 0.25f, ob, furn, 8); }        // back support
        { float c[3] = { 0, 0, 0.12f }; B.cylinder(f, c, 0.035f, 0.44f, ob, furn, 10); }             // column
        for (int k = 0; k < 5; k++) {                                                                // star base
            float ang = 6.2831853f * k / 5.f;
            Frame leg = frame_at(0, 0, 0, 0, 0);
            float dir[3] = { std::cos(ang), std::sin(ang), -0.12f }, wdir[3], o[3] = { 0, 0, 0.13f }, wo[3];
            to_world(f, dir, wdir, false); to_world(f, o, wo, true);
            float n = len(wdir); for (int q = 0; q < 3; q++) wdir[q] /= n;
            v_set(leg.o, wo[0], wo[1], wo[2]); v_set(leg.z, wdir[0], wdir[1], wdir[2]);
            float up[3] = { 0, 0, 1 }; float xx[3] = { wdir[1] * up[2] - wdir[2] * up[1], wdir[2] * up[0] - wdir[0] * up[2], wdir[0] * up[1] - wdir[1] * up[0] };
            float xn = len(xx); for (int q = 0; q < 3; q++) xx[q] /= xn;
            v_set(leg.x, xx[0], xx[1], xx[2]);
            v_set(leg.y, wdir[1] * xx[2] - wdir[2] * xx[1], wdir[2] * xx[0] - wdir[0] * xx[2], wdir[0] * xx[1] - wdir[1] * xx[0]);
            float c0[3] = { 0, 0, 0 }; B.cylinder(leg, c0, 0.022f, 0.36f, ob, furn, 8);
            float cs[3] = { 0, 0, 0.38f }; B.ellipsoid(leg, cs, 0.035f, 0.035f, 0.035f, ob, furn);
        }
        for (float side : { -1.f, 1.f }) {                                                           // arm rests
            float c[3] = { side * 0.44f, -0.05f, 0.62f }; B.cylinder(f, c, 0.018f, 0.22f, ob, furn, 8);
            Frame fa = f; float l[3] = { side * 0.44f, -0.25f, 0.86f }, w[3]; to_world(f, l, w, true);
            v_set(fa.o, w[0], w[1], w[2]); float tx[3], ty[3], tz[3];
            for (int q = 0; q < 3; q++) { tx[q] = f.x[q]; ty[q] = f.z[q]; tz[q] = f.y[q]; }
            v_set(fa.x, tx[0], tx[1], tx[2]); v_set(fa.y, -ty[0], -ty[1], -ty[2]); v_set(fa.z, tz[0], tz[1], tz[2]);
            float c0[3] = { 0, 0, 0 }; B.cylinder(fa, c0, 0.03f, 0.42f, ob, furn, 10);
        }
    };
    for (int i = 0; i < 7; i++) { chair(0.9f + 1.37f * i, -1.9f, 3.14159265f); chair(0.9f + 1.37f * i, 3.9f, 0.f); }
    chair(-1.0f, 1.0f, 1.5707963f); chair(11.0f, 1.0f, -1.5707963f);
    // cabinet, side board (boxes with cylindrical handles), a round column
    { int ob = nobj++; float lo[3] = { -7.99f, -4.f, 0.f }, hi[3] = { -7.f, 4.f, 2.2f }; B.box(world, lo, hi, ob, furn * 0.3f);
      for (int k = 0; k < 8; k++) { float c[3] = { -6.97f, -3.5f + k * 1.0f, 0.9f }; B.cylinder(world, c, 0.02f, 0.4f, ob, furn, 8); } }
    { int ob = nobj++; float lo[3] = { 2.f, 8.2f, 0.f }, hi[3] = { 9.f, 8.99f, 0.9f }; B.box(world, lo, hi, ob, furn * 0.3f); }
    { int ob = nobj++; float c[3] = { 13.4f, 5.4f, 0.f }; B.cylinder(world, c, 0.4f, 6.49f, ob, furn * 0.3f, 32); }
    // clutter ON the table: rotated books, cups, balls, a few laptops (two rotated thin boxes)
    for (int k = 0; k < 70; k++) {
        int ob = nobj + (k % 6);
        float x = uni(0.4f, 9.6f), y = uni(-0.5f, 2.5f);
        Frame f = frame_at(x, y, 1.15f, uni(0.f, 6.28f), 0.f);
        int kind = k % 4;
        if (kind == 0) { float lo[3] = { -0.11f, -0.15f, 0.f }, hi[3] = { 0.11f, 0.15f, uni(0.02f, 0.08f) }; B.box(f, lo, hi, ob, furn); }
        else if (kind == 1) { float c[3] = { 0, 0, 0 }; B.cylinder(f, c, 0.04f, uni(0.08f, 0.14f), ob, furn, 12); }
        else if (kind == 2) { float r = uni(0.04f, 0.09f); float c[3] = { 0, 0, r }; B.ellipsoid(f, c, r, r, r, ob, furn); }
        else { float lo[3] = { -0.17f, -0.12f, 0.f }, hi[3] = { 0.17f, 0.12f, 0.012f }; B.box(f, lo, hi, ob, furn);
               float l[3] = { 0, 0.12f, 0.012f }, w[3]; to_world(f, l, w, true);
               Frame sc = frame_at(w[0], w[1], w[2], 0.f, 0.f); sc.x[0] = f.x[0]; sc.x[1] = f.x[1]; sc.x[2] = f.x[2];
               float t = uni(1.2f, 1.5f), ct = std::cos(t), st = std::sin(t);
               for (int q = 0; q < 3; q++) { sc.y[q] = f.y[q] * ct + f.z[q] * st; sc.z[q] = -f.y[q] * st + f.z[q] * ct; }
               float lo2[3] = { -0.17f, 0.f, 0.f }, hi2[3] = { 0.17f, 0.24f, 0.008f }; B.box(sc, lo2, hi2, ob, furn); }
    }
    nobj += 6;
    // clutter UNDER the table and along the walls: bins, boxes, bags
    for (int k = 0; k < 40; k++) {
        int ob = nobj + (k % 5);
        bool under = k < 24;
        float x = under ? uni(0.5f, 9.5f) : uni(-6.f, 16.f), y = under ? uni(-0.4f, 2.4f) : (k & 1 ? uni(-6.6f, -5.5f) : uni(6.8f, 8.0f));
        Frame f = frame_at(x, y, 0.f, uni(0.f, 6.28f), 0.f);
        if (k % 3 == 0) { float c[3] = { 0, 0, 0 }; B.cylinder(f, c, uni(0.12f, 0.2f), uni(0.3f, 0.5f), ob, furn, 16); }
        else if (k % 3 == 1) { float lo[3] = { -0.25f, -0.18f, 0.f }, hi[3] = { 0.25f, 0.18f, uni(0.2f, 0.5f) }; B.box(f, lo, hi, ob, furn); }
        else { float c[3] = { 0, 0, 0.2f }; B.ellipsoid(f, c, uni(0.15f, 0.3f), uni(0.12f, 0.2f), 0.2f, ob, furn); }
    }
    nobj += 5;
    // two plants: a pot, a stem and ~1100 small thin leaves each, randomly oriented
    for (int pl = 0; pl < 2; pl++) {
        int ob = nobj++;
        float px = pl == 0 ? 12.2f : -5.5f, py = pl == 0 ? -5.6f : 6.5f;
        { float c[3] = { px, py, 0.f }; B.cylinder(world, c, 0.3f, 0.5f, ob, furn, 20); }
        { float c[3] = { px, py, 0.5f }; B.cylinder(world, c, 0.03f, 1.2f, ob, furn, 8); }
        int leaf_ob = nobj++;
        for (int k = 0; k < 1100; k++) {
            float r = uni(0.05f, 0.75f), ang = uni(0.f, 6.28f), hz = uni(0.7f, 2.3f);
            Frame f = frame_at(px + r * std::cos(ang), py + r * std::sin(ang), hz, uni(0.f, 6.28f), uni(-1.2f, 1.2f));
            float lo[3] = { -0.06f, -0.025f, -0.002f }, hi[3] = { 0.06f, 0.025f, 0.002f };
            B.box(f, lo, hi, leaf_ob, 0.0f);     // weight 0: exactly 12 triangles per leaf (the min_u/min_v floor)
        }
    }
    // a venetian blind in front of the +y wall: 160 tilted slats
    { int ob = nobj++;
      for (int k = 0; k < 160; k++) {
          Frame f = frame_at(-4.0f, 8.7f, 0.8f + k * 0.03f, 0.f, 0.6f);
          float lo[3] = { 0.f, -0.02f, -0.001f }, hi[3] = { 5.0f, 0.02f, 0.001f };
          B.box(f, lo, hi, ob, 0.05f);
      } }

    std::vector<Face> lights;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) {
        float cx = -0.5f + 3.7f * i, cy = -1.5f + 5.0f * j;
        float o[3] = { cx - 1.0f, cy - 0.4f, 6.45f }, u[3] = { 0.f, 0.8f, 0.f }, v[3] = { 2.0f, 0.f, 0.f };
        add_face(lights, o, u, v, 0);
    }
    FILE *f = std::fopen((base + ".obj").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    std::fprintf(f, "# synthetic furnished conference room (evplp_synth_scene_ex style 1, seed %u)\nmtllib %s.mtl\n", seed, name);
    // tessellation floors (thin cylinders, leaves) take triangles outside the area-proportional share: fix-point on the budget
    size_t budget = (size_t)target_triangles;
    if (FILE *nul = std::fopen("/dev/null", "w")) {
        for (int it = 0; it < 6; it++) {
            size_t got = write_patches(nul, B.patches, budget, "obj");
            if (got == 0) break;
            double next = (double)budget * (double)target_triangles / (double)got;
            budget = (size_t)std::max(1.0, next);
        }
        std::fclose(nul);
    }
    size_t ntri = write_patches(f, B.patches, budget, "obj");
    std::fclose(f);
    f = std::fopen((base + ".mtl").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    Rng32 mr{ seed * 2654435761u + 12345u };
    // style 2: image textures (map_Kd, one map_Ks / map_Ns) on the room shell and the table, like the reference's living room
    // (scene/livingroom/*.jpg through map_Kd): PNG files written with the product's own writer, decoded by its own decoder
    const int ntex = 3;
    if (textured) {
        const int TW = 128, TH = 128;
        std::vector<float> img((size_t)TW * TH * 3);
        for (int tex = 0; tex < ntex; tex++) {
            Rng32 tr{ seed * 97u + (uint32_t)tex * 7919u + 1u };
            for (int y = 0; y < TH; y++) for (int x = 0; x < TW; x++) {
                float *q = &img[((size_t)y * TW + x) * 3];
                const bool check = (((x >> 3) + (y >> 3)) & 1) != 0;
                const float n = 0.85f + 0.15f * tr.next();
                if (tex == 0) { q[0] = (check ? 0.55f : 0.25f) * n; q[1] = (check ? 0.5f : 0.22f) * n; q[2] = (check ? 0.45f : 0.2f) * n; }          // checker floor / walls
                else if (tex == 1) { const float g = 0.35f + 0.25f * std::sin(0.37f * (float)x + 2.0f * std::sin(0.11f * (float)y)); q[0] = 1.6f * g * n; q[1] = 1.0f * g * n; q[2] = 0.55f * g * n; }   // wood grain
                else { const float g = check ? 0.30f : 0.02f; q[0] = q[1] = q[2] = g; }                                                                   // specular mask / exponent map
            }
            const std::string tp = base + "_tex" + std::to_string(tex) + ".png";
            if (evplp::save_image(tp.c_str(), TW, TH, img.data()) != EVPLP_OK) return EVPLP_ERR_IO;
        }
    }
    for (int i = 0; i < nobj; i++) {
        float kd[3] = { 0.2f + 0.6f * mr.next(), 0.2f + 0.6f * mr.next(), 0.2f + 0.6f * mr.next() };
        bool glossy = (i % 5) == 4;
        std::fprintf(f, "newmtl obj%d\nKd %.9g %.9g %.9g\nKs %.9g %.9g %.9g\nNs %.9g\n", i, kd[0], kd[1], kd[2],
                     glossy ? 0.2f : 0.f, glossy ? 0.2f : 0.f, glossy ? 0.2f : 0.f, glossy ? 20.f : 0.f);
        if (textured && i < 6) std::fprintf(f, "map_Kd %s_tex0.png\n", name);                               // room shell
        if (textured && i == 6) std::fprintf(f, "map_Kd %s_tex1.png\nmap_Ks %s_tex2.png\nNs 40\n", name, name);   // table: wood + specular mask
        std::fprintf(f, "\n");
    }
    std::fclose(f);
    f = std::fopen((base + "_lights.obj").c_str(), "w");
    if (!f) return EVPLP_ERR_IO;
    std::fprintf(f, "# area light: 8 ceiling quads, one mesh\n");
    write_faces(f, lights, 128, "", false);
    std::fclose(f);
    if (int rc = write_scene_json(base, name, res_x, res_y)) return rc;
    return (int)std::min<size_t>(ntri, 0x7fffffff);
}

extern "C" int evplp_synth_scene_ex(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed, int32_t res_x, int32_t res_y, int32_t style) {
    if (!out_dir || !name || target_triangles < 12 || res_x <= 0 || res_y <= 0 || style < 0 || style > 2) return EVPLP_ERR_INVALID;
    mkdir(out_dir, 0755);
    return style == 0 ? synth_boxes(out_dir, name, target_triangles, seed, res_x, res_y) : synth_furnished(out_dir, name, target_triangles, seed, res_x, res_y, style == 2);
}
extern "C" int evplp_synth_scene(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed, int32_t res_x, int32_t res_y) {
    return evplp_synth_scene_ex(out_dir, name, target_triangles, seed, res_x, res_y, 0);
}
