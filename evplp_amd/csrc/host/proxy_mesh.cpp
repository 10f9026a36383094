// The proxy mesh of the reference's photon splat (setupPhotonSplatIcosohedron, rt/rtcomphoton/rtcomphoton.h:632-644, drawn by
// runPhotonSplat :789-837): host-side preparation for the tile kernel of EVPLP_FOOTPRINT_PROXY (kernels.h ProxyDev).
//   * default_splat_proxy: the mesh used when the caller gave none -- sphere/icosphere.obj is a 2178-byte Git-LFS stub in the
//     reference, the size of an icosahedron subdivided once and pushed onto the unit sphere (42 vertices, 80 faces).
//   * build_proxy_slabs: checks that a mesh is closed and convex around the origin and turns its faces into slabs.
#include "../context.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace evplp {

// Icosahedron with its poles on the y axis (two rings of five vertices at y = -+1/sqrt 5, the upper one turned by 36 degrees), every
// face split in four at its edge midpoints, midpoints normalised.  Vertices 0-11 are the icosahedron's; faces come four per
// icosahedron face (three corner triangles, then the middle one).
void default_splat_proxy(std::vector<float> &verts, std::vector<int32_t> &tris) {
    const double pi = 3.14159265358979323846, h = 1.0 / std::sqrt(5.0), ring = 2.0 / std::sqrt(5.0);
    std::vector<std::array<double, 3>> v;
    v.push_back({ 0.0, -1.0, 0.0 });
    for (int k = 0; k < 5; k++) { const double a = 2.0 * pi * k / 5.0; v.push_back({ ring * std::cos(a), -h, ring * std::sin(a) }); }
    for (int k = 0; k < 5; k++) { const double a = 2.0 * pi * (k + 0.5) / 5.0; v.push_back({ ring * std::cos(a), h, ring * std::sin(a) }); }
    v.push_back({ 0.0, 1.0, 0.0 });
    std::vector<std::array<int, 3>> base;
    for (int k = 0; k < 5; k++) base.push_back({ 0, 1 + k, 1 + (k + 1) % 5 });                  // cap around the lower pole
    for (int k = 0; k < 5; k++) base.push_back({ 1 + k, 6 + k, 1 + (k + 1) % 5 });              // belt, pointing up
    for (int k = 0; k < 5; k++) base.push_back({ 6 + k, 6 + (k + 1) % 5, 1 + (k + 1) % 5 });    // belt, pointing down
    for (int k = 0; k < 5; k++) base.push_back({ 11, 6 + (k + 1) % 5, 6 + k });                 // cap around the upper pole
    std::map<std::pair<int, int>, int> mid;
    auto midpoint = [&](int a, int b) {
        const std::pair<int, int> key(std::min(a, b), std::max(a, b));
        auto it = mid.find(key);
        if (it != mid.end()) return it->second;
        std::array<double, 3> m = { 0.5 * (v[a][0] + v[b][0]), 0.5 * (v[a][1] + v[b][1]), 0.5 * (v[a][2] + v[b][2]) };
        const double l = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        for (double &c : m) c /= l;
        v.push_back(m);
        return mid[key] = (int)v.size() - 1;
    };
    tris.clear();
    for (const auto &t : base) {
        const int m0 = midpoint(t[0], t[1]), m1 = midpoint(t[1], t[2]), m2 = midpoint(t[2], t[0]);
        const int f[4][3] = { { t[0], m0, m2 }, { m0, t[1], m1 }, { m2, m1, t[2] }, { m0, m1, m2 } };
        for (const auto &q : f) for (int k = 0; k < 3; k++) tris.push_back(q[k]);
    }
    verts.clear();
    for (const auto &p : v) for (double c : p) verts.push_back((float)c);
}

// Faces -> planes n . x <= h (n outward: h > 0 with the origin inside), coplanar faces once, opposite planes paired into slabs.
// Returns false and a reason for a mesh the entry / exit rule of the kernel does not describe.
bool build_proxy_slabs(const float *verts, int32_t nverts, const int32_t *tris, int32_t ntris, ProxyHost *out, std::string *why) {
    auto fail = [&](const std::string &m) { if (why) *why = m; return false; };
    if (!verts || !tris || nverts < 4 || ntris < 4) return fail("a closed mesh has at least 4 vertices and 4 triangles");
    for (int64_t i = 0; i < (int64_t)ntris * 3; i++) if (tris[i] < 0 || tris[i] >= nverts) return fail("vertex index out of range");
    // vertices welded by position: an OBJ exporter may repeat them per face
    std::map<std::array<float, 3>, int> weld; std::vector<int> id((size_t)nverts); std::vector<std::array<double, 3>> pos;
    for (int i = 0; i < nverts; i++) {
        const std::array<float, 3> key = { verts[3 * i], verts[3 * i + 1], verts[3 * i + 2] };
        if (!std::isfinite(key[0]) || !std::isfinite(key[1]) || !std::isfinite(key[2])) return fail("non-finite vertex");
        auto it = weld.find(key);
        if (it == weld.end()) { it = weld.emplace(key, (int)pos.size()).first; pos.push_back({ (double)key[0], (double)key[1], (double)key[2] }); }
        id[(size_t)i] = it->second;
    }
    double rout = 0.0;
    for (const auto &p : pos) rout = std::max(rout, std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
    if (!(rout > 0.0)) return fail("degenerate mesh");
    double cen[3] = { 0.0, 0.0, 0.0 };                                      // (inside any convex mesh: tells outward from inward)
    for (const auto &p : pos) for (int k = 0; k < 3; k++) cen[k] += p[k] / (double)pos.size();
    struct Plane { double n[3], h; };
    std::vector<Plane> planes;
    std::map<std::pair<int, int>, int> edges;
    const double tol = 1.0e-6 * rout;
    for (int t = 0; t < ntris; t++) {
        const int a = id[(size_t)tris[3 * t]], b = id[(size_t)tris[3 * t + 1]], c = id[(size_t)tris[3 * t + 2]];
        if (a == b || b == c || a == c) return fail("degenerate triangle " + std::to_string(t));
        for (const auto &e : { std::make_pair(a, b), std::make_pair(b, c), std::make_pair(c, a) }) edges[{ std::min(e.first, e.second), std::max(e.first, e.second) }]++;
        const double u[3] = { pos[b][0] - pos[a][0], pos[b][1] - pos[a][1], pos[b][2] - pos[a][2] }, w[3] = { pos[c][0] - pos[a][0], pos[c][1] - pos[a][1], pos[c][2] - pos[a][2] };
        double n[3] = { u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0] };
        const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(l > 1.0e-12 * rout * rout)) return fail("degenerate triangle " + std::to_string(t));
        for (double &x : n) x /= l;
        if (n[0] * (pos[a][0] - cen[0]) + n[1] * (pos[a][1] - cen[1]) + n[2] * (pos[a][2] - cen[2]) < 0.0) for (double &x : n) x = -x;   // (any winding)
        const double h = n[0] * pos[a][0] + n[1] * pos[a][1] + n[2] * pos[a][2];
        if (!(h > tol)) return fail("the origin is not strictly inside the mesh (face " + std::to_string(t) + ")");
        bool seen = false;
        for (const Plane &p : planes) if (std::fabs(p.h - h) <= tol && p.n[0] * n[0] + p.n[1] * n[1] + p.n[2] * n[2] >= 1.0 - 1.0e-10) { seen = true; break; }
        if (!seen) planes.push_back({ { n[0], n[1], n[2] }, h });
    }
    for (const auto &e : edges) if (e.second != 2) return fail("the mesh is not closed (an edge with " + std::to_string(e.second) + " faces)");
    for (size_t k = 0; k < planes.size(); k++)
        for (const auto &p : pos)
            if (planes[k].n[0] * p[0] + planes[k].n[1] * p[1] + planes[k].n[2] * p[2] > planes[k].h + tol) return fail("the mesh is not convex");
    if ((int)planes.size() > kMaxProxySlabs) return fail("more than " + std::to_string(kMaxProxySlabs) + " distinct face planes");
    out->slabs.clear(); out->hm.clear();
    std::vector<bool> used(planes.size(), false);
    double rin = planes[0].h;
    for (size_t i = 0; i < planes.size(); i++) {
        rin = std::min(rin, planes[i].h);
        if (used[i]) continue;
        used[i] = true;
        float hm = kProxyOpen;
        for (size_t j = i + 1; j < planes.size(); j++)
            if (!used[j] && planes[i].n[0] * planes[j].n[0] + planes[i].n[1] * planes[j].n[1] + planes[i].n[2] * planes[j].n[2] <= -1.0 + 1.0e-10) { used[j] = true; hm = (float)planes[j].h; break; }
        out->slabs.push_back({ (float)planes[i].n[0], (float)planes[i].n[1], (float)planes[i].n[2], (float)planes[i].h });
        out->hm.push_back(hm);
    }
    out->planes = (int32_t)planes.size();
    out->rin = (float)rin; out->rout = (float)rout;
    return true;
}

} // namespace evplp
