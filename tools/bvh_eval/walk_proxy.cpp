// Developer tool (CPU only): replay of the VPL gather's packet walk (device_common.hpp::occluded_wave) on the REAL walk population
// of the bench configuration -- G-buffer tiles and usable VPL records dumped on the GPU by tools/dump_proxy_data.py, the scene rebuilt
// here by the deterministic generator, the tree by the product's host builder (bvh_build.cpp, linked in).  Used to price changes of
// the walk's entry (per-VPL entry lists, plane culls) in node visits / triangle pairs per walk before they are written for the device.
//   g++ -O2 -std=c++17 -I evplp_amd/csrc -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bvh_eval/walk_proxy.cpp evplp_amd/csrc/bvh_build.cpp -o build/walk_proxy
//   build/walk_proxy proxy.bin scene.obj scene_lights.obj [walks] [mode] [flags...]
#include "evplp_types.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
using namespace evplp;
struct V { float x, y, z; };
static V operator-(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V operator+(V a, V b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V operator*(V a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

static bool tri_hit(const TriFlat &t, V o, V d, float tmin, float tmax) {
    float den = t.n[0] * d.x + t.n[1] * d.y + t.n[2] * d.z;
    float inv = 1.0f / den;
    float qx = (t.p0[0] - o.x) * inv, qy = (t.p0[1] - o.y) * inv, qz = (t.p0[2] - o.z) * inv;
    float ix = d.y * qz - d.z * qy, iy = d.z * qx - d.x * qz, iz = d.x * qy - d.y * qx;
    float beta = ix * t.e1[0] + iy * t.e1[1] + iz * t.e1[2], gamma = ix * t.e0[0] + iy * t.e0[1] + iz * t.e0[2];
    float tt = t.n[0] * qx + t.n[1] * qy + t.n[2] * qz;
    return tt < tmax && tt > tmin && beta >= 0.f && gamma >= 0.f && beta + gamma <= 1.f;
}
static float clamp01(float x) { return x < 0.f ? 0.f : x > 1.f ? 1.f : x; }
static float srcp(float d) { float a = std::fabs(d) < 1e-30f ? std::copysign(1e-30f, d) : d; return 1.0f / a; }

static void load_obj(const char *path, std::vector<float> &verts) {
    std::vector<V> vs; FILE *f = std::fopen(path, "r"); if (!f) { perror(path); exit(1); }
    char line[512];
    while (std::fgets(line, sizeof line, f)) {
        if (line[0] == 'v' && line[1] == ' ') { V v; std::sscanf(line + 2, "%f %f %f", &v.x, &v.y, &v.z); vs.push_back(v); }
        else if (line[0] == 'f' && line[1] == ' ') {
            int idx[8], n = 0; char *p = line + 2;
            while (*p && n < 8) { while (*p == ' ') p++; if (!*p || *p == '\n') break; idx[n++] = atoi(p); while (*p && *p != ' ') p++; }
            for (int k = 1; k + 1 < n; k++) { int t[3] = { idx[0], idx[k], idx[k + 1] }; for (int j = 0; j < 3; j++) { V v = vs[t[j] - 1]; verts.push_back(v.x); verts.push_back(v.y); verts.push_back(v.z); } }
        }
    }
    std::fclose(f);
}

struct Box { float c[3], h[3]; int32_t ref; };
static Box child_box(const BvhNode &n, int ch) { Box b; for (int k = 0; k < 3; k++) { b.c[k] = n.ctr[k][ch]; b.h[k] = n.hal[k][ch]; } b.ref = ch == 0 ? n.c0 : n.c1; return b; }
static bool holds(const Box &b, V p, float m) { return std::fabs(p.x - b.c[0]) <= b.h[0] + m && std::fabs(p.y - b.c[1]) <= b.h[1] + m && std::fabs(p.z - b.c[2]) <= b.h[2] + m; }

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: walk_proxy proxy.bin scene.obj lights.obj [walks] [mode] [cull] [order]\n"); return 2; }
    const int nwalks = argc > 4 ? atoi(argv[4]) : 20000;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;        // 0 root walk; 1 per-VPL entry list (siblings of the nodes that hold the VPL, two per synthetic node)
    const int cull = argc > 6 ? atoi(argv[6]) : 0;        // 1: drop siblings wholly behind the VPL's plane
    const int order = argc > 7 ? atoi(argv[7]) : 0;       // entry list: 0 = deepest first (popped first), 1 = top first, 2 = sorted by distance (near first)
    std::vector<float> verts; load_obj(argv[2], verts); if (std::strcmp(argv[3], "-") != 0) load_obj(argv[3], verts);
    int ntri = (int)(verts.size() / 9);
    BvhBuild bb; build_bvh(verts.data(), ntri, 1, &bb);
    std::printf("tris %d nodes %d leaves %d depth %d build %.0f ms\n", ntri, bb.nnodes, bb.nleaves, bb.depth, bb.build_ms);
    std::vector<int> depth((size_t)bb.nnodes, 0);
    for (int i = 0; i < bb.nnodes; i++) { const BvhNode &n = bb.nodes[i]; if (n.c0 >= 0) depth[n.c0] = depth[i] + 1; if (n.c1 >= 0) depth[n.c1] = depth[i] + 1; }   // pre-order: children after parents
    FILE *f = std::fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[2]; if (std::fread(hdr, 4, 2, f) != 2) return 1;
    const int ntiles = hdr[0], nvpl = hdr[1];
    std::vector<float> tiles((size_t)ntiles * 64 * 7), vpls((size_t)nvpl * 6);
    if (std::fread(tiles.data(), 4, tiles.size(), f) != tiles.size() || std::fread(vpls.data(), 4, vpls.size(), f) != vpls.size()) return 1;
    std::fclose(f);
    std::printf("tiles %d vpls %d mode %d cull %d order %d\n", ntiles, nvpl, mode, cull, order);

    // per-VPL entry lists (mode 1)
    std::vector<std::vector<BvhNode>> syn((size_t)nvpl);
    std::vector<double> list_len;
    double sib_total = 0, sib_culled = 0;
    if (mode == 1) {
        for (int v = 0; v < nvpl; v++) {
            const V vp = { vpls[6 * v], vpls[6 * v + 1], vpls[6 * v + 2] }, vn = { vpls[6 * v + 3], vpls[6 * v + 4], vpls[6 * v + 5] };
            std::vector<Box> sib;                   // in discovery order (top first)
            std::vector<int32_t> todo = { 0 };
            while (!todo.empty()) {
                const int32_t at = todo.back(); todo.pop_back();
                const BvhNode &n = bb.nodes[at];
                for (int ch = 0; ch < 2; ch++) {
                    const Box b = child_box(n, ch);
                    if (b.ref == kNoChild) continue;
                    if (b.ref >= 0 && holds(b, vp, 0.f)) { todo.push_back(b.ref); continue; }       // an inner node that holds the VPL: its children are tested instead
                    sib_total++;
                    if (cull) {
                        // wholly behind the VPL's plane (every active segment leaves the VPL into the half space n . (x - P) > 0)
                        const float top = vn.x * (b.c[0] - vp.x) + vn.y * (b.c[1] - vp.y) + vn.z * (b.c[2] - vp.z) + std::fabs(vn.x) * b.h[0] + std::fabs(vn.y) * b.h[1] + std::fabs(vn.z) * b.h[2];
                        if (top < 0.f) { sib_culled++; continue; }
                    }
                    sib.push_back(b);
                }
            }
            if (order == 2) {
                auto dist = [&](const Box &b) { float d = 0; const float p[3] = { vp.x, vp.y, vp.z }; for (int k = 0; k < 3; k++) { float e = std::max(std::fabs(p[k] - b.c[k]) - b.h[k], 0.f); d += e * e; } return d; };
                std::sort(sib.begin(), sib.end(), [&](const Box &a, const Box &b) { return dist(a) > dist(b); });   // far first = bottom of the stack
            } else if (order == 1) std::reverse(sib.begin(), sib.end());
            for (size_t k = 0; k < sib.size(); k += 2) {
                BvhNode m; std::memset(&m, 0, sizeof m);
                for (int a = 0; a < 3; a++) { m.ctr[a][0] = sib[k].c[a]; m.hal[a][0] = sib[k].h[a]; }
                m.c0 = sib[k].ref;
                if (k + 1 < sib.size()) { for (int a = 0; a < 3; a++) { m.ctr[a][1] = sib[k + 1].c[a]; m.hal[a][1] = sib[k + 1].h[a]; } m.c1 = sib[k + 1].ref; }
                else { for (int a = 0; a < 3; a++) { m.ctr[a][1] = 0; m.hal[a][1] = -3e38f; } m.c1 = kNoChild; }
                syn[v].push_back(m);
            }
            list_len.push_back((double)syn[v].size());
        }
        double s = 0, mx = 0; for (double l : list_len) { s += l; mx = std::max(mx, l); }
        std::printf("entry lists: %.1f synthetic nodes per VPL (max %.0f), siblings %.1f per VPL, %.1f %% culled by the VPL's plane\n", s / nvpl, mx, sib_total / nvpl, 100.0 * sib_culled / std::max(sib_total, 1.0));
    }

    std::mt19937 rng(4242);
    double S_walks = 0, S_nodes = 0, S_leaves = 0, S_pairs = 0, S_empty = 0, S_empty_nodes = 0, S_full = 0, S_syn = 0;
    double S_holdv = 0, S_holdt_all = 0, S_holdt_any = 0, S_holdboth = 0, S_none = 0;
    std::vector<double> by_depth(80, 0.0);
    std::vector<int32_t> stack(512);
    std::vector<int> cand_hist, fvis_hist;
    double C_n[4] = {}, C_nodes[4] = {}, C_pairs[4] = {}, C_alive0[4] = {}, C_alive1[4] = {}, C_after[4] = {};
    int done = 0;
    while (done < nwalks) {
        const int ti = (int)(rng() % (unsigned)ntiles), vi = (int)(rng() % (unsigned)nvpl);
        const float *T = &tiles[(size_t)ti * 64 * 7];
        const V vp = { vpls[6 * vi], vpls[6 * vi + 1], vpls[6 * vi + 2] }, vn = { vpls[6 * vi + 3], vpls[6 * vi + 4], vpls[6 * vi + 5] };
        V d[64], pp[64]; bool alive[64]; int nalive = 0;
        float ivx[64], ivy[64], ivz[64], nox[64], noy[64], noz[64];
        const float tmin = 1e-4f, tmax = 1.f - 1e-4f, ku = 1.0f / (tmax - tmin);
        for (int l = 0; l < 64; l++) {
            const V p1 = { T[l * 7], T[l * 7 + 1], T[l * 7 + 2] }, pn = { T[l * 7 + 4], T[l * 7 + 5], T[l * 7 + 6] };
            pp[l] = p1;
            const V v12 = vp - p1;
            const float c1 = std::max(dot(pn, v12), 0.f), c2 = std::max(-dot(vn, v12), 0.f);
            alive[l] = T[l * 7 + 3] != 0.f && !(c1 * c2 <= 0.f); if (alive[l]) nalive++;
            d[l] = p1 - vp;
            const float i0x = srcp(d[l].x), i0y = srcp(d[l].y), i0z = srcp(d[l].z);
            ivx[l] = i0x * ku; ivy[l] = i0y * ku; ivz[l] = i0z * ku;
            nox[l] = alive[l] ? (-(vp.x * i0x) - tmin) * ku : INFINITY; noy[l] = alive[l] ? (-(vp.y * i0y) - tmin) * ku : INFINITY; noz[l] = alive[l] ? (-(vp.z * i0z) - tmin) * ku : INFINITY;
        }
        if (nalive == 0) continue;     // (the kernel skips these before the walk)
        done++;
        if (mode == 2) {
            // ---- phase 1: one frustum per (tile, VPL) walks the tree alone (on the device: lane = VPL); candidates = the leaves it reaches
            const int planes_on = cull;      // flags: 1 side planes, 2 end-point planes (VPL normal / mean pixel normal), 4 slabs along m, u, v
            V cen = { 0, 0, 0 }; int na = 0; V pnm = { 0, 0, 0 };
            for (int l = 0; l < 64; l++) if (alive[l]) { cen = cen + pp[l]; na++; pnm = pnm + V{ T[l * 7 + 4], T[l * 7 + 5], T[l * 7 + 6] }; }
            cen = cen * (1.0f / na);
            V m = cen - vp; { float L = std::sqrt(dot(m, m)); m = m * (1.0f / L); }
            V ax = std::fabs(m.x) < 0.6f ? V{ 1, 0, 0 } : V{ 0, 1, 0 };
            V u = { m.y * ax.z - m.z * ax.y, m.z * ax.x - m.x * ax.z, m.x * ax.y - m.y * ax.x }; { float L = std::sqrt(dot(u, u)); u = u * (1.0f / L); }
            V w = { m.y * u.z - m.z * u.y, m.z * u.x - m.x * u.z, m.x * u.y - m.y * u.x };
            { float L = std::sqrt(dot(pnm, pnm)); if (L > 0) pnm = pnm * (1.0f / L); }
            float amin = 1e30f, amax = -1e30f, bmin = 1e30f, bmax = -1e30f; bool wide = false;
            float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
            float s_vpl = 1e30f, s_tile = 1e30f, mlo = 1e30f, mhi = -1e30f, ulo = 1e30f, uhi = -1e30f, wlo = 1e30f, whi = -1e30f;
            for (int l = 0; l < 64; l++) if (alive[l]) {
                const float dw = dot(m, d[l]), dl = std::sqrt(dot(d[l], d[l]));
                if (dw <= 0.05f * dl) wide = true;
                else { const float a = dot(u, d[l]) / dw, b = dot(w, d[l]) / dw; amin = std::min(amin, a); amax = std::max(amax, a); bmin = std::min(bmin, b); bmax = std::max(bmax, b); }
                const V e0 = vp + d[l] * tmin, e1 = vp + d[l] * tmax;
                const float q0[3] = { e0.x, e0.y, e0.z }, q1[3] = { e1.x, e1.y, e1.z };
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], std::min(q0[k], q1[k])); hi[k] = std::max(hi[k], std::max(q0[k], q1[k])); }
                s_vpl = std::min(s_vpl, dot(vn, d[l]) * tmin);                       // n_v . (x - P) >= this on every segment
                s_tile = std::min(s_tile, std::min(dot(pnm, e0), dot(pnm, e1)));     // pnm . x >= this on every segment
                mlo = std::min(mlo, std::min(dot(m, e0), dot(m, e1))); mhi = std::max(mhi, std::max(dot(m, e0), dot(m, e1)));
                ulo = std::min(ulo, std::min(dot(u, e0), dot(u, e1))); uhi = std::max(uhi, std::max(dot(u, e0), dot(u, e1)));
                wlo = std::min(wlo, std::min(dot(w, e0), dot(w, e1))); whi = std::max(whi, std::max(dot(w, e0), dot(w, e1)));
            }
            const float eps = 1e-5f;
            V pl[4] = { u - m * amin, m * amax - u, w - m * bmin, m * bmax - w };
            auto outside = [&](const Box &b) {
                const float c[3] = { b.c[0], b.c[1], b.c[2] };
                for (int k = 0; k < 3; k++) if (c[k] - b.h[k] > hi[k] + eps || c[k] + b.h[k] < lo[k] - eps) return true;
                const V cc = { c[0], c[1], c[2] }, rel = cc - vp;
                auto ext = [&](V n) { return std::fabs(n.x) * b.h[0] + std::fabs(n.y) * b.h[1] + std::fabs(n.z) * b.h[2]; };
                if ((planes_on & 1) && !wide) for (int k = 0; k < 4; k++) if (dot(pl[k], rel) + ext(pl[k]) < -eps) return true;
                if (planes_on & 2) {
                    if (dot(vn, rel) + ext(vn) < s_vpl - eps) return true;
                    if (dot(pnm, cc) + ext(pnm) < s_tile - eps) return true;
                }
                if (planes_on & 4) {
                    if (dot(m, cc) + ext(m) < mlo - eps || dot(m, cc) - ext(m) > mhi + eps) return true;
                    if (dot(u, cc) + ext(u) < ulo - eps || dot(u, cc) - ext(u) > uhi + eps) return true;
                    if (dot(w, cc) + ext(w) < wlo - eps || dot(w, cc) - ext(w) > whi + eps) return true;
                }
                return false;
            };
            std::vector<Box> cand; unsigned fvis = 0; int spf = 0; std::vector<int32_t> fst(256);
            fst[spf++] = 0;
            while (spf > 0) {
                const int32_t at = fst[--spf]; fvis++;
                const BvhNode &n = bb.nodes[at];
                for (int ch = 0; ch < 2; ch++) { const Box b = child_box(n, ch); if (b.ref == kNoChild || outside(b)) continue; if (b.ref >= 0) fst[spf++] = b.ref; else cand.push_back(b); }
            }
            // ---- phase 2: the wave tests the candidate leaves (64 rays): slab test of the leaf's box, then its triangle pairs
            unsigned entered = 0, prs = 0, tested = 0;
            for (const Box &b : cand) {
                if (nalive == 0) break;
                tested++;
                int hitn = 0;
                for (int l = 0; l < 64; l++) {
                    float axx = b.c[0] * ivx[l] + nox[l], ay = b.c[1] * ivy[l] + noy[l], az = b.c[2] * ivz[l] + noz[l];
                    float bx = b.h[0] * std::fabs(ivx[l]), by = b.h[1] * std::fabs(ivy[l]), bz = b.h[2] * std::fabs(ivz[l]);
                    float tn = clamp01(std::max(std::max(axx - bx, ay - by), az - bz)), tf = clamp01(std::min(std::min(axx + bx, ay + by), az + bz));
                    if (tn < tf) hitn++;
                }
                if (!hitn) continue;
                entered++;
                const uint32_t id = (uint32_t)~b.ref, block = id >> 2, cnt = (id & 3u) + 1u; prs += cnt > 2 ? 2 : 1;
                for (int l = 0; l < 64; l++) if (alive[l]) {
                    bool h = false;
                    for (uint32_t k = 0; k < cnt; k++) if (tri_hit(bb.tri_flat[block * 4 + k], vp, d[l], tmin, tmax)) h = true;
                    if (h) { alive[l] = false; nalive--; nox[l] = noy[l] = noz[l] = INFINITY; }
                }
            }
            cand_hist.push_back((int)cand.size()); fvis_hist.push_back((int)fvis);
            S_walks++; S_nodes += fvis; S_syn += (double)cand.size(); S_leaves += entered; S_pairs += prs; S_none += tested;
            if (cand.empty()) S_empty++; if (entered == 0) S_empty_nodes++;
            if (nalive == 0) S_full++;
            if (wide) S_holdv++;
            continue;
        }
        int sp = 0; int32_t cur = 0; unsigned nodes = 0, leaves = 0, pairs = 0, nodes_at_last_hit = 0;
        const int nalive0 = nalive;
        const std::vector<BvhNode> *sy = nullptr;
        if (mode == 1) {
            sy = &syn[vi];
            for (size_t k = 0; k < sy->size(); k++) stack[sp++] = bb.nnodes + (int32_t)k;
            if (sp == 0) cur = kNoChild; else cur = stack[--sp];
        }
        for (;;) {
            while (cur >= 0) {
                const bool is_syn = cur >= bb.nnodes;
                const BvhNode &n = is_syn ? (*sy)[cur - bb.nnodes] : bb.nodes[cur]; nodes++;
                if (is_syn) S_syn++;
                else {
                    by_depth[std::min(depth[cur], 79)]++;
                    bool hv = false, ht_all = false, ht_any = false;
                    for (int ch = 0; ch < 2; ch++) {
                        const Box b = child_box(n, ch); if (b.ref == kNoChild) continue;
                        if (holds(b, vp, 0.f)) hv = true;
                        int cnt = 0, tot = 0; for (int l = 0; l < 64; l++) if (alive[l]) { tot++; if (holds(b, pp[l], 0.f)) cnt++; }
                        if (cnt > 0) ht_any = true; if (cnt == tot) ht_all = true;
                    }
                    if (hv && ht_any) S_holdboth++; else if (hv) S_holdv++; else if (ht_all) S_holdt_all++; else if (ht_any) S_holdt_any++; else S_none++;
                }
                int p0 = 0, p1 = 0;
                for (int l = 0; l < 64; l++) for (int ch = 0; ch < 2; ch++) {
                    float ax = n.ctr[0][ch] * ivx[l] + nox[l], ay = n.ctr[1][ch] * ivy[l] + noy[l], az = n.ctr[2][ch] * ivz[l] + noz[l];
                    float bx = n.hal[0][ch] * std::fabs(ivx[l]), by = n.hal[1][ch] * std::fabs(ivy[l]), bz = n.hal[2][ch] * std::fabs(ivz[l]);
                    float tn = clamp01(std::max(std::max(ax - bx, ay - by), az - bz)), tf = clamp01(std::min(std::min(ax + bx, ay + by), az + bz));
                    if (tn < tf) { if (ch == 0) p0++; else p1++; }
                }
                if (p0 == 0 && p1 == 0) { cur = kNoChild; break; }
                if (p0 == 0) { cur = n.c1; continue; }
                if (p1 == 0) { cur = n.c0; continue; }
                const bool first0 = p0 >= p1;
                stack[sp++] = first0 ? n.c1 : n.c0; cur = first0 ? n.c0 : n.c1;
            }
            if (cur != kNoChild) {
                const uint32_t id = (uint32_t)~cur, block = id >> 2, cnt = (id & 3u) + 1u; leaves++; pairs += cnt > 2 ? 2 : 1;
                for (int l = 0; l < 64; l++) if (alive[l]) {
                    bool h = false;
                    for (uint32_t k = 0; k < cnt; k++) if (tri_hit(bb.tri_flat[block * 4 + k], vp, d[l], tmin, tmax)) h = true;
                    if (h) { alive[l] = false; nalive--; nox[l] = noy[l] = noz[l] = INFINITY; nodes_at_last_hit = nodes; }
                }
                if (nalive == 0) break;
            }
            if (sp == 0) break;
            cur = stack[--sp];
        }
        {
            const int cls = leaves == 0 ? 0 : nalive == 0 ? 1 : nalive == nalive0 ? 2 : 3;      // empty / fully occluded / leaves touched, nothing hit / partially occluded
            C_n[cls]++; C_nodes[cls] += nodes; C_pairs[cls] += pairs; C_alive0[cls] += nalive0; C_alive1[cls] += nalive; C_after[cls] += nodes - nodes_at_last_hit;
        }
        S_walks++; S_nodes += nodes; S_leaves += leaves; S_pairs += pairs;
        if (leaves == 0) { S_empty++; S_empty_nodes += nodes; }
        if (nalive == 0) S_full++;
    }
    if (mode == 2) {
        std::printf("frustum walk: %.2f node visits / packet, %.2f candidate leaves / packet (%.3f of the packets have none); wave phase: %.2f leaf boxes tested, %.2f entered, %.2f triangle pairs / packet; no leaf entered %.3f; fully occluded %.3f; wide packets %.4f\n",
                    S_nodes / S_walks, S_syn / S_walks, S_empty / S_walks, S_none / S_walks, S_leaves / S_walks, S_pairs / S_walks, S_empty_nodes / S_walks, S_full / S_walks, S_holdv / S_walks);
        std::sort(cand_hist.begin(), cand_hist.end()); std::sort(fvis_hist.begin(), fvis_hist.end());
        auto pc = [&](std::vector<int> &h, double q) { return h[(size_t)(q * (h.size() - 1))]; };
        std::printf("   candidates percentiles 50/75/90/95/99/max: %d %d %d %d %d %d;  frustum visits: %d %d %d %d %d %d\n", pc(cand_hist, .5), pc(cand_hist, .75), pc(cand_hist, .9), pc(cand_hist, .95), pc(cand_hist, .99), cand_hist.back(),
                    pc(fvis_hist, .5), pc(fvis_hist, .75), pc(fvis_hist, .9), pc(fvis_hist, .95), pc(fvis_hist, .99), fvis_hist.back());
        return 0;
    }
    std::printf("walks %.0f  nodes/walk %.2f (synthetic %.2f)  leaves/walk %.2f  pairs/walk %.2f  empty %.3f (nodes %.2f)  fully occluded %.3f  est VALU/walk %.0f\n",
                S_walks, S_nodes / S_walks, S_syn / S_walks, S_leaves / S_walks, S_pairs / S_walks, S_empty / S_walks, S_empty_nodes / std::max(S_empty, 1.0), S_full / S_walks,
                15.0 * S_nodes / S_walks + 46.0 * S_pairs / S_walks);
    std::printf("   real-node visits by what a child box holds: VPL+tile point %.2f  VPL only %.2f  all live tile points %.2f  some tile points %.2f  neither %.2f  /walk\n",
                S_holdboth / S_walks, S_holdv / S_walks, S_holdt_all / S_walks, S_holdt_any / S_walks, S_none / S_walks);
    { const char *nm[4] = { "empty", "fully occluded", "leaves touched, no hit", "partially occluded" };
      for (int c = 0; c < 4; c++) std::printf("   %-24s %.3f of walks: %.1f visits, %.1f pairs, %.1f lanes alive at start, %.1f at end, %.1f visits after the last hit; share of all visits %.3f, of all pairs %.3f\n", nm[c], C_n[c] / S_walks,
                  C_nodes[c] / std::max(C_n[c], 1.0), C_pairs[c] / std::max(C_n[c], 1.0), C_alive0[c] / std::max(C_n[c], 1.0), C_alive1[c] / std::max(C_n[c], 1.0), C_after[c] / std::max(C_n[c], 1.0), C_nodes[c] / S_nodes, C_pairs[c] / std::max(S_pairs, 1.0)); }
    std::printf("   visits by depth:");
    for (int k = 0; k < 40; k++) std::printf(" %.2f", by_depth[k] / S_walks);
    std::printf("\n");
    return 0;
}
