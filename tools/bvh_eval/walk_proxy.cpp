// Developer tool (CPU only): replay of the VPL gather's packet walk (device_common.hpp::occluded_wave) on the REAL walk population
// of the bench configuration -- G-buffer tiles and usable VPL records dumped on the GPU by tools/dump_proxy_data.py, the scene rebuilt
// here by the deterministic generator, the tree by the product's host builder (bvh_build.cpp, linked in).  Used to price changes of
// the walk's entry in node visits / triangle pairs per walk before they are written for the device:
//   mode 0  the walk from the root (the kernel as it is)
//   mode 1  per-VPL entry lists: the subtrees hanging off the nodes that hold the VPL, two per synthetic node
//   mode 2  one frustum per (tile, VPL) walks the tree down to the leaves (how loose is it against the 64 rays?)
//   mode 3  entry cuts per (group of G x G tiles, VPL): a frustum around the group's segments descends the tree to a cut; the
//           group's tile walks start from the cut (two cut nodes per synthetic node) instead of from the root
//   g++ -O2 -std=c++17 -I evplp_amd/csrc -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bvh_eval/walk_proxy.cpp evplp_amd/csrc/bvh_build.cpp -o build/walk_proxy
//   build/walk_proxy proxy.bin scene.obj scene_lights.obj [walks] [mode] [a] [b] [c] [d]
#include "evplp_types.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>
using namespace evplp;
struct V { float x, y, z; };
static V operator-(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V operator+(V a, V b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V operator*(V a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static V norm(V a) { float l = std::sqrt(dot(a, a)); return l > 0 ? a * (1.0f / l) : a; }

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
static std::vector<BvhNode> pair_up(const std::vector<Box> &sib) {
    std::vector<BvhNode> out;
    for (size_t k = 0; k < sib.size(); k += 2) {
        BvhNode m; std::memset(&m, 0, sizeof m);
        for (int a = 0; a < 3; a++) { m.ctr[a][0] = sib[k].c[a]; m.hal[a][0] = sib[k].h[a]; }
        m.c0 = sib[k].ref;
        if (k + 1 < sib.size()) { for (int a = 0; a < 3; a++) { m.ctr[a][1] = sib[k + 1].c[a]; m.hal[a][1] = sib[k + 1].h[a]; } m.c1 = sib[k + 1].ref; }
        else { for (int a = 0; a < 3; a++) { m.ctr[a][1] = 0; m.hal[a][1] = -3e38f; } m.c1 = kNoChild; }
        out.push_back(m);
    }
    return out;
}

static BvhBuild bb;
static std::vector<int> depth_of;
static const float kTmin = 1e-4f, kTmax = 1.f - 1e-4f;

// the 64 segments of one (tile, VPL) packet
struct Packet {
    V vp, vn, d[64], pp[64]; bool alive[64]; int nalive = 0;
    float ivx[64], ivy[64], ivz[64], nox[64], noy[64], noz[64];
    void setup(const float *T, const float *vpl) {
        vp = { vpl[0], vpl[1], vpl[2] }; vn = { vpl[3], vpl[4], vpl[5] }; nalive = 0;
        const float ku = 1.0f / (kTmax - kTmin);
        for (int l = 0; l < 64; l++) {
            const V p1 = { T[l * 7], T[l * 7 + 1], T[l * 7 + 2] }, pn = { T[l * 7 + 4], T[l * 7 + 5], T[l * 7 + 6] };
            pp[l] = p1;
            const V v12 = vp - p1;
            const float c1 = std::max(dot(pn, v12), 0.f), c2 = std::max(-dot(vn, v12), 0.f);
            alive[l] = T[l * 7 + 3] != 0.f && !(c1 * c2 <= 0.f); if (alive[l]) nalive++;
            d[l] = p1 - vp;
            const float i0x = srcp(d[l].x), i0y = srcp(d[l].y), i0z = srcp(d[l].z);
            ivx[l] = i0x * ku; ivy[l] = i0y * ku; ivz[l] = i0z * ku;
            nox[l] = alive[l] ? (-(vp.x * i0x) - kTmin) * ku : INFINITY; noy[l] = alive[l] ? (-(vp.y * i0y) - kTmin) * ku : INFINITY; noz[l] = alive[l] ? (-(vp.z * i0z) - kTmin) * ku : INFINITY;
        }
    }
    int enters(const float *c, const float *h) const {      // lanes whose clamped slab interval of the box is not empty
        int n = 0;
        for (int l = 0; l < 64; l++) {
            float ax = c[0] * ivx[l] + nox[l], ay = c[1] * ivy[l] + noy[l], az = c[2] * ivz[l] + noz[l];
            float bx = h[0] * std::fabs(ivx[l]), by = h[1] * std::fabs(ivy[l]), bz = h[2] * std::fabs(ivz[l]);
            float tn = clamp01(std::max(std::max(ax - bx, ay - by), az - bz)), tf = clamp01(std::min(std::min(ax + bx, ay + by), az + bz));
            if (tn < tf) n++;
        }
        return n;
    }
    void test_leaf(int32_t ref) {
        const uint32_t id = (uint32_t)~ref, block = id >> 2, cnt = (id & 3u) + 1u;
        for (int l = 0; l < 64; l++) if (alive[l]) {
            bool h = false;
            for (uint32_t k = 0; k < cnt; k++) if (tri_hit(bb.tri_flat[block * 4 + k], vp, d[l], kTmin, kTmax)) h = true;
            if (h) { alive[l] = false; nalive--; nox[l] = noy[l] = noz[l] = INFINITY; }
        }
    }
};
struct WalkOut { unsigned nodes = 0, syn = 0, leaves = 0, pairs = 0, after = 0; int alive0 = 0, alive1 = 0; };
static std::vector<double> g_by_depth(80, 0.0);
static double g_cls[5] = {};
// occluded_wave's control flow; `init` (may be null) = synthetic nodes preloaded on the stack (last = popped first)
static WalkOut walk(Packet &P, const std::vector<BvhNode> *init, bool classify) {
    WalkOut o; o.alive0 = P.nalive;
    static std::vector<int32_t> stack(1024);
    int sp = 0; int32_t cur = 0; unsigned at_last_hit = 0;
    if (init) { for (size_t k = 0; k < init->size(); k++) stack[sp++] = bb.nnodes + (int32_t)k; if (sp == 0) cur = kNoChild; else cur = stack[--sp]; }
    for (;;) {
        while (cur >= 0) {
            const bool is_syn = cur >= bb.nnodes;
            const BvhNode &n = is_syn ? (*init)[cur - bb.nnodes] : bb.nodes[cur]; o.nodes++;
            if (is_syn) o.syn++;
            else {
                g_by_depth[std::min(depth_of[cur], 79)]++;
                if (classify) {
                    bool hv = false, ht_all = false, ht_any = false;
                    for (int ch = 0; ch < 2; ch++) {
                        const Box b = child_box(n, ch); if (b.ref == kNoChild) continue;
                        if (holds(b, P.vp, 0.f)) hv = true;
                        int cnt = 0, tot = 0; for (int l = 0; l < 64; l++) if (P.alive[l]) { tot++; if (holds(b, P.pp[l], 0.f)) cnt++; }
                        if (cnt > 0) ht_any = true; if (cnt == tot) ht_all = true;
                    }
                    g_cls[hv && ht_any ? 0 : hv ? 1 : ht_all ? 2 : ht_any ? 3 : 4]++;
                }
            }
            float c0[3], h0[3], c1[3], h1[3];
            for (int k = 0; k < 3; k++) { c0[k] = n.ctr[k][0]; h0[k] = n.hal[k][0]; c1[k] = n.ctr[k][1]; h1[k] = n.hal[k][1]; }
            const int p0 = P.enters(c0, h0), p1 = P.enters(c1, h1);
            if (p0 == 0 && p1 == 0) { cur = kNoChild; break; }
            if (p0 == 0) { cur = n.c1; continue; }
            if (p1 == 0) { cur = n.c0; continue; }
            const bool first0 = p0 >= p1;
            stack[sp++] = first0 ? n.c1 : n.c0; cur = first0 ? n.c0 : n.c1;
        }
        if (cur != kNoChild) {
            const uint32_t cnt = (((uint32_t)~cur) & 3u) + 1u; o.leaves++; o.pairs += cnt > 2 ? 2 : 1;
            const int before = P.nalive;
            P.test_leaf(cur);
            if (P.nalive != before) at_last_hit = o.nodes;
            if (P.nalive == 0) break;
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    o.alive1 = P.nalive; o.after = o.nodes - at_last_hit;
    return o;
}

// conservative bounds of a bundle of segments that share the origin vp: the box of their end points, four planes through the origin
// around the directions, and two end-point planes (the VPL's own and a mean-normal plane under the far ends)
struct Frustum {
    V vp, vn, m, u, w, pnm, pl[4]; bool wide = false; int n = 0;
    float lo[3], hi[3], s_vpl, s_tile, amin, amax, bmin, bmax;
    V cen{ 0, 0, 0 };
    void begin(V vp_, V vn_) { vp = vp_; vn = vn_; n = 0; cen = { 0, 0, 0 }; pnm = { 0, 0, 0 }; }
    void centre(const Packet &P, const float *T) { for (int l = 0; l < 64; l++) if (P.alive[l]) { cen = cen + P.pp[l]; n++; pnm = pnm + V{ T[l * 7 + 4], T[l * 7 + 5], T[l * 7 + 6] }; } }
    void axes() {
        cen = cen * (1.0f / std::max(n, 1)); m = norm(cen - vp);
        const V ax = std::fabs(m.x) < 0.6f ? V{ 1, 0, 0 } : V{ 0, 1, 0 };
        u = norm(cross(m, ax)); w = cross(m, u); pnm = norm(pnm);
        amin = bmin = 1e30f; amax = bmax = -1e30f; s_vpl = s_tile = 1e30f; wide = false;
        for (int k = 0; k < 3; k++) { lo[k] = 1e30f; hi[k] = -1e30f; }
    }
    void bound(const Packet &P) {
        for (int l = 0; l < 64; l++) if (P.alive[l]) {
            const V d = P.d[l];
            const float dw = dot(m, d), dl = std::sqrt(dot(d, d));
            if (dw <= 0.05f * dl) wide = true;
            else { const float a = dot(u, d) / dw, b = dot(w, d) / dw; amin = std::min(amin, a); amax = std::max(amax, a); bmin = std::min(bmin, b); bmax = std::max(bmax, b); }
            const V e0 = vp + d * kTmin, e1 = vp + d * kTmax;
            const float q0[3] = { e0.x, e0.y, e0.z }, q1[3] = { e1.x, e1.y, e1.z };
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], std::min(q0[k], q1[k])); hi[k] = std::max(hi[k], std::max(q0[k], q1[k])); }
            s_vpl = std::min(s_vpl, dot(vn, d) * kTmin);
            s_tile = std::min(s_tile, std::min(dot(pnm, e0), dot(pnm, e1)));
        }
    }
    void finish() { pl[0] = u - m * amin; pl[1] = m * amax - u; pl[2] = w - m * bmin; pl[3] = m * bmax - w; }
    bool outside(const Box &b, int flags) const {
        const float eps = 1e-5f;
        for (int k = 0; k < 3; k++) if (b.c[k] - b.h[k] > hi[k] + eps || b.c[k] + b.h[k] < lo[k] - eps) return true;
        const V cc = { b.c[0], b.c[1], b.c[2] }, rel = cc - vp;
        auto ext = [&](V q) { return std::fabs(q.x) * b.h[0] + std::fabs(q.y) * b.h[1] + std::fabs(q.z) * b.h[2]; };
        if ((flags & 1) && !wide) for (int k = 0; k < 4; k++) if (dot(pl[k], rel) + ext(pl[k]) < -eps) return true;
        if (flags & 2) {
            if (dot(vn, rel) + ext(vn) < s_vpl - eps) return true;
            if (dot(pnm, cc) + ext(pnm) < s_tile - eps) return true;
        }
        return false;
    }
};

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: walk_proxy proxy.bin scene.obj lights.obj [walks] [mode] [a] [b] [c] [d]\n"); return 2; }
    const int nwalks = argc > 4 ? atoi(argv[4]) : 20000;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    const int pa = argc > 6 ? atoi(argv[6]) : 0, pb = argc > 7 ? atoi(argv[7]) : 0, pc = argc > 8 ? atoi(argv[8]) : 0, pd = argc > 9 ? atoi(argv[9]) : 0;
    std::vector<float> verts; load_obj(argv[2], verts); if (std::strcmp(argv[3], "-") != 0) load_obj(argv[3], verts);
    const int ntri = (int)(verts.size() / 9);
    build_bvh(verts.data(), ntri, 1, &bb);
    std::printf("tris %d nodes %d leaves %d depth %d build %.0f ms\n", ntri, bb.nnodes, bb.nleaves, bb.depth, bb.build_ms);
    depth_of.assign((size_t)bb.nnodes, 0);
    for (int i = 0; i < bb.nnodes; i++) { const BvhNode &n = bb.nodes[i]; if (n.c0 >= 0) depth_of[n.c0] = depth_of[i] + 1; if (n.c1 >= 0) depth_of[n.c1] = depth_of[i] + 1; }   // pre-order: children after parents
    FILE *f = std::fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[2]; if (std::fread(hdr, 4, 2, f) != 2) return 1;
    const int ntiles = hdr[0], nvpl = hdr[1];
    std::vector<float> tiles((size_t)ntiles * 64 * 7), vpls((size_t)nvpl * 6);
    if (std::fread(tiles.data(), 4, tiles.size(), f) != tiles.size() || std::fread(vpls.data(), 4, vpls.size(), f) != vpls.size()) return 1;
    std::fclose(f);
    std::printf("tiles %d vpls %d mode %d params %d %d %d %d\n", ntiles, nvpl, mode, pa, pb, pc, pd);
    std::mt19937 rng(4242);

    // ---------------------------------------------------------------- mode 1: per-VPL entry lists
    std::vector<std::vector<BvhNode>> syn((size_t)nvpl);
    if (mode == 1) {
        const int cull = pa, order = pb;     // cull 1: drop subtrees wholly behind the VPL's plane; order 0 deepest popped first, 1 top first, 2 nearest first
        double total = 0, culled = 0, len = 0, mx = 0;
        for (int v = 0; v < nvpl; v++) {
            const V vp = { vpls[6 * v], vpls[6 * v + 1], vpls[6 * v + 2] }, vn = { vpls[6 * v + 3], vpls[6 * v + 4], vpls[6 * v + 5] };
            std::vector<Box> sib; std::vector<int32_t> todo = { 0 };
            while (!todo.empty()) {
                const int32_t at = todo.back(); todo.pop_back();
                for (int ch = 0; ch < 2; ch++) {
                    const Box b = child_box(bb.nodes[at], ch);
                    if (b.ref == kNoChild) continue;
                    if (b.ref >= 0 && holds(b, vp, 0.f)) { todo.push_back(b.ref); continue; }       // an inner node that holds the VPL: its children are tested instead
                    total++;
                    if (cull) {
                        const float top = vn.x * (b.c[0] - vp.x) + vn.y * (b.c[1] - vp.y) + vn.z * (b.c[2] - vp.z) + std::fabs(vn.x) * b.h[0] + std::fabs(vn.y) * b.h[1] + std::fabs(vn.z) * b.h[2];
                        if (top < 0.f) { culled++; continue; }
                    }
                    sib.push_back(b);
                }
            }
            if (order == 2) {
                auto dist = [&](const Box &b) { float d = 0; const float p[3] = { vp.x, vp.y, vp.z }; for (int k = 0; k < 3; k++) { float e = std::max(std::fabs(p[k] - b.c[k]) - b.h[k], 0.f); d += e * e; } return d; };
                std::sort(sib.begin(), sib.end(), [&](const Box &a, const Box &b) { return dist(a) > dist(b); });
            } else if (order == 1) std::reverse(sib.begin(), sib.end());
            syn[v] = pair_up(sib); len += (double)syn[v].size(); mx = std::max(mx, (double)syn[v].size());
        }
        std::printf("entry lists: %.1f synthetic nodes per VPL (max %.0f), %.1f subtrees per VPL, %.1f %% behind the VPL's plane\n", len / nvpl, mx, total / nvpl, 100.0 * culled / std::max(total, 1.0));
    }

    double S_walks = 0, S_nodes = 0, S_syn = 0, S_leaves = 0, S_pairs = 0;
    double C_n[4] = {}, C_nodes[4] = {}, C_pairs[4] = {}, C_alive0[4] = {}, C_alive1[4] = {}, C_after[4] = {};
    auto account = [&](const WalkOut &o) {
        const int cls = o.leaves == 0 ? 0 : o.alive1 == 0 ? 1 : o.alive1 == o.alive0 ? 2 : 3;
        C_n[cls]++; C_nodes[cls] += o.nodes; C_pairs[cls] += o.pairs; C_alive0[cls] += o.alive0; C_alive1[cls] += o.alive1; C_after[cls] += o.after;
        S_walks++; S_nodes += o.nodes; S_syn += o.syn; S_leaves += o.leaves; S_pairs += o.pairs;
    };
    auto report = [&]() {
        std::printf("walks %.0f  nodes/walk %.2f (synthetic %.2f)  leaves/walk %.2f  pairs/walk %.2f  est VALU/walk %.0f (15 per visit, 46 per pair)\n",
                    S_walks, S_nodes / S_walks, S_syn / S_walks, S_leaves / S_walks, S_pairs / S_walks, 15.0 * S_nodes / S_walks + 46.0 * S_pairs / S_walks);
        const char *nm[4] = { "empty", "fully occluded", "leaves touched, no hit", "partially occluded" };
        for (int c = 0; c < 4; c++) std::printf("   %-24s %.3f of walks: %.1f visits, %.1f pairs, %.1f lanes alive at start, %.1f at end, %.1f visits after the last hit; share of all visits %.3f, of all pairs %.3f\n", nm[c], C_n[c] / S_walks,
                    C_nodes[c] / std::max(C_n[c], 1.0), C_pairs[c] / std::max(C_n[c], 1.0), C_alive0[c] / std::max(C_n[c], 1.0), C_alive1[c] / std::max(C_n[c], 1.0), C_after[c] / std::max(C_n[c], 1.0), C_nodes[c] / S_nodes, C_pairs[c] / std::max(S_pairs, 1.0));
    };

    if (mode == 0 || mode == 1) {
        int done = 0; Packet P;
        while (done < nwalks) {
            const int ti = (int)(rng() % (unsigned)ntiles), vi = (int)(rng() % (unsigned)nvpl);
            P.setup(&tiles[(size_t)ti * 64 * 7], &vpls[(size_t)vi * 6]);
            if (P.nalive == 0) continue;     // (the kernel skips these before the walk)
            done++;
            account(walk(P, mode == 1 ? &syn[vi] : nullptr, true));
        }
        report();
        std::printf("   real-node visits by what a child box holds: VPL+tile point %.2f  VPL only %.2f  all live tile points %.2f  some tile points %.2f  neither %.2f  /walk\n",
                    g_cls[0] / S_walks, g_cls[1] / S_walks, g_cls[2] / S_walks, g_cls[3] / S_walks, g_cls[4] / S_walks);
        std::printf("   visits by depth:"); for (int k = 0; k < 32; k++) std::printf(" %.2f", g_by_depth[k] / S_walks); std::printf("\n");
        return 0;
    }

    if (mode == 2) {
        const int flags = pa;
        double fvis = 0, cands = 0, none = 0, entered = 0, prs = 0, tested = 0, n = 0; std::vector<int> hist;
        Packet P; int done = 0;
        while (done < nwalks) {
            const int ti = (int)(rng() % (unsigned)ntiles), vi = (int)(rng() % (unsigned)nvpl);
            const float *T = &tiles[(size_t)ti * 64 * 7];
            P.setup(T, &vpls[(size_t)vi * 6]);
            if (P.nalive == 0) continue;
            done++; n++;
            Frustum F; F.begin(P.vp, P.vn); F.centre(P, T); F.axes(); F.bound(P); F.finish();
            std::vector<Box> cand; std::vector<int32_t> st = { 0 };
            while (!st.empty()) {
                const int32_t at = st.back(); st.pop_back(); fvis++;
                for (int ch = 0; ch < 2; ch++) { const Box b = child_box(bb.nodes[at], ch); if (b.ref == kNoChild || F.outside(b, flags)) continue; if (b.ref >= 0) st.push_back(b.ref); else cand.push_back(b); }
            }
            cands += (double)cand.size(); hist.push_back((int)cand.size()); if (cand.empty()) none++;
            for (const Box &b : cand) { if (P.nalive == 0) break; tested++; if (!P.enters(b.c, b.h)) continue; entered++; prs += ((((uint32_t)~b.ref) & 3u) + 1u) > 2 ? 2 : 1; P.test_leaf(b.ref); }
        }
        std::sort(hist.begin(), hist.end());
        auto pq = [&](double q) { return hist[(size_t)(q * (hist.size() - 1))]; };
        std::printf("frustum walk: %.2f node visits / packet, %.2f candidate leaves / packet (%.3f of the packets have none; percentiles 50/75/90/95/99/max %d %d %d %d %d %d); wave phase: %.2f leaf boxes tested, %.2f entered, %.2f triangle pairs / packet\n",
                    fvis / n, cands / n, none / n, pq(.5), pq(.75), pq(.9), pq(.95), pq(.99), hist.back(), tested / n, entered / n, prs / n);
        return 0;
    }

    if (mode == 5) {
        // (round 6) PACKET OR PER-LANE?  The packet walk visits the UNION of what its 64 segments touch; the walks that cost the most (tiles on
        // silhouettes: per-wave clocks show items 30 x the median) may be expensive because that union is large while every single segment
        // touches little -- the case in which a per-lane walk (one ray per lane, lockstep: the wave pays the LONGEST lane) would be cheaper.
        // For random (group of 2 x 2 tiles, VPL) samples with the device's group cut: per tile walk the packet's visits U and pairs, and, for
        // every lane alive at the start, the visits / triangle tests of that segment walking alone from the same cut (any-hit: stops at its
        // first hit).  Printed by class of U: the share of the packet cost, the mean and max single-segment visits, and what a hybrid
        // "packet up to T visits, then per-lane for the lanes still alive" would cost with a per-lane node visit priced at pa (default 40)
        // and a per-lane triangle test at pb (default 30) vector instructions per wave step.
        const int CN = pa ? pa : 40, CT = pb ? pb : 30, budget = pc ? pc : 8, flags = 3;
        const int tiles_x = (int)std::lround(std::sqrt((double)ntiles)), GX = tiles_x / 2;
        struct Cls { double n = 0, pk_cost = 0, U = 0, mean_v = 0, max_v = 0, max_t = 0, lane_cost = 0, alive0 = 0, alive1 = 0; };
        const int NC = 8; const unsigned edges[NC] = { 8, 16, 32, 64, 128, 256, 512, 0xffffffffu };
        Cls C[NC];
        const int thresholds[6] = { 32, 64, 96, 128, 192, 256 };
        double hybrid[6] = {}, total_pk = 0, walks = 0;
        // a single segment from the cut: visits and triangle tests until its first hit
        auto lane_walk = [&](const Packet &P0, int l, const std::vector<Box> &cut, bool is_root, unsigned &visits, unsigned &tris) {
            visits = tris = 0;
            const V o = P0.vp, d = P0.d[l];
            auto enters = [&](const float *c, const float *h) {
                float ax = c[0] * P0.ivx[l] + P0.nox[l], ay = c[1] * P0.ivy[l] + P0.noy[l], az = c[2] * P0.ivz[l] + P0.noz[l];
                float bx = h[0] * std::fabs(P0.ivx[l]), by = h[1] * std::fabs(P0.ivy[l]), bz = h[2] * std::fabs(P0.ivz[l]);
                float tn = clamp01(std::max(std::max(ax - bx, ay - by), az - bz)), tf = clamp01(std::min(std::min(ax + bx, ay + by), az + bz));
                return tn < tf;
            };
            std::vector<int32_t> st;
            if (is_root) st.push_back(0);
            else for (size_t k = cut.size(); k-- > 0;) { visits += (k & 1) == 0 ? 1 : 0; if (enters(cut[k].c, cut[k].h)) st.push_back(cut[k].ref); }   // (two cut entries per synthetic node)
            while (!st.empty()) {
                const int32_t cur = st.back(); st.pop_back();
                if (cur < 0) {
                    if (cur == kNoChild) continue;
                    const uint32_t id = (uint32_t)~cur, block = id >> 2, cnt = (id & 3u) + 1u;
                    for (uint32_t k = 0; k < cnt; k++) { tris++; if (tri_hit(bb.tri_flat[block * 4 + k], o, d, kTmin, kTmax)) return true; }
                    continue;
                }
                const BvhNode &n = bb.nodes[cur]; visits++;
                float c0[3], h0[3], c1[3], h1[3];
                for (int k = 0; k < 3; k++) { c0[k] = n.ctr[k][0]; h0[k] = n.hal[k][0]; c1[k] = n.ctr[k][1]; h1[k] = n.hal[k][1]; }
                if (n.c1 != kNoChild && enters(c1, h1)) st.push_back(n.c1);
                if (n.c0 != kNoChild && enters(c0, h0)) st.push_back(n.c0);
            }
            return false;
        };
        while (walks < nwalks) {
            const int gx = (int)(rng() % (unsigned)GX), gy = (int)(rng() % (unsigned)GX), vi = (int)(rng() % (unsigned)nvpl);
            const float *vpl = &vpls[(size_t)vi * 6];
            const float *T4[4]; for (int j = 0; j < 4; j++) T4[j] = &tiles[(size_t)((gy * 2 + j / 2) * tiles_x + gx * 2 + j % 2) * 64 * 7];
            Packet PK[4];
            Frustum F; F.begin({ vpl[0], vpl[1], vpl[2] }, { vpl[3], vpl[4], vpl[5] });
            int live = 0;
            for (int j = 0; j < 4; j++) { PK[j].setup(T4[j], vpl); if (PK[j].nalive) { live++; F.centre(PK[j], T4[j]); } }
            if (!live) continue;
            F.axes();
            {
                float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f }; int nv = 0;
                for (int j = 0; j < 4; j++) for (int l = 0; l < 64; l++) if (T4[j][l * 7 + 3] != 0.f) { nv++; for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], T4[j][l * 7 + k]); hi[k] = std::max(hi[k], T4[j][l * 7 + k]); } }
                Packet Cc; Cc.vp = F.vp; Cc.vn = F.vn; for (int l = 0; l < 64; l++) Cc.alive[l] = false;
                for (int q = 0; q < 8; q++) { const V pq = { (q & 1) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 4) ? hi[2] : lo[2] }; Cc.alive[q] = true; Cc.pp[q] = pq; Cc.d[q] = pq - F.vp; }
                if (nv) F.bound(Cc);
            }
            F.finish();
            std::vector<Box> cut; { Box r; std::memset(&r, 0, sizeof r); r.ref = 0; for (int k = 0; k < 3; k++) r.h[k] = 1e30f; cut.push_back(r); }
            size_t leaves_in_row = 0;
            while (!cut.empty() && leaves_in_row < cut.size()) {
                const Box e = cut.front();
                if (e.ref < 0) { cut.erase(cut.begin()); cut.push_back(e); leaves_in_row++; continue; }
                const BvhNode &n = bb.nodes[e.ref];
                Box kids[2]; int nk = 0;
                for (int ch = 0; ch < 2; ch++) { const Box b = child_box(n, ch); if (b.ref == kNoChild || F.outside(b, flags)) continue; kids[nk++] = b; }
                if ((int)cut.size() - 1 + nk > budget) break;
                cut.erase(cut.begin()); for (int k = 0; k < nk; k++) cut.push_back(kids[k]);
                leaves_in_row = 0;
            }
            const bool is_root = cut.size() == 1 && cut[0].ref == 0 && cut[0].h[0] > 1e29f;
            std::sort(cut.begin(), cut.end(), [&](const Box &a, const Box &b) {
                auto dist = [&](const Box &q) { const float vp[3] = { F.vp.x, F.vp.y, F.vp.z }; float s2 = 0; for (int k = 0; k < 3; k++) { float e = std::max(std::fabs(vp[k] - q.c[k]) - q.h[k], 0.f); s2 += e * e; } return s2; };
                return dist(a) > dist(b); });
            for (int j = 0; j < 4; j++) if (PK[j].nalive) {
                walks++;
                if (cut.empty()) { C[0].n++; C[0].alive0 += PK[j].nalive; C[0].alive1 += PK[j].nalive; continue; }
                const Packet P0 = PK[j];                     // (the packet walk kills lanes: the single-segment walks need the untouched copy)
                // single segments first
                unsigned mv = 0, mt = 0; double sv = 0; int na = 0;
                for (int l = 0; l < 64; l++) if (P0.alive[l]) { unsigned v, t; lane_walk(P0, l, cut, is_root, v, t); mv = std::max(mv, v); mt = std::max(mt, t); sv += v; na++; }
                const std::vector<BvhNode> init = pair_up(cut);
                const WalkOut o = walk(PK[j], is_root ? nullptr : &init, false);
                const double pk = 15.0 * o.nodes + 46.0 * o.pairs, ln = (double)CN * mv + (double)CT * mt;
                int c = 0; while (o.nodes > edges[c]) c++;
                C[c].n++; C[c].pk_cost += pk; C[c].U += o.nodes; C[c].mean_v += sv / std::max(na, 1); C[c].max_v += mv; C[c].max_t += mt; C[c].lane_cost += ln; C[c].alive0 += o.alive0; C[c].alive1 += o.alive1;
                total_pk += pk;
                // hybrid: the packet walk as it is while it stays within T visits; beyond, T visits' worth of packet work is lost and the lanes
                // walk alone (upper bound: all lanes that were alive at the START walk alone, from the cut)
                for (int q = 0; q < 6; q++) hybrid[q] += o.nodes <= (unsigned)thresholds[q] ? pk : std::min(pk, 15.0 * thresholds[q] + 46.0 * o.pairs * thresholds[q] / std::max(o.nodes, 1u) + ln);
            }
        }
        std::printf("per-lane prices: node visit %d, triangle test %d vector instructions per wave step; %0.f walks\n", CN, CT, walks);
        std::printf("  packet visits U  | walks   | share of packet cost | mean U | single segment: mean visits, longest lane's visits, longest lane's triangle tests | per-lane cost / packet cost | lanes alive start -> end\n");
        for (int c = 0; c < NC; c++) if (C[c].n > 0)
            std::printf("  U <= %-10u | %.4f | %.3f | %.1f | %.1f  %.1f  %.1f | %.2f | %.1f -> %.1f\n", edges[c], C[c].n / walks, C[c].pk_cost / std::max(total_pk, 1.0), C[c].U / C[c].n,
                        C[c].mean_v / C[c].n, C[c].max_v / C[c].n, C[c].max_t / C[c].n, C[c].lane_cost / std::max(C[c].pk_cost, 1.0), C[c].alive0 / C[c].n, C[c].alive1 / C[c].n);
        for (int q = 0; q < 6; q++) std::printf("hybrid, switch after %3d packet visits: %.3f of the packet-only cost\n", thresholds[q], hybrid[q] / std::max(total_pk, 1.0));
        return 0;
    }

    if (mode == 4) {
        // (round 6) COHERENT RE-BINNING: lane <-> pixel is free (per-pixel sums, fixed VPL order), so the 256 pixels of a group of 2 x 2 tiles can
        // be dealt to its four wavefronts by WORLD position instead of by screen quadrant: a tile that straddles a depth discontinuity (chair in
        // front of floor) makes a fat bundle of segments for every VPL.  Same group cut for both (it bounds all 256 pixels; budget pc, breadth
        // first, bounds from the union box: the device's), then the four walks.  pa = clustering: 0 k-d (median split of the largest extent, twice),
        // 1 sorted by distance from the eye (argv eye = the first VPL-independent guess: centroid of all tile points is NOT the eye, so the key is
        // the distance to `EYE=x,y,z`), 2 k-d over position + 0.5 x normal (six dimensions).  Prints both totals and, per decile of the screen
        // tiles' box diagonal relative to the group's smallest, where the cost sits.
        const int clustering = pa, budget = pc ? pc : 8, flags = 3;
        const int tiles_x = (int)std::lround(std::sqrt((double)ntiles));
        if (tiles_x * tiles_x != ntiles) { std::fprintf(stderr, "mode 4 needs the full tile grid\n"); return 1; }
        float eye[3] = { 0, 0, 0 }; if (getenv("EYE")) std::sscanf(getenv("EYE"), "%f,%f,%f", &eye[0], &eye[1], &eye[2]);
        const int GX = tiles_x / 2;
        std::vector<std::vector<float>> rebinned((size_t)GX * GX);       // per group: 4 x 64 x 7 floats, built on first use
        auto build_rebinned = [&](int gx, int gy) {
            std::vector<float> &R = rebinned[(size_t)gy * GX + gx];
            if (!R.empty()) return;
            struct Px { float v[7]; };
            std::vector<Px> px;
            for (int j = 0; j < 4; j++) { const int ti = (gy * 2 + j / 2) * tiles_x + gx * 2 + j % 2; const float *T = &tiles[(size_t)ti * 64 * 7]; for (int l = 0; l < 64; l++) { Px q; std::memcpy(q.v, T + l * 7, 28); px.push_back(q); } }
            auto key_dim = [&](const Px &q, int dim) { return dim < 3 ? q.v[dim] : 0.5f * q.v[4 + dim - 3]; };
            std::function<void(int, int, int)> split = [&](int lo, int hi, int levels) {
                if (levels == 0) return;
                if (clustering == 1) {
                    auto dist = [&](const Px &q) { float s = 0; for (int k = 0; k < 3; k++) s += (q.v[k] - eye[k]) * (q.v[k] - eye[k]); return q.v[3] != 0.f ? s : 3e38f; };
                    std::sort(px.begin() + lo, px.begin() + hi, [&](const Px &a, const Px &b) { return dist(a) < dist(b); });
                    return;                                          // one sort, chunks of 64
                }
                const int dims = clustering == 2 ? 6 : 3;
                int best = 0; float best_e = -1.f;
                for (int dim = 0; dim < dims; dim++) { float mn = 3e38f, mx = -3e38f; for (int i = lo; i < hi; i++) if (px[(size_t)i].v[3] != 0.f) { mn = std::min(mn, key_dim(px[(size_t)i], dim)); mx = std::max(mx, key_dim(px[(size_t)i], dim)); } if (mx - mn > best_e) { best_e = mx - mn; best = dim; } }
                // (pixels outside the scene -- w = 0 -- sort to the end: they are dead lanes wherever they sit)
                std::sort(px.begin() + lo, px.begin() + hi, [&](const Px &a, const Px &b) { const float ka = a.v[3] != 0.f ? key_dim(a, best) : 3e38f, kb = b.v[3] != 0.f ? key_dim(b, best) : 3e38f; return ka < kb; });
                const int mid = (lo + hi) / 2;
                split(lo, mid, levels - 1); split(mid, hi, levels - 1);
            };
            split(0, 256, 2);
            R.resize(4 * 64 * 7);
            for (int i = 0; i < 256; i++) std::memcpy(&R[(size_t)i * 7], px[(size_t)i].v, 28);
        };
        struct Acc { double walks = 0, nodes = 0, pairs = 0, lanes = 0, groups = 0, cutsz = 0; double cost() const { return 15.0 * nodes + 46.0 * pairs; } };
        Acc A[2]; Acc D[2][10];
        auto eval_group = [&](const float *const T4[4], const float *vpl, Acc &acc, Acc *dec) {
            Packet PK[4];
            Frustum F; F.begin({ vpl[0], vpl[1], vpl[2] }, { vpl[3], vpl[4], vpl[5] });
            int live = 0;
            for (int j = 0; j < 4; j++) { PK[j].setup(T4[j], vpl); if (PK[j].nalive) { live++; F.centre(PK[j], T4[j]); } }
            if (!live) return false;
            F.axes();
            {   // bounds from the eight corners of the union box of all valid pixels (the device's)
                float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f }; int nv = 0;
                for (int j = 0; j < 4; j++) for (int l = 0; l < 64; l++) if (T4[j][l * 7 + 3] != 0.f) { nv++; for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], T4[j][l * 7 + k]); hi[k] = std::max(hi[k], T4[j][l * 7 + k]); } }
                Packet C; C.vp = F.vp; C.vn = F.vn; for (int l = 0; l < 64; l++) C.alive[l] = false;
                for (int q = 0; q < 8; q++) { const V pq = { (q & 1) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 4) ? hi[2] : lo[2] }; C.alive[q] = true; C.pp[q] = pq; C.d[q] = pq - F.vp; }
                if (nv) F.bound(C);
            }
            F.finish();
            std::vector<Box> cut; { Box r; std::memset(&r, 0, sizeof r); r.ref = 0; for (int k = 0; k < 3; k++) r.h[k] = 1e30f; cut.push_back(r); }
            size_t leaves_in_row = 0;
            while (!cut.empty() && leaves_in_row < cut.size()) {
                const Box e = cut.front();
                if (e.ref < 0) { cut.erase(cut.begin()); cut.push_back(e); leaves_in_row++; continue; }
                const BvhNode &n = bb.nodes[e.ref];
                Box kids[2]; int nk = 0;
                for (int ch = 0; ch < 2; ch++) { const Box b = child_box(n, ch); if (b.ref == kNoChild || F.outside(b, flags)) continue; kids[nk++] = b; }
                if ((int)cut.size() - 1 + nk > budget) break;
                cut.erase(cut.begin()); for (int k = 0; k < nk; k++) cut.push_back(kids[k]);
                leaves_in_row = 0;
            }
            const bool is_root = cut.size() == 1 && cut[0].ref == 0 && cut[0].h[0] > 1e29f;
            std::sort(cut.begin(), cut.end(), [&](const Box &a, const Box &b) {
                auto dist = [&](const Box &q) { const float vp[3] = { F.vp.x, F.vp.y, F.vp.z }; float s2 = 0; for (int k = 0; k < 3; k++) { float e = std::max(std::fabs(vp[k] - q.c[k]) - q.h[k], 0.f); s2 += e * e; } return s2; };
                return dist(a) > dist(b); });
            acc.groups++; acc.cutsz += (double)cut.size();
            for (int j = 0; j < 4; j++) if (PK[j].nalive) {
                acc.walks++; acc.lanes += PK[j].nalive; if (dec) { dec->walks++; dec->lanes += PK[j].nalive; }
                if (cut.empty()) continue;
                const std::vector<BvhNode> init = pair_up(cut);
                const WalkOut o = walk(PK[j], is_root ? nullptr : &init, false);
                acc.nodes += o.nodes; acc.pairs += o.pairs; if (dec) { dec->nodes += o.nodes; dec->pairs += o.pairs; }
            }
            return true;
        };
        // decile of a group by how much deeper it is than a flat patch: diagonal of the group's position box / the smallest of its tiles' diagonals
        auto group_spread = [&](int gx, int gy) {
            float glo[3] = { 1e30f, 1e30f, 1e30f }, ghi[3] = { -1e30f, -1e30f, -1e30f }, dmin = 1e30f;
            for (int j = 0; j < 4; j++) {
                const int ti = (gy * 2 + j / 2) * tiles_x + gx * 2 + j % 2; const float *T = &tiles[(size_t)ti * 64 * 7];
                float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
                for (int l = 0; l < 64; l++) if (T[l * 7 + 3] != 0.f) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], T[l * 7 + k]); hi[k] = std::max(hi[k], T[l * 7 + k]); glo[k] = std::min(glo[k], T[l * 7 + k]); ghi[k] = std::max(ghi[k], T[l * 7 + k]); }
                if (lo[0] <= hi[0]) dmin = std::min(dmin, std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2])));
            }
            const float dg = std::sqrt((ghi[0] - glo[0]) * (ghi[0] - glo[0]) + (ghi[1] - glo[1]) * (ghi[1] - glo[1]) + (ghi[2] - glo[2]) * (ghi[2] - glo[2]));
            return dg / std::max(dmin, 1e-6f);
        };
        // the groups' spread deciles first (all groups)
        std::vector<float> spread((size_t)GX * GX); for (int gy = 0; gy < GX; gy++) for (int gx = 0; gx < GX; gx++) spread[(size_t)gy * GX + gx] = group_spread(gx, gy);
        std::vector<float> sorted_spread = spread; std::sort(sorted_spread.begin(), sorted_spread.end());
        auto decile = [&](float v) { int d = (int)(std::lower_bound(sorted_spread.begin(), sorted_spread.end(), v) - sorted_spread.begin()) * 10 / (int)sorted_spread.size(); return std::min(d, 9); };
        while (A[0].walks < nwalks) {
            const int gx = (int)(rng() % (unsigned)GX), gy = (int)(rng() % (unsigned)GX), vi = (int)(rng() % (unsigned)nvpl);
            const float *vpl = &vpls[(size_t)vi * 6];
            const float *S4[4]; for (int j = 0; j < 4; j++) S4[j] = &tiles[(size_t)((gy * 2 + j / 2) * tiles_x + gx * 2 + j % 2) * 64 * 7];
            const int dc = decile(spread[(size_t)gy * GX + gx]);
            if (!eval_group(S4, vpl, A[0], &D[0][dc])) continue;
            build_rebinned(gx, gy);
            const std::vector<float> &R = rebinned[(size_t)gy * GX + gx];
            const float *R4[4] = { &R[0], &R[64 * 7], &R[128 * 7], &R[192 * 7] };
            eval_group(R4, vpl, A[1], &D[1][dc]);
        }
        const char *nm[2] = { "screen quadrants", "re-binned" };
        for (int v = 0; v < 2; v++)
            std::printf("%-17s: %.0f (group, VPL) samples, %.0f walks (%.3f per sample), %.1f lanes alive per walk; per walk %.2f visits %.2f pairs est VALU %.0f;  per SAMPLE est VALU %.0f;  per live (pixel, VPL) pair %.2f\n",
                        nm[v], A[v].groups, A[v].walks, A[v].walks / A[v].groups, A[v].lanes / A[v].walks, A[v].nodes / A[v].walks, A[v].pairs / A[v].walks, A[v].cost() / A[v].walks, A[v].cost() / A[v].groups, A[v].cost() / A[v].lanes);
        std::printf("re-binned / screen: est VALU per live pair %.4f (the kill criterion: < 0.92), walks %.4f\n", (A[1].cost() / A[1].lanes) / (A[0].cost() / A[0].lanes), A[1].walks / A[0].walks);
        std::printf("by decile of the group's depth spread (group box diagonal / smallest tile box diagonal): share of the screen-quadrant cost, re-binned / screen cost\n");
        for (int dcl = 0; dcl < 10; dcl++) std::printf("   decile %d (spread <= %.2f): share %.3f  ratio %.3f  walks ratio %.3f\n", dcl, sorted_spread[std::min((size_t)((dcl + 1) * sorted_spread.size() / 10), sorted_spread.size() - 1)],
                                                        D[0][dcl].cost() / A[0].cost(), D[1][dcl].cost() / std::max(D[0][dcl].cost(), 1.0), D[1][dcl].walks / std::max(D[0][dcl].walks, 1.0));
        return 0;
    }

    if (mode == 3) {
        // pa = G (tiles per group edge), pb = frustum flags, pc = cut budget (entries; the descent stops refining when the cut would exceed it),
        // pd = 1: refine the group's cut per tile with the tile's own frustum (no further descent, just drop what the tile cannot reach)
        const int G = std::max(pa, 1), flags = pb ? pb : 3, budget = pc ? pc : 32, per_tile = pd;
        const int policy = getenv("POLICY") ? atoi(getenv("POLICY")) : 0, corner_bounds = getenv("CORNERS") ? atoi(getenv("CORNERS")) : 0, nosort = getenv("NOSORT") ? 1 : 0;
        const int tiles_x = (int)std::lround(std::sqrt((double)ntiles));
        if (tiles_x * tiles_x != ntiles) { std::fprintf(stderr, "mode 3 needs the full tile grid (dump with --stride 1)\n"); return 1; }
        double groups = 0, fvis = 0, cut_sum = 0, cut_empty = 0, fallback = 0, tile_cut = 0, tile_walks = 0, tile_skipped = 0; std::vector<int> hist;
        std::vector<Packet> PK((size_t)G * G);
        while (S_walks < nwalks) {
            const int gx = (int)(rng() % (unsigned)(tiles_x / G)), gy = (int)(rng() % (unsigned)(tiles_x / G)), vi = (int)(rng() % (unsigned)nvpl);
            const float *vpl = &vpls[(size_t)vi * 6];
            Frustum F; F.begin({ vpl[0], vpl[1], vpl[2] }, { vpl[3], vpl[4], vpl[5] });
            int live_tiles = 0;
            for (int j = 0; j < G * G; j++) {
                const int ti = (gy * G + j / G) * tiles_x + gx * G + j % G;
                const float *T = &tiles[(size_t)ti * 64 * 7];
                PK[j].setup(T, vpl); if (PK[j].nalive) { live_tiles++; F.centre(PK[j], T); }
            }
            if (!live_tiles) continue;
            groups++;
            F.axes();
            if (!corner_bounds) { for (int j = 0; j < G * G; j++) if (PK[j].nalive) F.bound(PK[j]); }
            else if (corner_bounds == 2) {
                // bounds from the eight corners of the UNION of the tiles' position boxes (a quarter of the device's set-up)
                float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f }; int nv = 0;
                for (int j = 0; j < G * G; j++) {
                    const int ti = (gy * G + j / G) * tiles_x + gx * G + j % G; const float *T = &tiles[(size_t)ti * 64 * 7];
                    for (int l = 0; l < 64; l++) if (T[l * 7 + 3] != 0.f) { nv++; for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], T[l * 7 + k]); hi[k] = std::max(hi[k], T[l * 7 + k]); } }
                }
                if (nv) {
                    Packet C; C.vp = F.vp; C.vn = F.vn; for (int l = 0; l < 64; l++) C.alive[l] = false;
                    for (int q = 0; q < 8; q++) { const V p = { (q & 1) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 4) ? hi[2] : lo[2] }; C.alive[q] = true; C.pp[q] = p; C.d[q] = p - F.vp; }
                    F.bound(C);
                }
            } else {
                // bounds from the eight corners of every tile's position box (what primary_kernel already writes per tile), all valid pixels
                for (int j = 0; j < G * G; j++) {
                    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f }; int nv = 0;
                    const int ti = (gy * G + j / G) * tiles_x + gx * G + j % G; const float *T = &tiles[(size_t)ti * 64 * 7];
                    for (int l = 0; l < 64; l++) if (T[l * 7 + 3] != 0.f) { nv++; for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], T[l * 7 + k]); hi[k] = std::max(hi[k], T[l * 7 + k]); } }
                    if (!nv) continue;
                    Packet C; C.vp = F.vp; C.vn = F.vn; for (int l = 0; l < 64; l++) C.alive[l] = false;
                    for (int q = 0; q < 8; q++) { const V p = { (q & 1) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 4) ? hi[2] : lo[2] }; C.alive[q] = true; C.pp[q] = p; C.d[q] = p - F.vp; }
                    F.bound(C);
                }
            }
            F.finish();
            // the cut: refinement of the largest inner entry (policy 0) or of the entries in FIFO order (policy 1: breadth first), while
            // the cut stays within the budget
            std::vector<Box> cut; { Box r; std::memset(&r, 0, sizeof r); r.ref = 0; for (int k = 0; k < 3; k++) r.h[k] = 1e30f; cut.push_back(r); }
            if (policy == 0) {
                for (;;) {
                    int best = -1; float best_a = -1.f;
                    for (size_t k = 0; k < cut.size(); k++) if (cut[k].ref >= 0) { const float a = cut[k].h[0] * cut[k].h[1] + cut[k].h[1] * cut[k].h[2] + cut[k].h[2] * cut[k].h[0]; if (a > best_a) { best_a = a; best = (int)k; } }
                    if (best < 0) break;
                    const BvhNode &n = bb.nodes[cut[best].ref]; fvis++;
                    Box kids[2]; int nk = 0;
                    for (int ch = 0; ch < 2; ch++) { const Box b = child_box(n, ch); if (b.ref == kNoChild || F.outside(b, flags)) continue; kids[nk++] = b; }
                    if ((int)cut.size() - 1 + nk > budget) break;
                    cut.erase(cut.begin() + best); for (int k = 0; k < nk; k++) cut.push_back(kids[k]);
                }
            } else {
                // ring: pop the front; an inner entry is replaced by its surviving children at the back, a leaf goes to the back as it is;
                // stop when an expansion does not fit, or after a full round without an inner entry
                size_t leaves_in_row = 0;
                while (!cut.empty() && leaves_in_row < cut.size()) {
                    const Box e = cut.front();
                    if (e.ref < 0) { cut.erase(cut.begin()); cut.push_back(e); leaves_in_row++; continue; }
                    const BvhNode &n = bb.nodes[e.ref]; fvis++;
                    Box kids[2]; int nk = 0;
                    for (int ch = 0; ch < 2; ch++) { const Box b = child_box(n, ch); if (b.ref == kNoChild || F.outside(b, flags)) continue; kids[nk++] = b; }
                    if ((int)cut.size() - 1 + nk > budget) break;
                    cut.erase(cut.begin()); for (int k = 0; k < nk; k++) cut.push_back(kids[k]);
                    leaves_in_row = 0;
                }
            }
            const bool is_root = cut.size() == 1 && cut[0].ref == 0 && cut[0].h[0] > 1e29f;
            if (is_root) fallback++;
            cut_sum += (double)cut.size(); hist.push_back((int)cut.size()); if (cut.empty()) cut_empty++;
            // far entries at the bottom of the stack, near ones popped first
            // SORTKEY: 0 distance of the box to the VPL (the device's), 1 distance to the centroid of the group's pixels, 2 the smaller of the two,
            // 3 surface area of the box (largest first), 4 distance to the VPL, farthest first
            static const int sortkey = getenv("SORTKEY") ? atoi(getenv("SORTKEY")) : 0;
            float gc[3] = { 0, 0, 0 }; { int nn = 0; for (int j = 0; j < G * G; j++) for (int l = 0; l < 64; l++) if (PK[j].alive[l]) { gc[0] += PK[j].pp[l].x; gc[1] += PK[j].pp[l].y; gc[2] += PK[j].pp[l].z; nn++; } for (int k = 0; k < 3; k++) gc[k] /= (float)std::max(nn, 1); }
            if (!nosort) std::sort(cut.begin(), cut.end(), [&](const Box &a, const Box &b) {
                auto dist = [&](const Box &q, const float *p) { float s = 0; for (int k = 0; k < 3; k++) { float e = std::max(std::fabs(p[k] - q.c[k]) - q.h[k], 0.f); s += e * e; } return s; };
                const float vp[3] = { F.vp.x, F.vp.y, F.vp.z };
                auto key = [&](const Box &q) {
                    if (sortkey == 1) return dist(q, gc);
                    if (sortkey == 2) return std::min(dist(q, gc), dist(q, vp));
                    if (sortkey == 3) return -(q.h[0] * q.h[1] + q.h[1] * q.h[2] + q.h[2] * q.h[0]);
                    if (sortkey == 4) return -dist(q, vp);
                    return dist(q, vp); };
                return key(a) > key(b); });
            for (int j = 0; j < G * G; j++) if (PK[j].nalive) {
                if (is_root) { account(walk(PK[j], nullptr, false)); tile_walks++; continue; }
                std::vector<Box> mine = cut;
                if (per_tile) {
                    const int ti = (gy * G + j / G) * tiles_x + gx * G + j % G;
                    Frustum Ft; Ft.begin(F.vp, F.vn); Ft.centre(PK[j], &tiles[(size_t)ti * 64 * 7]); Ft.axes(); Ft.bound(PK[j]); Ft.finish();
                    mine.clear(); for (const Box &b : cut) if (!Ft.outside(b, flags)) mine.push_back(b);
                }
                tile_cut += (double)mine.size();
                if (mine.empty()) { tile_skipped++; WalkOut o; o.alive0 = o.alive1 = PK[j].nalive; account(o); continue; }
                const std::vector<BvhNode> init = pair_up(mine);
                account(walk(PK[j], &init, false)); tile_walks++;
            }
        }
        std::sort(hist.begin(), hist.end());
        auto pq = [&](double q) { return hist[(size_t)(q * (hist.size() - 1))]; };
        std::printf("groups of %dx%d tiles: %.0f (group, VPL) cuts, %.1f refinement steps each, cut size %.2f (percentiles 50/75/90/99/max %d %d %d %d %d), empty %.3f, left at the root %.3f;  per tile walk: %.2f cut entries, %.3f of the walks skipped (empty cut)\n",
                    G, G, groups, fvis / groups, cut_sum / groups, pq(.5), pq(.75), pq(.9), pq(.99), hist.back(), cut_empty / groups, fallback / groups, tile_cut / std::max(S_walks, 1.0), tile_skipped / std::max(S_walks, 1.0));
        report();
        std::printf("   refinement steps per tile walk %.2f\n", fvis / S_walks);
        return 0;
    }
    return 0;
}
