// Developer tool (CPU only): how many node visits / leaf blocks does the gather's packet walk need per (tile, VPL) on a given
// tree?  Builds the tree with the product's host builder (bvh_build.cpp, linked in), draws VPL-like and tile-like surface
// points (area weighted; a "tile" is 64 points on a small disc around a surface point) and replays occluded_wave's control
// flow in plain C++.  A proxy for tools/traversal_stats.py that needs no GPU: use it to compare builder heuristics, then
// confirm on the device.
//   g++ -O2 -std=c++17 -DEVPLP_DEV_KNOBS -I evplp_amd/csrc -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bvh_eval/bvh_eval.cpp evplp_amd/csrc/bvh_build.cpp -o build/bvh_eval
//   build/bvh_eval scene.obj [builder 0..2] [walks]
#include "evplp_types.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <algorithm>
using namespace evplp;
struct V { float x, y, z; };
static V operator-(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V operator+(V a, V b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V operator*(V a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static V norm(V a) { float l = std::sqrt(dot(a, a)); return a * (1.0f / l); }

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

struct Stats { double chain_len = 0, walks = 0, nodes = 0, leaves = 0, pairs = 0, empty = 0, empty_nodes = 0, full = 0, leaf_at_vpl = 0, leaf_at_tile = 0, leaf_hit = 0, node_at_vpl = 0, node_at_tile = 0; };

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: bvh_eval scene.obj [builder] [walks]\n"); return 2; }
    int builder = argc > 2 ? atoi(argv[2]) : 0; int nwalks = argc > 3 ? atoi(argv[3]) : 20000;
    std::vector<V> vs; std::vector<float> verts;
    FILE *f = std::fopen(argv[1], "r"); if (!f) { perror("obj"); return 1; }
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
    int ntri = (int)(verts.size() / 9);
    BvhBuild bb; build_bvh(verts.data(), ntri, builder, &bb);
    std::printf("tris %d nodes %d leaves %d depth %d build %.0f ms\n", ntri, bb.nnodes, bb.nleaves, bb.depth, bb.build_ms);
    // area CDF
    std::vector<double> cdf(ntri); double acc = 0;
    auto tv = [&](int t, int k) { return V{ verts[9 * (size_t)t + 3 * k], verts[9 * (size_t)t + 3 * k + 1], verts[9 * (size_t)t + 3 * k + 2] }; };
    for (int t = 0; t < ntri; t++) { V n = cross(tv(t, 1) - tv(t, 0), tv(t, 2) - tv(t, 0)); acc += 0.5 * std::sqrt(dot(n, n)); cdf[t] = acc; }
    std::mt19937 rng(12345); std::uniform_real_distribution<float> U(0.f, 1.f);
    auto sample = [&](V &p, V &n) {
        double r = U(rng) * acc; int t = (int)(std::lower_bound(cdf.begin(), cdf.end(), r) - cdf.begin()); if (t >= ntri) t = ntri - 1;
        float a = std::sqrt(U(rng)), b = U(rng); float u = 1 - a, v = a * (1 - b), w = a * b;
        p = tv(t, 0) * u + tv(t, 1) * v + tv(t, 2) * w; n = norm(cross(tv(t, 1) - tv(t, 0), tv(t, 2) - tv(t, 0)));
    };
    const float tile_r = argc > 4 ? (float)atof(argv[4]) : 0.07f;
    const int chain = argc > 6 ? atoi(argv[6]) : 0; const float margin = argc > 7 ? (float)atof(argv[7]) : 1e-3f;
    const int order = argc > 5 ? atoi(argv[5]) : 0;   // descent order when both children are hit: 0 = more lanes first (the kernel), 1.. see below
    Stats S; int done = 0, tries = 0;
    std::vector<int32_t> stack(256);
    while (done < nwalks && tries < nwalks * 50) {
        tries++;
        V vp, vn, pp, pn; sample(vp, vn); sample(pp, pn);
        // tile: 64 points on a disc around pp in its tangent plane
        V tu = norm(std::fabs(pn.x) > 0.5f ? cross(pn, V{ 0, 1, 0 }) : cross(pn, V{ 1, 0, 0 })), tw = cross(pn, tu);
        V d[64]; bool alive[64]; int nalive = 0;
        float ivx[64], ivy[64], ivz[64], nox[64], noy[64], noz[64];
        const float tmin = 1e-4f, tmax = 1.f - 1e-4f, ku = 1.0f / (tmax - tmin);
        for (int l = 0; l < 64; l++) {
            float gx = ((l & 7) - 3.5f) / 3.5f * tile_r, gy = ((l >> 3) - 3.5f) / 3.5f * tile_r;
            V p1 = pp + tu * gx + tw * gy; V v12 = vp - p1;
            float c1 = dot(pn, v12), c2 = -dot(vn, v12);
            alive[l] = c1 > 0 && c2 > 0; if (alive[l]) nalive++;
            d[l] = p1 - vp;
            float i0x = srcp(d[l].x), i0y = srcp(d[l].y), i0z = srcp(d[l].z);
            ivx[l] = i0x * ku; ivy[l] = i0y * ku; ivz[l] = i0z * ku;
            const float dead = INFINITY;
            nox[l] = alive[l] ? (-(vp.x * i0x) - tmin) * ku : dead; noy[l] = alive[l] ? (-(vp.y * i0y) - tmin) * ku : dead; noz[l] = alive[l] ? (-(vp.z * i0z) - tmin) * ku : dead;
        }
        if (nalive == 0) continue;
        done++;
        int sp = 0; int32_t cur = 0; unsigned nodes = 0, leaves = 0, pairs = 0;
        std::vector<BvhNode> syn;                   // chain mode: synthetic nodes, indices >= bb.nnodes
        if (chain) {
            // follow the child whose box holds the VPL strictly inside (by `margin`); the siblings, two at a time, become synthetic nodes
            std::vector<std::pair<int32_t, int>> sib;          // (parent node, which child) of every sibling subtree
            int32_t at = 0; int32_t last = 0;
            for (;;) {
                const BvhNode &n = bb.nodes[at];
                auto inside = [&](int ch) { return std::fabs(vp.x - n.ctr[0][ch]) < n.hal[0][ch] - margin && std::fabs(vp.y - n.ctr[1][ch]) < n.hal[1][ch] - margin && std::fabs(vp.z - n.ctr[2][ch]) < n.hal[2][ch] - margin; };
                const bool i0 = inside(0), i1 = inside(1);
                if (i0 == i1) { last = at; break; }
                const int in = i0 ? 0 : 1; const int32_t nxt = in == 0 ? n.c0 : n.c1;
                sib.push_back({ at, 1 - in });
                if (nxt < 0) { sib.push_back({ at, in }); last = -1; break; }     // reached a leaf: it is a subtree of its own
                at = nxt;
            }
            S.chain_len += sib.size();
            auto make = [&](std::pair<int32_t, int> a, const std::pair<int32_t, int> *b) {
                BvhNode m; std::memset(&m, 0, sizeof m);
                const BvhNode &pa = bb.nodes[a.first];
                for (int k = 0; k < 3; k++) { m.ctr[k][0] = pa.ctr[k][a.second]; m.hal[k][0] = pa.hal[k][a.second]; }
                m.c0 = a.second == 0 ? pa.c0 : pa.c1;
                if (b) { const BvhNode &pb = bb.nodes[b->first]; for (int k = 0; k < 3; k++) { m.ctr[k][1] = pb.ctr[k][b->second]; m.hal[k][1] = pb.hal[k][b->second]; } m.c1 = b->second == 0 ? pb.c0 : pb.c1; }
                else { for (int k = 0; k < 3; k++) { m.ctr[k][1] = 0; m.hal[k][1] = -3e38f; } m.c1 = kNoChild; }
                return m;
            };
            for (size_t k = 0; k < sib.size(); k += 2) syn.push_back(make(sib[k], k + 1 < sib.size() ? &sib[k + 1] : nullptr));
            // initial stack: the synthetic nodes and the node where the chain stopped
            for (size_t k = 0; k < syn.size(); k++) stack[sp++] = bb.nnodes + (int32_t)k;
            if (last >= 0) cur = last; else cur = stack[--sp];
        }
        for (;;) {
            while (cur >= 0) {
                const BvhNode &n = cur >= bb.nnodes ? syn[cur - bb.nnodes] : bb.nodes[cur]; nodes++;
                {   // is this node one of the chain that contains an end point?
                    bool at_v = false, at_t = false;
                    for (int ch = 0; ch < 2; ch++) {
                        auto in = [&](V q, float m) { return std::fabs(q.x - n.ctr[0][ch]) <= n.hal[0][ch] + m && std::fabs(q.y - n.ctr[1][ch]) <= n.hal[1][ch] + m && std::fabs(q.z - n.ctr[2][ch]) <= n.hal[2][ch] + m; };
                        if (in(vp, 0.f)) at_v = true; if (in(pp, 0.f)) at_t = true;
                    }
                    if (at_v) S.node_at_vpl++; if (at_t) S.node_at_tile++;
                }
                int p0 = 0, p1 = 0;
                for (int l = 0; l < 64; l++) {
                    for (int ch = 0; ch < 2; ch++) {
                        float ax = n.ctr[0][ch] * ivx[l] + nox[l], ay = n.ctr[1][ch] * ivy[l] + noy[l], az = n.ctr[2][ch] * ivz[l] + noz[l];
                        float bx = n.hal[0][ch] * std::fabs(ivx[l]), by = n.hal[1][ch] * std::fabs(ivy[l]), bz = n.hal[2][ch] * std::fabs(ivz[l]);
                        float tn = clamp01(std::max(std::max(ax - bx, ay - by), az - bz)), tf = clamp01(std::min(std::min(ax + bx, ay + by), az + bz));
                        if (tn < tf) { if (ch == 0) p0++; else p1++; }
                    }
                }
                if (p0 == 0 && p1 == 0) { cur = kNoChild; break; }
                if (p0 == 0) { cur = n.c1; continue; }
                if (p1 == 0) { cur = n.c0; continue; }
                bool first0 = p0 >= p1;
                if (order == 1) {            // larger surface area first (SATO)
                    auto area = [&](int ch) { float dx = n.hal[0][ch], dy = n.hal[1][ch], dz = n.hal[2][ch]; return dx * dy + dy * dz + dz * dx; };
                    first0 = area(0) >= area(1);
                } else if (order == 2) {     // always child 0 (the builder's order)
                    first0 = true;
                } else if (order == 3) {     // nearer to the VPL first (box centre distance)
                    auto dist = [&](int ch) { float dx = n.ctr[0][ch] - vp.x, dy = n.ctr[1][ch] - vp.y, dz = n.ctr[2][ch] - vp.z; return dx * dx + dy * dy + dz * dz; };
                    first0 = dist(0) <= dist(1);
                } else if (order == 4) {     // nearer to the tile first
                    auto dist = [&](int ch) { float dx = n.ctr[0][ch] - pp.x, dy = n.ctr[1][ch] - pp.y, dz = n.ctr[2][ch] - pp.z; return dx * dx + dy * dy + dz * dz; };
                    first0 = dist(0) <= dist(1);
                } else if (order == 5) {     // smaller box first (denser geometry: more likely to occlude per visit)
                    auto area = [&](int ch) { float dx = n.hal[0][ch], dy = n.hal[1][ch], dz = n.hal[2][ch]; return dx * dy + dy * dz + dz * dx; };
                    first0 = area(0) <= area(1);
                }
                stack[sp++] = first0 ? n.c1 : n.c0; cur = first0 ? n.c0 : n.c1;
            }
            if (cur != kNoChild) {
                uint32_t id = (uint32_t)~cur, block = id >> 2, cnt = (id & 3u) + 1u; leaves++; pairs += cnt > 2 ? 2 : 1;
                {
                    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
                    for (uint32_t k = 0; k < cnt; k++) { const TriFlat &t = bb.tri_flat[block * 4 + k];
                        for (int c = 0; c < 3; c++) { float a = t.p0[c], b = t.p0[c] + t.e0[c], e = t.p0[c] - t.e1[c]; lo[c] = std::min(lo[c], std::min(a, std::min(b, e))); hi[c] = std::max(hi[c], std::max(a, std::max(b, e))); } }
                    auto inside = [&](V q, float m) { return q.x >= lo[0] - m && q.x <= hi[0] + m && q.y >= lo[1] - m && q.y <= hi[1] + m && q.z >= lo[2] - m && q.z <= hi[2] + m; };
                    if (inside(vp, 0.01f)) S.leaf_at_vpl++; else if (inside(pp, tile_r * 1.5f)) S.leaf_at_tile++;
                }
                bool leaf_counted = false;
                for (int l = 0; l < 64; l++) if (alive[l]) {
                    bool h = false;
                    for (uint32_t k = 0; k < cnt; k++) if (tri_hit(bb.tri_flat[block * 4 + k], vp, d[l], tmin, tmax)) h = true;
                    if (h && !leaf_counted) { S.leaf_hit++; leaf_counted = true; }
                    if (h) { alive[l] = false; nalive--; nox[l] = noy[l] = noz[l] = INFINITY; }
                }
                if (nalive == 0) break;
            }
            if (sp == 0) break;
            cur = stack[--sp];
        }
        S.walks++; S.nodes += nodes; S.leaves += leaves; S.pairs += pairs;
        if (leaves == 0) { S.empty++; S.empty_nodes += nodes; }
        if (nalive == 0) S.full++;
    }
    std::printf("walks %.0f  nodes/walk %.2f  leaves/walk %.2f  pairs/walk %.2f  empty %.3f (nodes %.2f)  fully occluded %.3f  est VALU/walk %.0f\n",
                S.walks, S.nodes / S.walks, S.leaves / S.walks, S.pairs / S.walks, S.empty / S.walks, S.empty_nodes / std::max(S.empty, 1.0), S.full / S.walks,
                15.7 * S.nodes / S.walks + 54.0 * S.pairs / S.walks);
    if (chain) std::printf("   chain: %.2f sibling subtrees per VPL walk\n", S.chain_len / S.walks);
    std::printf("   leaves: at VPL %.2f  at tile %.2f  with a hit %.2f /walk;  node visits with a child box holding the VPL %.2f, the tile centre %.2f\n", S.leaf_at_vpl / S.walks, S.leaf_at_tile / S.walks, S.leaf_hit / S.walks, S.node_at_vpl / S.walks, S.node_at_tile / S.walks);
    return 0;
}
