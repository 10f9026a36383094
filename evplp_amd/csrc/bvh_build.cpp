// Host builders for the flattened BVH consumed by the HIP traversal kernels.
// Replaces OptiX's closed-source "Trbvh" acceleration (rt/rtcomphoton/rtcomphoton.h:705-707) and the
// meshBound program (rt/triangleintersect.cu:62-81).  Two builders, one node format:
//   LBVH : 63-bit Morton codes of triangle centroids, sorted, hierarchy split at the highest
//          differing code bit (the Karras radix-tree topology, built top-down on the host).
//   SAH  : top-down binned surface-area heuristic (16 bins).
//   SBVH : the same with spatial splits (triangle references clipped at the split plane; a triangle may sit in several leaves).
// Static scenes are built once per run like the reference's accel, so the host is acceptable.
#include "evplp_types.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace evplp {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; } }
    void grow(const Box &b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float *p) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

struct TempNode { Box box; int32_t left = -1, right = -1; int32_t first = 0, count = 0; };

struct Builder {
    const float *verts;
    std::vector<int32_t> ids;        // valid original triangle ids, permuted during the build
    std::vector<Box> tbox;           // per original triangle (unpadded)
    std::vector<float> centroid;     // 3 per original triangle
    std::vector<uint64_t> morton;    // per position in ids (LBVH)
    std::vector<TempNode> nodes;
    int depth = 0;

    int32_t make_leaf(int32_t first, int32_t count) {
        TempNode n; n.first = first; n.count = count; n.box.reset();
        for (int32_t i = 0; i < count; i++) n.box.grow(tbox[ids[first + i]]);
        nodes.push_back(n);
        return (int32_t)nodes.size() - 1;
    }
    int32_t finish_inner(int32_t id) {
        TempNode &n = nodes[id];
        n.box = nodes[n.left].box; n.box.grow(nodes[n.right].box);
        return id;
    }

    // ---- LBVH -------------------------------------------------------------------------
    static uint64_t expand21(uint64_t v) {
        v &= 0x1fffffull;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    }
    int32_t lbvh_rec(int32_t first, int32_t last /*inclusive*/, int d) {
        depth = std::max(depth, d);
        int32_t count = last - first + 1;
        if (count <= kMaxLeafTris) return make_leaf(first, count);
        uint64_t a = morton[first], b = morton[last];
        int32_t split;
        if (a == b) split = (first + last) >> 1;
        else {
            int prefix = __builtin_clzll(a ^ b);
            // last index whose code shares more than `prefix` leading bits with the first code
            int32_t lo = first, hi = last;
            while (lo + 1 < hi) {
                int32_t mid = (lo + hi) >> 1;
                uint64_t x = a ^ morton[mid];
                int p = x ? __builtin_clzll(x) : 64;
                if (p > prefix) lo = mid; else hi = mid;
            }
            split = lo;
        }
        int32_t id = (int32_t)nodes.size(); nodes.emplace_back();
        int32_t l = lbvh_rec(first, split, d + 1);
        int32_t r = lbvh_rec(split + 1, last, d + 1);
        nodes[id].left = l; nodes[id].right = r;
        return finish_inner(id);
    }
    int32_t build_lbvh() {
        Box cb; cb.reset();
        for (int32_t id : ids) cb.grow(&centroid[3 * (size_t)id]);
        float ext[3]; for (int k = 0; k < 3; k++) ext[k] = std::max(cb.hi[k] - cb.lo[k], 1e-30f);
        std::vector<std::pair<uint64_t, int32_t>> keyed(ids.size());
        for (size_t i = 0; i < ids.size(); i++) {
            const float *c = &centroid[3 * (size_t)ids[i]];
            uint64_t q[3];
            for (int k = 0; k < 3; k++) {
                double t = ((double)c[k] - cb.lo[k]) / ext[k];
                t = std::min(std::max(t, 0.0), 1.0);
                q[k] = (uint64_t)std::min(t * 2097152.0, 2097151.0);
            }
            keyed[i] = { (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]), ids[i] };
        }
        std::sort(keyed.begin(), keyed.end());
        morton.resize(ids.size());
        for (size_t i = 0; i < ids.size(); i++) { morton[i] = keyed[i].first; ids[i] = keyed[i].second; }
        return lbvh_rec(0, (int32_t)ids.size() - 1, 0);
    }

    // ---- binned SAH ---------------------------------------------------------------------
    float sah_ct = 1.0f;
    int32_t peel_min = 16, peel_min_flat = 8; float peel_eps = 1e-4f, peel_frac = 0.2f, peel_cov = 0.7f, peel_big = 0.05f, root_area_sah = 0.f;
    std::vector<float> tarea;
    int32_t sah_rec(int32_t first, int32_t count, int d) {
        depth = std::max(depth, d);
        if (count <= kMaxLeafTris && count <= 2) return make_leaf(first, count);
        Box cb; cb.reset(); Box bb; bb.reset();
        for (int32_t i = 0; i < count; i++) { cb.grow(&centroid[3 * (size_t)ids[first + i]]); bb.grow(tbox[ids[first + i]]); }
        if (peel_min > 0 && count >= peel_min) {
            // face peel: triangles that lie flat on one face of this node's box go into a child of their own
            int best_face = -1; int32_t best_n = 0; float best_cov = 0.f;
            const bool big = bb.area() >= peel_big * root_area_sah;
            for (int face = 0; face < 6; face++) {
                const int axis = face >> 1; const bool hi_side = face & 1;
                const float plane = hi_side ? bb.hi[axis] : bb.lo[axis];
                const float eps = peel_eps * (bb.hi[axis] - bb.lo[axis]);
                if (!(bb.hi[axis] - bb.lo[axis] > 0.f)) continue;
                int32_t n = 0; double area = 0.0;
                for (int32_t i = 0; i < count; i++) { const int32_t id = ids[first + i]; const Box &t = tbox[id]; if (std::fabs(t.lo[axis] - plane) <= eps && std::fabs(t.hi[axis] - plane) <= eps) { n++; area += tarea[id]; } }
                if (n == 0 || n == count) continue;
                const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
                const float face_area = (bb.hi[a1] - bb.lo[a1]) * (bb.hi[a2] - bb.lo[a2]);
                const float cov = face_area > 0.f ? (float)(area / face_area) : 0.f;
                const bool ok = cov >= peel_cov && n >= peel_min_flat && (big || (float)n >= peel_frac * (float)count);
                if (ok && n > best_n) { best_n = n; best_face = face; best_cov = cov; }
            }
            if (best_face >= 0) {
                const int axis = best_face >> 1; const float plane = (best_face & 1) ? bb.hi[axis] : bb.lo[axis];
                const float eps = peel_eps * (bb.hi[axis] - bb.lo[axis]);
                auto it = std::partition(ids.begin() + first, ids.begin() + first + count, [&](int32_t id) { const Box &t = tbox[id]; return std::fabs(t.lo[axis] - plane) <= eps && std::fabs(t.hi[axis] - plane) <= eps; });
                const int32_t mid = (int32_t)(it - ids.begin());
                const int32_t id = (int32_t)nodes.size(); nodes.emplace_back();
                const int32_t l = sah_rec(first, mid - first, d + 1);
                const int32_t r = sah_rec(mid, first + count - mid, d + 1);
                nodes[id].left = l; nodes[id].right = r;
                return finish_inner(id);
            }
        }
        constexpr int NB = 16, NBMAX = NB;
        float best_cost = 3.0e38f; int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; axis++) {
            float lo = cb.lo[axis], ext = cb.hi[axis] - lo;
            if (!(ext > 0.f)) continue;
            Box bins[NBMAX]; int cnt[NBMAX];
            for (int b = 0; b < NB; b++) { bins[b].reset(); cnt[b] = 0; }
            float scale = NB / ext;
            for (int32_t i = 0; i < count; i++) {
                int32_t id = ids[first + i];
                int b = std::min(NB - 1, (int)((centroid[3 * (size_t)id + axis] - lo) * scale));
                bins[b].grow(tbox[id]); cnt[b]++;
            }
            float ra[NBMAX]; int rc[NBMAX]; Box acc; acc.reset(); int c = 0;
            for (int b = NB - 1; b > 0; b--) { acc.grow(bins[b]); c += cnt[b]; ra[b] = acc.area(); rc[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < NB - 1; b++) {
                acc.grow(bins[b]); c += cnt[b];
                if (c == 0 || rc[b + 1] == 0) continue;
                float cost = acc.area() * (float)c + ra[b + 1] * (float)rc[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        // leaf if cheaper and it fits.  Cost model in units of one triangle test: a node visit of the packet walk costs
        // sah_ct of them (20 vector + 27 scalar instructions and a dependent 64-byte fetch against ~25 vector instructions per
        // triangle with the plane-distance pre-test)
        if (count <= kMaxLeafTris) {
            float leaf_cost = (float)count * bb.area();
            if (best_axis < 0 || best_cost + sah_ct * bb.area() >= leaf_cost) return make_leaf(first, count);
        }
        int32_t mid;
        if (best_axis >= 0) {
            float lo = cb.lo[best_axis], scale = NB / (cb.hi[best_axis] - lo);
            auto it = std::partition(ids.begin() + first, ids.begin() + first + count, [&](int32_t id) {
                int b = std::min(NB - 1, (int)((centroid[3 * (size_t)id + best_axis] - lo) * scale));
                return b <= best_bin;
            });
            mid = (int32_t)(it - ids.begin());
        } else mid = first;
        if (mid == first || mid == first + count) {
            // all centroids coincide (or partition failed): median split on index order
            mid = first + count / 2;
        }
        int32_t id = (int32_t)nodes.size(); nodes.emplace_back();
        int32_t l = sah_rec(first, mid - first, d + 1);
        int32_t r = sah_rec(mid, first + count - mid, d + 1);
        nodes[id].left = l; nodes[id].right = r;
        return finish_inner(id);
    }

    // ---- SAH with spatial splits (SBVH) ---------------------------------------------------
    // Stich, Friedrich, Dietrich 2009: besides the object split (partition of the references by centroid), every node
    // tries a SPATIAL split -- a plane that cuts straddling triangles into two references with clipped boxes -- whenever the
    // children of the best object split overlap.  Long or diagonal triangles (table tops, blinds, walls next to clutter)
    // otherwise force large, overlapping boxes that every shadow segment in the neighbourhood has to enter.  A triangle
    // may end up in several leaves; the walks do not mind (any-hit: the same answer; closest hit: ties keep the lowest
    // original index).
    struct Ref { int32_t tri; Box box; };
    float root_area = 0.f;
    size_t ref_budget = 0, refs_total = 0;       // duplication is capped
    // bounds of triangle `tri` clipped to lo <= x[axis] <= hi (Sutherland-Hodgman against the two planes)
    Box clip_tri(int32_t tri, int axis, float lo, float hi) const {
        const float *v = verts + 9 * (size_t)tri;
        float poly[8][3], tmp[8][3]; int n = 3;
        for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) poly[i][k] = v[3 * i + k];
        for (int side = 0; side < 2; side++) {
            const float plane = side == 0 ? lo : hi, sign = side == 0 ? 1.f : -1.f;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const float *a = poly[i], *b = poly[(i + 1) % n];
                const float da = sign * (a[axis] - plane), db = sign * (b[axis] - plane);
                if (da >= 0.f) { for (int k = 0; k < 3; k++) tmp[m][k] = a[k]; m++; }
                if ((da > 0.f && db < 0.f) || (da < 0.f && db > 0.f)) {
                    const float t = da / (da - db);
                    for (int k = 0; k < 3; k++) tmp[m][k] = a[k] + t * (b[k] - a[k]);
                    tmp[m][axis] = plane; m++;
                }
            }
            n = m;
            for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) poly[i][k] = tmp[i][k];
            if (n == 0) break;
        }
        Box b; b.reset();
        for (int i = 0; i < n; i++) b.grow(poly[i]);
        return b;
    }
    static Box intersect(const Box &a, const Box &b) {
        Box r; for (int k = 0; k < 3; k++) { r.lo[k] = std::max(a.lo[k], b.lo[k]); r.hi[k] = std::min(a.hi[k], b.hi[k]); }
        return r;
    }
    static bool valid(const Box &b) { return b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]; }
    int32_t sbvh_leaf(const std::vector<Ref> &refs) {
        TempNode n; n.first = (int32_t)ids.size(); n.count = (int32_t)refs.size(); n.box.reset();
        for (const Ref &r : refs) { ids.push_back(r.tri); n.box.grow(r.box); }
        nodes.push_back(n);
        return (int32_t)nodes.size() - 1;
    }
    int32_t sbvh_rec(std::vector<Ref> &refs, int d) {
        depth = std::max(depth, d);
        const int32_t count = (int32_t)refs.size();
        if (count <= 2) return sbvh_leaf(refs);
        Box cb; cb.reset(); Box bb; bb.reset();
        for (const Ref &r : refs) {
            float c[3]; for (int k = 0; k < 3; k++) c[k] = 0.5f * (r.box.lo[k] + r.box.hi[k]);
            cb.grow(c); bb.grow(r.box);
        }
        constexpr int NB = 16;
        // object split
        float obj_cost = 3.0e38f; int obj_axis = -1, obj_bin = -1; Box obj_l, obj_r;
        for (int axis = 0; axis < 3; axis++) {
            const float lo = cb.lo[axis], ext = cb.hi[axis] - lo;
            if (!(ext > 0.f)) continue;
            Box bins[NB]; int cnt[NB];
            for (int b = 0; b < NB; b++) { bins[b].reset(); cnt[b] = 0; }
            const float scale = NB / ext;
            for (const Ref &r : refs) {
                const int b = std::min(NB - 1, (int)((0.5f * (r.box.lo[axis] + r.box.hi[axis]) - lo) * scale));
                bins[b].grow(r.box); cnt[b]++;
            }
            Box rb[NB]; int rc[NB]; Box acc; acc.reset(); int c = 0;
            for (int b = NB - 1; b > 0; b--) { acc.grow(bins[b]); c += cnt[b]; rb[b] = acc; rc[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < NB - 1; b++) {
                acc.grow(bins[b]); c += cnt[b];
                if (c == 0 || rc[b + 1] == 0) continue;
                const float cost = acc.area() * (float)c + rb[b + 1].area() * (float)rc[b + 1];
                if (cost < obj_cost) { obj_cost = cost; obj_axis = axis; obj_bin = b; obj_l = acc; obj_r = rb[b + 1]; }
            }
        }
        // spatial split: only where the object split leaves overlapping children, and while the duplication budget lasts
        float sp_cost = 3.0e38f; int sp_axis = -1; float sp_pos = 0.f;
        bool try_spatial = refs_total < ref_budget;
        if (try_spatial && obj_axis >= 0) {
            const Box ov = intersect(obj_l, obj_r);
            try_spatial = valid(ov) && ov.area() > 1e-5f * root_area;
        }
        if (try_spatial) {
            for (int axis = 0; axis < 3; axis++) {
                const float lo = bb.lo[axis], ext = bb.hi[axis] - lo;
                if (!(ext > 0.f)) continue;
                Box bins[NB]; int enter[NB], leave[NB];
                for (int b = 0; b < NB; b++) { bins[b].reset(); enter[b] = leave[b] = 0; }
                const float scale = NB / ext, width = ext / NB;
                for (const Ref &r : refs) {
                    const int b0 = std::min(NB - 1, std::max(0, (int)((r.box.lo[axis] - lo) * scale)));
                    const int b1 = std::min(NB - 1, std::max(b0, (int)((r.box.hi[axis] - lo) * scale)));
                    enter[b0]++; leave[b1]++;
                    if (b0 == b1) { bins[b0].grow(r.box); continue; }
                    for (int b = b0; b <= b1; b++) {
                        Box c = intersect(clip_tri(r.tri, axis, lo + width * (float)b, lo + width * (float)(b + 1)), r.box);
                        if (valid(c)) bins[b].grow(c);
                    }
                }
                Box rb[NB]; int rc[NB]; Box acc; acc.reset(); int c = 0;
                for (int b = NB - 1; b > 0; b--) { acc.grow(bins[b]); c += leave[b]; rb[b] = acc; rc[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < NB - 1; b++) {
                    acc.grow(bins[b]); c += enter[b];
                    if (c == 0 || rc[b + 1] == 0 || c == count || rc[b + 1] == count) continue;   // a plane that separates nothing
                    const float cost = acc.area() * (float)c + rb[b + 1].area() * (float)rc[b + 1];
                    if (cost < sp_cost) { sp_cost = cost; sp_axis = axis; sp_pos = lo + width * (float)(b + 1); }
                }
            }
        }
        const float best_cost = std::min(obj_cost, sp_cost);
        if (count <= kMaxLeafTris) {
            const float leaf_cost = (float)count * bb.area();
            if (best_cost >= 3.0e38f || best_cost + sah_ct * bb.area() >= leaf_cost) return sbvh_leaf(refs);
        }
        std::vector<Ref> left, right;
        if (sp_cost < obj_cost) {
            for (const Ref &r : refs) {
                if (r.box.hi[sp_axis] <= sp_pos) left.push_back(r);
                else if (r.box.lo[sp_axis] >= sp_pos) right.push_back(r);
                else {
                    Ref a = { r.tri, intersect(clip_tri(r.tri, sp_axis, -3.0e38f, sp_pos), r.box) };
                    Ref b = { r.tri, intersect(clip_tri(r.tri, sp_axis, sp_pos, 3.0e38f), r.box) };
                    const bool va = valid(a.box), vb = valid(b.box);
                    if (va) left.push_back(a);
                    if (vb) right.push_back(b);
                    if (va && vb) refs_total++;
                    if (!va && !vb) left.push_back(r);      // (numerically empty on both sides: keep it whole)
                }
            }
            if (left.empty() || right.empty() || ((int32_t)left.size() == count && (int32_t)right.size() == count)) { left.clear(); right.clear(); }
        }
        if (left.empty() && obj_axis >= 0) {
            const float lo = cb.lo[obj_axis], scale = NB / (cb.hi[obj_axis] - lo);
            for (const Ref &r : refs) {
                const int b = std::min(NB - 1, (int)((0.5f * (r.box.lo[obj_axis] + r.box.hi[obj_axis]) - lo) * scale));
                (b <= obj_bin ? left : right).push_back(r);
            }
        }
        if (left.empty() || right.empty()) {       // all centroids coincide: split the list in the middle
            left.assign(refs.begin(), refs.begin() + count / 2); right.assign(refs.begin() + count / 2, refs.end());
        }
        std::vector<Ref>().swap(refs);
        const int32_t id = (int32_t)nodes.size(); nodes.emplace_back();
        const int32_t l = sbvh_rec(left, d + 1);
        const int32_t r = sbvh_rec(right, d + 1);
        nodes[id].left = l; nodes[id].right = r;
        return finish_inner(id);
    }
    int32_t build_sbvh() {
        std::vector<Ref> refs; refs.reserve(ids.size());
        Box bb; bb.reset();
        for (int32_t t : ids) { refs.push_back({ t, tbox[t] }); bb.grow(tbox[t]); }
        root_area = bb.area();
        float dup = 0.3f;
#ifdef EVPLP_DEV_KNOBS
        if (const char *e = std::getenv("EVPLP_SBVH_DUP")) dup = (float)atof(e);
#endif
        refs_total = refs.size(); ref_budget = refs.size() + (size_t)(dup * (float)refs.size());
        ids.clear();
        return sbvh_rec(refs, 0);
    }
};

inline void precompute_tri(const float *v, TriPair *tp, int half) {
    // same operation order as the oracle's tri_test: e0 = p1-p0, e1 = p0-p2, n = cross(e1, e0)
    float e0[3], e1[3];
    for (int k = 0; k < 3; k++) { e0[k] = v[3 + k] - v[k]; e1[k] = v[k] - v[6 + k]; }
    float n[3] = { e1[1] * e0[2] - e1[2] * e0[1], e1[2] * e0[0] - e1[0] * e0[2], e1[0] * e0[1] - e1[1] * e0[0] };
    for (int k = 0; k < 3; k++) { tp->p0[k][half] = v[k]; tp->e0[k][half] = e0[k]; tp->e1[k][half] = e1[k]; tp->n[k][half] = n[k]; }
}

} // namespace

int build_bvh(const float *verts, int32_t ntri, int builder, BvhBuild *out) {
    auto t0 = std::chrono::steady_clock::now();
    Builder B; B.verts = verts;
    B.tbox.resize((size_t)std::max(ntri, 1)); B.centroid.resize(3 * (size_t)std::max(ntri, 1)); B.tarea.resize((size_t)std::max(ntri, 1));
    Box scene; scene.reset();
    for (int32_t i = 0; i < ntri; i++) {
        const float *v = verts + 9 * (size_t)i;
        // rt/triangleintersect.cu:62-81 meshBound: area = |cross(v1-v0, v2-v0)| must be > 0 and finite
        float a[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] }, b[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
        float c[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
        float area = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        Box &tb = B.tbox[i]; tb.reset(); tb.grow(v); tb.grow(v + 3); tb.grow(v + 6);
        for (int k = 0; k < 3; k++) B.centroid[3 * (size_t)i + k] = 0.5f * (tb.lo[k] + tb.hi[k]);
        B.tarea[i] = 0.5f * area;
        if (area > 0.0f && !std::isinf(area)) { B.ids.push_back(i); scene.grow(tb); }
    }
    int32_t nvalid = (int32_t)B.ids.size();
    B.nodes.reserve((size_t)3 * std::max(nvalid, 1) + 2);
    int32_t root = -1;
#ifdef EVPLP_DEV_KNOBS   // builder tuning (tools/bvh_eval is compiled with it; the product library is not)
    if (const char *e = std::getenv("EVPLP_SAH_CT")) B.sah_ct = (float)atof(e);
    if (const char *e = std::getenv("EVPLP_PEEL_MIN")) B.peel_min = atoi(e);
    if (const char *e = std::getenv("EVPLP_PEEL_FLAT")) B.peel_min_flat = atoi(e);
    if (const char *e = std::getenv("EVPLP_PEEL_FRAC")) B.peel_frac = (float)atof(e);
    if (const char *e = std::getenv("EVPLP_PEEL_EPS")) B.peel_eps = (float)atof(e);
    if (const char *e = std::getenv("EVPLP_PEEL_COV")) B.peel_cov = (float)atof(e);
    if (const char *e = std::getenv("EVPLP_PEEL_BIG")) B.peel_big = (float)atof(e);
#endif
    B.root_area_sah = scene.area();
    if (nvalid > 0) root = builder == EVPLP_BVH_SBVH ? B.build_sbvh() : builder == EVPLP_BVH_SAH ? B.sah_rec(0, nvalid, 0) : B.build_lbvh();

    // conservative padding: the device slab test is inexact, the triangle test is exact; a padded
    // box guarantees no triangle the exact test accepts is ever culled.  Error budget of the slab tests, with S = the larger of
    // the scene diagonal and the largest coordinate magnitude (the slab form plane * inv - o * inv cancels terms of the size of
    // the coordinates): ~6 roundings of terms of size <= S (6e-8 S each) + the 1-ulp reciprocal applied to a difference <= S:
    // < 7e-7 S in position.  The pad is 3x that (2e-6 S; 4e-6: +2 % / +8 % gather time on the furnished / box scene, 1e-6:
    // -1 % / -4 % but only 1.4x the budget).  It is kept this small on purpose: a shadow segment stops
    // 1e-4 of its length short of the surface it ends on (lighttracing.cu:292), and only while pad < 1e-4 |d_perp| does it stay
    // out of the leaf boxes of that surface -- with the former 2e-5 D every segment shorter than 6 units entered the leaf under
    // its end point, and a beam shaft every leaf under its tile's footprint.
    const float pad_scale = bvh_pad_scale();
    float diag = 0.f;
    if (nvalid > 0) { float dx = scene.hi[0] - scene.lo[0], dy = scene.hi[1] - scene.lo[1], dz = scene.hi[2] - scene.lo[2]; diag = std::sqrt(dx * dx + dy * dy + dz * dz); }
    float coord = 0.f;
    if (nvalid > 0) for (int k = 0; k < 3; k++) coord = std::max(coord, std::max(std::abs(scene.lo[k]), std::abs(scene.hi[k])));
    const float pad = pad_scale * std::max(diag, coord) + 1e-30f;

    // flatten: inner nodes in DFS pre-order; each inner node carries both child boxes
    std::vector<BvhNode> flat; std::vector<int32_t> order; order.reserve((size_t)nvalid);   // 4 slots per leaf, -1 = empty
    int32_t nleaves = 0;
    // centre / half-size form; the half-size is rounded up so that [ctr - hal, ctr + hal] contains the padded box
    auto set_box = [&](BvhNode &f, int child, const Box &b) {
        for (int k = 0; k < 3; k++) {
            float lo = b.lo[k] - pad, hi = b.hi[k] + pad;
            float c = 0.5f * (lo + hi);
            float h = std::max(hi - c, c - lo);
            h = h + std::abs(h) * 1e-6f + 1e-30f;
            f.ctr[k][child] = c; f.hal[k][child] = h;
        }
    };
    auto set_empty = [&](BvhNode &f, int child) { for (int k = 0; k < 3; k++) { f.ctr[k][child] = 0.f; f.hal[k][child] = -3.0e38f; } };
    struct Item { int32_t temp; int32_t parent; int which; };
    std::vector<Item> stack;
    auto emit_leaf = [&](const TempNode &n) -> int32_t {
        int32_t block = nleaves++;
        for (int32_t i = 0; i < kMaxLeafTris; i++) order.push_back(i < n.count ? B.ids[n.first + i] : -1);
        return ~((block << 2) | (n.count - 1));
    };
    if (root >= 0) {
        if (B.nodes[root].left < 0) {
            // a single leaf: wrap it into a root with an absent second child
            BvhNode r; std::memset(&r, 0, sizeof(r));
            set_box(r, 0, B.nodes[root].box); set_empty(r, 1);
            r.c0 = emit_leaf(B.nodes[root]); r.c1 = kNoChild;
            flat.push_back(r);
        } else {
            stack.push_back({ root, -1, 0 });
            while (!stack.empty()) {
                Item it = stack.back(); stack.pop_back();
                const TempNode &n = B.nodes[it.temp];
                int32_t ref;
                if (n.left < 0) ref = emit_leaf(n);
                else {
                    ref = (int32_t)flat.size();
                    BvhNode f; std::memset(&f, 0, sizeof(f));
                    set_box(f, 0, B.nodes[n.left].box); set_box(f, 1, B.nodes[n.right].box);
                    flat.push_back(f);
                    // push right first so the left subtree is emitted next (pre-order, leaves contiguous)
                    stack.push_back({ n.right, ref, 1 }); stack.push_back({ n.left, ref, 0 });
                }
                if (it.parent >= 0) { if (it.which == 0) flat[it.parent].c0 = ref; else flat[it.parent].c1 = ref; }
            }
        }
    } else {
        BvhNode r; std::memset(&r, 0, sizeof(r));
        set_empty(r, 0); set_empty(r, 1); r.c0 = r.c1 = kNoChild;
        flat.push_back(r);
    }
    out->nnodes = (int32_t)flat.size();
    out->nodes = (BvhNode *)std::malloc(sizeof(BvhNode) * flat.size());
    std::memcpy(out->nodes, flat.data(), sizeof(BvhNode) * flat.size());
    out->ntris = nvalid;
    out->leaves = (LeafBlock *)std::calloc((size_t)std::max(nleaves, 1), sizeof(LeafBlock));
    out->tri_index = (int32_t *)std::malloc(sizeof(int32_t) * std::max<size_t>(order.size(), 4));
    out->tri_flat = (TriFlat *)std::calloc(std::max<size_t>(order.size(), 4), sizeof(TriFlat));
    for (size_t i = 0; i < order.size(); i++) {
        out->tri_index[i] = order[i];
        if (order[i] >= 0) {
            TriPair &tp = out->leaves[i >> 2].pair[(i >> 1) & 1]; const int h = (int)(i & 1);
            precompute_tri(verts + 9 * (size_t)order[i], &tp, h);
            TriFlat &tf = out->tri_flat[i];
            for (int k = 0; k < 3; k++) { tf.p0[k] = tp.p0[k][h]; tf.e0[k] = tp.e0[k][h]; tf.e1[k] = tp.e1[k][h]; tf.n[k] = tp.n[k][h]; }
        }
    }
    out->nleaves = nleaves; out->depth = B.depth + 1;
    out->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

float bvh_pad_scale() {
    float pad_scale = 2e-6f;
#ifdef EVPLP_DEV_KNOBS
    if (const char *e = std::getenv("EVPLP_BVH_PAD")) pad_scale = (float)atof(e);
#endif
    return pad_scale;
}

void free_bvh(BvhBuild *b) {
    std::free(b->nodes); std::free(b->leaves); std::free(b->tri_flat); std::free(b->tri_index);
    *b = BvhBuild();
}

} // namespace evplp
