// Entry cuts of the VPL / VSL gather (kernels.h CutArgs): where the packet walks of a tile group start, per VPL.
//
// The shadow segments between one VPL and the pixels of 2 x 2 neighbouring tiles lie inside a thin pyramid with its apex at the VPL.
// Walking the tree once with that pyramid -- lane = VPL, one box-against-pyramid test per child instead of 64 ray-against-box tests
// -- finds the few subtrees any of the group's 4 x 64 segments can reach; the four (tile, VPL) packet walks then start there.  Half
// of all walks of the bench configuration find their cut empty (nothing between the VPL and the tiles) and never touch the tree.
// The reference has no counterpart: OptiX traced every shadow ray of vplSplat from the root (rt/lighttracing.cu:290-294).
//
// Conservative by construction: a segment point x that lies on a triangle lies in that triangle's leaf box and in every ancestor's
// box (boxes are padded, bvh_build.cpp), inside the box of the segments' end points and on the inner side of the four planes, so no
// subtree that holds an occluder of any of the segments is ever dropped; all bounds carry margins far above the rounding of their
// arithmetic.  The walks' results therefore stay bit-identical to a walk from the root (tests/test_gpu_parity.py::
// test_visibility_is_bit_exact, tests/test_gpu_bvh.py, the configuration tests).
#include "device_common.hpp"
#include "kernels.h"

namespace evplp {

struct CutFrustum {
    V3 p;                        // apex (the VPL)
    float lo[3], hi[3];          // box of all segment end points, padded
    V3 pl[4]; float tol[4];      // inner normals of the four side planes (through the apex) and the slack of their tests
    bool planes;                 // false: the bundle is too wide for a pyramid (a VPL beside or inside the group): box only
};
// is the box (centre c, half-size h) outside?
EV_DEV bool cut_outside(const CutFrustum &F, const float c[3], const float h[3]) {
    bool out = false;
#pragma unroll
    for (int k = 0; k < 3; k++) out = out || (c[k] - h[k] > F.hi[k]) || (c[k] + h[k] < F.lo[k]);
    if (F.planes) {
        const V3 rel = v3(c[0] - F.p.x, c[1] - F.p.y, c[2] - F.p.z);
        // slack: the rounding of `far` below is a few ulps of |n|_1 (|rel|_1 + |h|_1)
        const float mag = fabsf(rel.x) + fabsf(rel.y) + fabsf(rel.z) + h[0] + h[1] + h[2];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const V3 n = F.pl[q];
            const float far = __builtin_fmaf(fabsf(n.z), h[2], __builtin_fmaf(fabsf(n.y), h[1], __builtin_fmaf(fabsf(n.x), h[0], __builtin_fmaf(n.z, rel.z, __builtin_fmaf(n.y, rel.y, n.x * rel.x)))));
            out = out || (__builtin_fmaf(mag, F.tol[q], far) < 0.0f);
        }
    }
    return out;
}

constexpr int kCutRing = kCutEntries;
// The descent and the output of one lane: breadth-first (ring) refinement of the cut of frustum F over the tree, then the cut -- nearest
// entry first -- as synthetic nodes in `slot`.  s_ref / s_src: the wave's ring buffers in LDS ([entry][lane]).
EV_DEV void cut_descend_and_store(const BvhNode *nodes, const CutFrustum &F, char *slot, bool work, int lane, int32_t (*s_ref)[64], uint32_t (*s_src)[64]) {
    // ---- the cut: breadth-first (ring) refinement
    int head = 0, count = 0;
    auto ring_push = [&](int32_t ref, uint32_t src) { const int at = (head + count) & (kCutRing - 1); s_ref[at][lane] = ref; s_src[at][lane] = src; count++; };
    // surviving children of an inner node (no arrays indexed by a count: they would live in scratch)
    struct Kids { int32_t r0, r1; bool k0, k1; };
    auto expand = [&](int32_t node) -> Kids {
        const float4 *q4 = reinterpret_cast<const float4 *>(nodes + node);
        const float4 n0 = q4[0], n1 = q4[1], n2 = q4[2], n3 = q4[3];
        // BvhNode: ctr[3][2] = n0.xyzw n1.xy ; hal[3][2] = n1.zw n2.xyzw ; c0 c1 = n3.xy
        const float c0[3] = { n0.x, n0.z, n1.x }, c1[3] = { n0.y, n0.w, n1.y };
        const float h0[3] = { n1.z, n2.x, n2.z }, h1[3] = { n1.w, n2.y, n2.w };
        Kids k; k.r0 = __float_as_int(n3.x); k.r1 = __float_as_int(n3.y);
        k.k0 = k.r0 != kNoChild && !cut_outside(F, c0, h0);
        k.k1 = k.r1 != kNoChild && !cut_outside(F, c1, h1);
        return k;
    };
    if (work) {
        {
            const Kids k = expand(0);
            if (k.k0) ring_push(k.r0, 0u);
            if (k.k1) ring_push(k.r1, 1u);
        }
        int leaves_in_row = 0;
        while (count > 0 && leaves_in_row < count) {
            const int32_t ref = s_ref[head][lane]; const uint32_t src = s_src[head][lane];
            if (ref < 0) { head = (head + 1) & (kCutRing - 1); count--; ring_push(ref, src); leaves_in_row++; continue; }    // a leaf stays in the cut
            const Kids k = expand(ref);
            if (count - 1 + (int)k.k0 + (int)k.k1 > kCutEntries) break;
            head = (head + 1) & (kCutRing - 1); count--;
            if (k.k0) ring_push(k.r0, (uint32_t)ref << 1);
            if (k.k1) ring_push(k.r1, ((uint32_t)ref << 1) | 1u);
            leaves_in_row = 0;
        }
        // ---- the cut, nearest entry first (the walk visits the synthetic nodes in slot order and ends as soon as every lane is occluded:
        // 14.8 against 16.0 node visits per walk in the CPU replay): distance of every entry's box from the apex, sorting network on
        // (distance, entry) in registers
        auto box_of = [&](int k, float c[3], float h[3], int32_t &ref) {
            const int at = (head + k) & (kCutRing - 1);
            ref = s_ref[at][lane]; const uint32_t src = s_src[at][lane];
            const float4 *q4 = reinterpret_cast<const float4 *>(nodes + (src >> 1));
            const float4 n0 = q4[0], n1 = q4[1], n2 = q4[2];
            const bool ch = (src & 1u) != 0u;
            c[0] = ch ? n0.y : n0.x; c[1] = ch ? n0.w : n0.z; c[2] = ch ? n1.y : n1.x;
            h[0] = ch ? n1.w : n1.z; h[1] = ch ? n2.y : n2.x; h[2] = ch ? n2.w : n2.z;
        };
        float dist[kCutEntries]; int order[kCutEntries];
#pragma unroll
        for (int k = 0; k < kCutEntries; k++) { order[k] = k; dist[k] = 3.0e38f; }
        // (wave-uniform exits: half of the cuts are empty, and whole waves of them -- neighbouring groups, one VPL -- are common)
        const bool sort_needed = __builtin_amdgcn_ballot_w64(count > 1) != 0ull;
#pragma unroll
        for (int k = 0; k < kCutEntries; k++) {
            // (guards, not breaks: the loop must unroll completely or dist[] / order[] are indexed dynamically and land in scratch)
            if (sort_needed && __builtin_amdgcn_ballot_w64(k < count) != 0ull && k < count) {
                float c[3], h[3]; int32_t r; box_of(k, c, h, r);
                const float ex = fmaxf(fabsf(F.p.x - c[0]) - h[0], 0.f), ey = fmaxf(fabsf(F.p.y - c[1]) - h[1], 0.f), ez = fmaxf(fabsf(F.p.z - c[2]) - h[2], 0.f);
                dist[k] = ex * ex + ey * ey + ez * ez;
            }
        }
#define EV_CX(i_, j_) { const bool sw = dist[j_] < dist[i_]; const float td = sw ? dist[j_] : dist[i_]; dist[j_] = sw ? dist[i_] : dist[j_]; dist[i_] = td; \
                        const int to = sw ? order[j_] : order[i_]; order[j_] = sw ? order[i_] : order[j_]; order[i_] = to; }
        static_assert(kCutEntries == 8, "the sorting network below is the 19-comparator network for 8 keys");
        if (sort_needed) {
            EV_CX(0, 1) EV_CX(2, 3) EV_CX(4, 5) EV_CX(6, 7)
            EV_CX(0, 2) EV_CX(1, 3) EV_CX(4, 6) EV_CX(5, 7)
            EV_CX(1, 2) EV_CX(5, 6) EV_CX(0, 4) EV_CX(3, 7)
            EV_CX(1, 5) EV_CX(2, 6)
            EV_CX(1, 4) EV_CX(3, 6)
            EV_CX(2, 4) EV_CX(3, 5)
            EV_CX(3, 4)
        }
#undef EV_CX
        // ---- synthetic nodes: entries (2 s, 2 s + 1) of the sorted cut -> node s; node 0 carries the node count in its first padding word
        const int nsyn = (count + 1) >> 1;
        float4 *out = reinterpret_cast<float4 *>(slot);
        if (nsyn == 0) out[3] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s2 = 0; s2 < kCutNodes; s2++) {
            if (__builtin_amdgcn_ballot_w64(s2 < nsyn) != 0ull && s2 < nsyn) {
                float c[2][3], h[2][3]; int32_t r[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (2 * s2 + e < count) box_of(order[2 * s2 + e], c[e], h[e], r[e]);
                    else { r[e] = kNoChild; c[e][0] = c[e][1] = c[e][2] = 0.f; h[e][0] = h[e][1] = h[e][2] = -3.0e38f; }
                }
                out[4 * s2 + 0] = make_float4(c[0][0], c[1][0], c[0][1], c[1][1]);
                out[4 * s2 + 1] = make_float4(c[0][2], c[1][2], h[0][0], h[1][0]);
                out[4 * s2 + 2] = make_float4(h[0][1], h[1][1], h[0][2], h[1][2]);
                out[4 * s2 + 3] = make_float4(__int_as_float(r[0]), __int_as_float(r[1]), __int_as_float(nsyn), 0.f);
            }
        }
    }
}

// One lane = one (tile group, VPL): build the pyramid, refine the cut breadth-first (ring buffer of its own in LDS), write the
// synthetic nodes.  A wave = ONE VPL and a block of 8 x 8 neighbouring tile groups: the 64 pyramids share their apex and point at
// neighbouring parts of the screen, so they walk the same part of the tree for about the same number of steps (first version, lane =
// VPL for one group: 64 unrelated apexes per wave, half of them done after a few steps -- 39.5 % of the lanes busy, 7.6 ms; the kernel
// is bound by vector-instruction issue, profiles/r04a_bench_ir_pmc_first_cuts.txt).  Blocks are ordered VPL-fastest: the waves of a
// block of groups run back to back.
__global__ __launch_bounds__(64) void gather_cut_kernel(CutArgs a) {
    __shared__ int32_t s_ref[kCutRing][64];      // child reference of a cut entry (inner node index >= 0, leaf < 0)
    __shared__ uint32_t s_src[kCutRing][64];     // where its box is: parent node index << 1 | child
    const int lane = threadIdx.x;
    const uint32_t nvpl = *a.nvpl;
    const uint32_t i = blockIdx.x % a.vpl_stride, gb = blockIdx.x / a.vpl_stride;
    if (i >= nvpl) return;                                               // (wave-uniform)
    const int gblocks_x = (a.groups_x + 7) >> 3;
    const int gx = (int)(gb % (uint32_t)gblocks_x) * 8 + (lane & 7), gyl = (int)(gb / (uint32_t)gblocks_x) * 8 + (lane >> 3);
    const bool live = gx < a.groups_x && gyl < a.groups_y;
    if (__builtin_amdgcn_ballot_w64(live) == 0ull) return;
    const uint32_t g = (uint32_t)(min(gyl, a.groups_y - 1) * a.groups_x + min(gx, a.groups_x - 1));
    const int gy = a.group_row_first + min(gyl, a.groups_y - 1);
    const int gxc = min(gx, a.groups_x - 1);
    const float4 pv = reinterpret_cast<const float4 *>(a.vpls + i)[0];
    CutFrustum F; F.p = v3(pv.x, pv.y, pv.z);

    // ---- the group's tiles: union box, then the pyramid around the 8 corners of every tile's box
    const int gw = 1 << a.gw_log2, gh = 1 << a.gh_log2;
    float ulo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, uhi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int ty = gy * gh; ty < min((gy + 1) * gh, a.tiles_y); ty++)
        for (int tx = gxc * gw; tx < min((gxc + 1) * gw, a.tiles_x); tx++) {
            const float4 lo = a.tile_box[2 * (ty * a.tiles_x + tx)], hi = a.tile_box[2 * (ty * a.tiles_x + tx) + 1];
            if (lo.x > hi.x) continue;                                   // a tile without a pixel in the image
            ulo[0] = fminf(ulo[0], lo.x); ulo[1] = fminf(ulo[1], lo.y); ulo[2] = fminf(ulo[2], lo.z);
            uhi[0] = fmaxf(uhi[0], hi.x); uhi[1] = fmaxf(uhi[1], hi.y); uhi[2] = fmaxf(uhi[2], hi.z);
        }
    const bool any_tile = ulo[0] <= uhi[0];
    char *const slot = a.cuts + ((size_t)g * a.vpl_stride + i) * (size_t)kCutSlotBytes;
    if (live && !any_tile) reinterpret_cast<float4 *>(slot)[3] = make_float4(0.f, 0.f, 0.f, 0.f);    // count 0 (dwords 12..15: c0, c1, count, -)
    const bool work = live && any_tile;
    if (!any_tile) { ulo[0] = ulo[1] = ulo[2] = 0.f; uhi[0] = uhi[1] = uhi[2] = 0.f; }              // (keeps the arithmetic below finite; nothing is written)
    {
        // end points of the segments: e = P (1 - t) + t x for t = tmin and t = tmax, x in the union box -- linear in x, so the box of
        // the end points follows from the box of the pixels; padded by 1e-5 of its size and of the coordinates' magnitude
        const float pp[3] = { F.p.x, F.p.y, F.p.z };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float t0 = 0.0001f, t1 = 1.0f - 0.0001f;
            const float a0 = pp[k] * (1.0f - t0) + t0 * ulo[k], b0 = pp[k] * (1.0f - t0) + t0 * uhi[k];
            const float a1 = pp[k] * (1.0f - t1) + t1 * ulo[k], b1 = pp[k] * (1.0f - t1) + t1 * uhi[k];
            const float l = fminf(a0, a1), h = fmaxf(b0, b1);
            const float pad = 1.0e-5f * ((h - l) + fabsf(l) + fabsf(h)) + 1.0e-30f;
            F.lo[k] = l - pad; F.hi[k] = h + pad;
        }
    }
    {
        const V3 cen = v3(0.5f * (ulo[0] + uhi[0]), 0.5f * (ulo[1] + uhi[1]), 0.5f * (ulo[2] + uhi[2]));
        V3 m = cen - F.p;
        const float ml2 = dot(m, m);
        bool planes = ml2 > 1.0e-30f;
        m = m * __builtin_amdgcn_rsqf(fmaxf(ml2, 1.0e-30f));
        const V3 ax = fabsf(m.x) < 0.6f ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f);
        V3 u = cross(m, ax); u = u * __builtin_amdgcn_rsqf(fmaxf(dot(u, u), 1.0e-30f));
        const V3 w = cross(m, u);
        float amin = 3.0e38f, amax = -3.0e38f, bmin = 3.0e38f, bmax = -3.0e38f;
        // the pyramid around the 8 corners of the UNION box (round 4: the 32 corners of the four tile boxes were 40 % of this kernel and
        // bought 0.4 % fewer node visits -- 16.73 against 16.79 per walk in the CPU replay)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const V3 d = v3(((q & 1) ? uhi[0] : ulo[0]) - F.p.x, ((q & 2) ? uhi[1] : ulo[1]) - F.p.y, ((q & 4) ? uhi[2] : ulo[2]) - F.p.z);
            const float dw = dot(m, d), dl2 = dot(d, d);
            // a direction more than ~87 degrees off the axis: no pyramid for this bundle
            if (!(dw > 0.0f) || dw * dw <= 0.0025f * dl2) planes = false;
            const float r = __builtin_amdgcn_rcpf(dw);
            const float ca = dot(u, d) * r, cb = dot(w, d) * r;
            amin = fminf(amin, ca); amax = fmaxf(amax, ca); bmin = fminf(bmin, cb); bmax = fmaxf(bmax, cb);
        }
        // the tangents carry ~1e-6 of relative rounding (1-ulp reciprocal, three dot products): opened by 1e-4 (|a| <= 20 here)
        const float oa = 1.0e-4f * (1.0f + fmaxf(fabsf(amin), fabsf(amax))), ob = 1.0e-4f * (1.0f + fmaxf(fabsf(bmin), fabsf(bmax)));
        amin -= oa; amax += oa; bmin -= ob; bmax += ob;
        F.planes = planes;
        F.pl[0] = u - m * amin; F.pl[1] = m * amax - u; F.pl[2] = w - m * bmin; F.pl[3] = m * bmax - w;
#pragma unroll
        for (int q = 0; q < 4; q++) F.tol[q] = 4.0e-6f * (fabsf(F.pl[q].x) + fabsf(F.pl[q].y) + fabsf(F.pl[q].z));
    }

    cut_descend_and_store(a.nodes, F, slot, work, lane, s_ref, s_src);
}

// The eye's cuts (kernels.h PrimaryCutArgs): lane = tile group, 8 x 8 neighbouring groups per wave.  The pyramid is exact: a pixel's ray
// is eye + t (S jx + U jy + F) with jx = (ndc_x - jitter_x) aspect tan(fovy / 2), so in the basis (m, u, w) = (F, S, U) its tangents ARE
// (jx, jy); the group's range of them is opened by one pixel (the jitter moves a ray by at most half a pixel, rtcomphoton.h:949).  No
// end-point box: the rays end wherever the scene is.
__global__ __launch_bounds__(64) void primary_cut_kernel(PrimaryCutArgs a) {
    __shared__ int32_t s_ref[kCutRing][64];
    __shared__ uint32_t s_src[kCutRing][64];
    const int lane = threadIdx.x;
    const int gblocks_x = (a.groups_x + 7) >> 3;
    const int gx = (int)(blockIdx.x % (uint32_t)gblocks_x) * 8 + (lane & 7), gy = (int)(blockIdx.x / (uint32_t)gblocks_x) * 8 + (lane >> 3);
    const bool live = gx < a.groups_x && gy < a.groups_y;
    const int gxc = min(gx, a.groups_x - 1), gyc = min(gy, a.groups_y - 1);
    const int gw = 8 << a.gw_log2, gh = 8 << a.gh_log2;                 // pixels per group
    const int x0 = gxc * gw, x1 = min(x0 + gw, a.st.W) - 1;
    const int ly0 = gyc * gh, ly1 = min(ly0 + gh, a.st.local_rows) - 1;
    // the group's image rows (a strip's local rows of one group are consecutive image rows: groups never span strip blocks)
    const int y0 = a.st.global_row(ly0), y1 = a.st.global_row(ly1);
    CutFrustum F; F.p = v3(a.cam.eye);
    const V3 S = v3(a.cam.s), U = v3(a.cam.u), Fw = v3(a.cam.f);
    const float px = 2.0f / (float)a.st.W, py = 2.0f / (float)a.st.H;   // one pixel in NDC
    const float ax0 = (((float)x0 + 0.5f) * px - 1.0f - px) * a.cam.aspect * a.cam.tan_half, ax1 = (((float)x1 + 0.5f) * px - 1.0f + px) * a.cam.aspect * a.cam.tan_half;
    const float by0 = (((float)min(y0, y1) + 0.5f) * py - 1.0f - py) * a.cam.tan_half, by1 = (((float)max(y0, y1) + 0.5f) * py - 1.0f + py) * a.cam.tan_half;
#pragma unroll
    for (int k = 0; k < 3; k++) { F.lo[k] = -3.0e38f; F.hi[k] = 3.0e38f; }
    F.planes = true;
    F.pl[0] = S - Fw * ax0; F.pl[1] = Fw * ax1 - S; F.pl[2] = U - Fw * by0; F.pl[3] = Fw * by1 - U;
#pragma unroll
    for (int q = 0; q < 4; q++) F.tol[q] = 4.0e-6f * (fabsf(F.pl[q].x) + fabsf(F.pl[q].y) + fabsf(F.pl[q].z));
    char *const slot = a.cuts + (size_t)(gyc * a.groups_x + gxc) * (size_t)kCutSlotBytes;
    cut_descend_and_store(a.nodes, F, slot, live, lane, s_ref, s_src);
}
void launch_primary_cuts(const PrimaryCutArgs &a, hipStream_t s) {
    const uint32_t gblocks = (uint32_t)(((a.groups_x + 7) >> 3) * ((a.groups_y + 7) >> 3));
    if (gblocks) hipLaunchKernelGGL(primary_cut_kernel, dim3(gblocks), dim3(64), 0, s, a);
}

void launch_gather_cuts(const CutArgs &a, hipStream_t s) {
    const uint32_t gblocks = (uint32_t)(((a.groups_x + 7) >> 3) * ((a.groups_y + 7) >> 3));
    if (a.vpl_stride == 0u || gblocks == 0u) return;
    hipLaunchKernelGGL(gather_cut_kernel, dim3(gblocks * a.vpl_stride), dim3(64), 0, s, a);
}

} // namespace evplp
