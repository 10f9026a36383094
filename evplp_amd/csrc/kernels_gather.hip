// The gather: per-pixel sum over all usable virtual lights with shadow-ray visibility.
//   gather_vpl_kernel <- splatColor + vplSplat   (rt/lighttracing.cu:348-379, 275-346)
//   gather_vsl_kernel <- splatSplotch + vslSplat (rt/lighttracing.cu:689-722, 596-686, 395-594)
//
// Work decomposition.  One work item = one wavefront = (8x8 pixel tile, split s of kVplSplit): lane =
// pixel, and the item sums the compacted VPLs i with i % kVplSplit == s.  Items write float4 partial
// sums; gather_reduce_kernel adds the kVplSplit partials of a pixel in split order and applies
// out = sum / numVplLightPaths + doAccumulate * out (lighttracing.cu:378).  The split is a compile-time
// constant, so every pixel is summed in the same order on any GPU count (bitwise reproducible), and
// the launch has 128x more, 128x shorter items than one-tile-per-workgroup: 2 M items at 1024^2, still
// 262k per GPU on an 8-GPU strip partition (item cost varies 5x across the image; short items keep the
// launch tail small).
//
// For one VPL the 64 shadow segments of a wave share their origin (the VPL) and end on neighbouring
// surface points, so the wave walks the BVH as a packet (device_common.hpp::occluded_wave): control
// state in SGPRs, node/leaf fetches are scalar loads, stack in the lanes of a VGPR, descent decided by
// ballots, lanes whose un-normalised cosine product is <= 0 (lighttracing.cu:288) never enter.
// The VPL record is wave-uniform and is fetched with scalar loads through the scalar cache (LDS
// staging with a barrier per chunk coupled the waves of a workgroup and measured 5% slower).
#include "device_common.hpp"
#include "kernels.h"

namespace evplp {

constexpr int kRecF4 = sizeof(evplp_record) / 16; // 6 float4 per record

struct Pixel {
    V3 p1, n1, rd, rs; float e; V3 wi10;
};

struct Vpl {
    V3 pos, n, flux, fdir, rd, rs; float psel, e;
};
EV_DEV Vpl load_vpl(const float4 *r) {
    Vpl v; float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5];
    v.pos = v3(a); v.n = v3(b); v.psel = b.w; v.flux = v3(c); v.fdir = v3(d); v.rd = v3(e); v.rs = v3(f); v.e = f.w;
    return v;
}

// vplSplat after the visibility test (rt/lighttracing.cu:296-345)
EV_DEV V3 vpl_shade(const evplp_frame_params &fp, float pdf_mc2, const Pixel &px, const Vpl &v, V3 v12, float c1c2) {
    // Radiance is toleranced arithmetic (stated bars: rel. L2 1e-5, 2e-4 per pixel; powf already differs between
    // glibc and ocml): 1-ulp hardware rsq / rcp instead of the IEEE-correct sqrt + 4 divisions (~50 instructions).
    float dist2 = dot(v12, v12);
    V3 wi12 = v12 * __builtin_amdgcn_rsqf(dist2);
    // Phong lobes only where they exist: rho_s = 0 makes the term exactly 0 (0 * finite) and e = 0 makes
    // powf(d, 0) exactly 1, so both shortcuts return the bits the general expression would; they remove the
    // two powf calls for Lambertian receivers / Lambertian bounce VPLs / the on-light VPL (e = I.w = 0)
    float ph2 = 0.0f;
    if (v.rs.x != 0.0f || v.rs.y != 0.0f || v.rs.z != 0.0f) {              // wave-uniform (the VPL is)
        if (v.e == 0.0f) { V3 r = reflect(-v.fdir, v.n); ph2 = fmaxf(dot(-wi12, r), 0.0f) <= 0.000001f ? 0.0f : (v.e + 2.0f) * 1.0f * EV_INV_PI * 0.5f; }
        else ph2 = phong_eval_f(-wi12, v.fdir, v.n, v.e);
    }
    float ph1 = 0.0f;
    if (ballot64(px.rs.x != 0.0f || px.rs.y != 0.0f || px.rs.z != 0.0f) != 0ull) ph1 = phong_eval_f(px.wi10, wi12, px.n1, px.e);
    V3 brdf2 = v.rd * EV_INV_PI + v.rs * ph2;
    V3 brdf1 = px.rd * EV_INV_PI + px.rs * ph1;
    float g21 = c1c2 * __builtin_amdgcn_rcpf(dist2 * dist2);
    const uint32_t mode = fp.mis_mode;
    if (mode == 0u) return v.flux * brdf1 * brdf2 * g21;
    if (mode <= 3u) {
        // pdf of having sampled this segment from the VPL side (:318-323), from terms that are already here:
        //   LambertPdfA(n2, n1, -v12) = cos1 cos2 / d^4 / pi = g21 / pi                      (rtmaterial.cuh:46-54)
        //   PhongPdfA(n2, n1, -v12, fdir) = (e + 1) / (2 pi) c^e * max(n1 . w12, 0) / d^2    (:87-102), and c is the
        //   cosine of the VPL's own Phong lobe: c^e = ph2 * 2 pi / (e + 2); zero for rho_s.x <= 1e-6 (:92), wave-uniform
        float pdf_de = g21 * EV_INV_PI * v.psel;
        if (!(v.rs.x <= 0.000001f))
            pdf_de += ph2 * ((v.e + 1.0f) * __builtin_amdgcn_rcpf(v.e + 2.0f)) * fmaxf(dot(px.n1, wi12), 0.0f) * __builtin_amdgcn_rcpf(dist2) * (1.0f - v.psel);
        float w;
        if (mode == 1u) w = fp.pdf_mc * __builtin_amdgcn_rcpf(fp.pdf_mc + pdf_de);
        else if (mode == 2u) w = fp.pdf_mc > pdf_de ? 1.0f : 0.0f;
        else { float a2 = pdf_mc2, b2 = pdf_de * pdf_de; w = a2 * __builtin_amdgcn_rcpf(a2 + b2); }   // pdfMc^2 squared on the host: a kernel-argument SGPR
        return (v.flux * w) * brdf1 * brdf2 * g21;
    }
    if (mode == 4u) return (v.flux * fminf(g21, fp.clamping_value)) * brdf1 * brdf2;
    V3 x = (brdf1 * g21) * brdf2;
    x = v3(fminf(x.x, fp.clamping_value), fminf(x.y, fp.clamping_value), fminf(x.z, fp.clamping_value));
    return v.flux * x;
}

// Tile enumeration.  Tiles are grouped into super-tiles of SW x SH tiles (SW * SH = 64, SH = as many tile rows as a row
// strip keeps adjacent); tile id = super-tile * 64 + (ty % SH) * SW + tx % SW.  One launch covers a band of super-tiles.
struct TileXY { int tx, ty; bool exists; };
EV_DEV TileXY tile_of(const GatherArgs &a, uint32_t tid) {
    const int tiles_x = (a.st.W + 7) >> 3, tiles_y = (a.st.local_rows + 7) >> 3;
    const int swl = a.super_w_log2, sw = 1 << swl;
    const int st = (int)(tid >> 6), l = (int)(tid & 63u);
    const int stx = st % a.nsx, sty = st / a.nsx;
    TileXY t;
    t.tx = (stx << swl) + (l & (sw - 1)); t.ty = sty * (64 >> swl) + (l >> swl);
    t.exists = sty < a.nsy && t.tx < tiles_x && t.ty < tiles_y;
    return t;
}
// Item order.  Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8; speed only, never
// correctness).  All items of a tile run back to back on one XCD (they share the tile's G-buffer lines, its shaft lists
// and BVH neighbourhood in that L2); consecutive TILES go to different XCDs (item cost varies by 5x across the image --
// furniture silhouettes vs open floor -- and balance beats L2 locality: per-XCD super-tiles of 2x2 / 4x4 tiles
// measured 3 % / 13 % slower in round 1).
struct Item { int x, ly, gy, group; bool in_image, has_tile; uint32_t p, tile_in_band; };   // p: pixel index in the strip (W * local_rows < 2^32)
EV_DEV Item item_setup(const GatherArgs &a, int lane) {
    const StripDev &st = a.st;
    const int groups = kVplSplit / a.splits_per_wave;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int tile_j = j / groups;
    Item t;
    t.group = j - tile_j * groups;
    t.tile_in_band = (uint32_t)(tile_j * 8 + xcd);
    const TileXY xy = tile_of(a, (uint32_t)a.band_first_super * 64u + t.tile_in_band);
    t.has_tile = t.tile_in_band < (uint32_t)a.band_supers * 64u && xy.exists;
    t.x = xy.tx * 8 + (lane & 7); t.ly = xy.ty * 8 + (lane >> 3);
    const int cly = max(min(t.ly, st.local_rows - 1), 0);
    t.gy = st.global_row(cly);
    t.in_image = t.has_tile && t.x < st.W && t.ly < st.local_rows && t.gy < st.H;
    t.p = (uint32_t)cly * (uint32_t)st.W + (uint32_t)min(t.x, st.W - 1);
    return t;
}

#ifndef EVPLP_GATHER_WAVES
#define EVPLP_GATHER_WAVES 7   // waves per SIMD (1-wave workgroups)
#endif
typedef int v8i __attribute__((ext_vector_type(8)));

// wave-uniform scalar fetch of one 96-byte record (s_load_dwordx16 + s_load_dwordx8)
EV_DEV Vpl fetch_vpl(const evplp_record *r) {
    const v16i ra = *reinterpret_cast<const v16i *>(r);
    const v8i rb = *reinterpret_cast<const v8i *>(reinterpret_cast<const int *>(r) + 16);
    Vpl v;
    v.pos = v3(f_of(ra[0]), f_of(ra[1]), f_of(ra[2])); v.n = v3(f_of(ra[4]), f_of(ra[5]), f_of(ra[6])); v.psel = f_of(ra[7]);
    v.flux = v3(f_of(ra[8]), f_of(ra[9]), f_of(ra[10])); v.fdir = v3(f_of(ra[12]), f_of(ra[13]), f_of(ra[14]));
    v.rd = v3(f_of(rb[0]), f_of(rb[1]), f_of(rb[2])); v.rs = v3(f_of(rb[4]), f_of(rb[5]), f_of(rb[6])); v.e = f_of(rb[7]);
    return v;
}

// ---------------------------------------------------------------------------------- shaft lists
// Phase 0: sub-tile boxes of every tile (one wave per tile id; see SubBound).
__global__ __launch_bounds__(64) void tile_clusters_kernel(GatherArgs a) {
    __shared__ float s_z[64];
    const int lane = threadIdx.x;
    const uint32_t tid = blockIdx.x;
    const TileXY xy = tile_of(a, tid);
    const int x = xy.tx * 8 + (lane & 7), ly = xy.ty * 8 + (lane >> 3);
    const int cly = max(min(ly, a.st.local_rows - 1), 0);
    const bool in_image = xy.exists && x < a.st.W && ly < a.st.local_rows && a.st.global_row(cly) < a.st.H;
    const size_t p = (size_t)cly * a.st.W + min(x, a.st.W - 1);
    float4 gp = make_float4(0.f, 0.f, 0.f, 0.f), gn = gp;
    if (in_image) { gp = a.g_pos[p]; gn = a.g_nrm[p]; }
    const bool lit = in_image && gp.w != 0.0f && (gn.x != 0.0f || gn.y != 0.0f || gn.z != 0.0f);
    const unsigned long long lm = ballot64(lit);
    const int nlit = (int)__builtin_popcountll(lm);
    const float big = 3.0e38f;
    // sort key: squared distance to the camera; rank among the lit pixels (ties by lane)
    const V3 cp = v3(gp) - v3(a.fp.camera_pos);
    const float z = lit ? dot(cp, cp) : big;
    int rank = 0;
    for (int j = 0; j < 64; j++) {
        const float zj = __shfl(z, j);
        rank += (zj < z || (zj == z && j < lane)) ? 1 : 0;
    }
    s_z[rank] = z;                                    // ranks are a permutation of 0..63; lit pixels come first
    __builtin_amdgcn_wave_barrier();
    // three cuts: always split the group with the largest depth extent, at its largest gap when that gap is a real
    // discontinuity (> 30 % of the group's extent), at the middle of its depth range otherwise
    int cut[kSubs + 1] = { 0, nlit, nlit, nlit, nlit };      // group g = ranks [cut[g], cut[g + 1]); unused groups are empty
    const float zr = s_z[lane], zr1 = s_z[min(lane + 1, 63)];
    for (int step = 1; step < kSubs; step++) {
        int best = -1; float best_ext = 0.0f;
        for (int g = 0; g < step; g++) {
            const int ga = cut[g], gb = cut[g + 1];
            if (gb - ga < 2) continue;
            const float e = s_z[gb - 1] - s_z[ga];
            if (e > best_ext) { best_ext = e; best = g; }
        }
        if (best < 0) break;
        const int ga = cut[best], gb = cut[best + 1];
        const bool inside = lane >= ga && lane + 1 < gb;
        float gap = inside ? zr1 - zr : -1.0f, gmax = gap;
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off));
        int c;
        if (gmax > 0.3f * best_ext) c = (int)__builtin_ctzll(ballot64(inside && gap == gmax)) + 1;
        else {
            const float mid = 0.5f * (s_z[ga] + s_z[gb - 1]);
            c = ga + (int)__builtin_popcountll(ballot64(lane >= ga && lane < gb && zr < mid));
        }
        c = max(ga + 1, min(c, gb - 1));
        for (int g = step; g > best + 1; g--) cut[g] = cut[g - 1];      // insert the cut, keep the groups ordered by depth
        cut[best + 1] = c; cut[step + 1] = nlit;
        for (int g = step + 1; g <= kSubs; g++) cut[g] = max(cut[g], cut[g - 1]);
    }
    // smallest spacing of horizontally / vertically adjacent lit pixels (the scale a compact box is measured against)
    float s2 = big;
    {
        const float rx = __shfl_down(gp.x, 1), ry = __shfl_down(gp.y, 1), rz = __shfl_down(gp.z, 1);
        const float ux = __shfl_down(gp.x, 8), uy = __shfl_down(gp.y, 8), uz = __shfl_down(gp.z, 8);
        const bool lit_r = ((lm >> ((lane + 1) & 63)) & 1ull) != 0ull && (lane & 7) != 7;
        const bool lit_u = ((lm >> ((lane + 8) & 63)) & 1ull) != 0ull && lane < 56;
        if (lit && lit_r) s2 = fminf(s2, (rx - gp.x) * (rx - gp.x) + (ry - gp.y) * (ry - gp.y) + (rz - gp.z) * (rz - gp.z));
        if (lit && lit_u) s2 = fminf(s2, (ux - gp.x) * (ux - gp.x) + (uy - gp.y) * (uy - gp.y) + (uz - gp.z) * (uz - gp.z));
    }
    for (int off = 32; off > 0; off >>= 1) s2 = fminf(s2, __shfl_xor(s2, off));
    const float lim = a.fat_ratio * 8.0f;
    SubBound mine; bool any_fat = false;
    for (int g = 0; g < kSubs; g++) {
        const bool mem = lit && rank >= cut[g] && rank < cut[g + 1];
        const unsigned long long mm = ballot64(mem);
        float lo[3] = { mem ? gp.x : big, mem ? gp.y : big, mem ? gp.z : big }, hi[3] = { mem ? gp.x : -big, mem ? gp.y : -big, mem ? gp.z : -big };
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
        uint32_t flags = 0u; float n[3] = { 0.f, 0.f, 0.f };
        if (mm != 0ull) {
            flags = kTileLit;
            const int first = (int)__builtin_ctzll(mm);
            n[0] = __shfl(gn.x, first); n[1] = __shfl(gn.y, first); n[2] = __shfl(gn.z, first);
            if (ballot64(mem && (gn.x != n[0] || gn.y != n[1] || gn.z != n[2])) == 0ull) flags |= kTileFlat;
            const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
            if (ext > 0.0f && (s2 >= big || ext * ext > lim * lim * s2)) any_fat = true;
        }
        if (lane == g) {
            for (int k = 0; k < 3; k++) { mine.lo[k] = lo[k]; mine.hi[k] = hi[k]; mine.n[k] = n[k]; }
            mine.flags = flags; mine.mem_lo = (uint32_t)mm; mine.mem_hi = (uint32_t)(mm >> 32);
        }
    }
    if (lane < kSubs) {
        if (any_fat) mine.flags |= kTileFat;
        a.tile_bounds[(size_t)tid * kSubs + lane] = mine;
    }
#if EVPLP_TRAVERSAL_STATS
    if (lane == 0 && nlit > 0) { atomicAdd(&a.counters->hist[44], 1ull); if (any_fat) atomicAdd(&a.counters->hist[43], 1ull); }
#endif
}

// Phase 1, the beam pass: one wave = (16 tiles of a super-tile, VPL).  While it walks the tree a lane is a SUB-TILE (kSubs boxes
// per tile, SubBound) and carries the SHAFT of its box -- the union of the segments from the VPL to every point of the box,
// o + t (p - o), p in [lo, hi], t in [tmin, tmax] -- so one node visit (18 vector instructions, the two-child packed slab test of
// occluded_wave with separate entry / exit reciprocals) serves 16 (tile, VPL) pairs instead of one.  Per axis the shaft occupies
// o + t [dlo, dhi] (dlo = lo - o, dhi = hi - o) and meets the node slab [ctr - hal, ctr + hal] while  t dhi >= ctr - hal - o  and
// t dlo <= ctr + hal - o:
//   dlo > 0          entry (ctr - o - hal) / dhi,  exit (ctr - o + hal) / dlo
//   dhi < 0          entry (ctr - o + hal) / dlo,  exit (ctr - o - hal) / dhi
//   dlo <= 0 <= dhi  the box straddles the origin on this axis (the VPL's coordinate lies inside the box's range: common for the
//                    metre-wide boxes of surfaces seen at grazing angles): the cross-section widens to both sides, both conditions
//                    are ENTRY conditions -- (ctr - o - hal) / dhi and (ctr - o + hal) / dlo -- and there is no exit
// i.e.  entry = max(ctr rE + cE - hal |rE|, ctr rF + cF - hal |rF|),  exit = ctr rX + cX + hal |rX|  with per-lane constants
// (the second entry form is -inf off the straddling case; waves without a straddling lane skip it: 18 instead of 26 instructions).
// A leaf is not entered.  For every tile one of whose shafts meets the leaf's (padded) box the wave switches roles -- lane =
// PIXEL of that tile -- and runs the exact any-hit predicate on the leaf's triangles for the tile's shadow segments, OR-ing the
// hits into the tile's 64-bit occlusion mask (kept in the lane of the tile's first sub-tile).  A segment can only hit a
// triangle whose leaf box it meets, and then the shaft of its sub-tile meets that box too: every (segment, triangle) pair that
// can hit is tested with the same predicate as the per-item walk, so the masks are bit-identical to it.  A tile whose lit pixels
// are all occluded stops driving the walk.  The gather then reads one 8-byte mask per (tile, VPL) and never touches the tree.
#ifndef EVPLP_BEAM_WAVES
#define EVPLP_BEAM_WAVES 6
#endif
constexpr int kBeamTiles = 64 / kSubs;      // tiles per beam wave
__global__ __launch_bounds__(64, EVPLP_BEAM_WAVES) void beam_visibility_kernel(GatherArgs a) {
    const int lane = threadIdx.x;
    const uint32_t nvpl = *a.nvpl;
    // super-tiles are dealt to XCDs (block b runs on XCD b % 8): the beams of one super-tile stay on one XCD, whose L2 then
    // holds the super-tile's G-buffer positions for all its VPLs
    const uint32_t b = blockIdx.x, xcd = b & 7u, j = b >> 3;
    const uint32_t part = j % (uint32_t)kSubs, j2 = j / (uint32_t)kSubs;            // which 16 tiles of the super-tile
    const uint32_t i = j2 % a.max_vpls, sb = (j2 / a.max_vpls) * 8u + xcd;
    if (i >= nvpl || sb >= (uint32_t)a.band_supers) return;
    const uint32_t st = (uint32_t)a.band_first_super + sb;
    const int tl = (int)part * kBeamTiles + (lane / kSubs);                          // this lane's tile inside the super-tile (0..63)
    const int first_lane = lane & ~(kSubs - 1);                                      // lane of the tile's first sub-tile: owns the tile's mask
    const uint32_t tid = st * 64u + (uint32_t)tl;
    const float4 *tbp = reinterpret_cast<const float4 *>(a.tile_bounds + (size_t)tid * kSubs + (lane & (kSubs - 1)));
    const float4 t0 = tbp[0], t1 = tbp[1], t2 = tbp[2];
    const uint32_t tflags = __float_as_uint(t0.w);
    const uint32_t mem_lo = __float_as_uint(t1.w), mem_hi = __float_as_uint(t2.w);
    // lit pixels of the whole tile = union of its sub-tiles' members (every lane of the tile gets it)
    uint32_t lit_lo = mem_lo, lit_hi = mem_hi;
    for (int off = 1; off < kSubs; off <<= 1) { lit_lo |= (uint32_t)__shfl_xor((int)lit_lo, off); lit_hi |= (uint32_t)__shfl_xor((int)lit_hi, off); }
    // the VPL: position and normal (first 32 bytes of the record), wave-uniform
    const v8i ra = *reinterpret_cast<const v8i *>(a.vpls + i);
    const V3 o = v3(f_of(ra[0]), f_of(ra[1]), f_of(ra[2])), vn = v3(f_of(ra[4]), f_of(ra[5]), f_of(ra[6]));
    const V3 lo = v3(t0), hi = v3(t1), tn = v3(t2);
    bool live = (tflags & kTileLit) != 0u && (tflags & kTileFat) == 0u;    // fat tiles are served by the per-item walk
    {
        // cosine culls with a rounding margin (the per-pixel test of lighttracing.cu:284-288 is evaluated in fp32; a box is
        // only dropped when every member's cosine is negative by far more than that arithmetic can err):
        //   VPL side:   max over the box of  n2 . (p - o)  <= -eps   ->  c2 = max(-n2 . v12, 0) = 0 for every member
        //   pixel side: boxes with one common normal:  max of  n1 . (o - p)  <= -eps  ->  c1 = 0 for every member
        const V3 c = (lo + hi) * 0.5f, h = (hi - lo) * 0.5f, co = c - o;
        const V3 aco = v3(fabsf(co.x), fabsf(co.y), fabsf(co.z));
        const V3 an2 = v3(fabsf(vn.x), fabsf(vn.y), fabsf(vn.z));
        const float m2 = dot(vn, co) + dot(an2, h), s2 = dot(an2, aco) + dot(an2, h);
        if (m2 <= -1.0e-5f * s2) live = false;      // (s2 = 0: every product n2_k (p_k - o_k) is exactly 0 for every member -> c2 = 0)
        if (tflags & kTileFlat) {
            const V3 an1 = v3(fabsf(tn.x), fabsf(tn.y), fabsf(tn.z));
            const float m1 = -dot(tn, co) + dot(an1, h), s1 = dot(an1, aco) + dot(an1, h);
            if (m1 <= -1.0e-5f * s1) live = false;
        }
    }
    // pixels that can still be lit by this VPL = members of the live sub-tiles; everything else is reported blocked:
    // unlit pixels, members of culled sub-tiles (their cosine product is 0), and -- for the gather's skip test -- whole culled tiles
    uint32_t want_lo = live ? mem_lo : 0u, want_hi = live ? mem_hi : 0u;
    for (int off = 1; off < kSubs; off <<= 1) { want_lo |= (uint32_t)__shfl_xor((int)want_lo, off); want_hi |= (uint32_t)__shfl_xor((int)want_hi, off); }
    uint32_t occ_lo = ~want_lo, occ_hi = ~want_hi;      // the tile's mask (authoritative copy in the tile's first lane)
#if EVPLP_TRAVERSAL_STATS
    uint32_t st_nodes = 0, st_leaves = 0, st_tests = 0, st_exact = 0, st_pairs = 0, st_dead = 0, my_tests = 0;
    const uint32_t st_culled = (uint32_t)__builtin_popcountll(ballot64(!live && (tflags & kTileLit) != 0u && (tflags & kTileFat) == 0u));
#endif
    if (ballot64(live) != 0ull) {
        // shaft constants in the segment's own parameter u = (t - tmin) / (tmax - tmin), as occluded_wave
        const float tmin = 0.0001f, tmax = 1.0f - 0.0001f, ku = 1.0f / (tmax - tmin);
        const float dead = __builtin_inff();
        float rE[3], cE[3], rX[3], cX[3], rF[3], cF[3];
        const float dl[3] = { lo.x - o.x, lo.y - o.y, lo.z - o.z }, dh[3] = { hi.x - o.x, hi.y - o.y, hi.z - o.z }, oo[3] = { o.x, o.y, o.z };
        bool straddles = false;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float il = safe_rcp(dl[k]), ih = safe_rcp(dh[k]);
            const bool pos = dl[k] > 0.0f, neg = dh[k] < 0.0f, mid = !(pos || neg);
            // straddling: dlo <= 0 <= dhi, the reciprocals keep those signs even when an end is exactly (+-)0
            const float e = pos ? ih : neg ? il : fabsf(ih);
            const float x = pos ? il : neg ? ih : 0.0f;
            const float f = -fabsf(il);
            rE[k] = e * ku; cE[k] = live ? (-(oo[k] * e) - tmin) * ku : dead;
            rX[k] = x * ku; cX[k] = mid ? dead : (-(oo[k] * x) - tmin) * ku;
            rF[k] = mid ? f * ku : 0.0f; cF[k] = mid ? (-(oo[k] * f) - tmin) * ku : -dead;
            straddles = straddles || (mid && live);
        }
        const bool any_straddle = ballot64(straddles) != 0ull;                 // wave-uniform
        const v2f rEx = bc(rE[0]), rEy = bc(rE[1]), rEz = bc(rE[2]), aEx = bc(fabsf(rE[0])), aEy = bc(fabsf(rE[1])), aEz = bc(fabsf(rE[2]));
        const v2f rXx = bc(rX[0]), rXy = bc(rX[1]), rXz = bc(rX[2]), aXx = bc(fabsf(rX[0])), aXy = bc(fabsf(rX[1])), aXz = bc(fabsf(rX[2]));
        const v2f rFx = bc(rF[0]), rFy = bc(rF[1]), rFz = bc(rF[2]), aFx = bc(fabsf(rF[0])), aFy = bc(fabsf(rF[1])), aFz = bc(fabsf(rF[2]));
        v2f cEx = bc(cE[0]), cEy = bc(cE[1]), cEz = bc(cE[2]);
        const v2f cXx = bc(cX[0]), cXy = bc(cX[1]), cXz = bc(cX[2]);
        const v2f cFx = bc(cF[0]), cFy = bc(cF[1]), cFz = bc(cF[2]);
        // pixel role: where this lane's pixel sits inside a tile, and the super-tile's origin in tiles
        const int swl = a.super_w_log2, sw = 1 << swl;
        const int stx = (int)(st % (uint32_t)a.nsx), sty = (int)(st / (uint32_t)a.nsx);
        const int px_x = lane & 7, px_y = lane >> 3;
        const int W = a.st.W, max_row = a.st.local_rows - 1;
        const char *leaf_base = reinterpret_cast<const char *>(a.sc.leaves);
        int sp = 0, vstack = 0;
        int32_t cur = 0;  // root is always an inner node
        const char *node_base = reinterpret_cast<const char *>(a.sc.nodes);
        for (;;) {
            const v16i n = *reinterpret_cast<const v16i *>(node_base + ((uint32_t)cur << 6));
#if EVPLP_TRAVERSAL_STATS
            st_nodes++;
#endif
            const v2f cx = pk(n[0], n[1]), cy = pk(n[2], n[3]), cz = pk(n[4], n[5]);
            const v2f hx = pk(n[6], n[7]), hy = pk(n[8], n[9]), hz = pk(n[10], n[11]);
            const v2f enx = pk_fma(hx, -aEx, pk_fma(cx, rEx, cEx)), eny = pk_fma(hy, -aEy, pk_fma(cy, rEy, cEy)), enz = pk_fma(hz, -aEz, pk_fma(cz, rEz, cEz));
            const v2f exx = pk_fma(hx, aXx, pk_fma(cx, rXx, cXx)), exy = pk_fma(hy, aXy, pk_fma(cy, rXy, cXy)), exz = pk_fma(hz, aXz, pk_fma(cz, rXz, cXz));
            float e0 = fmaxf(fmaxf(enx.x, eny.x), enz.x), e1 = fmaxf(fmaxf(enx.y, eny.y), enz.y);
            if (any_straddle) {
                const v2f fx = pk_fma(hx, -aFx, pk_fma(cx, rFx, cFx)), fy = pk_fma(hy, -aFy, pk_fma(cy, rFy, cFy)), fz = pk_fma(hz, -aFz, pk_fma(cz, rFz, cFz));
                e0 = fmaxf(e0, fmaxf(fmaxf(fx.x, fy.x), fz.x)); e1 = fmaxf(e1, fmaxf(fmaxf(fx.y, fy.y), fz.y));
            }
            const float tn0 = clamp01(e0), tf0 = clamp01(fminf(fminf(exx.x, exy.x), exz.x));
            const float tn1 = clamp01(e1), tf1 = clamp01(fminf(fminf(exx.y, exy.y), exz.y));
            const bool h0 = tn0 < tf0, h1 = tn1 < tf1;
            unsigned long long m0 = ballot64(h0), m1 = ballot64(h1);
            const int32_t c0 = n[12], c1 = n[13];
            // leaf children: exact tests with pixel lanes for every tile one of whose shafts meets the leaf box
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const int32_t cc = side ? c1 : c0;
                unsigned long long m = side ? m1 : m0;
                if (cc >= 0 || m == 0ull) continue;
                if (side) m1 = 0ull; else m0 = 0ull;
                const LeafOps L = fetch_leaf(leaf_base, (uint32_t)cc);
                // pixels of the sub-tiles whose shafts met the box, per tile (OR over the tile's lanes)
                const bool hs = side ? h1 : h0;
                uint32_t hit_lo = hs ? mem_lo : 0u, hit_hi = hs ? mem_hi : 0u;
                for (int off = 1; off < kSubs; off <<= 1) { hit_lo |= (uint32_t)__shfl_xor((int)hit_lo, off); hit_hi |= (uint32_t)__shfl_xor((int)hit_hi, off); }
                // one bit per tile: its first lane
                unsigned long long mt = m;
                for (int off = 1; off < kSubs; off <<= 1) mt |= mt >> off;
                mt &= 0x1111111111111111ull;
#if EVPLP_TRAVERSAL_STATS
                st_leaves++;
#endif
                while (mt != 0ull) {
                    const int fl = (int)__builtin_ctzll(mt);       // first lane of the tile
                    mt &= mt - 1ull;
                    // the tile's pixels that still need an answer: members of the shafts that met the box, not blocked yet
                    const unsigned long long occ_t = (unsigned long long)(uint32_t)lane_read((int)occ_lo, fl) | ((unsigned long long)(uint32_t)lane_read((int)occ_hi, fl) << 32);
                    const unsigned long long hit_t = (unsigned long long)(uint32_t)lane_read((int)hit_lo, fl) | ((unsigned long long)(uint32_t)lane_read((int)hit_hi, fl) << 32);
                    const unsigned long long need = hit_t & ~occ_t;
                    if (need == 0ull) continue;
                    const int t = (int)part * kBeamTiles + fl / kSubs;
                    const int tx = (stx << swl) + (t & (sw - 1)), ty = sty * (64 >> swl) + (t >> swl);
                    const int x = min(tx * 8 + px_x, W - 1), ly = min(ty * 8 + px_y, max_row);
                    const float4 gp = a.g_pos[(size_t)ly * W + x];
                    const V3 d = v3(gp) - o;                       // == -(v.pos - p1) bit for bit (Ray(o, -v12), lighttracing.cu:292)
#if EVPLP_TRAVERSAL_STATS
                    st_tests++; st_pairs += L.cnt > 2u ? 2u : 1u; if (first_lane == fl) my_tests++;
                    bool hit = tri_pair_any(L.A, o, d, tmin, tmax, need, &st_exact);
                    if (L.cnt > 2u) hit = hit | tri_pair_any(L.B, o, d, tmin, tmax, need, &st_exact);
#else
                    bool hit = tri_pair_any(L.A, o, d, tmin, tmax, need);
                    if (L.cnt > 2u) hit = hit | tri_pair_any(L.B, o, d, tmin, tmax, need);
#endif
                    const unsigned long long hm = ballot64(hit) & need;
                    if (hm != 0ull) {
                        const bool mine = first_lane == fl;       // every lane of the tile keeps the mask up to date
                        occ_lo = mine ? (occ_lo | (uint32_t)hm) : occ_lo;
                        occ_hi = mine ? (occ_hi | (uint32_t)(hm >> 32)) : occ_hi;
                        if ((~(occ_t | hm)) == 0ull) {             // every pixel of the tile is blocked: its shafts leave the walk
                            if (mine) { cEx = bc(dead); cEy = bc(dead); cEz = bc(dead); }
#if EVPLP_TRAVERSAL_STATS
                            st_dead++;
#endif
                        }
                    }
                }
            }
            const uint32_t a0 = (uint32_t)m0 | (uint32_t)(m0 >> 32), a1 = (uint32_t)m1 | (uint32_t)(m1 >> 32);
            if ((a0 | a1) != 0u) {
                if (a0 == 0u) { cur = c1; continue; }
                if (a1 == 0u) { cur = c0; continue; }
                const uint32_t p0 = (uint32_t)__builtin_popcountll(m0), p1 = (uint32_t)__builtin_popcountll(m1);
                const bool first0 = p0 >= p1;
                vstack = lane_write(first0 ? c1 : c0, sp, vstack);
                sp++;
                cur = first0 ? c0 : c1;
                continue;
            }
            if (sp == 0) break;
            sp--;
            cur = lane_read(vstack, sp);
        }
    }
    if ((lane & (kSubs - 1)) == 0)
        a.vis[(size_t)i * ((size_t)a.band_supers * 64u) + (size_t)sb * 64u + (size_t)tl] = (unsigned long long)occ_lo | ((unsigned long long)occ_hi << 32);
#if EVPLP_TRAVERSAL_STATS
    if (lane == 0) {
        atomicAdd(&a.counters->hist[35], 1ull); atomicAdd(&a.counters->hist[36], (unsigned long long)st_nodes);
        atomicAdd(&a.counters->hist[37], (unsigned long long)st_leaves); atomicAdd(&a.counters->hist[38], (unsigned long long)st_tests);
        atomicAdd(&a.counters->hist[39], (unsigned long long)st_exact); atomicAdd(&a.counters->hist[40], (unsigned long long)(st_pairs - st_exact));
        atomicAdd(&a.counters->hist[41], (unsigned long long)st_dead); atomicAdd(&a.counters->hist[42], (unsigned long long)st_culled);
        if (a.dbg) atomicAdd(&a.dbg[(size_t)a.nsx * a.nsy * 64 + i], st_tests);
    }
    if (a.dbg && (lane & (kSubs - 1)) == 0) atomicAdd(&a.dbg[tid], my_tests);
#endif
}

// Phase 2.  One item = (tile, group of splits_per_wave consecutive splits): lane = pixel.  With the beam pass the wave reads the
// (tile, VPL) occlusion mask (one s_load_dwordx2) and only shades; without it (a.vis == nullptr) it walks the tree itself.
template <bool kBeam>
__global__ __launch_bounds__(64, kBeam ? 8 : EVPLP_GATHER_WAVES) void gather_vpl_kernel(GatherArgs a) {
    __shared__ float s_lvl[6 * 192];
    const int lane = threadIdx.x;
    const Item t = item_setup(a, lane);
    if (!t.has_tile) return;   // padding of the super-tile grid
    // with the beam pass two launches cover the image: <true> shades the tiles the beams served, <false> walks for the fat tiles
    if (a.vis) {
        const uint32_t tflags = a.tile_bounds[((size_t)a.band_first_super * 64u + t.tile_in_band) * kSubs].flags;
        if (((tflags & kTileFat) != 0u) == kBeam) return;
    }
    const uint32_t p = t.p;

    Pixel px;
    float4 gp = a.g_pos[p], gn = a.g_nrm[p], gd = a.g_dif[p], gs = a.g_phg[p];
    px.p1 = v3(gp); px.n1 = v3(gn); px.rd = v3(gd); px.rs = v3(gs); px.e = gs.w;
    px.wi10 = normalize(v3(a.fp.camera_pos) - px.p1);  // lighttracing.cu:363
    const bool valid = t.in_image && gp.w != 0.0f;      // stencil test, lighttracing.cu:354

    const uint32_t nvpl = *a.nvpl;
    const int k = a.splits_per_wave;
    const unsigned long long *vis_row = kBeam ? a.vis + t.tile_in_band : nullptr;
    const size_t vis_stride = (size_t)a.band_supers * 64u;    // masks per VPL
    const uint32_t bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane >= 32 ? 1u << (lane - 32) : 0u;
    V3 total = v3(0.f, 0.f, 0.f);
    uint32_t rays = 0, shaded = 0;
    for (int jj = 0; jj < k; jj++) {
        const uint32_t split = (uint32_t)(t.group * k + jj);
        V3 result = v3(0.f, 0.f, 0.f);
        for (uint32_t i = split; i < nvpl; i += kVplSplit) {
            const Vpl v = fetch_vpl(a.vpls + i);
            V3 v12 = v.pos - px.p1;                                         // :282
            float c1 = fmaxf(dot(px.n1, v12), 0.0f);
            float c2 = fmaxf(-dot(v.n, v12), 0.0f);
            float c1c2 = c1 * c2;
            bool active = valid && !(c1c2 <= 0.0f);                         // :288
            if (ballot64(active) == 0ull) continue;
            rays += active ? 1u : 0u;
            // Ray(photon.mPosition, -v12, 1, 0.0001, 1 - 0.0001)  :292
            bool occ;
            if (kBeam) {
                const unsigned long long m = vis_row[(size_t)i * vis_stride];
                occ = (((uint32_t)m & bit_lo) | ((uint32_t)(m >> 32) & bit_hi)) != 0u;
                if (ballot64(active && !occ) == 0ull) continue;
            } else {
#if EVPLP_TRAVERSAL_STATS
                WalkStats ws = { 0u, 0u, 0u };
                occ = occluded_wave(a.sc, v.pos, -v12, 0.0001f, 1.0f - 0.0001f, active, &ws);
                if (lane == 0) {
                    atomicAdd(&a.counters->nodes, (unsigned long long)ws.nodes);
                    atomicAdd(&a.counters->hist[min(ws.leaves, 31u)], 1ull);
                    atomicAdd(&a.counters->hist[32], 1ull);
                    atomicAdd(&a.counters->hist[33], (unsigned long long)ws.pairs);
                    if (ballot64(active && !occ) == 0ull) atomicAdd(&a.counters->hist[34], 1ull);
                }
#else
                occ = occluded_wave(a.sc, v.pos, -v12, 0.0001f, 1.0f - 0.0001f, active);
#endif
            }
            if (active && !occ) { result = result + vpl_shade(a.fp, a.pdf_mc2, px, v, v12, c1c2); shaded++; }
        }
        // fold the split sums in the fixed balanced-tree order: a binary counter whose level j holds the sum of 2^j splits.
        // The levels live in LDS ([level][component][lane], touched once per split): registers are what limits occupancy here.
        {
            int lev = 0;
            while ((jj >> lev) & 1) {       // wave-uniform
                float *q = s_lvl + lev * 192 + lane;
                result = v3(q[0], q[64], q[128]) + result;
                lev++;
            }
            float *q = s_lvl + lev * 192 + lane;
            q[0] = result.x; q[64] = result.y; q[128] = result.z;
        }
        total = result;     // after the last jj (k - 1 = all ones) this is the sum of all k splits
    }
    // per-lane statistics ride in the unused fourth component: shadow rays | unoccluded pairs << 16 (both < 65536 per item)
    if (t.in_image) a.partial[(size_t)t.group * a.partial_stride + p] = make_float4(total.x, total.y, total.z, __uint_as_float(rays | (shaded << 16)));
}

// out = (balanced-tree sum of the per-group partials) / numVplLightPaths + doAccumulate * out   (lighttracing.cu:378);
// the shadow-ray / unoccluded-pair counts of the items are summed here too (64 counter shards, summed by the host)
__global__ __launch_bounds__(256) void gather_reduce_kernel(GatherArgs a, int stencil_test) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)a.st.W * a.st.local_rows;
    unsigned long long rays = 0ull, shaded = 0ull;
    bool writes = i < n;
    if (writes) {
        const int ly = (int)(i / a.st.W);
        if (a.st.global_row(ly) >= a.st.H) writes = false;
        else if (stencil_test && a.g_pos[i].w == 0.0f) writes = false;      // splatColor returns before writing (:354)
    }
    if (writes) {
        const int groups = kVplSplit / a.splits_per_wave;
        V3 r = v3(0.f, 0.f, 0.f), lv0 = r, lv1 = r, lv2 = r, lv3 = r, lv4 = r, lv5 = r, lv6 = r;
        for (int g = 0; g < groups; g++) {
            float4 q = a.partial[(size_t)g * a.partial_stride + i];
            const uint32_t st = __float_as_uint(q.w);
            rays += st & 0xffffu; shaded += st >> 16;
            r = v3(q.x, q.y, q.z);
            // binary counter over g (level j holds the sum of 2^j consecutive partials): merge while the low bits of g are ones
#define EV_MERGE(L, NEXT) if (((g >> L) & 1) == 0) lv##L = r; else { r = lv##L + r; NEXT }
            EV_MERGE(0, EV_MERGE(1, EV_MERGE(2, EV_MERGE(3, EV_MERGE(4, EV_MERGE(5, EV_MERGE(6, ;)))))))
#undef EV_MERGE
        }
        const float inv = (float)a.fp.num_vpl_light_paths, acc = (float)a.fp.do_accumulate;
        float4 old = a.out[i];
        a.out[i] = make_float4(r.x / inv + acc * old.x, r.y / inv + acc * old.y, r.z / inv + acc * old.z, 0.0f + acc * old.w);
    }
    __shared__ unsigned long long s_sum[2];
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0ull;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); shaded += __shfl_down(shaded, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_sum[0], rays); atomicAdd(&s_sum[1], shaded); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int shard = blockIdx.x & (kCounterShards - 1);
        atomicAdd(&a.counters->shard_rays[shard], s_sum[0]);
        atomicAdd(&a.counters->shard_shaded[shard], s_sum[1]);
    }
}

// ------------------------------------------------------------------- light-subpath windows
// "lvcphotonfam" (rt/lvclighttracing.cu:348-384): every pixel gathers the usable records of numVplLightPaths
// consecutive light paths starting at its own random path.  Neighbouring pixels see different records, so
// there is no shared origin to build a packet on: one shadow ray per lane, per-lane LDS stack (the authors
// note the variant is experimental and slower than the plain gather for the same reason).
#ifndef EVPLP_LVC_WAVES
#define EVPLP_LVC_WAVES 6   // 5 = 56.7 ms, 6 = 53.5, 7 = 54.4, 8 = 55.9 (1024^2, 64-path windows)
#endif
__global__ __launch_bounds__(64, EVPLP_LVC_WAVES) void gather_lvc_kernel(GatherArgs a, const evplp_record *records) {
    extern __shared__ int32_t lds_stack[];   // [bvh_depth + 2][64 lanes]
    __shared__ unsigned long long s_stats[2];
    const int lane = threadIdx.x;
    if (lane < 2) s_stats[lane] = 0ull;      // single-wave workgroup: in-order LDS, no barrier needed
    const int tiles_x = (a.st.W + 7) >> 3;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    if (x >= a.st.W || ly >= a.st.local_rows) return;
    const int y = a.st.global_row(ly);
    if (y >= a.st.H) return;
    const size_t p = (size_t)ly * a.st.W + x;
    Pixel px;
    float4 gp = a.g_pos[p], gn = a.g_nrm[p], gd = a.g_dif[p], gs = a.g_phg[p];
    if (gp.w == 0.0f) return;                                            // stencil (:354)
    px.p1 = v3(gp); px.n1 = v3(gn); px.rd = v3(gd); px.rs = v3(gs); px.e = gs.w;
    px.wi10 = normalize(v3(a.fp.camera_pos) - px.p1);
    int32_t *stack = lds_stack + lane;

    Rng rng; rng_init(rng, (uint32_t)y * (uint32_t)a.st.W + (uint32_t)x, a.fp.rng_seed, 0x4c564300u);   // :369-370
    const uint32_t offset = (uint32_t)(fminf(rng_uniform(rng), 0.999999f) * (float)a.fp.num_light_paths);  // :372
    V3 result = v3(0.f, 0.f, 0.f);
    unsigned long long rays = 0, pairs = 0;
    for (uint32_t i = 0; i < a.fp.num_vpl_light_paths; i++) {
        const uint32_t path = (i + offset) % a.fp.num_light_paths;
        for (uint32_t j = 0; j < a.fp.photons_per_path; j++) {
            const evplp_record *r = records + (size_t)path * a.fp.photons_per_path + j;
            if ((r->flags & EVPLP_USABLE_VPL) == 0u) continue;
            pairs++;
            const Vpl v = load_vpl(reinterpret_cast<const float4 *>(r));
            V3 v12 = v.pos - px.p1;
            float c1 = fmaxf(dot(px.n1, v12), 0.0f);
            float c2 = fmaxf(-dot(v.n, v12), 0.0f);
            float c1c2 = c1 * c2;
            if (c1c2 <= 0.0f) continue;
            rays++;
            if (occluded_lane<64>(a.sc, v.pos, -v12, 0.0001f, 1.0f - 0.0001f, stack)) continue;
            result = result + vpl_shade(a.fp, a.pdf_mc2, px, v, v12, c1c2);
        }
    }
    const float inv = (float)a.fp.num_vpl_light_paths, acc = (float)a.fp.do_accumulate;
    float4 old = a.out[p];
    a.out[p] = make_float4(result.x / inv + acc * old.x, result.y / inv + acc * old.y, result.z / inv + acc * old.z, 0.0f + acc * old.w);
    // statistics: per-wave partial sums through LDS (lanes of masked-out pixels have left), one global atomic each
    atomicAdd(&s_stats[0], rays); atomicAdd(&s_stats[1], pairs);
    __threadfence_block();
    if ((int)__ffsll((long long)__ballot(1)) - 1 == lane) { atomicAdd(&a.counters->rays, s_stats[0]); atomicAdd(&a.counters->pairs, s_stats[1]); }
}

void launch_gather_lvc(const GatherArgs &a, const evplp_record *records, hipStream_t s) {
    int tiles_x = (a.st.W + 7) / 8, tiles_y = (a.st.local_rows + 7) / 8;
    if (tiles_x * tiles_y == 0) return;
    hipLaunchKernelGGL(gather_lvc_kernel, dim3(tiles_x * tiles_y), dim3(64), lane_stack_bytes(a.sc), s, a, records);
}

// ------------------------------------------------------------------------------------ VSL
// The VSL estimators are ALU-bound (up to 101 sample iterations x 3 estimators per lit pair, each with
// powf / sinf / cosf).  They use hardware transcendentals (v_log_f32 / v_exp_f32 / v_sin_f32 / v_cos_f32):
// x^y = exp2(y log2 x), sin(2 pi u) = v_sin(u).  The estimator is Monte-Carlo noise-limited; the result is
// compared with the oracle (accurate libm) under the VSL tolerance (rel. L2 1e-3, 2 % per pixel).  Light
// tracing and the VPL gather keep the accurate functions.
#ifndef EVPLP_VSL_FAST
#define EVPLP_VSL_FAST 1
#endif
namespace vslm {
#if EVPLP_VSL_FAST
EV_DEV float fpow(float x, float y) { return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
EV_DEV float sin2pi(float u) { return __builtin_amdgcn_sinf(u); }   // v_sin_f32 takes revolutions
EV_DEV float cos2pi(float u) { return __builtin_amdgcn_cosf(u); }
// 1-ulp hardware reciprocal / square roots instead of the IEEE-correct expansions (~10 instructions each)
EV_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
EV_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
EV_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
EV_DEV float rcp(float x) { return 1.0f / x; }
EV_DEV float rsq(float x) { return 1.0f / sqrtf(x); }
EV_DEV float fsqrt(float x) { return sqrtf(x); }
EV_DEV float fpow(float x, float y) { return powf(x, y); }
EV_DEV float sin2pi(float u) { return sinf(2.0f * EV_PI * u); }
EV_DEV float cos2pi(float u) { return cosf(2.0f * EV_PI * u); }
#endif
EV_DEV V3 fnormalize(V3 v) { return v * rsq(dot(v, v)); }
EV_DEV Onb fonb_make(V3 n) {                                         // optixu Onb
    Onb o; o.n = n;
    if (fabsf(n.x) > fabsf(n.z)) o.b = v3(-n.y, n.x, 0.0f);
    else o.b = v3(0.0f, -n.z, n.y);
    o.b = fnormalize(o.b);
    o.t = cross(o.b, o.n);
    return o;
}
EV_DEV V3 lambert_sample(V3 &out, float &pdfw, V3 normal, V3 rho_d, Rng &rng) {   // :56-66
    float u1 = rng_uniform(rng);
    float u2 = rng_uniform(rng);
    float r = fsqrt(u1);
    V3 p; p.x = r * cos2pi(u2); p.y = r * sin2pi(u2);
    p.z = fsqrt(fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    Onb o = fonb_make(normal);
    out = onb_inverse(o, p);
    pdfw = fmaxf(dot(out, normal), 0.f) * EV_INV_PI;
    return rho_d;
}
EV_DEV V3 phong_sample(V3 &out, float &pdfw, V3 in, V3 normal, V3 rho_s, float e, Rng &rng) {   // :120-154
    V3 r = reflect(-in, normal);
    float sx = rng_uniform(rng);
    float sy = rng_uniform(rng);
    float cos_t = fpow(sx, rcp(e + 1.f));
    float sin_t = fsqrt(fmaxf(1.0f - cos_t * cos_t, 0.0f));
    V3 p = v3(sin_t * cos2pi(sy), sin_t * sin2pi(sy), cos_t);
    Onb o = fonb_make(r);
    out = onb_inverse(o, p);
    float unsafe_cos = dot(out, normal);
    float cos_n = fmaxf(unsafe_cos, 0.f);
    float cos_r = fmaxf(dot(out, r), 0.f);
    if (unsafe_cos > 0.0f) pdfw = (e + 1.0f) * 0.5f * fpow(cos_r, e) * EV_INV_PI;
    else pdfw = 0.0f;
    return rho_s * ((e + 2.0f) * rcp(e + 1.0f) * cos_n);
}
} // namespace vslm
EV_DEV V3 square_to_solid_angle(float sx, float sy, float cos_half_angle_max) {  // lighttracing.cu:382-390
    float z = 1.0f - sy * (1.0f - cos_half_angle_max);
    float l = vslm::fsqrt(fmaxf(1.0f - z * z, 0.0f));
    return v3(vslm::cos2pi(sx) * l, vslm::sin2pi(sx) * l, z);
}
struct VslCtx {
    float half_cone, cos_half_cone, solid_angle, inv_solid_angle, inv_pi_r2; V3 nd12;
};
// Per-pixel and per-VSL invariants of the three estimators.  Both Phong lobes are functions of one cosine each:
//   pixel:  PhongEvalF(wi10, w, n1) and PhongPdfW(n1, w, wi10) both raise  max(w . R1, 0)  to e1,  R1 = reflect(-wi10, n1)
//           (out . reflect(-in, n) == in . reflect(-out, n));
//   VSL:    PhongEvalF(-w, fdir, n2) and PhongPdfW(n2, -w, fdir) both raise  max(-w . R2, 0)  to e2,  R2 = reflect(-fdir, n2);
// so every sampled direction costs one power per lobe instead of two, the reflections are hoisted, and a
// lobe with rho_s = 0 is skipped altogether (exact: its terms are multiplied by 0 / PhongPdfW returns 0 for
// rho_s.x <= 1e-6).  The reference normalises already-unit vectors again inside its pdfs; that is dropped
// here (last-ulp differences, far below the VSL tolerance).
struct VslPixel { V3 R1; float psel, inv_psel, inv_1mpsel; bool dead, glossy, pdf_glossy; };
struct VslLight { V3 R2; float psel, inv_psel, inv_1mpsel; bool dead, glossy, pdf_glossy; };
EV_DEV float lobe_pow(float c, float e) { return c <= 0.000001f ? 0.0f : vslm::fpow(c, e); }
// brdf1, brdf2 and the shared MIS denominators (:433-443, 508-518, 581-591; reference quirk of SURVEY A.6: pdf2 uses the
// PIXEL's lobe-selection probability for its Lambert term and no (1 - psel) on its Phong term) for direction w = wi12
EV_DEV void vsl_terms(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, V3 w, float c1, float c2,
                      V3 *brdf1, V3 *brdf2, float &pdf1, float &pdf2) {
    float pw1 = 0.0f, pw2 = 0.0f;
    if (P.glossy) pw1 = lobe_pow(fmaxf(dot(w, P.R1), 0.0f), px.e);
    if (L.glossy) pw2 = lobe_pow(fmaxf(-dot(w, L.R2), 0.0f), v.e);
    if (brdf1) *brdf1 = px.rd * EV_INV_PI + px.rs * ((px.e + 2.0f) * pw1 * EV_INV_PI * 0.5f);
    if (brdf2) *brdf2 = v.rd * EV_INV_PI + v.rs * ((v.e + 2.0f) * pw2 * EV_INV_PI * 0.5f);
    float pp1 = P.pdf_glossy ? (px.e + 1.0f) * 0.5f * EV_INV_PI * pw1 : 0.0f;
    float pp2 = L.pdf_glossy ? (v.e + 1.0f) * 0.5f * EV_INV_PI * pw2 : 0.0f;
    pdf1 = c1 * P.psel + pp1 * (1.0f - P.psel);
    pdf2 = c2 * P.psel + pp2;
}
EV_DEV V3 vsl_sample_cone(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, const VslCtx &c, float &w, Rng &rng) {  // :395-446
    const V3 zero = v3(0.f, 0.f, 0.f);
    if (P.dead) return zero;
    (void)rng_uniform(rng);
    float ua = rng_uniform(rng);
    float ub = rng_uniform(rng);
    V3 wi12 = vslm::fnormalize(square_to_solid_angle(ua, ub, c.cos_half_cone));
    Onb o = vslm::fonb_make(c.nd12);
    wi12 = vslm::fnormalize(onb_inverse(o, wi12));
    float c1 = fmaxf(dot(px.n1, wi12), 0.0f), c2 = fmaxf(-dot(v.n, wi12), 0.0f);
    float c1c2 = c1 * c2;
    if (c1c2 <= 0.000000001f) return zero;
    V3 brdf1, brdf2; float pdf1, pdf2;
    vsl_terms(px, v, P, L, wi12, c1, c2, &brdf1, &brdf2, pdf1, pdf2);
    w = c.inv_solid_angle * vslm::rcp(pdf1 + pdf2 + c.inv_solid_angle);
    return (((v.flux * c.inv_pi_r2) * c1c2) * brdf1 * brdf2) * c.solid_angle;
}
EV_DEV V3 vsl_sample_brdf1(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, const VslCtx &c, float &w, Rng &rng) {  // :448-521
    const V3 zero = v3(0.f, 0.f, 0.f);
    if (P.dead) return zero;
    float choose = fminf(rng_uniform(rng), 0.999999f);
    V3 wi12, brdf1; float pdfw;
    if (choose < P.psel) brdf1 = vslm::lambert_sample(wi12, pdfw, px.n1, px.rd, rng) * P.inv_psel;
    else brdf1 = vslm::phong_sample(wi12, pdfw, px.wi10, px.n1, px.rs, px.e, rng) * P.inv_1mpsel;
    if (dot(wi12, c.nd12) <= c.cos_half_cone) return zero;
    float cos1 = fmaxf(dot(px.n1, wi12), 0.0f);
    if (cos1 <= 0.000000001f) return zero;
    float cos2 = fmaxf(-dot(v.n, wi12), 0.0f);
    (void)rng_uniform(rng);  // :506
    V3 brdf2; float pdf1, pdf2;
    vsl_terms(px, v, P, L, wi12, cos1, cos2, nullptr, &brdf2, pdf1, pdf2);
    w = pdf1 * vslm::rcp(pdf1 + pdf2 + c.inv_solid_angle);
    return ((v.flux * c.inv_pi_r2) * cos2) * brdf1 * brdf2;
}
EV_DEV V3 vsl_sample_brdf2(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, const VslCtx &c, float &w, Rng &rng) {  // :523-594
    const V3 zero = v3(0.f, 0.f, 0.f);
    if (L.dead) return zero;
    V3 wi21, brdf2; float pdfw;
    float choose = fminf(rng_uniform(rng), 0.999999f);
    if (choose < L.psel) brdf2 = vslm::lambert_sample(wi21, pdfw, v.n, v.rd, rng) * L.inv_psel;
    else brdf2 = vslm::phong_sample(wi21, pdfw, v.fdir, v.n, v.rs, v.e, rng) * L.inv_1mpsel;
    if (-dot(wi21, c.nd12) <= c.cos_half_cone) return zero;
    float cos2 = fmaxf(dot(v.n, wi21), 0.0f);
    if (cos2 <= 0.00000001f) return zero;
    float cos1 = fmaxf(-dot(px.n1, wi21), 0.0f);
    if (P.dead) return zero;
    (void)rng_uniform(rng);  // :579
    V3 brdf1; float pdf1, pdf2;
    vsl_terms(px, v, P, L, -wi21, cos1, cos2, &brdf1, nullptr, pdf1, pdf2);
    w = pdf2 * vslm::rcp(pdf1 + pdf2 + c.inv_solid_angle);
    return ((v.flux * c.inv_pi_r2) * cos1) * brdf1 * brdf2;
}

#ifndef EVPLP_VSL_WAVES
#define EVPLP_VSL_WAVES 8   // 5: 56.4 ms, 6: 51.0, 7: 48.0, 8: 46.0 (512^2, 760 VSLs): occupancy beats the spills it costs
#endif
__global__ __launch_bounds__(64, EVPLP_VSL_WAVES) void gather_vsl_kernel(GatherArgs a) {
    const int lane = threadIdx.x;
    const int W = a.st.W;
    const Item t = item_setup(a, lane);      // splits_per_wave = 1: group = split
    if (!t.has_tile) return;
    const uint32_t p = t.p;
    const bool in_image = t.in_image;

    Pixel px;
    float4 gp = a.g_pos[p], gn = a.g_nrm[p], gd = a.g_dif[p], gs = a.g_phg[p];
    px.p1 = v3(gp); px.n1 = v3(gn); px.rd = v3(gd); px.rs = v3(gs); px.e = gs.w;
    px.wi10 = normalize(v3(a.fp.camera_pos) - px.p1);   // :704
    const bool valid = in_image;                        // no stencil test in splatSplotch (:694-695)
    const uint32_t pixel_id = (uint32_t)t.gy * (uint32_t)W + (uint32_t)t.x;  // launchIndex.y * dim.x + launchIndex.x (:711)

    VslPixel P;
    {
        float ml = max_color(px.rd), mp = max_color(px.rs);
        P.dead = ml + mp <= 0.000001f; P.psel = ml / (mp + ml); P.inv_psel = 1.0f / P.psel; P.inv_1mpsel = 1.0f / (1.0f - P.psel);
        P.glossy = px.rs.x != 0.0f || px.rs.y != 0.0f || px.rs.z != 0.0f; P.pdf_glossy = !(px.rs.x <= 0.000001f);
        P.R1 = reflect(-px.wi10, px.n1);
    }
    const uint32_t nvpl = *a.nvpl;
    V3 result = v3(0.f, 0.f, 0.f);
    uint32_t rays = 0, nlit = 0;
    for (uint32_t i = (uint32_t)t.group; i < nvpl; i += kVplSplit) {
        const Vpl v = fetch_vpl(a.vpls + i);
        V3 v12 = v.pos - px.p1;                                       // :605
        float dist2 = dot(v12, v12);
        float dist = sqrtf(dist2);
        rays += valid ? 1u : 0u;
        bool occ = occluded_wave(a.sc, v.pos, -v12, 0.0001f, 1.0f - 0.0001f, valid);  // :612-614
        V3 nv12 = v12 / dist;
        float c1c2 = fmaxf(dot(px.n1, nv12), 0.0f) * fmaxf(-dot(v.n, nv12), 0.0f);
        bool lit = valid && !occ && !(c1c2 <= 0.000000001f);        // :619
        if (ballot64(lit) == 0ull) continue;
        if (lit) {
            nlit++;
            VslCtx c;
            float rdratio = a.fp.vsl_radius / dist;
            c.half_cone = (rdratio >= 1.0f) ? EV_PI / 2.0f : asinf(rdratio);   // :623
            c.cos_half_cone = (rdratio >= 1.0f) ? cosf(EV_PI / 2.0f) : vslm::fsqrt(1.0f - rdratio * rdratio);   // cos(asin x)
            c.solid_angle = EV_PI * 2.0f * (1.0f - c.cos_half_cone);
            c.inv_solid_angle = vslm::rcp(c.solid_angle);
            c.inv_pi_r2 = a.fp.vsl_inv_pi_radius2; c.nd12 = nv12;
            int num_samples = (int)(c.half_cone / EV_PI * 2.0f * 100.0f) + 1;  // :632
            // one RNG substream per (pixel, record): any decomposition reproduces the same numbers
            Rng rng; rng_init(rng, pixel_id, a.fp.rng_seed, 1u + a.vpl_src_index[i]);
            VslLight L;
            {
                float ml = max_color(v.rd), mp = max_color(v.rs);
                L.dead = ml + mp <= 0.000001f; L.psel = ml / (mp + ml); L.inv_psel = 1.0f / L.psel; L.inv_1mpsel = 1.0f / (1.0f - L.psel);
                L.glossy = v.rs.x != 0.0f || v.rs.y != 0.0f || v.rs.z != 0.0f; L.pdf_glossy = !(v.rs.x <= 0.000001f);
                L.R2 = reflect(-v.fdir, v.n);
            }
            V3 acc = v3(0.f, 0.f, 0.f);
            for (int s = 0; s < num_samples; s++) {
                float wc = 0.f, w1 = 0.f, w2 = 0.f;
                V3 rc = vsl_sample_cone(px, v, P, L, c, wc, rng);
                V3 r1 = vsl_sample_brdf1(px, v, P, L, c, w1, rng);
                V3 r2 = vsl_sample_brdf2(px, v, P, L, c, w2, rng);
                acc = acc + rc * wc;
                acc = acc + r1 * w1;
                acc = acc + r2 * w2;
            }
            result = result + acc * vslm::rcp((float)num_samples);
        }
    }
    if (in_image) a.partial[(size_t)t.group * a.partial_stride + p] = make_float4(result.x, result.y, result.z, __uint_as_float(rays | (nlit << 16)));
}

static dim3 gather_grid(const GatherArgs &a) {
    const int groups = kVplSplit / a.splits_per_wave;
    return dim3((unsigned)(a.band_supers * 64 * groups));       // (band tiles rounded to 8) x groups: tile = tile_j * 8 + xcd
}
void launch_tile_bounds(const GatherArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(tile_clusters_kernel, dim3((unsigned)(a.nsx * a.nsy * 64)), dim3(64), 0, s, a);
}
void launch_gather_reduce(const GatherArgs &a, int stencil_test, hipStream_t s) {
    size_t n = (size_t)a.st.W * a.st.local_rows;
    hipLaunchKernelGGL(gather_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, stencil_test);
}
void launch_beam_visibility(const GatherArgs &a, hipStream_t s) {
    const unsigned per_xcd = ((unsigned)a.band_supers + 7u) / 8u;       // super-tiles per XCD (rounded up)
    hipLaunchKernelGGL(beam_visibility_kernel, dim3(per_xcd * a.max_vpls * 8u * (unsigned)kSubs), dim3(64), 0, s, a);
}
void launch_gather_vpl_items(const GatherArgs &a, hipStream_t s) {
    if (a.vis) hipLaunchKernelGGL(gather_vpl_kernel<true>, gather_grid(a), dim3(64), 0, s, a);
    hipLaunchKernelGGL(gather_vpl_kernel<false>, gather_grid(a), dim3(64), 0, s, a);      // everything without the beam pass, the fat tiles with it
}
void launch_gather_vsl(const GatherArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(gather_vsl_kernel, gather_grid(a), dim3(64), 0, s, a);
}

} // namespace evplp
