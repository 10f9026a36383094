// The gather: per-pixel sum over all usable virtual lights with shadow-ray visibility.
//   gather_vpl_kernel <- splatColor + vplSplat   (rt/lighttracing.cu:348-379, 275-346)
//   gather_vsl_walk_kernel + gather_vsl_shade_kernel <- splatSplotch + vslSplat (rt/lighttracing.cu:689-722, 596-686, 395-594)
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
#include <algorithm>

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

// PhongEvalF (rtmaterial.cuh:112-118) with the power on the hardware transcendentals: d^e = exp2(e log2 d), d in (1e-6, 1].
// Relative error ~ 0.7 e |log2 d| 2^-22 -- 2e-6 where the lobe is still 1e-4 of its peak, less near it -- against the stated bars of
// 1e-5 (image, relative L2) and 2e-4 (pixel).  The library powf is ~170 instructions; the two of a glossy pair were 8 % of the
// gather's vector instructions on the furnished scene (every fifth object carries a Phong lobe).
EV_DEV float phong_eval_f_hw(V3 out, V3 in, V3 n, float e) {
    V3 r = reflect(-in, n);
    float d = fmaxf(dot(out, r), 0.0f);
    if (d <= 0.000001f) return 0.0f;
    return (e + 2.0f) * __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(d)) * EV_INV_PI * 0.5f;
}

// vplSplat after the visibility test (rt/lighttracing.cu:296-345)
// wi10_lds: where the caller parked the pixel's view direction ([component][lane], LDS) instead of holding it in three VGPRs
// across the walks -- it is only read for tiles with a glossy pixel.  PXL (gather_vpl_kernel, EVPLP_PX_LDS): the column continues with
// the pixel's reflectances -- [3..5] rho_d, [6..8] rho_s, [9] e -- and px.rd / px.rs / px.e are not read; `px_glossy` = some pixel of the
// tile has a Phong lobe (wave-uniform, computed once per item instead of one ballot per lit VPL).
template <bool PXL = false>
EV_DEV V3 vpl_shade(const evplp_frame_params &fp, float pdf_mc2, const Pixel &px, const Vpl &v, V3 v12, float c1c2, const float *wi10_lds = nullptr, bool px_glossy = true) {
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
        else ph2 = phong_eval_f_hw(-wi12, v.fdir, v.n, v.e);
    }
    float ph1 = 0.0f;
    V3 prs = PXL ? v3(0.f, 0.f, 0.f) : px.rs;
    if (PXL ? px_glossy : ballot64(px.rs.x != 0.0f || px.rs.y != 0.0f || px.rs.z != 0.0f) != 0ull) {
        ph1 = phong_eval_f_hw(wi10_lds ? v3(wi10_lds[0], wi10_lds[64], wi10_lds[128]) : px.wi10, wi12, px.n1, PXL ? wi10_lds[576] : px.e);
        if (PXL) prs = v3(wi10_lds[384], wi10_lds[448], wi10_lds[512]);
    }
    const V3 prd = PXL ? v3(wi10_lds[192], wi10_lds[256], wi10_lds[320]) : px.rd;
    V3 brdf2 = v.rd * EV_INV_PI + v.rs * ph2;
    V3 brdf1 = prd * EV_INV_PI + prs * ph1;
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

// Item order.  Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8; speed only, never
// correctness).  All items of a tile run back to back on one XCD (they share the tile's G-buffer lines and BVH
// neighbourhood in that L2); consecutive TILES go to different XCDs (item cost varies by 5x across the image --
// furniture silhouettes vs open floor -- and balance beats L2 locality: per-XCD super-tiles of 2x2 / 4x4 tiles
// measured 3 % / 13 % slower in round 1).  Tiles are enumerated block by block -- 8 x SH tiles, SH = as many tile rows
// as a row strip keeps adjacent (8 for a whole image) -- so that the tiles in flight at one time are 2-D neighbours and
// walk the same part of the tree: 107.6 ms against 110.8 ms for row-major order (cfg2, hard scene, same GPU).
struct Item { int x, ly, gy, group; bool in_image, has_tile; uint32_t p; };   // p: pixel index in the strip (W * local_rows < 2^32)
// Block index -> (tile, group of splits).  EVPLP_GROUP_ORDER (round 4, measured, off): with entry cuts the unit dealt to an XCD is a CUT
// GROUP of tiles, and the items that read the same cut slots -- the same splits of the group's 2 x 2 tiles -- are neighbours in launch
// order on one XCD, so that a slot comes from HBM once instead of once per tile (12.7 GB per config-#2 frame).  57.2 ms against 53.6 ms
// (box scene 21.7 / 20.5, VSL gather at 1024^2 116.2 / 113.2): as in round 1, balance beats locality -- the slots arrive through a ring
// two walks ahead, nobody waits for them, while four neighbouring tiles of equal cost on one XCD are four times the granularity
// the XCDs are balanced with.  (The other direction -- a tile's items dealt over ALL XCDs, perfect balance, no tile owns an L2 -- 54.0 / 21.5 ms.)
// RANGE: the launch covers the groups [group_first, group_first + group_count) only (the VSL kernels; the VPL gather always launches all)
#ifndef EVPLP_GROUP_ORDER
#define EVPLP_GROUP_ORDER 0
#endif
#ifndef EVPLP_XCD_TIMES
#define EVPLP_XCD_TIMES 0
#endif
#ifndef EVPLP_XCD_ROTATE
#define EVPLP_XCD_ROTATE 2
#endif
struct ItemIx { int tx, ty, sg, tile_l, blk; };      // tile, group of splits within the launch, tile index in launch order, tile block
template <bool RANGE = false>
EV_DEV ItemIx item_index(const GatherArgs &a, int b) {
    const int tiles_x = (a.st.W + 7) >> 3;
    const int groups = RANGE ? a.group_count : kVplSplit / a.splits_per_wave;      // groups of this launch
    const bool grp = EVPLP_GROUP_ORDER == 1 && a.cuts != nullptr;
    const int gwl = grp ? a.cut_gw_log2 : 0, ghl = grp ? a.cut_gh_log2 : 0, gl = gwl + ghl;
    const int xcd = b & 7, j = b >> 3;
    const int q = j & ((1 << gl) - 1);                         // tile within its cut group (fastest), then the group of splits, then the unit
    // (round 6) a.item_deal: the groups of ONE tile are consecutive workgroups and so go to all eight XCDs -- no tile owns an L2, but every XCD
    // gets an eighth of every tile.  With tiles dealt to XCDs (the default for whole images: a tile's items share its G-buffer lines and tree
    // neighbourhood in one L2, 0.7 % faster at 1024 x 1024) an XCD's share is a sum over (tiles / 8) tile costs, and those vary tenfold: over
    // the 2 048 tiles of an eight-way partition's strip the XCDs finish 10-22 % of the launch apart (tools/xcd_balance.py --strip-count 8).
    const int r = a.item_deal ? b : j >> gl;
    const int unit_j = r / groups;
    const int unit = a.item_deal ? unit_j : unit_j * 8 + xcd;  // cut group (or tile) in block order
    const int shl = a.block_h_log2, bwl = 3 - gwl, bhl = shl - ghl, nbx = (tiles_x + 7) >> 3;   // block = 8 x (1 << shl) tiles
    const int l = unit & ((1 << (bwl + bhl)) - 1);
    ItemIx ix;
    ix.blk = a.band_first * nbx + (unit >> (bwl + bhl));       // (a launch may cover a band of block rows only)
    // (EVPLP_XCD_ROTATE: the column of a block a given XCD takes rotates with the row and the block -- with XCD = column every XCD owned
    // 8-pixel-wide vertical stripes of the image, and the stripes' costs differ systematically: tools/xcd_balance.py)
    // (round 6: ... with the block's ROW counted in tile rows.  "+ blk" alone repeats with the blocks of a row -- 16 at 1024 pixels, a multiple of
    // 8 -- so that a row strip, whose blocks are one or two tile rows high, kept the stripes: EVPLP_XCD_ROTATE=1 is that formula.  Whole images
    // -- blocks of 8 tile rows -- are dealt exactly as before.)
    const int rot = EVPLP_XCD_ROTATE == 0 ? 0 : EVPLP_XCD_ROTATE == 1 ? (l >> bwl) + ix.blk : (l >> bwl) + (ix.blk % nbx) + ((ix.blk / nbx) << bhl);
    const int ux = ((ix.blk % nbx) << bwl) | ((l + rot) & ((1 << bwl) - 1)), uy = ((ix.blk / nbx) << bhl) | (l >> bwl);
    ix.tx = (ux << gwl) | (q & ((1 << gwl) - 1)); ix.ty = (uy << ghl) | (q >> gwl);
    ix.sg = r - unit_j * groups;
    ix.tile_l = (unit << gl) | q;
    return ix;
}
template <bool RANGE = false>
EV_DEV Item item_setup(const GatherArgs &a, int lane, int b) {
    const StripDev &st = a.st;
    const int tiles_x = (st.W + 7) >> 3, tiles_y = (st.local_rows + 7) >> 3;
    const int sh = 1 << a.block_h_log2, nbx = (tiles_x + 7) >> 3, nby = (tiles_y + sh - 1) >> a.block_h_log2;
    const ItemIx ix = item_index<RANGE>(a, b);
    Item t;
    t.group = (RANGE ? a.group_first : 0) + ix.sg;
    t.has_tile = ix.blk < nbx * (a.band_rows > 0 ? min(nby, a.band_first + a.band_rows) : nby) && ix.tx < tiles_x && ix.ty < tiles_y;
    t.x = ix.tx * 8 + (lane & 7); t.ly = ix.ty * 8 + (lane >> 3);
    const int cly = max(min(t.ly, st.local_rows - 1), 0);
    t.gy = st.global_row(cly);
    // (a tile whose first row lies outside the image -- a local block that holds nothing under a dealt block table, the padding blocks of the
    // round-robin deal, the rows behind a band -- has no pixel in it: rows ascend within a tile)
    t.has_tile = t.has_tile && __builtin_amdgcn_readfirstlane(t.gy) < st.H;
    t.in_image = t.has_tile && t.x < st.W && t.ly < st.local_rows && t.gy < st.H;
    t.p = (uint32_t)cly * (uint32_t)st.W + (uint32_t)min(t.x, st.W - 1);
    return t;
}

// the texel index of item_setup alone (no row-strip division): what a kernel re-derives late instead of carrying it in registers
template <bool RANGE = false>
EV_DEV uint32_t item_texel(const GatherArgs &a, int lane, int b) {
    const StripDev &st = a.st;
    const ItemIx ix = item_index<RANGE>(a, b);
    const int x = ix.tx * 8 + (lane & 7), ly = ix.ty * 8 + (lane >> 3);
    return (uint32_t)max(min(ly, st.local_rows - 1), 0) * (uint32_t)st.W + (uint32_t)min(x, st.W - 1);
}

// index of the item's tile group in the entry-cut buffer (wave-uniform)
EV_DEV uint32_t item_cut_group(const GatherArgs &a, const Item &t) {
    const int tx = __builtin_amdgcn_readfirstlane(t.x) >> 3, ty = __builtin_amdgcn_readfirstlane(t.ly) >> 3;
    return (uint32_t)(((ty >> a.cut_gh_log2) - a.cut_group_row_first) * a.cut_groups_x + (tx >> a.cut_gw_log2));
}

#ifndef EVPLP_CUT_RING
#define EVPLP_CUT_RING 3          // LDS buffers of the cut-slot ring of a gather wave (512 B each): EVPLP_CUT_RING - 1 slots in flight ahead of the walk
#endif
#ifndef EVPLP_PX_LDS
#define EVPLP_PX_LDS 1            // the pixel's reflectances parked in LDS beside its view direction: seven registers fewer across the walks
#endif
constexpr int kGatherPxFloats = 64 * (EVPLP_PX_LDS ? 10 : 3);      // [192] view direction (+ [192] rho_d, [192] rho_s, [64] e) per wavefront
#ifndef EVPLP_GATHER_WAVES
// waves per SIMD (1-wave workgroups).  cfg2 hard / easy scene -- round 3 (no cuts, 64 registers at either): 7 = 70.6 / 29.3 ms, 8 = 70.9 / 30.9.
// Round 4, with the entry cuts the walks are half as long and wait relatively more for their node fetches (76 % of the SIMD cycles issue
// a vector instruction at 7 waves): 8 waves = 53.5 / 20.5 ms against 56.6 / 21.2 -- once the kernel FITS 64 registers (EVPLP_PX_LDS above:
// with six spilled registers reloaded from scratch per lit VPL 8 waves gave 55.0 / 22.0).  LDS per wavefront: 2.5 KB of pixel data + 768 B
// per fold level + the 768-byte cut ring = 4864 B at k = 2, inside the 5120 B that 32 wavefronts per CU leave each (k >= 4: 7 waves).
#define EVPLP_GATHER_WAVES 8
#endif

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
// ... in two parts around the walk (gather_vpl_kernel): position + normal first (all the cosine test and the walk need), the
// shading fields afterwards.  Held across the walk the 24 dwords cost the walk its SGPRs: eight were spilled to VGPR lanes for
// every VPL and the node base pointer was re-loaded from the kernel arguments at every node visit.
EV_DEV void fetch_vpl_head(const evplp_record *vpls, uint32_t i, V3 &pos, V3 &n, float &psel) {
    const v8i ra = sload8(vpls, i * (uint32_t)sizeof(evplp_record));
    pos = v3(f_of(ra[0]), f_of(ra[1]), f_of(ra[2])); n = v3(f_of(ra[4]), f_of(ra[5]), f_of(ra[6])); psel = f_of(ra[7]);
}
EV_DEV void fetch_vpl_tail(const evplp_record *vpls, uint32_t i, Vpl &v) {
    const v16i rb = sload16(vpls, i * (uint32_t)sizeof(evplp_record) + 32u);
    v.flux = v3(f_of(rb[0]), f_of(rb[1]), f_of(rb[2])); v.fdir = v3(f_of(rb[4]), f_of(rb[5]), f_of(rb[6]));
    v.rd = v3(f_of(rb[8]), f_of(rb[9]), f_of(rb[10])); v.rs = v3(f_of(rb[12]), f_of(rb[13]), f_of(rb[14])); v.e = f_of(rb[15]);
}

#if EVPLP_GATHER_TIMES       // developer build (tools/gather_times.py): start / end clock (100 MHz) of every 16th item's wavefront
__device__ unsigned long long g_gather_times[2 * 131072];
extern "C" int evplp_debug_gather_times(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gather_times), sizeof(unsigned long long) * (size_t)n); }
#endif
// One item = (tile, group of splits_per_wave consecutive splits): lane = pixel.
// COST (calibration launches, evplp_calibrate_blocks): the item adds the clock ticks it was resident to its local block's counter
EV_DEV void item_cost_add(const GatherArgs &a, int ty, unsigned long long t0, int waves) {
    const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
    // (kernels of different residency add up as launch time does: ticks x 8 / waves per SIMD)
    if (threadIdx.x == 0) atomicAdd(&a.block_cost[(ty * 8) / a.st.strip_rows], dt * 8ull / (unsigned long long)waves);
}
template <bool CUT, bool COST = false>
#if EVPLP_WALK_ASM && !EVPLP_TRAVERSAL_STATS
__attribute__((amdgpu_num_vgpr(52)))      // v[52:63] belong to the hand-written node visit (device_common.hpp)
#endif
__global__ __launch_bounds__(64, EVPLP_GATHER_WAVES) void gather_vpl_kernel(GatherArgs a) {
    unsigned long long t_cost = 0ull;
    if constexpr (COST) t_cost = __builtin_amdgcn_s_memrealtime();
#if EVPLP_GATHER_TIMES
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    // dynamic LDS: [192] the view directions, then one [192] block per level of the k-split fold (log2 k + 1 of them).  Sized by k
    // because LDS is what limits occupancy next: 7 single-wavefront workgroups per SIMD fit while a workgroup stays within
    // 5120 bytes (the allocation granule of this part is 1280 bytes: 5376 bytes measured 6 % slower, one wave per SIMD fewer)
    extern __shared__ float s_dyn[];
    float *const s_wi10 = s_dyn, *const s_lvl = s_dyn + kGatherPxFloats;
    const int lane = threadIdx.x;
    const Item t = item_setup(a, lane, (int)blockIdx.x);
    if (!t.has_tile) return;   // padding of the block grid
    const uint32_t p = t.p;

    Pixel px;
    float4 gp = a.g_pos[p], gn = a.g_nrm[p], gd = a.g_dif[p], gs = a.g_phg[p];
    px.p1 = v3(gp); px.n1 = v3(gn); px.rd = v3(gd); px.rs = v3(gs); px.e = gs.w;
#ifndef EVPLP_WI10_LDS
#define EVPLP_WI10_LDS 1
#endif
    {   // lighttracing.cu:363; parked in LDS (single-wavefront workgroup: in order, no barrier)
        const V3 w = normalize(v3(a.fp.camera_pos) - px.p1);
        s_wi10[lane] = w.x; s_wi10[64 + lane] = w.y; s_wi10[128 + lane] = w.z;
        px.wi10 = EVPLP_WI10_LDS ? v3(0.f, 0.f, 0.f) : w;
    }
    bool px_glossy = true;
    if constexpr (EVPLP_PX_LDS != 0) {
        static_assert(EVPLP_WI10_LDS != 0, "EVPLP_PX_LDS continues the LDS column of the view direction");
        s_wi10[192 + lane] = px.rd.x; s_wi10[256 + lane] = px.rd.y; s_wi10[320 + lane] = px.rd.z;
        s_wi10[384 + lane] = px.rs.x; s_wi10[448 + lane] = px.rs.y; s_wi10[512 + lane] = px.rs.z; s_wi10[576 + lane] = px.e;
        px_glossy = ballot64(px.rs.x != 0.0f || px.rs.y != 0.0f || px.rs.z != 0.0f) != 0ull;
        px.rd = v3(0.f, 0.f, 0.f); px.rs = v3(0.f, 0.f, 0.f); px.e = 0.0f;
    }
    const bool valid = t.in_image && gp.w != 0.0f;      // stencil test, lighttracing.cu:354

    const uint32_t nvpl = *a.nvpl;
    const int k = a.splits_per_wave;
    const char *node_base = pinned(reinterpret_cast<const char *>(a.sc.nodes)), *leaf_base = pinned(reinterpret_cast<const char *>(a.sc.leaves));
    const evplp_record *vpls = pinned(a.vpls);
    // entry cuts of this tile's group (kernels.h CutArgs): slot i belongs to VPL i
    const char *cuts_g = nullptr;
    if constexpr (CUT) cuts_g = pinned(a.cuts + (size_t)item_cut_group(a, t) * a.cut_vpl_stride * (size_t)kCutSlotBytes);
    V3 total = v3(0.f, 0.f, 0.f);
    uint32_t rays = t.in_image ? 0x80000000u : 0u, shaded = 0;      // (rays: bit 31 = the lane's pixel is in the image, see the end)
#if EVPLP_TRAVERSAL_STATS
    int32_t cache_leaf = kNoChild; bool prev_all_occ = false;
#endif
    // The cut slots of the item's VPLs stream through a small LDS ring ahead of the walks (global_load_lds: asynchronous, counted by vmcnt,
    // no register in between).  A walk that fetched its slot itself began with two dependent scalar loads from memory no cache holds
    // (8.6 GB of slots per frame, each read once per tile): 65.6 ms against 70.6 ms without cuts although the node visits had halved.
    constexpr bool kRing = CUT && EVPLP_WALK_ASM && !EVPLP_TRAVERSAL_STATS;
    constexpr int kRingSlots = EVPLP_CUT_RING, kAhead = kRingSlots - 1;
    __shared__ float4 s_cut[kRing ? kRingSlots : 1][kCutSlotBytes / 16];
    int pj = 0; uint32_t pi = (uint32_t)(t.group * k), qslot = 0u;       // the prefetch position in the item's VPL sequence (split pj, VPL pi)
    auto ring_skip_empty = [&]() { while (pj < k && pi >= nvpl) { pj++; pi = (uint32_t)(t.group * k + pj); } };
    auto ring_issue = [&](uint32_t slot) {
        if (lane < kCutSlotBytes / 16) lds_dma16(cuts_g + (size_t)pi * kCutSlotBytes + (size_t)lane * 16u, lds_offset(&s_cut[slot][0]));
        pi += (uint32_t)kVplSplit; ring_skip_empty();
    };
    if constexpr (kRing) {
        ring_skip_empty();
        for (int q = 0; q < kAhead; q++) if (pj < k) ring_issue((uint32_t)q);
    }
    for (int jj = 0; jj < k; jj++) {
        const uint32_t split = (uint32_t)(t.group * k + jj);
        V3 result = v3(0.f, 0.f, 0.f);
        for (uint32_t i = split; i < nvpl; i += kVplSplit) {
            const float4 *cut_lds = nullptr;
            if constexpr (kRing) {
                // one more slot goes into flight (into the buffer the previous walk has read), then this walk's slot is waited for
                if (pj < k) { ring_issue((qslot + (uint32_t)kAhead) % (uint32_t)kRingSlots); wait_vmcnt<kAhead>(); }
                else wait_vmcnt0();
                cut_lds = &s_cut[qslot][0];
                qslot = (qslot + 1u) % (uint32_t)kRingSlots;
            }
            V3 vpos, vn; float vpsel; fetch_vpl_head(vpls, i, vpos, vn, vpsel);
            V3 v12 = vpos - px.p1;                                         // :282
            float c1 = fmaxf(dot_exact(px.n1, v12), 0.0f);
            float c2 = fmaxf(-dot_exact(vn, v12), 0.0f);
            float c1c2 = c1 * c2;
            bool active = valid && !(c1c2 <= 0.0f);                         // :288
            if (ballot64(active) == 0ull) continue;
            rays += active ? 1u : 0u;
            // Ray(photon.mPosition, -v12, 1, 0.0001, 1 - 0.0001)  :292
            bool occ;
            {
#if EVPLP_TRAVERSAL_STATS
                // cache simulation: would the leaf block that last occluded a lane of this item occlude this VPL's lanes too?
                unsigned long long cache_kill = 0ull;
                const unsigned long long act_m = ballot64(active);
                if (cache_leaf != kNoChild) {
                    const LeafOps L = fetch_leaf(leaf_base, (uint32_t)cache_leaf);
                    const V3 dd = -v12;
                    bool any = tri_pair_any(L.A, vpos, dd, 0.0001f, 1.0f - 0.0001f);
                    if (L.cnt > 2u) any = any | tri_pair_any(L.B, vpos, dd, 0.0001f, 1.0f - 0.0001f);
                    cache_kill = ballot64(any) & act_m;
                }
                WalkStats ws = { 0u, 0u, 0u, 0u, kNoChild, 0u };
                occ = occluded_wave<0, CUT>(node_base, leaf_base, vpos, -v12, 0.0001f, 1.0f - 0.0001f, active, &ws, cuts_g, i * (uint32_t)kCutSlotBytes);
                const bool all_occ = ballot64(active && !occ) == 0ull;
                const unsigned long long occ_m = ballot64(active && occ);
                if (lane == 0) {
                    atomicAdd(&a.counters->hist[41], (unsigned long long)__builtin_popcountll(occ_m));
                    atomicAdd(&a.counters->nodes, (unsigned long long)ws.nodes);
                    atomicAdd(&a.counters->aux, (unsigned long long)ws.syn);
                    atomicAdd(&a.counters->hist[min(ws.leaves, 31u)], 1ull);
                    atomicAdd(&a.counters->hist[32], 1ull);
                    atomicAdd(&a.counters->hist[33], (unsigned long long)ws.pairs);
                    if (all_occ) { atomicAdd(&a.counters->hist[34], 1ull); atomicAdd(&a.counters->hist[36], (unsigned long long)ws.nodes); atomicAdd(&a.counters->hist[37], (unsigned long long)ws.leaves); }
                    if (ws.leaves == 0u) atomicAdd(&a.counters->hist[35], (unsigned long long)ws.nodes);
                    if (cache_leaf != kNoChild) {
                        atomicAdd(&a.counters->hist[42], 1ull);
                        atomicAdd(&a.counters->hist[40], (unsigned long long)__builtin_popcountll(cache_kill));
                        if (cache_kill == act_m) { atomicAdd(&a.counters->hist[38], 1ull); atomicAdd(&a.counters->hist[39], (unsigned long long)ws.nodes); }
                        if (prev_all_occ) { atomicAdd(&a.counters->hist[44], 1ull); if (cache_kill == act_m) atomicAdd(&a.counters->hist[43], 1ull); }
                    }
                    atomicAdd(&a.counters->hist[45 + min(ws.nodes / 16u, 18u)], 1ull);
                }
                if (ws.hit_leaf != kNoChild) cache_leaf = ws.hit_leaf;
                prev_all_occ = all_occ;
#else
                occ = occluded_wave<EVPLP_WALK_ASM ? 52 : 0, CUT>(node_base, leaf_base, vpos, -v12, 0.0001f, 1.0f - 0.0001f, active, nullptr, cuts_g, i * (uint32_t)kCutSlotBytes, cut_lds);
#endif
            }
            const bool lit = active && !occ;
            if (ballot64(lit) == 0ull) continue;
            // (the record's head is fetched AGAIN for the shading -- it is in the scalar cache -- instead of holding its normal and lobe
            // probability in four scalar registers across the walk, which has none to spare)
            Vpl v; fetch_vpl_head(vpls, i, v.pos, v.n, v.psel); fetch_vpl_tail(vpls, i, v);
            // ... and so are the four frame parameters the shading reads (MIS mode, pdfMc, its square, the clamp): from the kernel-argument
            // segment, right here -- held in scalar registers from the top of the kernel they are spilled to VGPR lanes and read back
            // lane by lane in every branch of vpl_shade
            evplp_frame_params fps; float pdf_mc2;
            {
                const void *ka = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
                typedef int v4i_ __attribute__((ext_vector_type(4)));
                v4i_ q; int q2;
                asm volatile("s_load_dwordx4 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(q), "=&s"(q2)
                             : "s"(ka), "s"((uint32_t)(offsetof(GatherArgs, fp) + offsetof(evplp_frame_params, mis_mode))), "s"((uint32_t)offsetof(GatherArgs, pdf_mc2)) : "memory");
                fps.mis_mode = (uint32_t)q[0]; fps.pdf_mc = f_of(q[1]); fps.clamping_value = f_of(q[2]); pdf_mc2 = f_of(q2);
            }
            if (lit) { result = result + vpl_shade<EVPLP_PX_LDS != 0>(fps, pdf_mc2, px, v, v12, c1c2, EVPLP_WI10_LDS ? s_wi10 + lane : nullptr, px_glossy); shaded++; }
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
#if EVPLP_DEBUG_NAN
    if ((rays >> 31) != 0u && !(isfinite(total.x) && isfinite(total.y) && isfinite(total.z))) atomicAdd(&a.counters->nonfinite, 1ull);
#endif
#if EVPLP_XCD_TIMES
    // (developer probe, tools/xcd_balance.py: when did the last item of every XCD end?  s_memrealtime ticks at 100 MHz)
    if (lane == 0) atomicMax(&a.counters->hist[16 + (blockIdx.x & 7u)], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
#if EVPLP_GATHER_TIMES
    if (lane == 0 && (blockIdx.x & 15u) == 0u && (blockIdx.x >> 4) < 131072u) { g_gather_times[2 * (blockIdx.x >> 4)] = t_start; g_gather_times[2 * (blockIdx.x >> 4) + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // per-lane statistics ride in the unused fourth component: shadow rays | unoccluded pairs << 16 (both < 65536 per item)
    int lane_out = lane, blk_out = (int)blockIdx.x;
    asm volatile("" : "+v"(lane_out), "+s"(blk_out));   // (the store address is formed here, not carried through the walks)
    const bool in_image = (rays >> 31) != 0u;           // (parked in the counter's top bit at the start: one register fewer across the walks)
    rays &= 0xffffu;
    if (in_image) a.partial[(size_t)item_index(a, blk_out).sg * a.partial_stride + item_texel(a, lane_out, blk_out)] = make_float4(total.x, total.y, total.z, __uint_as_float(rays | (shaded << 16)));
    if constexpr (COST) item_cost_add(a, item_index(a, blk_out).ty, t_cost, EVPLP_GATHER_WAVES);
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
#ifndef EVPLP_LVC_WIDE
#define EVPLP_LVC_WIDE 0
#endif
#if EVPLP_LVC_WIDE
#define LVC_OCCLUDED occluded_lane4
#else
#define LVC_OCCLUDED occluded_lane
#endif
#ifndef EVPLP_LVC_SPEC
#define EVPLP_LVC_SPEC 0   // speculative while-while (device_common.hpp): 245 ms against 222 ms (512^2 window gather) -- an any-hit walk wants its leaf NOW; off
#endif
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
#ifndef EVPLP_LVC_NO_OFFSET
#define EVPLP_LVC_NO_OFFSET 0     // developer probe (tools/quick_bench.py --lvc, round 6): every pixel's window starts at path 0, so that the lanes of a wave walk towards the SAME record at the same time -- the per-lane walk on the packet walk's own ray population
#endif
    const uint32_t offset = EVPLP_LVC_NO_OFFSET ? 0u : (uint32_t)(fminf(rng_uniform(rng), 0.999999f) * (float)a.fp.num_light_paths);  // :372
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
            float c1 = fmaxf(dot_exact(px.n1, v12), 0.0f);
            float c2 = fmaxf(-dot_exact(v.n, v12), 0.0f);
            float c1c2 = c1 * c2;
            if (c1c2 <= 0.0f) continue;
            rays++;
            if (LVC_OCCLUDED<64, EVPLP_LVC_SPEC>(a.sc, v.pos, -v12, 0.0001f, 1.0f - 0.0001f, stack)) continue;
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
    hipLaunchKernelGGL(gather_lvc_kernel, dim3(tiles_x * tiles_y), dim3(64), EVPLP_LVC_WIDE ? lane_stack_bytes4(a.sc) : lane_stack_bytes(a.sc), s, a, records);
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
struct VslPixel { V3 R1; float psel; bool dead, glossy, pdf_glossy; };
struct VslLight { V3 R2; float psel; bool dead, glossy, pdf_glossy; };
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
// (round 4) The three estimators of one sample-iteration add their MIS-weighted terms into `acc` WITHOUT the factor every term of a
// pair shares, flux / (pi r^2): the kernel applies it once per pair after the loop.  What a sample-iteration costs is what the 99.6 %
// of BRDF samples that miss the VSL's cone cost (half-angles of ~0.06 rad), so those leave as early as possible: the cone test is
// first taken in the SAMPLING frame -- dot(p, local(nd12)) with nd12 brought into the frame once per pair -- with 1e-5 of slack, and
// only a sample that passes is turned into a world direction and takes the reference's own test (:470-474, :545-549) on it, so the
// set of accepted samples is the one the one-step code had.  The cone sample drops the two re-normalisations of vectors that are
// unit vectors by construction (:401-404: last-ulp differences, toleranced like the transcendentals).
struct VslFrames { V3 nd_n1, nd_r1, nd_n2, nd_r2; };   // nd12 in the frames of n1 / R1, -nd12 in the frames of n2 / R2 (local coordinates)
EV_DEV V3 onb_local(const Onb &o, V3 d) { return v3(dot(d, o.t), dot(d, o.b), dot(d, o.n)); }
constexpr float kConeSlack = 0.00001f;
EV_DEV V3 select3(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }   // (by value: a select between two struct members by address would put the struct into scratch)
EV_DEV V3 lambert_local(float u1, float u2) {                          // LambertSample's direction in its own frame (rtmaterial.cuh:56-66)
    float r = vslm::fsqrt(u1);
    V3 p; p.x = r * vslm::cos2pi(u2); p.y = r * vslm::sin2pi(u2);
    p.z = vslm::fsqrt(fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    return p;
}
EV_DEV V3 phong_local(float e, float sx, float sy) {                  // PhongSample's direction in the frame of the reflected direction (:120-154)
    float cos_t = vslm::fpow(sx, vslm::rcp(e + 1.f));
    float sin_t = vslm::fsqrt(fmaxf(1.0f - cos_t * cos_t, 0.0f));
    return v3(sin_t * vslm::cos2pi(sy), sin_t * vslm::sin2pi(sy), cos_t);
}
// The estimators' random numbers (round 5; the oracle's evo_vsl_rng_*, where the choice is argued): xoroshiro64, one stream per
// (pixel, record), ONE STEP PER SAMPLE -- six vector instructions and no multiply, where the PCG stream of the other kernels
// (device_common.hpp) spent 4 instructions with three integer multiplies on every draw, used or not, and 5 more on each one used:
// half of a sample-iteration's issue cycles.  The state after the step IS the sample's draws: top 24 bits of each word = its two
// uniforms, the 16 bits left over = its lobe choice.
struct VslRng { uint32_t s0, s1; };
EV_DEV void vsl_rng_step(VslRng &r) {
    const uint32_t t = r.s1 ^ r.s0;
    r.s0 = __builtin_rotateleft32(r.s0, 26) ^ t ^ (t << 9);
    r.s1 = __builtin_rotateleft32(t, 13);
}
EV_DEV float vsl_uniform(uint32_t w) { return __builtin_fmaf((float)(w >> 8), 1.0f / 16777216.0f, 1.0f / 16777216.0f); }      // (0, 1] like curand_uniform
EV_DEV float vsl_choose(const VslRng &r) { return __builtin_fmaf((float)(((r.s0 & 0xffu) << 8) | (r.s1 & 0xffu)), 1.0f / 65536.0f, 0.5f / 65536.0f); }
EV_DEV uint64_t vsl_key(uint64_t x) { return splitmix64(x); }
// seed of the pair's stream from the pixel's key (splitmix64(seed << 32 | pixel), once per lane) and the record's (once per VSL, scalar)
EV_DEV VslRng vsl_rng_seed(uint64_t pixel_key, uint64_t record_key) {
    const uint32_t a = (uint32_t)pixel_key ^ (uint32_t)record_key, b = (uint32_t)(pixel_key >> 32) ^ (uint32_t)(record_key >> 32);
    const uint64_t t = (uint64_t)a * (uint64_t)(b | 1u);
    VslRng r; r.s0 = (uint32_t)t ^ b; r.s1 = ((uint32_t)(t >> 32) ^ a) | 0x80000000u;
    return r;
}
// DIFF (wave-uniform): neither the VSL nor any lit pixel of the tile has a Phong lobe.  Then brdf1 brdf2 = rho_d1 rho_d2 / pi^2 is the same
// for every sample of a pair, both lobe-selection probabilities are 1 (the `choose` draws only advance the generator) and the MIS
// denominators are c1 + c2 + 1 / solid angle: a cone sample adds ONE scalar, c1 c2 / (c1 + c2 + 1 / solid angle), and the colour is
// applied once per pair after the loop.
template <bool DIFF>
EV_DEV void vsl_sample_cone(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, const VslCtx &c, const Onb &cone, V3 &acc, float &acc_d, VslRng &rng) {  // :395-446
    if (P.dead) return;
    vsl_rng_step(rng);
    const float ua = vsl_uniform(rng.s0), ub = vsl_uniform(rng.s1);
    const V3 wi12 = onb_inverse(cone, square_to_solid_angle(ua, ub, c.cos_half_cone));
    float c1 = fmaxf(dot(px.n1, wi12), 0.0f), c2 = fmaxf(-dot(v.n, wi12), 0.0f);
    float c1c2 = c1 * c2;
    if (c1c2 <= 0.000000001f) return;
    if constexpr (DIFF) { acc_d += c1c2 * vslm::rcp((c1 + c2) + c.inv_solid_angle); return; }
    V3 brdf1, brdf2; float pdf1, pdf2;
    vsl_terms(px, v, P, L, wi12, c1, c2, &brdf1, &brdf2, pdf1, pdf2);
    const float w = c.inv_solid_angle * vslm::rcp(pdf1 + pdf2 + c.inv_solid_angle);
    acc = acc + (brdf1 * brdf2) * ((c.solid_angle * c1c2) * w);
}
template <bool DIFF>
EV_DEV void vsl_sample_brdf1(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, const VslCtx &c, const VslFrames &F, V3 &acc, VslRng &rng) {  // :448-521
    if (P.dead) return;
    vsl_rng_step(rng);
    bool lam = true;                                    // (DIFF: choose < pSel = 1 whatever it is)
    if constexpr (!DIFF) lam = vsl_choose(rng) < P.psel;
    const float u1 = vsl_uniform(rng.s0), u2 = vsl_uniform(rng.s1);
    V3 p;
    if (DIFF || lam) p = lambert_local(u1, u2); else p = phong_local(px.e, u1, u2);
    if (!(dot(p, select3(lam, F.nd_n1, F.nd_r1)) > c.cos_half_cone - kConeSlack)) return;
    V3 wi12, brdf1;
    if (lam) {
        wi12 = onb_inverse(vslm::fonb_make(px.n1), p);
        brdf1 = px.rd * vslm::rcp(P.psel);
    } else {
        wi12 = onb_inverse(vslm::fonb_make(P.R1), p);
        brdf1 = (px.rs * ((px.e + 2.0f) * vslm::rcp(px.e + 1.0f) * fmaxf(dot(wi12, px.n1), 0.f))) * vslm::rcp(1.0f - P.psel);
    }
    if (dot(wi12, c.nd12) <= c.cos_half_cone) return;
    float cos1 = fmaxf(dot(px.n1, wi12), 0.0f);
    if (cos1 <= 0.000000001f) return;
    float cos2 = fmaxf(-dot(v.n, wi12), 0.0f);
    V3 brdf2; float pdf1, pdf2;
    vsl_terms(px, v, P, L, wi12, cos1, cos2, nullptr, &brdf2, pdf1, pdf2);
    const float w = pdf1 * vslm::rcp(pdf1 + pdf2 + c.inv_solid_angle);
    acc = acc + (brdf1 * brdf2) * (cos2 * w);
}
template <bool DIFF>
EV_DEV void vsl_sample_brdf2(const Pixel &px, const Vpl &v, const VslPixel &P, const VslLight &L, const VslCtx &c, const VslFrames &F, V3 &acc, VslRng &rng) {  // :523-594
    if (L.dead) return;
    vsl_rng_step(rng);
    bool lam = true;
    if constexpr (!DIFF) lam = vsl_choose(rng) < L.psel;
    const float u1 = vsl_uniform(rng.s0), u2 = vsl_uniform(rng.s1);
    V3 p;
    if (DIFF || lam) p = lambert_local(u1, u2); else p = phong_local(v.e, u1, u2);
    if (!(dot(p, select3(lam, F.nd_n2, F.nd_r2)) > c.cos_half_cone - kConeSlack)) return;
    V3 wi21, brdf2;
    if (lam) {
        wi21 = onb_inverse(vslm::fonb_make(v.n), p);
        brdf2 = v.rd * vslm::rcp(L.psel);
    } else {
        wi21 = onb_inverse(vslm::fonb_make(L.R2), p);
        brdf2 = (v.rs * ((v.e + 2.0f) * vslm::rcp(v.e + 1.0f) * fmaxf(dot(wi21, v.n), 0.f))) * vslm::rcp(1.0f - L.psel);
    }
    if (-dot(wi21, c.nd12) <= c.cos_half_cone) return;
    float cos2 = fmaxf(dot(v.n, wi21), 0.0f);
    if (cos2 <= 0.00000001f) return;
    float cos1 = fmaxf(-dot(px.n1, wi21), 0.0f);
    if (P.dead) return;
    V3 brdf1; float pdf1, pdf2;
    vsl_terms(px, v, P, L, -wi21, cos1, cos2, &brdf1, nullptr, pdf1, pdf2);
    const float w = pdf2 * vslm::rcp(pdf1 + pdf2 + c.inv_solid_angle);
    acc = acc + (brdf1 * brdf2) * (cos1 * w);
}

#ifndef EVPLP_VSL_WAVES
#define EVPLP_VSL_WAVES 4   // estimator kernel: 128 VGPRs (it needs ~125), zero scratch
#endif
// The VSL gather is TWO kernels, because the walk and the estimators have disjoint register sets and different needs (round 2: one
// loop, 65 VGPR + 72 SGPR spills at 64 registers, 2-3 TB/s of scratch traffic; round 3 first as two phases of one kernel, then split):
//   gather_vsl_walk_kernel   the packet walk of every VSL of the item, with the VPL gather's register budget (64 registers, 7 waves
//                            per SIMD) and its hand-scheduled node visit; all it keeps per lane is the pixel's position and normal.
//                            Lanes whose pair has a zero geometry term (c1 c2 <= 1e-9, lighttracing.cu:619) do not enter the walk:
//                            the reference traces their shadow ray and then discards the pair -- same radiance, fewer rays (the
//                            `rays` statistic counts the rays actually traced).  The 64-bit masks of lit lanes go to HBM, 1 KB per
//                            128 VSLs, coalesced (8 bytes per (tile, VSL): 0.4 % of the estimators' time to write and read back);
//   gather_vsl_shade_kernel  the estimators of the lit pairs: 128 registers, zero scratch, the VSL record in SGPRs.
// Items are the VPL gather's: (tile, k consecutive splits), folded in its fixed tree (the partial sums shrink by k); a launch
// covers a range of groups so that the mask buffer stays small (context.cpp).
constexpr int kVslChunk = 128;     // lit masks staged in LDS, 1 KB per wavefront
EV_DEV size_t vsl_mask_base(const GatherArgs &a, int tile_in_launch_order, int group) {
    const int groups = a.group_count;
    return ((size_t)tile_in_launch_order * groups + (size_t)(group - a.group_first)) * (size_t)(a.splits_per_wave * a.masks_per_split);
}
EV_DEV int launch_tile(const GatherArgs &a) { return item_index<true>(a, (int)blockIdx.x).tile_l; }      // the tile's index in launch order

template <bool CUT, bool COST = false>
#if EVPLP_WALK_ASM && !EVPLP_TRAVERSAL_STATS
__attribute__((amdgpu_num_vgpr(52)))      // v[52:63] belong to the hand-written node visit (device_common.hpp)
#endif
__global__ __launch_bounds__(64, EVPLP_GATHER_WAVES) void gather_vsl_walk_kernel(GatherArgs a) {
    unsigned long long t_cost = 0ull;
    if constexpr (COST) t_cost = __builtin_amdgcn_s_memrealtime();
    __shared__ unsigned long long s_lit[kVslChunk];
    const int lane = threadIdx.x;
    const Item t = item_setup<true>(a, lane, (int)blockIdx.x);
    if (!t.has_tile) return;
    const bool valid = t.in_image;                      // no stencil test in splatSplotch (:694-695)
    const uint32_t nvpl = *a.nvpl;
    const int k = a.splits_per_wave;
    const char *node_base = pinned(reinterpret_cast<const char *>(a.sc.nodes)), *leaf_base = pinned(reinterpret_cast<const char *>(a.sc.leaves));
    const evplp_record *vpls = pinned(a.vpls);
    V3 p1, n1;
    { const float4 gp = a.g_pos[t.p], gn = a.g_nrm[t.p]; p1 = v3(gp); n1 = v3(gn); }
    const int tile_l = launch_tile(a);
    unsigned long long *masks = a.vsl_masks + vsl_mask_base(a, tile_l, t.group);
    const char *cuts_g = nullptr;
    if constexpr (CUT) cuts_g = pinned(a.cuts + (size_t)item_cut_group(a, t) * a.cut_vpl_stride * (size_t)kCutSlotBytes);
    uint32_t rays = 0;
    // the cut slots of the item's VSLs stream through an LDS ring ahead of the walks (as in gather_vpl_kernel)
    constexpr bool kRing = CUT && EVPLP_WALK_ASM && !EVPLP_TRAVERSAL_STATS;
    constexpr int kRingSlots = EVPLP_CUT_RING, kAhead = kRingSlots - 1;
    __shared__ float4 s_cut[kRing ? kRingSlots : 1][kCutSlotBytes / 16];
    int pj = 0; uint32_t pi = (uint32_t)(t.group * k), qslot = 0u;
    auto ring_skip_empty = [&]() { while (pj < k && pi >= nvpl) { pj++; pi = (uint32_t)(t.group * k + pj); } };
    auto ring_issue = [&](uint32_t slot) {
        if (lane < kCutSlotBytes / 16) lds_dma16(cuts_g + (size_t)pi * kCutSlotBytes + (size_t)lane * 16u, lds_offset(&s_cut[slot][0]));
        pi += (uint32_t)kVplSplit; ring_skip_empty();
    };
    if constexpr (kRing) {
        ring_skip_empty();
        for (int q = 0; q < kAhead; q++) if (pj < k) ring_issue((uint32_t)q);
    }
    for (int jj = 0; jj < k; jj++) {
        const uint32_t split = (uint32_t)(t.group * k + jj);
        uint32_t cbase = 0;
        for (uint32_t first = split; first < nvpl; first += (uint32_t)(kVplSplit * kVslChunk), cbase += (uint32_t)kVslChunk) {
            const uint32_t n = min((uint32_t)kVslChunk, (nvpl - first + (uint32_t)kVplSplit - 1u) / (uint32_t)kVplSplit);
            for (uint32_t c = 0; c < n; c++) {
                const uint32_t i = first + c * (uint32_t)kVplSplit;
                const float4 *cut_lds = nullptr;
                if constexpr (kRing) {
                    if (pj < k) { ring_issue((qslot + (uint32_t)kAhead) % (uint32_t)kRingSlots); wait_vmcnt<kAhead>(); }
                    else wait_vmcnt0();
                    cut_lds = &s_cut[qslot][0];
                    qslot = (qslot + 1u) % (uint32_t)kRingSlots;
                }
                V3 vpos, vn; float vpsel; fetch_vpl_head(vpls, i, vpos, vn, vpsel);
                const V3 v12 = vpos - p1;                                     // :605
                // c1 c2 <= 1e-9 (:619) decides which pairs exist at all, so it is taken with the reference's roundings (IEEE square root and
                // three divisions, ~45 instructions) -- but only when some lane is within 1e-3 (relative) of the threshold; everywhere
                // else the same product through two 1-ulp hardware operations gives the same answer
                const float d2 = dot(v12, v12), a1 = fmaxf(dot(n1, v12), 0.0f), a2 = fmaxf(-dot(vn, v12), 0.0f);
                const float quick = a1 * a2 * __builtin_amdgcn_rcpf(d2);
                // (written positively: a VSL that sits exactly ON the pixel's point gives 0 * 0 * rcp(0) = NaN, which must not pass --
                // the reference's fmaxf(NaN, 0) * fmaxf(NaN, 0) = 0 <= 1e-9 drops the pair, and a lit pair at distance 0 would put
                // rsq(0) = inf into the accumulating image for good)
                bool pre = valid && (quick > 0.000000001f);
                if (ballot64(valid && fabsf(quick - 0.000000001f) <= 0.000000000001f) != 0ull) {
                    const float dist = sqrtf(d2);
                    const V3 nv12 = v12 / dist;
                    const float c1c2 = fmaxf(dot(n1, nv12), 0.0f) * fmaxf(-dot(vn, nv12), 0.0f);
                    pre = valid && !(c1c2 <= 0.000000001f);                  // taken before the shadow ray instead of after it
                }
                unsigned long long lit = 0ull;
                if (ballot64(pre) != 0ull) {
                    rays += pre ? 1u : 0u;
                    const bool occ = occluded_wave<EVPLP_WALK_ASM ? 52 : 0, CUT>(node_base, leaf_base, vpos, -v12, 0.0001f, 1.0f - 0.0001f, pre, nullptr, cuts_g, i * (uint32_t)kCutSlotBytes, cut_lds);   // :612-614
                    lit = ballot64(pre && !occ);
                }
                if (lane == 0) s_lit[c] = lit;          // (single-wavefront workgroup: LDS is in order, no barrier)
            }
            // the chunk's masks leave as one coalesced store (two per lane)
            unsigned long long *dst = masks + (size_t)jj * a.masks_per_split + cbase;
            for (uint32_t q = (uint32_t)lane; q < n; q += 64u) dst[q] = s_lit[q];
        }
    }
#if EVPLP_XCD_TIMES
    if (lane == 0) atomicMax(&a.counters->hist[24 + (blockIdx.x & 7u)], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    // shadow rays of the item, summed over the wavefront: one add per item into the pass's 64 counter shards (an item's count can reach
    // 64 lanes x 4095 VSLs -- it does not fit the 16-bit field the VPL gather's per-pixel statistics word has for it)
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (lane == 0 && rays != 0u) atomicAdd(&a.counters->shard_rays[blockIdx.x & (kCounterShards - 1)], (unsigned long long)rays);
    if constexpr (COST) item_cost_add(a, item_index<true>(a, (int)blockIdx.x).ty, t_cost, EVPLP_GATHER_WAVES);
}

template <bool COST = false>
__global__ __launch_bounds__(64, EVPLP_VSL_WAVES) void gather_vsl_shade_kernel(GatherArgs a) {
    unsigned long long t_cost = 0ull;
    if constexpr (COST) t_cost = __builtin_amdgcn_s_memrealtime();
    extern __shared__ float s_lvl[];                   // one [192] block per level of the k-split fold
    __shared__ unsigned long long s_lit[kVslChunk];
    const int lane = threadIdx.x;
    const int W = a.st.W;
    const Item t = item_setup<true>(a, lane, (int)blockIdx.x);
    if (!t.has_tile) return;
    const bool valid = t.in_image;
    const uint32_t pixel_id = (uint32_t)t.gy * (uint32_t)W + (uint32_t)t.x;  // launchIndex.y * dim.x + launchIndex.x (:711)
    const uint64_t pixel_key = vsl_key(((uint64_t)a.fp.rng_seed << 32) | (uint64_t)pixel_id);
    const uint32_t nvpl = *a.nvpl;
    const int k = a.splits_per_wave;
    const evplp_record *vpls = pinned(a.vpls);
    const int tile_l = launch_tile(a);
    const unsigned long long *masks = a.vsl_masks + vsl_mask_base(a, tile_l, t.group);
    Pixel px;
    { const float4 gp = a.g_pos[t.p], gn = a.g_nrm[t.p], gd = a.g_dif[t.p], gs = a.g_phg[t.p]; px.p1 = v3(gp); px.n1 = v3(gn); px.rd = v3(gd); px.rs = v3(gs); px.e = gs.w; }
    px.wi10 = normalize(v3(a.fp.camera_pos) - px.p1);   // :704
    VslPixel P;
    {
        const float ml = max_color(px.rd), mp = max_color(px.rs);
        P.dead = ml + mp <= 0.000001f; P.psel = ml / (mp + ml);
        P.glossy = px.rs.x != 0.0f || px.rs.y != 0.0f || px.rs.z != 0.0f; P.pdf_glossy = !(px.rs.x <= 0.000001f);
        P.R1 = reflect(-px.wi10, px.n1);
    }
    V3 total = v3(0.f, 0.f, 0.f);
    uint32_t cnt = 0;                                   // lit pairs (low 12 bits: context.cpp keeps an item within 4095 VSLs) | sample-iterations << 12 (<= 101 x 4095 < 2^20)
    for (int jj = 0; jj < k; jj++) {
        const uint32_t split = (uint32_t)(t.group * k + jj);
        V3 result = v3(0.f, 0.f, 0.f);
        uint32_t cbase = 0;
        for (uint32_t first = split; first < nvpl; first += (uint32_t)(kVplSplit * kVslChunk), cbase += (uint32_t)kVslChunk) {
            const uint32_t n = min((uint32_t)kVslChunk, (nvpl - first + (uint32_t)kVplSplit - 1u) / (uint32_t)kVplSplit);
            uint32_t moff = (uint32_t)jj * (uint32_t)a.masks_per_split + cbase;
            asm volatile("" : "+s"(moff));             // (a scalar offset formed here: as an induction variable the per-lane address was the one spill left)
            const unsigned long long *src = masks + moff;
            uint32_t q0 = (uint32_t)lane;
            asm volatile("" : "+v"(q0));
            for (uint32_t q = q0; q < n; q += 64u) s_lit[q] = src[q];        // coalesced; single-wavefront workgroup: no barrier
            for (uint32_t c = 0; c < n; c++) {
                const unsigned long long lit = s_lit[c];
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)lit), hi = __builtin_amdgcn_readfirstlane((uint32_t)(lit >> 32));
                if ((lo | hi) == 0u) continue;
                const uint32_t i = first + c * (uint32_t)kVplSplit;
                Vpl v; fetch_vpl_head(vpls, i, v.pos, v.n, v.psel); fetch_vpl_tail(vpls, i, v);
                const bool lit_lane = (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) != 0u;
                const uint64_t record_key = vsl_key((uint64_t)(1u + a.vpl_src_index[i]) * 0xD1B54A32D192ED03ull);   // (scalar arithmetic)
                // (wave-uniform) no Phong lobe on either side of any lit pair of this VSL: the estimators' diffuse form
                const bool diffuse_wave = !(v.rs.x != 0.0f || v.rs.y != 0.0f || v.rs.z != 0.0f) && ballot64(lit_lane && (P.glossy || P.psel < 1.0f)) == 0ull;
                if (lit_lane) {
                    const V3 v12 = v.pos - px.p1;
                    const float inv_dist = vslm::rsq(dot(v12, v12));         // (1-ulp hardware operations: the IEEE square root and four divisions were ~55 instructions per pair)
                    const V3 nv12 = v12 * inv_dist;
                    VslCtx cx;
                    const float rdratio = a.fp.vsl_radius * inv_dist;
                    cx.half_cone = (rdratio >= 1.0f) ? EV_PI / 2.0f : asinf(rdratio);   // :623
                    cx.cos_half_cone = (rdratio >= 1.0f) ? cosf(EV_PI / 2.0f) : vslm::fsqrt(1.0f - rdratio * rdratio);   // cos(asin x)
                    cx.solid_angle = EV_PI * 2.0f * (1.0f - cx.cos_half_cone);
                    cx.inv_solid_angle = vslm::rcp(cx.solid_angle);
                    cx.inv_pi_r2 = a.fp.vsl_inv_pi_radius2; cx.nd12 = nv12;
                    const int num_samples = (int)(cx.half_cone / EV_PI * 2.0f * 100.0f) + 1;  // :632
                    cnt += ((uint32_t)num_samples << 12) + 1u;
                    // one stream per (pixel, record): any decomposition reproduces the same numbers
                    VslRng rng = vsl_rng_seed(pixel_key, record_key);
                    VslLight L;
                    {
                        const float ml = max_color(v.rd), mp = max_color(v.rs);
                        L.dead = ml + mp <= 0.000001f; L.psel = ml * vslm::rcp(mp + ml);
                        L.glossy = v.rs.x != 0.0f || v.rs.y != 0.0f || v.rs.z != 0.0f; L.pdf_glossy = !(v.rs.x <= 0.000001f);
                        L.R2 = reflect(-v.fdir, v.n);
                    }
                    // nd12 in the four sampling frames (the BRDF samples' cone pre-test) and the frame of the cone itself
                    VslFrames F;
                    F.nd_n1 = onb_local(vslm::fonb_make(px.n1), nv12); F.nd_r1 = onb_local(vslm::fonb_make(P.R1), nv12);
                    F.nd_n2 = onb_local(vslm::fonb_make(v.n), -nv12);  F.nd_r2 = onb_local(vslm::fonb_make(L.R2), -nv12);
                    const Onb cone = vslm::fonb_make(nv12);
                    V3 acc = v3(0.f, 0.f, 0.f);
                    if (diffuse_wave) {
                        float acc_d = 0.0f;
                        for (int sidx = 0; sidx < num_samples; sidx++) {
                            vsl_sample_cone<true>(px, v, P, L, cx, cone, acc, acc_d, rng);
                            vsl_sample_brdf1<true>(px, v, P, L, cx, F, acc, rng);
                            vsl_sample_brdf2<true>(px, v, P, L, cx, F, acc, rng);
                        }
                        acc = acc + ((px.rd * EV_INV_PI) * (v.rd * EV_INV_PI)) * acc_d;
                    } else {
                        float unused = 0.0f;
                        for (int sidx = 0; sidx < num_samples; sidx++) {
                            vsl_sample_cone<false>(px, v, P, L, cx, cone, acc, unused, rng);
                            vsl_sample_brdf1<false>(px, v, P, L, cx, F, acc, rng);
                            vsl_sample_brdf2<false>(px, v, P, L, cx, F, acc, rng);
                        }
                    }
                    acc = (v.flux * cx.inv_pi_r2) * acc;
                    result = result + acc * vslm::rcp((float)num_samples);
                }
            }
        }
        {   // fold the split sums in the fixed balanced-tree order (as gather_vpl_kernel)
            int lev = 0;
            while ((jj >> lev) & 1) {
                float *q = s_lvl + lev * 192 + lane;
                result = v3(q[0], q[64], q[128]) + result;
                lev++;
            }
            float *q = s_lvl + lev * 192 + lane;
            q[0] = result.x; q[64] = result.y; q[128] = result.z;
        }
        total = result;
    }
#if EVPLP_DEBUG_NAN
    if (valid && !(isfinite(total.x) && isfinite(total.y) && isfinite(total.z))) atomicAdd(&a.counters->nonfinite, 1ull);
#endif
#if EVPLP_XCD_TIMES
    if (lane == 0) atomicMax(&a.counters->hist[16 + (blockIdx.x & 7u)], (unsigned long long)__builtin_amdgcn_s_memrealtime());   // (tools/xcd_balance.py --vsl; the walk kernel: hist[24 ..])
#endif
    // sample-iterations of the item (the unit the estimators' work is priced in): one add per wavefront into 64 counter shards
    const uint32_t nlit = cnt & 4095u;
    uint32_t nsamp = cnt >> 12;
    for (int off = 32; off > 0; off >>= 1) nsamp += __shfl_down(nsamp, off);
    if (lane == 0) atomicAdd(&a.counters->hist[blockIdx.x & 63u], (unsigned long long)nsamp);
    // (the item's shadow rays were counted by the walk kernel; the per-pixel word carries the lit pairs only: < 4096 per item, context.cpp)
    int lane_out = lane, blk_out = (int)blockIdx.x;
    asm volatile("" : "+v"(lane_out), "+s"(blk_out));   // (the store address is formed here, not carried through the estimators)
    const int group_out = a.group_first + item_index<true>(a, blk_out).sg;
    if (valid) a.partial[(size_t)group_out * a.partial_stride + item_texel<true>(a, lane_out, blk_out)] = make_float4(total.x, total.y, total.z, __uint_as_float(nlit << 16));
    if constexpr (COST) item_cost_add(a, item_index<true>(a, blk_out).ty, t_cost, EVPLP_VSL_WAVES);
}

int gather_launch_tiles(const GatherArgs &a) {                    // tiles of a launch: whole blocks of 8 x (1 << block_h_log2)
    const int tiles_x = (a.st.W + 7) / 8, tiles_y = (a.st.local_rows + 7) / 8, sh = 1 << a.block_h_log2;
    const int nby = (tiles_y + sh - 1) / sh, rows = a.band_rows > 0 ? std::min(a.band_rows, nby - a.band_first) : nby;
    return ((tiles_x + 7) / 8) * std::max(rows, 0) * 8 * sh;
}
static dim3 gather_grid(const GatherArgs &a) {
    const int groups = a.group_count > 0 ? a.group_count : kVplSplit / a.splits_per_wave;
    return dim3((unsigned)(gather_launch_tiles(a) * groups));     // tile = tile_j * 8 + xcd
}
void launch_gather_reduce(const GatherArgs &a, int stencil_test, hipStream_t s) {
    size_t n = (size_t)a.st.W * a.st.local_rows;
    hipLaunchKernelGGL(gather_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, stencil_test);
}
static size_t fold_lds_bytes(const GatherArgs &a, int extra_floats) {
    int levels = 1; while ((1 << (levels - 1)) < a.splits_per_wave) levels++;       // log2 k + 1
    return (size_t)(levels * 192 + extra_floats) * sizeof(float);
}
// (a.block_cost set: the calibration variants of the same kernels -- same results, every item also clocks itself)
void launch_gather_vpl_items(const GatherArgs &a, hipStream_t s) {
    const size_t lds = fold_lds_bytes(a, kGatherPxFloats);
    if (a.block_cost) {
        if (a.cuts) hipLaunchKernelGGL((gather_vpl_kernel<true, true>), gather_grid(a), dim3(64), lds, s, a);
        else hipLaunchKernelGGL((gather_vpl_kernel<false, true>), gather_grid(a), dim3(64), lds, s, a);
    } else if (a.cuts) hipLaunchKernelGGL((gather_vpl_kernel<true>), gather_grid(a), dim3(64), lds, s, a);
    else hipLaunchKernelGGL((gather_vpl_kernel<false>), gather_grid(a), dim3(64), lds, s, a);
}
void launch_gather_vsl(const GatherArgs &a, hipStream_t s) {
    if (a.block_cost) {
        if (a.cuts) hipLaunchKernelGGL((gather_vsl_walk_kernel<true, true>), gather_grid(a), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((gather_vsl_walk_kernel<false, true>), gather_grid(a), dim3(64), 0, s, a);
        hipLaunchKernelGGL((gather_vsl_shade_kernel<true>), gather_grid(a), dim3(64), fold_lds_bytes(a, 0), s, a);
        return;
    }
    if (a.cuts) hipLaunchKernelGGL((gather_vsl_walk_kernel<true>), gather_grid(a), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((gather_vsl_walk_kernel<false>), gather_grid(a), dim3(64), 0, s, a);
    hipLaunchKernelGGL((gather_vsl_shade_kernel<false>), gather_grid(a), dim3(64), fold_lds_bytes(a, 0), s, a);
}

} // namespace evplp
