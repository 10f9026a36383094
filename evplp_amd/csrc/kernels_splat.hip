// Image-space photon splat and the final composite.
//   runPhotonSplat + shaders/photonsplatinstanced.{vert,geom,frag}
//       (rt/rtcomphoton/rtcomphoton.h:789-837, frag:146-240)        -> tile_box / bin / scatter / big / tiles
//   shaders/final.frag:19-35, rtcomphoton.h:756-787                  -> resolve_kernel
//
// The reference rasterises one icosphere proxy per record and lets the ROP blend every fragment
// (one RMW of HBM per fragment).  Here photon i adds to pixel p iff |X_p - P_i|^2 <= r^2 (frag:152-154) -- once
// (EVPLP_FOOTPRINT_IDEAL) or once per face of the proxy mesh in front of the surface (EVPLP_FOOTPRINT_PROXY, the
// reference's coverage; splat_tiles_kernel<.., true>):
//   0. splat_tile_box : world-space bounding box of the G-buffer positions of every 8x8-pixel tile -- only for G-buffers that
//      were uploaded: evplp_primary writes the boxes as it writes the G-buffer.
//   1. splat_bin : one lane per record (records staged through LDS).  Everything of the fragment shader that does not depend
//      on the pixel (w12, the MIS/clamp weight, 1/(pi r^2 N) scaling) is folded into a 64-byte compact photon; the photon gets
//      an entry for every 8x8-pixel tile of the conservative screen rectangle of its radius-r sphere whose position box the
//      sphere reaches.  Entries are binned in two levels WITHOUT contended atomics (kernels.h "Two-level binning"):
//      splat_bin sorts a workgroup's entries by coarse bucket into a segment of its own, splat_scatter ranks a bucket's
//      entries per tile and reserves bin slots with one atomic per (workgroup, tile), splat_big handles photons with
//      rectangles of more than 3x3 tiles.  Bins are fixed slabs of bin_stride slots per tile; a bin that wants more than its
//      slab raises the overflow flag and the host doubles the slabs and runs the pass again (context.cpp settle_splat).
//      (deterministic mode: + rank sort so every pixel accumulates in ascending record order, like the oracle.)
//   2. splat_tiles   : one wavefront per tile (four when bins are very full), lane = pixel with its G-buffer texel in
//      registers; the bin streams through LDS 64 photons at a time (each lane fetches one compact photon); every pixel first
//      takes the radius test of the 64 photons (a 64-bit mask), then shades ITS photons in ascending order; RGB accumulates in
//      registers and is written once per pixel with coalesced 16-byte stores -- no atomics, no per-fragment RMW.
#include "device_common.hpp"
#include "kernels.h"
#include <cstring>

#ifndef EVPLP_PROXY_MIXED_WAVES
#define EVPLP_PROXY_MIXED_WAVES 6
#endif
#ifndef EVPLP_PROXY_WAVES
#define EVPLP_PROXY_WAVES 7
#endif
namespace evplp {

// d^e on the hardware transcendentals (d in (1e-5, 1]): exp2(e log2 d), relative error ~0.7 e |log2 d| 2^-22, i.e. 2e-6 where the lobe is
// still 1e-4 of its peak, against the stated bars of 1e-5 (image) / 2e-4 (pixel); the library powf is ~170 instructions and ran twice per
// photon in the bin kernel and once per shaded (photon, pixel) pair of a glossy tile
// ... and 1-ulp hardware reciprocals / reciprocal square roots instead of the IEEE-correct expansions (~10 instructions each; the bin kernel
// had eleven divisions and two square roots per photon)
EV_DEV float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
EV_DEV V3 normalize_hw(V3 v) { return v * __builtin_amdgcn_rsqf(dot(v, v)); }
EV_DEV float pow_hw(float d, float e) { return e == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(d)); }
// GLSL flavours of the BRDF helpers (photonsplatinstanced.frag:42-98; they differ from the CUDA ones)
EV_DEV V3 g_lambert_eval(V3 w10, V3 w12, V3 n, V3 rd) {
    if (dot(w10, n) <= 0.0f || dot(w12, n) <= 0.0f) return v3(0.f, 0.f, 0.f);
    return rd * EV_INV_PI;
}
EV_DEV V3 g_phong_eval(V3 outv, V3 inv_, V3 n, V3 rs, float e) {
    V3 r = reflect(-inv_, n);
    float d = dot(outv, r);
    if (d <= 0.00001f) return v3(0.f, 0.f, 0.f);
    return rs * (e + 2.0f) * pow_hw(d, e) * EV_INV_PI * 0.5f;
}
EV_DEV float g_lambert_pdf_w(V3 n1, V3 v12) { return fmaxf(dot(n1, normalize_hw(v12)), 0.f) * EV_INV_PI; }
EV_DEV float g_phong_pdf_w(V3 n1, V3 wi12, V3 inv_, V3 rs, float e) {
    V3 r = reflect(-inv_, n1);
    float d = fmaxf(dot(wi12, r), 0.f);
    if (d <= 0.00001f || rs.x <= 0.00001f) return 0.0f;
    return (e + 1.0f) * 0.5f * EV_INV_PI * pow_hw(d, e);
}

constexpr int kRecF4 = sizeof(evplp_record) / 16;   // 6 float4 per record
struct Rec { V3 pos, n, flux, fdir, rd, rs; float psel, e; uint32_t flags; };
EV_DEV Rec load_rec(const float4 *q) {
    float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5];
    Rec v; v.pos = v3(a); v.flags = __float_as_uint(a.w); v.n = v3(b); v.psel = b.w; v.flux = v3(c);
    v.fdir = v3(d); v.rd = v3(e); v.rs = v3(f); v.e = f.w;
    return v;
}

// Conservative rectangle of 8x8-px tiles (GLOBAL tile rows) of every visible point within r of `pos`, through the (jittered)
// camera of this iteration (uMVP of runPhotonSplat is the jittered matrix, :982), packed (x0 | x1 << 16, y0 | y1 << 16);
// x0 > x1 = nothing to splat.
EV_DEV uint2 photon_rect(const SplatArgs &a, V3 pos) {
    const uint2 none = make_uint2(1u, 0u);
    const float r = a.fp.photon_radius;
    V3 q = pos - v3(a.cam.eye);
    float vx = dot(q, v3(a.cam.s)), vy = dot(q, v3(a.cam.u)), vz = dot(q, v3(a.cam.f));
    // visible surface points have view depth in [near, far] = [0.1, 100] (rtcommon.h:586): clip the
    // sphere's depth range to it -- a photon closer than r to the camera plane then needs no
    // whole-screen fallback (those few photons used to produce most of the bin entries)
    const float zlo = fmaxf(vz - r, 0.1f), zhi = fminf(vz + r, 100.0f);
    if (zlo > zhi) return none;
    float sx = rcp_hw(a.cam.aspect * a.cam.tan_half), sy = rcp_hw(a.cam.tan_half);
    float il = rcp_hw(zlo), ih = rcp_hw(zhi);            // (1 ulp: 1e-4 px at 4 k pixels, the guard below is 1/32 px)
    float nx0 = fminf((vx - r) * il, (vx - r) * ih) * sx + a.fp.jitter[0];
    float nx1 = fmaxf((vx + r) * il, (vx + r) * ih) * sx + a.fp.jitter[0];
    float ny0 = fminf((vy - r) * il, (vy - r) * ih) * sy + a.fp.jitter[1];
    float ny1 = fmaxf((vy + r) * il, (vy + r) * ih) * sy + a.fp.jitter[1];
    float fx0 = (nx0 * 0.5f + 0.5f) * (float)a.st.W - 0.5f, fx1 = (nx1 * 0.5f + 0.5f) * (float)a.st.W - 0.5f;
    float fy0 = (ny0 * 0.5f + 0.5f) * (float)a.st.H - 0.5f, fy1 = (ny1 * 0.5f + 0.5f) * (float)a.st.H - 0.5f;
    // guard against rounding in the projection (the G-buffer point of a pixel is seen exactly through the pixel centre of the
    // jittered camera; the arithmetic above is good to ~1e-4 px at 4 k pixels).  A whole pixel of guard made one photon in
    // five touch a third tile per axis.
    const float guard = 1.0f / 32.0f;
    fx0 = fminf(fmaxf(fx0 - guard, -1.0f), (float)a.st.W); fx1 = fminf(fmaxf(fx1 + guard, -1.0f), (float)a.st.W);
    fy0 = fminf(fmaxf(fy0 - guard, -1.0f), (float)a.st.H); fy1 = fminf(fmaxf(fy1 + guard, -1.0f), (float)a.st.H);
    int x0 = max((int)ceilf(fx0), 0), x1 = min((int)floorf(fx1), a.st.W - 1);
    int y0 = max((int)ceilf(fy0), 0), y1 = min((int)floorf(fy1), a.st.H - 1);
    if (x0 > x1 || y0 > y1) return none;
    int tx0 = x0 >> 3, tx1 = x1 >> 3, ty0 = y0 >> 3, ty1 = y1 >> 3;
    return make_uint2((uint32_t)tx0 | ((uint32_t)tx1 << 16), (uint32_t)ty0 | ((uint32_t)ty1 << 16));
}
// squared reach of a photon at `pos`: the radius with a guard for the rounding of the distance computation (relative 1e-5 and
// an absolute term scaled by the coordinates): a tile whose box is farther than that cannot hold a pixel within r of it
EV_DEV float photon_reach2(const SplatArgs &a, V3 pos) {
    const float pmag = fmaxf(fmaxf(fabsf(pos.x), fabsf(pos.y)), fmaxf(fabsf(pos.z), 1.0f));
    const float reach = a.fp.photon_radius * (1.0f + 1.0e-5f) + 4.0e-6f * pmag;
    return reach * reach;
}
EV_DEV bool box_within(float4 blo, float4 bhi, V3 c, float reach2) {
    const float dx = fmaxf(fmaxf(blo.x - c.x, c.x - bhi.x), 0.0f), dy = fmaxf(fmaxf(blo.y - c.y, c.y - bhi.y), 0.0f),
                dz = fmaxf(fmaxf(blo.z - c.z, c.z - bhi.z), 0.0f);
    return !(dx * dx + dy * dy + dz * dz > reach2);
}
EV_DEV int local_tile_row(const SplatArgs &a, int ty) {                  // -1: row strip of another GPU
    if (a.st.band_rows > 0) { const int l = ty * 8 - a.st.band_first; return (l >= 0 && l < a.st.band_rows) ? (l >> 3) : -1; }
    if (a.st.strip_count == 1) return ty;
    const int tiles_per_block = a.st.strip_rows >> 3, blk = ty / tiles_per_block, lb = a.st.local_block(blk);      // (ty is an image tile row: blk < image blocks)
    return lb < 0 ? -1 : lb * tiles_per_block + (ty - blk * tiles_per_block);
}

// compact photon: [0] pos.xyz, cpn   [1] w12.xyz, d2   [2] wflux.xyz, alive   [3] brdf2.xyz (misMode 5 only)
// Everything of one photon that does not depend on the pixel (compact record); returns its rectangle of tiles.
EV_DEV uint2 splat_prepare_one(const SplatArgs &a, uint32_t i, const float4 *s_ph, const float4 *s_prev, V3 &photon_pos) {
    const uint2 none = make_uint2(1u, 0u);
    Rec ph = load_rec(s_ph);
    if (!(ph.flags & EVPLP_USABLE_PHOTON)) return none;  // vert:31, geom:20
    photon_pos = ph.pos;
    // a photon whose sphere shows nowhere on the screen (about half of the usable ones at the BASELINE configurations) needs no
    // compact record: nothing will ever read it
    const uint2 rect = photon_rect(a, ph.pos);
    if ((rect.x & 0xffffu) > (rect.x >> 16)) return none;
    Rec prev = load_rec(s_prev);                    // frag:163

    const float r = a.fp.photon_radius;
    V3 v12 = prev.pos - ph.pos;                                           // frag:170
    float d2 = dot(v12, v12);
    V3 w12 = normalize_hw(v12);
    float mix_w = g_lambert_pdf_w(prev.n, -w12) * prev.psel;              // frag:184-187
    mix_w += g_phong_pdf_w(prev.n, -w12, prev.fdir, prev.rs, prev.e) * (1.0f - prev.psel);
    float mix_a = mix_w * fmaxf(dot(ph.n, w12), 0.0f) * rcp_hw(d2);       // frag:189
    bool alive = mix_w > 0.0f;                                            // frag:191
    float k = EV_INV_PI * rcp_hw(r * r);                                  // InvPi * uInvPhotonRadius2
    float inv_n = rcp_hw((float)a.fp.num_light_paths);                    // uInvNumLightPaths
    float wgt = 1.0f;
    const uint32_t mode = a.fp.mis_mode;
    if (mode == 1u) wgt = mix_a * rcp_hw(mix_a + a.fp.pdf_mc);
    else if (mode == 2u) wgt = mix_a > a.fp.pdf_mc ? 1.0f : 0.0f;
    else if (mode == 3u) { float a2 = mix_a * mix_a, b2 = a.fp.pdf_mc * a.fp.pdf_mc; wgt = a2 * rcp_hw(a2 + b2); }
    V3 wflux = ph.flux * (k * inv_n * wgt);
    float cpn = fmaxf(-dot(prev.n, w12), 0.0f);
    V3 brdf2 = g_lambert_eval(-w12, prev.fdir, prev.n, prev.rd) + g_phong_eval(-w12, prev.fdir, prev.n, prev.rs, prev.e);  // frag:182
    float4 *c = a.compact + (size_t)i * kCompactF4;
    c[0] = make_float4(ph.pos.x, ph.pos.y, ph.pos.z, cpn);
    c[1] = make_float4(w12.x, w12.y, w12.z, d2);
    c[2] = make_float4(wflux.x, wflux.y, wflux.z, alive ? 1.0f : 0.0f);
    if (mode == 5u) c[3] = make_float4(brdf2.x, brdf2.y, brdf2.z, 0.f);       // read by misMode 5 only
    return rect;
}

// World-space bounding box of every 8x8-px tile's G-buffer positions (one wave per tile).  A pixel can only receive a
// photon whose centre lies within r of its position: splat_bin drops the (photon, tile) entries whose sphere does not reach
// the tile's box.  Screen-space bins otherwise collect every photon along the tile's frustum -- floor under the table,
// wall behind the chairs -- several times more than ever pass the radius test.  Every in-image pixel counts, background
// included (its position is the clear colour, which is what the radius test of frag:152-154 sees too).
__global__ __launch_bounds__(256) void splat_tile_box_kernel(SplatArgs a) {
    // four tiles per wave, their texels fetched together
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile0 = (blockIdx.x * 4 + wave) * 4;
    float4 gp[4]; bool in_image[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int tile = min(tile0 + q, ntiles - 1);
        const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
        const int x = tx * 8 + (lane & 7), ly = lty * 8 + (lane >> 3);
        in_image[q] = x < a.st.W && ly < a.st.local_rows && a.st.global_row(min(ly, a.st.local_rows - 1)) < a.st.H;
        gp[q] = a.g_pos[(size_t)min(ly, a.st.local_rows - 1) * a.st.W + min(x, a.st.W - 1)];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int tile = tile0 + q;
        if (tile >= ntiles) break;
        float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        if (in_image[q]) { lo[0] = hi[0] = gp[q].x; lo[1] = hi[1] = gp[q].y; lo[2] = hi[2] = gp[q].z; }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
        if (lane == 0) {
            a.tile_box[2 * tile] = make_float4(lo[0], lo[1], lo[2], 0.f); a.tile_box[2 * tile + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
}

// Bin kernel (kernels.h "Two-level binning").  The 96-byte AoS records are read ONCE, as a coalesced 16 B/lane stream, into LDS
// (slot k = record base - 1 + k: every photon also needs its predecessor on the light path, frag:163) -- all seven loads of
// a thread in flight together, the next chunk's issued before this chunk is worked on; a lane then picks its two records
// from LDS (one lane per record reading its own 6 float4 at a 96-byte stride touched every line six times and every
// record twice).
// (plain variables: a struct or an array of these ends up in scratch memory)
#define EV_BIN_FETCH(base_)                                                                                             \
    {                                                                                                                   \
        const float4 *src = reinterpret_cast<const float4 *>(a.records) + (size_t)(base_) * kRecF4;                      \
        const uint32_t n4 = (min((base_) + 256u, a.num_records) - (base_)) * kRecF4, last = n4 - 1u;                      \
        r0_ = src[min(tid, last)]; r1_ = src[min(tid + 256u, last)]; r2_ = src[min(tid + 512u, last)];                    \
        r3_ = src[min(tid + 768u, last)]; r4_ = src[min(tid + 1024u, last)]; r5_ = src[min(tid + 1280u, last)];           \
        pv = *(src + (((base_) != 0u && tid < (uint32_t)kRecF4) ? (int)tid - kRecF4 : 0));                                \
    }
#define EV_BIN_STAGE(base_)                                                                                             \
    {                                                                                                                   \
        const uint32_t n4 = (min((base_) + 256u, a.num_records) - (base_)) * kRecF4;                                      \
        float4 *dst = s_rec + kRecF4;                                                                                   \
        if (tid < n4) dst[tid] = r0_;                                                                                   \
        if (tid + 256u < n4) dst[tid + 256u] = r1_;                                                                     \
        if (tid + 512u < n4) dst[tid + 512u] = r2_;                                                                     \
        if (tid + 768u < n4) dst[tid + 768u] = r3_;                                                                     \
        if (tid + 1024u < n4) dst[tid + 1024u] = r4_;                                                                   \
        if (tid + 1280u < n4) dst[tid + 1280u] = r5_;                                                                   \
        if ((base_) != 0u && tid < (uint32_t)kRecF4) s_rec[tid] = pv;                                                   \
    }
// LDS entry: record - group base (10 bits; kBinGroup <= 1024) | tile within the bucket (<= 9 bits) << 10 | bucket (10) << 19
__global__ __launch_bounds__(256) void splat_bin_kernel(SplatArgs a) {
    __shared__ float4 s_rec[257 * kRecF4];
    __shared__ uint32_t s_pair[kSegCap];
    // bucket counters and offsets: sized by the launch (num_buckets rounded up to 4: 128 at 1024^2, 254 -> 256 at 1080p) -- as two static
    // kMaxBuckets arrays they were 8 of the kernel's 37 KB and the fifth workgroup of a CU did not fit (round 5)
    extern __shared__ uint32_t s_bucket_lds[];
    const uint32_t nb4 = ((uint32_t)a.num_buckets + 3u) & ~3u;
    uint32_t *const s_cnt = s_bucket_lds, *const s_off = s_bucket_lds + nb4;
    __shared__ uint32_t s_n, s_nbig, s_wsum[4];
    const uint32_t tid = threadIdx.x, group = blockIdx.x, wg_base = group * (uint32_t)kBinGroup;
    for (uint32_t b = tid; b < nb4; b += 256u) s_cnt[b] = 0u;
    if (tid == 0u) { s_n = 0u; s_nbig = 0u; }
    // this launch also clears what the NEXT two use: the tiles' bin cursors, the pass summary, the overflow flag
    for (uint32_t k = blockIdx.x * 256u + tid; k < (uint32_t)(a.tiles_x * a.tiles_y); k += gridDim.x * 256u) a.tile_cursor[k] = 0u;
    if (blockIdx.x == 0u) {
        for (uint32_t k = tid; k < (uint32_t)kSummaryShards; k += 256u) { a.summary[k * kSummaryStride] = 0u; a.summary[k * kSummaryStride + 1] = 0u; }
        if (tid == 0u) { *a.overflow = 0u; a.summary[kSummaryHeavy] = 0u; }
    }
    const uint32_t bw_mask = (1u << a.bucket_w_log2) - 1u, bh_mask = (1u << a.bucket_h_log2) - 1u;

    float4 r0_, r1_, r2_, r3_, r4_, r5_, pv;
    if (wg_base < a.num_records) EV_BIN_FETCH(wg_base)
    for (int c = 0; c < kBinChunks; c++) {
        const uint32_t base = wg_base + 256u * (uint32_t)c;
        if (base >= a.num_records) break;
        EV_BIN_STAGE(base)
        __syncthreads();
        if (c + 1 < kBinChunks && base + 256u < a.num_records) EV_BIN_FETCH(base + 256u)
        const uint32_t i = base + tid;
        uint2 rc = make_uint2(1u, 0u);
        V3 c0 = v3(0.f, 0.f, 0.f);
        if (i < a.num_records && i != 0u) rc = splat_prepare_one(a, i, &s_rec[(tid + 1u) * kRecF4], &s_rec[tid * kRecF4], c0);
        const int tx0 = rc.x & 0xffff, tx1 = rc.x >> 16, ty0 = rc.y & 0xffff, ty1 = rc.y >> 16;
        const float reach2 = photon_reach2(a, c0);
#if EVPLP_TRAVERSAL_STATS
        { const int nx_ = tx1 - tx0 + 1, ny_ = ty1 - ty0 + 1; int cls = tx0 > tx1 ? 0 : (nx_ <= 2 && ny_ <= 2) ? 1 : (nx_ <= 3 && ny_ <= 3) ? 2 : 3;
          atomicAdd(&a.counters->hist[cls], 1ull); if (cls) { atomicAdd(&a.counters->hist[4], (unsigned long long)(nx_ * ny_)); atomicAdd(&a.counters->hist[8 + min(nx_ * ny_, 16)], 1ull); } }
#endif
        if (tx0 > tx1) {
        } else if (tx1 - tx0 <= 2 && ty1 - ty0 <= 2) {
            // the usual case at the radii of a converging run: up to 3x3 tiles, their boxes fetched together
            int tx[9], lty[9]; float4 blo[9], bhi[9];
#pragma unroll
            for (int q = 0; q < 9; q++) {
                tx[q] = tx0 + (q % 3);
                const int ty = ty0 + (q / 3);
                lty[q] = (tx[q] <= tx1 && ty <= ty1) ? local_tile_row(a, ty) : -1;
            }
#pragma unroll
            for (int q = 0; q < 9; q++) if (lty[q] >= 0) { const int t = lty[q] * a.tiles_x + tx[q]; blo[q] = a.tile_box[2 * t]; bhi[q] = a.tile_box[2 * t + 1]; }
            uint32_t hit = 0u;
#pragma unroll
            for (int q = 0; q < 9; q++) if (lty[q] >= 0 && box_within(blo[q], bhi[q], c0, reach2)) hit |= 1u << q;
            // the group's segment holds kSegCap entries (4 per photon on average; the depth cull leaves 1-2): a photon that
            // does not fit any more goes the slow way
            const uint32_t cnt = (uint32_t)__builtin_popcount(hit);
            uint32_t at = cnt ? atomicAdd(&s_n, cnt) : 0u;
            if (at + cnt > (uint32_t)kSegCap) {
                for (uint32_t k = at; k < min(at + cnt, (uint32_t)kSegCap); k++) s_pair[k] = 0xffffffffu;
                a.big_list[(size_t)group * kBinGroup + atomicAdd(&s_nbig, 1u)] = i;
            } else {
#pragma unroll
                for (int q = 0; q < 9; q++) if (hit & (1u << q)) {
                    const uint32_t b = (uint32_t)(lty[q] >> a.bucket_h_log2) * (uint32_t)a.buckets_x + (uint32_t)(tx[q] >> a.bucket_w_log2);
                    const uint32_t tib = (((uint32_t)lty[q] & bh_mask) << a.bucket_w_log2) | ((uint32_t)tx[q] & bw_mask);
                    atomicAdd(&s_cnt[b], 1u);
                    s_pair[at++] = (i - wg_base) | (tib << 10) | (b << 19);
                }
            }
        } else {
            a.big_list[(size_t)group * kBinGroup + atomicAdd(&s_nbig, 1u)] = i;      // -> splat_big_group
        }
        __syncthreads();
    }
    // counting sort of the entries by bucket: exclusive scan of the bucket counts (4 buckets per thread) ...
    const uint32_t b0 = tid * 4u;
    const bool mine = b0 < nb4;                        // (this thread's four buckets exist)
    const uint32_t c0_ = mine ? s_cnt[b0] : 0u, c1_ = mine ? s_cnt[b0 + 1u] : 0u, c2_ = mine ? s_cnt[b0 + 2u] : 0u, c3_ = mine ? s_cnt[b0 + 3u] : 0u, tsum = c0_ + c1_ + c2_ + c3_;
    uint32_t x = tsum;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if ((int)(tid & 63u) >= off) x += y; }
    if ((tid & 63u) == 63u) s_wsum[tid >> 6] = x;
    __syncthreads();
    uint32_t excl = x - tsum;
    for (uint32_t w = 0; w < (tid >> 6); w++) excl += s_wsum[w];
    if (mine) {
        s_off[b0] = excl; s_off[b0 + 1u] = excl + c0_; s_off[b0 + 2u] = excl + c0_ + c1_; s_off[b0 + 3u] = excl + c0_ + c1_ + c2_;
        s_cnt[b0] = 0u; s_cnt[b0 + 1u] = 0u; s_cnt[b0 + 2u] = 0u; s_cnt[b0 + 3u] = 0u;       // now: entries placed per bucket
    }
    __syncthreads();
    const uint32_t n_raw = min(s_n, (uint32_t)kSegCap), n = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];   // slots used, entries
    uint16_t *off_row = a.seg_off + (size_t)group * (a.num_buckets + 1);
    for (uint32_t b = tid; b <= (uint32_t)a.num_buckets; b += 256u) off_row[b] = (uint16_t)(b < (uint32_t)a.num_buckets ? s_off[b] : n);
    // ... and the entries leave for the group's segment in bucket order
    uint32_t *seg = a.seg + (size_t)group * kSegCap;
    for (uint32_t k = tid; k < n_raw; k += 256u) {
        const uint32_t e = s_pair[k], b = e >> 19;
        if (e != 0xffffffffu) seg[s_off[b] + atomicAdd(&s_cnt[b], 1u)] = e & 0x7ffffu;
    }
    if (tid == 0u) a.big_count[group] = s_nbig;
}

// Scatter kernel (kernels.h "Two-level binning"): workgroup (slice, bucket).  Also the pass summary: total entries and the
// fullest bin, one atomic per workgroup on one of 1024 shard lines.
EV_DEV void splat_big_group(const SplatArgs &a, uint32_t *items, uint32_t group);
#if EVPLP_SCATTER_TIMES      // developer build (tools/tile_times.py --scatter): start / end clock of every workgroup, and its entries
__device__ unsigned long long g_scatter_times[3 * 65536];
extern "C" int evplp_debug_scatter_times(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scatter_times), sizeof(unsigned long long) * (size_t)n); }
#endif
__global__ __launch_bounds__(256) void splat_scatter_kernel(SplatArgs a, uint32_t *items) {
#if EVPLP_SCATTER_TIMES
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    struct Stamp { unsigned long long t0; uint32_t id; uint32_t *n; __device__ ~Stamp() { if (threadIdx.x == 0 && id < 65536u) { g_scatter_times[3 * id] = t0; g_scatter_times[3 * id + 1] = __builtin_amdgcn_s_memrealtime(); g_scatter_times[3 * id + 2] = n ? *n : 0xffffffffull; } } };
#endif
    if (blockIdx.y >= (uint32_t)a.num_buckets) {                           // the rows beyond the buckets: photons with large rectangles
        splat_big_group(a, items, (blockIdx.y - (uint32_t)a.num_buckets) * gridDim.x + blockIdx.x);
        return;
    }
    constexpr int kTilesMax = 1 << kMaxBucketTilesLog2;
    constexpr uint32_t kNG = 256u * kScatterG;                            // bin-groups of a slice
    __shared__ uint32_t s_cnt[kTilesMax], s_base[kTilesMax], s_max, s_total;
#if EVPLP_SCATTER_TIMES
    Stamp stamp{ t_start, blockIdx.y * gridDim.x + blockIdx.x, &s_total };
#endif
    __shared__ uint32_t s_start[kNG + 1];                                 // first entry of every group's run in the workgroup's entry numbering
    __shared__ uint16_t s_beg[kNG];                                       // ... and where that run starts in the group's segment
    const uint32_t ntile = 1u << (a.bucket_w_log2 + a.bucket_h_log2);   // tiles of a bucket
    const uint32_t tid = threadIdx.x, b = blockIdx.y, group0 = blockIdx.x * kNG;
    // The runs of this bucket in the segments of the slice's groups are short and uneven (1.6 entries on average at config #3, tens in
    // the hot buckets): one thread per run (round 2) waited for its entries one after the other, and the longest run set the time of
    // the launch.  Now the runs are numbered through (a scan of their lengths) and the workgroup walks the ENTRIES, 256 at a time,
    // each thread finding its run by bisection in LDS: balanced, and every load of a sweep is in flight at once.
#pragma unroll
    for (int g = 0; g < kScatterG; g++) {
        const uint32_t gl = tid + 256u * (uint32_t)g, group = group0 + gl;
        uint32_t beg = 0u, end = 0u;
        if (group < (uint32_t)a.num_bin_groups) {
            const uint16_t *o = a.seg_off + (size_t)group * (a.num_buckets + 1) + b;
            beg = o[0]; end = o[1];
        }
        s_beg[gl] = (uint16_t)beg; s_start[gl] = end - beg;               // (lengths first, scanned below)
    }
    for (uint32_t t = tid; t < ntile; t += 256u) s_cnt[t] = 0u;
    if (tid == 0u) { s_max = 0u; s_total = 0u; }
    __syncthreads();
    if (tid < 64u) {                                                      // exclusive scan of the kNG lengths by one wave
        constexpr uint32_t per = kNG / 64u;
        uint32_t v[per], sum = 0u;
#pragma unroll
        for (uint32_t q = 0; q < per; q++) { v[q] = s_start[tid * per + q]; sum += v[q]; }
        uint32_t x = sum;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if ((int)tid >= off) x += y; }
        uint32_t run = x - sum;
#pragma unroll
        for (uint32_t q = 0; q < per; q++) { s_start[tid * per + q] = run; run += v[q]; }
        if (tid == 63u) s_start[kNG] = run;
    }
    __syncthreads();
    const uint32_t total = s_start[kNG];
    auto entry_of = [&](uint32_t e, uint32_t &group) -> uint32_t {        // entry e of the workgroup: its group and its word
        uint32_t lo = 0u, hi = kNG;                                       // last gl with s_start[gl] <= e
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= e) lo = mid; else hi = mid; }
        group = group0 + lo;
        return a.seg[(size_t)group * kSegCap + s_beg[lo] + (e - s_start[lo])];
    };
    // (round 5) a thread keeps the first kKeep entries it finds -- word and group -- in registers for the placing sweep below: the bisection
    // and the load of the entry were done twice per entry, and the launch is as long as the workgroups of its hottest buckets (3 400 entries at
    // config #3 against a mean of 750: 39 us against 16, tools/scatter_times.py)
    constexpr int kKeep = 14;
    uint32_t kw[kKeep], kg[kKeep];
#pragma unroll
    for (int q = 0; q < kKeep; q++) {
        const uint32_t e = tid + 256u * (uint32_t)q;
        kw[q] = 0u; kg[q] = 0u;
        if (e < total) { kw[q] = entry_of(e, kg[q]); atomicAdd(&s_cnt[kw[q] >> 10], 1u); }
    }
    for (uint32_t e = tid + 256u * (uint32_t)kKeep; e < total; e += 256u) { uint32_t group; atomicAdd(&s_cnt[entry_of(e, group) >> 10], 1u); }
    __syncthreads();
    const uint32_t bx = b % (uint32_t)a.buckets_x, by = b / (uint32_t)a.buckets_x, bw_mask = (1u << a.bucket_w_log2) - 1u;
    auto tile_of = [&](uint32_t t) -> uint32_t {
        return ((by << a.bucket_h_log2) + (t >> a.bucket_w_log2)) * (uint32_t)a.tiles_x + (bx << a.bucket_w_log2) + (t & bw_mask);
    };
    for (uint32_t t = tid; t < ntile; t += 256u) {
        const uint32_t c = s_cnt[t];
        if (c != 0u) {
            const uint32_t at = atomicAdd(&a.tile_cursor[tile_of(t)], c);
            s_base[t] = at; s_cnt[t] = 0u;                               // now: entries placed per tile
            atomicMax(&s_max, at + c); atomicAdd(&s_total, c);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kKeep; q++) {
        const uint32_t e = tid + 256u * (uint32_t)q;
        if (e < total) {
            const uint32_t w = kw[q], t = w >> 10, pos = s_base[t] + atomicAdd(&s_cnt[t], 1u);
            if (pos < a.bin_stride) items[(size_t)tile_of(t) * a.bin_stride + pos] = kg[q] * (uint32_t)kBinGroup + (w & 1023u);
        }
    }
    for (uint32_t e = tid + 256u * (uint32_t)kKeep; e < total; e += 256u) {
        uint32_t group;
        const uint32_t w = entry_of(e, group), t = w >> 10, pos = s_base[t] + atomicAdd(&s_cnt[t], 1u);
        if (pos < a.bin_stride) items[(size_t)tile_of(t) * a.bin_stride + pos] = group * (uint32_t)kBinGroup + (w & 1023u);
    }
    if (tid == 0u && s_total != 0u) {
        uint32_t *sh = a.summary + ((blockIdx.y * gridDim.x + blockIdx.x) & (uint32_t)(kSummaryShards - 1)) * kSummaryStride;
        atomicAdd(&sh[0], s_total);
        const uint32_t fullest = s_max;
        if (fullest > __builtin_nontemporal_load(&sh[1])) atomicMax(&sh[1], fullest);
        if (fullest > a.bin_stride) atomicMax(a.overflow, fullest);
    }
}

// Photons whose rectangle is larger than 3x3 tiles (huge radii, or closer to the eye than a few radii) and those that did not
// fit into their group's segment: workgroup g takes the list of bin-group g, one wave per photon, lanes over the tiles of the
// rectangle, one atomic per entry.  Rectangles of more than 64 tiles are not depth-culled.  (Rows of the scatter launch beyond
// the buckets: one launch less.)
EV_DEV void splat_big_group(const SplatArgs &a, uint32_t *items, uint32_t group) {
    __shared__ uint32_t s_sum[4][2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (group >= (uint32_t)a.num_bin_groups) return;
    const uint32_t nbig = a.big_count[group];
    if (nbig == 0u) return;
    uint32_t entries = 0u, fullest = 0u;
    for (uint32_t k = wave; k < nbig; k += 4u) {
        const uint32_t i = a.big_list[(size_t)group * kBinGroup + k];
        const V3 c0 = v3(a.compact[(size_t)i * kCompactF4]);
        const uint2 rc = photon_rect(a, c0);
        const int tx0 = rc.x & 0xffff, tx1 = rc.x >> 16, ty0 = rc.y & 0xffff, ty1 = rc.y >> 16;
        const float reach2 = photon_reach2(a, c0);
        const uint32_t nx = (uint32_t)(tx1 - tx0 + 1), total = nx * (uint32_t)(ty1 - ty0 + 1);
        const bool small = total <= 64u;
        for (uint32_t q = lane; q < total; q += 64u) {
            const int tx = tx0 + (int)(q % nx), ly = local_tile_row(a, ty0 + (int)(q / nx));
            if (ly < 0) continue;
            const int t = ly * a.tiles_x + tx;
            if (small && !box_within(a.tile_box[2 * t], a.tile_box[2 * t + 1], c0, reach2)) continue;
            const uint32_t slot = atomicAdd(&a.tile_cursor[t], 1u);
            entries++; fullest = max(fullest, slot + 1u);
            if (slot < a.bin_stride) items[(size_t)t * a.bin_stride + slot] = i;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { entries += __shfl_xor(entries, off); fullest = max(fullest, (uint32_t)__shfl_xor((int)fullest, off)); }
    if (lane == 0u) { s_sum[wave][0] = entries; s_sum[wave][1] = fullest; }
    __syncthreads();
    if (tid == 0u) {
        entries = s_sum[0][0] + s_sum[1][0] + s_sum[2][0] + s_sum[3][0];
        fullest = max(max(s_sum[0][1], s_sum[1][1]), max(s_sum[2][1], s_sum[3][1]));
        if (entries != 0u) {
            uint32_t *sh = a.summary + (group & (uint32_t)(kSummaryShards - 1)) * kSummaryStride;
            atomicAdd(&sh[0], entries);
            if (fullest > __builtin_nontemporal_load(&sh[1])) atomicMax(&sh[1], fullest);
            if (fullest > a.bin_stride) atomicMax(a.overflow, fullest);
        }
    }
}

// deterministic mode: rank sort of every bin (ids are unique) so pixels accumulate in record order
__global__ __launch_bounds__(256) void splat_sort_kernel(const uint32_t *cursor, uint32_t stride, const uint32_t *src, uint32_t *dst, const uint32_t *overflow) {
    const uint32_t tile = blockIdx.x;
    if (*overflow != 0u) return;
    const uint32_t b = tile * stride, e = b + min(cursor[tile], stride);
    for (uint32_t i = b + threadIdx.x; i < e; i += 256) {
        uint32_t v = src[i], rank = 0;
        for (uint32_t j = b; j < e; j++) rank += src[j] < v ? 1u : 0u;
        dst[b + rank] = v;
    }
}

// PROXY (EVPLP_FOOTPRINT_PROXY, include/evplp.h): a (pixel, photon) pair inside the radius counts once per face of the scaled proxy mesh
// that the pixel's eye ray crosses between the near plane and the visible surface -- the reference's instanced, un-culled, depth-tested
// draw (rtcomphoton.h:653-655, 789-837; photonsplatinstanced.vert:28-33, .geom:16-32).  For a convex mesh that is an interval question:
// with R = eye + depth x direction (the point of the pixel's ray at the depth of its G-buffer point), D = depth x direction and
// x(s) = R + s D (s = 0 at the surface depth, s = -1 at the eye) the ray is inside the mesh for s in [enter, exit], the intersection of its
// intervals with the mesh's slabs (kernels.h ProxyDev); fragments = [near <= enter <= 0] + [near <= exit <= 0] when enter < exit, else 0.
// Written around R because R - P_i is small (<= r) where eye - P_i is metres long.
//   * A pair whose surface point lies within the sphere inscribed in the mesh (|X - P|^2 <= (rin r)^2: 87 % of the pairs for the
//     icosphere) has it inside the mesh, its entry face in front of the surface and its exit face behind it: one fragment, no test (pixels
//     closer to the near plane than the proxy is large, and pixels whose G-buffer point is off their ray, take the full test for every pair).
//   * The others ("rim" pairs) are compacted into a list of up to 64 per trip -- rounds of "every pixel's next rim photon", placed by the
//     round's ballot -- and tested ONE PAIR PER GROUP OF L LANES (L = 16, 8, 4, 2 or 1, the largest that fits the list into the wave), each
//     lane of a group taking every L-th slab: 13 vector instructions and one LDS read per (pair, slab), the group's interval folded by
//     log2 L cross-lane steps.  (With lane = pixel every lane would walk all slabs for the 0.4 rim pairs per pixel and batch.)
//   * The verdicts go back to the pixels' lanes as two 64-bit masks through LDS (kill: no fragment, dbl: two), and the shading loop below
//     runs as in the ideal kernel on the pairs that are left, adding a doubled pair's value twice.
// Measured (configs #3 / #4, tile kernel alone): ideal 82 / 56 us, proxy 136 / 97 us.  What the proxy rule costs, by compiling parts out:
// the second mask of the radius pass and the per-pixel ray ~15 us, the rim lists ~15, the slab loops 40 / 20, and the occupancy the
// rest -- the kernel follows its occupancy (the ideal kernel padded to six workgroups per CU: 117 / 66 us), which is why the LDS is sized
// by the launch, the list and the verdicts share 1 KB per wave, and the variant is held to 72 registers.
#if EVPLP_TILE_TIMES         // developer build (tools/tile_times.py): start / end clock (100 MHz) and bin entries of every tile (its first wave)
__device__ unsigned long long g_tile_times[3 * 65536];
extern "C" int evplp_debug_tile_times(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_times), sizeof(unsigned long long) * (size_t)n); }
#endif
// Which tiles get four waves (MIXED launches of splat_tiles_kernel): a tile whose bin holds at least heavy_threshold entries goes on
// the heavy list (at most heavy_cap of them; the count sits behind the pass summary, cleared by splat_bin) and is flagged, so that the
// one-wave workgroup that would have had it leaves it alone.  One thread per tile, between the scatter and the tile kernel.
__global__ __launch_bounds__(256) void splat_heavy_kernel(SplatArgs a) {
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    if (tile >= (uint32_t)(a.tiles_x * a.tiles_y)) return;
    bool heavy = min(a.tile_cursor[tile], a.bin_stride) >= a.heavy_threshold;
    if (heavy) { const uint32_t pos = atomicAdd(&a.summary[kSummaryHeavy], 1u); if (pos < a.heavy_cap) a.heavy_list[pos] = tile; else heavy = false; }
    a.tile_flags[tile] = heavy ? (uint8_t)1 : (uint8_t)0;
}
// WAVES = waves per tile: 1 (four tiles per workgroup), 4 (one tile per workgroup), or 0 = MIXED (round 5): the first heavy_cap
// workgroups take the tiles of the heavy list with four waves each, the others four light tiles with one wave each.  Per-wave clocks
// (tools/tile_times.py) showed why neither pure variant is right at config #3: with one wave per tile the 5 % of the tiles whose bins
// hold 300 - 1 100 entries run for 105 - 118 us while the median tile takes 29 -- the last third of the launch belongs to them alone --
// and with four waves per tile the other 95 % pay three idle waves each.
template <int WAVES, bool PROXY>
__global__ __launch_bounds__(256, PROXY ? (WAVES == 4 ? 6 : WAVES == 0 ? EVPLP_PROXY_MIXED_WAVES : EVPLP_PROXY_WAVES) : 8) void splat_tiles_kernel(SplatArgs a) {
    // One workgroup = one tile; its four waves share the bin (wave w takes the 64-photon batches w, w + 4, ...) and
    // their per-pixel sums are folded in wave order: the fullest bins (tiles that see a floor at grazing angle) set
    // the duration of the launch.
    // Dynamic LDS, sized by the launch (splat_tiles_lds_bytes): the tile kernel's speed follows its occupancy (measured with a padded copy of
    // the ideal kernel: 82 -> 117 us at config #3 when the LDS leaves six workgroups per CU instead of eight), so nothing is reserved that
    // the pass does not use.  [4 waves][rows x 64] float4 of staged photons (rows = 3; 4 when misMode 5 needs brdf2; after its last batch a
    // wave's stage carries its sums to the fold), then for PROXY [4 waves][64] float4 shared by the rim list and the verdicts of a trip, and
    // the mesh's slabs.
    extern __shared__ float4 dyn_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the tile, its bin and every loop bound below are wave-uniform)
    const int stage_rows = a.fp.mis_mode == 5u ? 4 : 3;
    float4 *const stage_base = dyn_lds;
    float4 *const aux = dyn_lds + 4 * stage_rows * 64 + wave * 64;          // PROXY: this wave's rim list (float4 entries) / verdicts (4 words per pixel)
    float4 *const s_slab = dyn_lds + 4 * stage_rows * 64 + 4 * 64;          // PROXY: (n, r h+) ...
    float *const s_slabw = reinterpret_cast<float *>(s_slab + a.proxy_count);   // ... and r (h+ + h-)
    if (PROXY) {
        for (int i = tid; i < a.proxy_count; i += 256) {
            const float4 sl = a.proxy_slabs[i]; const float hm = a.proxy_hm[i];
            s_slab[i] = make_float4(sl.x, sl.y, sl.z, sl.w * a.fp.photon_radius);
            s_slabw[i] = (sl.w + hm) * a.fp.photon_radius;
        }
        __syncthreads();
    }
#if EVPLP_TILE_TIMES
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr bool MIXED = WAVES == 0;
    const bool heavy_wg = MIXED && blockIdx.x < a.heavy_cap;      // (workgroup-uniform)
    const int W = MIXED ? (heavy_wg ? 4 : 1) : WAVES;             // waves of this tile
    const int part = W == 4 ? wave : 0;                           // this wave's share of the bin
    int tile = W == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (MIXED && !heavy_wg) tile = ((int)blockIdx.x - (int)a.heavy_cap) * 4 + wave;
    if (blockIdx.x == 0 && wave == 0) {                                   // fold the bin kernel's summary shards
        uint32_t sum = 0u, mx = 0u;
        for (int k = lane; k < kSummaryShards; k += 64) { sum += a.summary[k * kSummaryStride]; mx = max(mx, a.summary[k * kSummaryStride + 1]); }
        for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); }
        if (lane == 0) { a.summary[kSummaryFinal] = sum; a.summary[kSummaryFinal + 1] = mx; a.summary[kSummaryFinal + 2] = *a.overflow; }   // one read-back for the host
    }
    if (MIXED && heavy_wg) {                                      // (the whole workgroup leaves together: no barrier is left behind)
        if (blockIdx.x >= min(a.summary[kSummaryHeavy], a.heavy_cap)) return;
        tile = (int)a.heavy_list[blockIdx.x];
    }
    if (tile >= a.tiles_x * a.tiles_y || *a.overflow != 0u) return;                    // (one-wave tiles only: whole waves leave, no barrier below)
    if (MIXED && !heavy_wg && a.tile_flags[tile] != 0) return;                         // (a heavy tile: its own workgroup has it)
    const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
    const int x = tx * 8 + (lane & 7), ly = lty * 8 + (lane >> 3);
    const bool in_image = x < a.st.W && ly < a.st.local_rows && a.st.global_row(min(ly, a.st.local_rows - 1)) < a.st.H;
    const size_t p = (size_t)min(ly, a.st.local_rows - 1) * a.st.W + min(x, a.st.W - 1);
    const uint32_t b = (uint32_t)tile * a.bin_stride, e = b + min((uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_cursor[tile]), a.bin_stride);
#if EVPLP_TILE_TIMES
    if (b >= e && lane == 0 && part == 0 && tile < 65536) { g_tile_times[3 * tile] = t_start; g_tile_times[3 * tile + 1] = __builtin_amdgcn_s_memrealtime(); g_tile_times[3 * tile + 2] = 0ull; }
#endif
    if (b >= e) { if (lane == 0 && part == 0) { a.tile_pairs[tile] = 0u; if (PROXY) a.tile_frags[tile] = 0u; } return; }

    // a wave without a batch of its own (most bins hold one or two) only takes part in the fold below
    const bool has_work = b + 64u * (uint32_t)part < e;
    float4 gp = make_float4(0.f, 0.f, 0.f, 0.f), gn = gp, gd = gp, gs = gp;
    if (has_work) { gp = a.g_pos[p]; gn = a.g_nrm[p]; gd = a.g_dif[p]; gs = a.g_phg[p]; }
    V3 X = v3(gp), sn = v3(gn), sd = v3(gd), sps = v3(gs); float se = gs.w;
    V3 w10 = normalize(v3(a.fp.camera_pos) - X);                          // frag:177
    const float r2 = a.fp.photon_radius * a.fp.photon_radius;             // frag:152
    const uint32_t mode = a.fp.mis_mode;
    const float clampv = a.fp.clamping_value;
    V3 sum = v3(0.f, 0.f, 0.f);
    uint32_t pairs = 0, frags = 0;
    float4 *stage = stage_base + wave * stage_rows * 64;
    // PROXY: the pixel's eye ray is the one through its centre under the jittered matrix of this iteration (uMVP of runPhotonSplat,
    // rtcomphoton.h:982), the depth it is tested against that of the G-buffer point.  The two differ where the pixel shows the emitter:
    // the light mesh is drawn through the UN-jittered matrix (:720-727), so its G-buffer point lies a jitter off the pixel's ray.
    // Dray = depth x direction (the ray's point at the surface depth is eye + Dray); rin2 = the squared radius inside which a pair
    // needs no test (see above) -- none for a pixel closer to the near plane than the proxy can reach (its entry face may be
    // clipped) or whose G-buffer point is off its ray.
    float rin2 = -1.0f;
    V3 dj = v3(0.f, 0.f, 0.f); double tsd = 0.0;           // the pixel's ray direction (depth 1) and the depth of its G-buffer point
    if (PROXY) {
        // The direction in the oracle's operation order, unfused (cam_dir of oracle/evplp_oracle.c; = the deferred pass's ray, kernels_trace.hip).
        // Whether a rim pair's surface point is inside its proxy is decided within ~1e-7 r of a face for ~1e-7 of the pairs; the point itself is
        // metres from the origin, so eye + depth x direction - photon is formed in DOUBLE (full rate on this part, six operations per rim pair)
        // and only the small difference goes on in fp32: in fp32 the point's own rounding (1e-6 m against r ~ 3e-2 m, along a boundary
        // ~10 r long) moved 2e-5 of the pairs across a face (measured: tools/debug_footprint.py).
        const int gy = a.st.global_row(min(ly, a.st.local_rows - 1));
        const float ndx = __fsub_rn(__fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)x, 0.5f), (float)a.st.W), 2.0f), 1.0f), a.fp.jitter[0]);
        const float ndy = __fsub_rn(__fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn((float)gy, 0.5f), (float)a.st.H), 2.0f), 1.0f), a.fp.jitter[1]);
        const float dxs = __fmul_rn(__fmul_rn(ndx, a.cam.aspect), a.cam.tan_half), dys = __fmul_rn(ndy, a.cam.tan_half);
        dj = v3(__fadd_rn(__fadd_rn(__fmul_rn(a.cam.s[0], dxs), __fmul_rn(a.cam.u[0], dys)), a.cam.f[0]),
                __fadd_rn(__fadd_rn(__fmul_rn(a.cam.s[1], dxs), __fmul_rn(a.cam.u[1], dys)), a.cam.f[1]),
                __fadd_rn(__fadd_rn(__fmul_rn(a.cam.s[2], dxs), __fmul_rn(a.cam.u[2], dys)), a.cam.f[2]));
        tsd = ((double)X.x - (double)a.cam.eye[0]) * (double)a.cam.f[0] + ((double)X.y - (double)a.cam.eye[1]) * (double)a.cam.f[1] +
              ((double)X.z - (double)a.cam.eye[2]) * (double)a.cam.f[2];
        const float tsurf = (float)tsd;
        const V3 off = dj * tsurf - (X - v3(a.cam.eye));
        const float rin = a.proxy_rin * a.fp.photon_radius * (1.0f - 1.0e-5f) - 4.0e-6f * tsurf;
        if (tsurf >= 0.1f + (1.0f + a.proxy_rout) * a.fp.photon_radius * 1.001f && dot(off, off) <= 1.0e-11f * tsurf * tsurf && rin > 0.0f) rin2 = rin * rin;
    }
    // no specular lobe anywhere in the tile: PhongEval is rho_s * (...) = exactly 0, skip its powf (wave-uniform)
    const bool tile_glossy = __ballot(sps.x != 0.0f || sps.y != 0.0f || sps.z != 0.0f) != 0ull;

    for (uint32_t base = b + 64u * (uint32_t)part; base < e; base += 64u * (uint32_t)W) {
        uint32_t n = min(64u, e - base);
        if ((uint32_t)lane < n) {
            uint32_t id = a.bin_items[base + lane];
            const float4 *c = a.compact + (size_t)id * kCompactF4;
            float4 c0 = c[0], c1 = c[1], c2 = c[2];
            stage[0 * 64 + lane] = c0; stage[1 * 64 + lane] = c1;         // [field][photon]: a lane-per-photon read is conflict-free
            stage[2 * 64 + lane] = c2;
            if (mode == 5u) stage[3 * 64 + lane] = c[3];
        } else stage[lane] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);       // (an empty slot of the batch: infinitely far from every pixel)
        __builtin_amdgcn_wave_barrier();
        // Pass 1: the radius test of every (pixel, photon) of the batch -> one 64-bit mask per pixel.  A photon reaches ~3 of a
        // tile's 64 pixels at the radii of a converging run: running the shading under `if (inside)` for every photon that
        // reaches ANY pixel kept 3 lanes of 64 busy.
        // (eight photons per trip with literal bits, the trip's byte shifted into place: as one counted loop with a 64-bit shift per
        // photon the compiler spent 16 vector instructions per photon, seven of them on the loop counter and the shift; unrolled over
        // all 64 it keeps every photon's position in registers at once and spills)
        uint32_t mlo = 0u, mhi = 0u, ilo = 0u, ihi = 0u;
#pragma unroll 1
        for (uint32_t o = 0; o * 8u < n; o++) {
            const float4 *sp = stage + o * 8u;
            uint32_t b8 = 0u, i8 = 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const V3 dv = v3(sp[j]) - X;
                const float d2 = dot(dv, dv);
                if (!(d2 > r2)) b8 |= 1u << j;                            // frag:153-154
                if (PROXY && !(d2 > rin2)) i8 |= 1u << j;
            }
            if (o < 4u) { mlo |= b8 << (8u * o); ilo |= i8 << (8u * o); } else { mhi |= b8 << (8u * (o - 4u)); ihi |= i8 << (8u * (o - 4u)); }
        }
        uint64_t mask = ((uint64_t)mhi << 32) | (uint64_t)mlo;
        if (!in_image) mask = 0ull;
        pairs += (uint32_t)__builtin_popcountll(mask);
        uint64_t dbl = 0ull;
        if (PROXY) {
            uint64_t rim = mask & ~(((uint64_t)ihi << 32) | (uint64_t)ilo);
            uint64_t kill = 0ull;
            while (__ballot(rim != 0ull) != 0ull) {
                // The next up to 64 rim pairs of the wave: rounds of "every pixel's next rim photon", compacted by the round's ballot.  The
                // pixel's lane leaves its pair as (ray point - photon, who) in the list: the pair's lanes need nothing else from it but the ray.
                uint32_t np = 0u;
                for (;;) {
                    const uint64_t have = __ballot(rim != 0ull);
                    const uint32_t nh = (uint32_t)__builtin_popcountll(have);
                    if (nh == 0u || np + nh > 64u) break;
                    if (rim != 0ull) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(rim);
                        rim &= rim - 1ull;
                        const float4 pc = stage[j];
                        const uint32_t at = np + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0u));
                        aux[at] = make_float4((float)(__builtin_fma(tsd, (double)dj.x, (double)a.cam.eye[0]) - (double)pc.x),
                                                       (float)(__builtin_fma(tsd, (double)dj.y, (double)a.cam.eye[1]) - (double)pc.y),
                                                       (float)(__builtin_fma(tsd, (double)dj.z, (double)a.cam.eye[2]) - (double)pc.z), __uint_as_float((uint32_t)lane | (j << 6)));
                    }
                    np += nh;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const int lg = np <= 4u ? 4 : np <= 8u ? 3 : np <= 16u ? 2 : np <= 32u ? 1 : 0;      // lanes per pair = 1 << lg
                const uint32_t pair = (uint32_t)lane >> lg, sub = (uint32_t)lane & ((1u << lg) - 1u), step = 1u << lg;
                const bool act = pair < np;
                const float4 ent = aux[act ? pair : 0u];
                const uint32_t who = __float_as_uint(ent.w), pl = who & 63u, pj = who >> 6;
                // (every pair has its entry in registers: the same 1 KB now takes the verdicts of this trip, four words per pixel)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                aux[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                uint32_t *const verdict = reinterpret_cast<uint32_t *>(aux);
                const V3 q = v3(ent);
                const float tsf = (float)tsd;
                const V3 D = v3(__shfl(dj.x * tsf, (int)pl), __shfl(dj.y * tsf, (int)pl), __shfl(dj.z * tsf, (int)pl));
                float enter = -3.0e38f, exitp = 3.0e38f;
                for (uint32_t i = sub; i < (uint32_t)a.proxy_count; i += step) {
                    const float4 sl = s_slab[i]; const float w = s_slabw[i];
                    const V3 nn = v3(sl);
                    const float av = dot(nn, q), bv = dot(nn, D);
                    // (a ray parallel to the slab: the clamped reciprocal keeps both parameters finite and on the right sides)
                    const float rb = __builtin_amdgcn_fmed3f(rcp_hw(bv), -1.0e30f, 1.0e30f);
                    const float tp = (sl.w - av) * rb, tm = __builtin_fmaf(-w, rb, tp);
                    enter = fmaxf(enter, fminf(tp, tm)); exitp = fminf(exitp, fmaxf(tp, tm));
                }
                for (uint32_t off = 1u; off < step; off <<= 1) {
                    enter = fmaxf(enter, __shfl_xor(enter, (int)off)); exitp = fminf(exitp, __shfl_xor(exitp, (int)off));
                }
                if (act && sub == 0u) {
                    // s = -1 at the eye; the near plane (view depth 0.1, rtcommon.h:586) at 0.1 / depth(X) - 1; LEQUAL with the
                    // oracle's relative slack
                    const float nearp = 0.1f * rcp_hw(dot(D, v3(a.cam.f))) - 1.0f, slack = 1.0e-7f;
                    int count = 0;
                    if (enter < exitp) count = ((enter >= nearp && enter <= slack) ? 1 : 0) + ((exitp >= nearp && exitp <= slack) ? 1 : 0);
                    const uint32_t bit = 1u << (pj & 31u), word = pj >> 5;
                    if (count == 0) atomicOr(&verdict[pl * 4u + word], bit);
                    else if (count == 2) atomicOr(&verdict[pl * 4u + 2u + word], bit);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const float4 vd = aux[lane];
                kill |= ((uint64_t)__float_as_uint(vd.y) << 32) | (uint64_t)__float_as_uint(vd.x);
                dbl |= ((uint64_t)__float_as_uint(vd.w) << 32) | (uint64_t)__float_as_uint(vd.z);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            mask &= ~kill;
            frags += (uint32_t)__builtin_popcountll(mask) + (uint32_t)__builtin_popcountll(dbl);
        }
        // Pass 2: every pixel walks ITS photons in ascending order (the accumulation order of the one-photon-at-a-time loop);
        // the wave runs max-over-pixels iterations instead of one per photon.
        while (__ballot(mask != 0ull) != 0ull) {
            if (mask != 0ull) {
                const uint32_t j = (uint32_t)__builtin_ctzll(mask);
                mask &= mask - 1ull;
                float4 c0 = stage[j], c1 = stage[64 + j], c2 = stage[128 + j];
                V3 w12 = v3(c1);
                V3 brdf1 = g_lambert_eval(w10, w12, sn, sd);                                       // frag:181
                if (tile_glossy) brdf1 = brdf1 + g_phong_eval(w10, w12, sn, sps, se);
                if (c2.w != 0.0f) {                                       // mixPdfW > 0, frag:191
                    V3 col;
                    if (mode <= 3u) col = brdf1 * v3(c2);
                    else {
                        float cc = fmaxf(dot(sn, w12), 0.0f) * c0.w;      // frag:216,226
                        if (cc <= 0.0f) col = v3(0.f, 0.f, 0.f);          // discard
                        else {
                            float g = cc * rcp_hw(c1.w);
                            if (mode == 4u) col = (brdf1 * v3(c2)) * (fmaxf(g - clampv, 0.0f) * rcp_hw(g));
                            else {
                                V3 brdf2 = v3(stage[192 + j]);
                                V3 num = (brdf1 * brdf2) * g;
                                num = v3(fmaxf(num.x - clampv, 0.f), fmaxf(num.y - clampv, 0.f), fmaxf(num.z - clampv, 0.f));
                                V3 den = brdf2 * g;
                                V3 pre = v3(c2);
                                // zero denominator contributes 0 (the reference produces NaN here)
                                col = v3(den.x != 0.f ? pre.x * num.x * rcp_hw(den.x) : 0.f, den.y != 0.f ? pre.y * num.y * rcp_hw(den.y) : 0.f,
                                         den.z != 0.f ? pre.z * num.z * rcp_hw(den.z) : 0.f);
                            }
                        }
                    }
                    sum = sum + col;
                    if (PROXY && ((dbl >> j) & 1ull) != 0ull) sum = sum + col;   // both faces of the proxy lie in front of the surface: two fragments, blended one after the other
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int off = 32; off > 0; off >>= 1) { pairs += __shfl_down(pairs, off); if (PROXY) frags += __shfl_down(frags, off); }
    if (W == 4) {
        // (.w: lane 0 carries the wave's pairs; lane 1 -- whose own count went into lane 0's -- the wave's proxy fragments)
        const uint32_t frags0 = (uint32_t)__shfl((int)frags, 0);
        if (wave != 0) stage[lane] = make_float4(sum.x, sum.y, sum.z, lane == 0 ? __uint_as_float(pairs) : lane == 1 ? __uint_as_float(frags0) : 0.f);
        __syncthreads();
    }
    if (part == 0) {
        if (W == 4) for (int w = 1; w < 4; w++) {
            const float4 *other = stage_base + w * stage_rows * 64;
            float4 q = other[lane]; sum = sum + v3(q);
            if (lane == 0) { pairs += __float_as_uint(q.w); if (PROXY) frags += __float_as_uint(other[1].w); }
        }
#if EVPLP_DEBUG_NAN
        if (in_image && !(isfinite(sum.x) && isfinite(sum.y) && isfinite(sum.z))) atomicAdd(&a.counters->nonfinite, 1ull);
#endif
        if (in_image) {
            float4 o = a.out[p];
            a.out[p] = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w);   // additive blend ONE, ONE (:793)
        }
#if EVPLP_TILE_TIMES
        if (lane == 0 && tile < 65536) { g_tile_times[3 * tile] = t_start; g_tile_times[3 * tile + 1] = __builtin_amdgcn_s_memrealtime(); g_tile_times[3 * tile + 2] = (unsigned long long)(e - b); }
#endif
        if (lane == 0) {
            a.tile_pairs[tile] = pairs;
            if (PROXY) a.tile_frags[tile] = frags;
            // running total of the context (never cleared; words 4-5 of the 1024 summary lines): lets a caller count pairs over many
            // passes without reading anything back in between
            if (pairs) atomicAdd(reinterpret_cast<unsigned long long *>(&a.summary[((uint32_t)tile & (uint32_t)(kSummaryShards - 1)) * kSummaryStride + 4]), (unsigned long long)pairs);
        }
    }
}

// Phase A: tile boxes (+ cleared cursors and summary), compact photons + bucket-sorted segments, segments -> tile bins
// (+ summary), large photons.
void launch_tile_boxes(const StripDev &st, const float4 *g_pos, float4 *tile_box, int tiles_x, int tiles_y, hipStream_t s) {
    SplatArgs a; std::memset(&a, 0, sizeof(a));
    a.st = st; a.g_pos = g_pos; a.tile_box = tile_box; a.tiles_x = tiles_x; a.tiles_y = tiles_y;
    const uint32_t ntiles = (uint32_t)(tiles_x * tiles_y);
    if (ntiles) hipLaunchKernelGGL(splat_tile_box_kernel, dim3((ntiles + 15) / 16), dim3(256), 0, s, a);
}
void launch_splat_bin(const SplatArgs &a, hipStream_t s) {
    const uint32_t ntiles = (uint32_t)(a.tiles_x * a.tiles_y);
    if (!a.boxes_valid) hipLaunchKernelGGL(splat_tile_box_kernel, dim3((ntiles + 15) / 16), dim3(256), 0, s, a);   // (else: written by primary_kernel)
    uint32_t *items = a.deterministic ? a.bin_items_tmp : a.bin_items;
    hipLaunchKernelGGL(splat_bin_kernel, dim3((uint32_t)a.num_bin_groups), dim3(256), sizeof(uint32_t) * 2u * (size_t)(((uint32_t)a.num_buckets + 3u) & ~3u), s, a);
    const uint32_t per = 256u * kScatterG, slices = ((uint32_t)a.num_bin_groups + per - 1u) / per, big_rows = ((uint32_t)a.num_bin_groups + slices - 1u) / slices;
    hipLaunchKernelGGL(splat_scatter_kernel, dim3(slices, (uint32_t)a.num_buckets + big_rows), dim3(256), 0, s, a, items);
}
// Phase B: (deterministic: sort the bins) and accumulate the tiles.  Both do nothing when a bin overflowed.
void launch_splat_tiles(const SplatArgs &a, bool split_tiles, hipStream_t s, hipEvent_t dom_begin, hipEvent_t dom_end) {
    const uint32_t ntiles = (uint32_t)(a.tiles_x * a.tiles_y);
    if (a.deterministic) hipLaunchKernelGGL(splat_sort_kernel, dim3(ntiles), dim3(256), 0, s, a.tile_cursor, a.bin_stride, a.bin_items_tmp, a.bin_items, a.overflow);
    if (dom_begin) hipEventRecord(dom_begin, s);
    const bool proxy = a.fp.splat_footprint == (uint32_t)EVPLP_FOOTPRINT_PROXY;
    // dynamic LDS of the tile kernel (its layout is at the top of splat_tiles_kernel)
    const size_t lds_bytes = sizeof(float4) * (size_t)(4 * (a.fp.mis_mode == 5u ? 4 : 3) * 64) + (proxy ? sizeof(float4) * (size_t)(4 * 64 + a.proxy_count) + sizeof(float) * (size_t)a.proxy_count : 0);
    if (a.heavy_list && !a.deterministic) {          // MIXED: heavy tiles with four waves (the first heavy_cap workgroups), the others with one
        hipLaunchKernelGGL(splat_heavy_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, s, a);
        const uint32_t grid = a.heavy_cap + (ntiles + 3) / 4;
        if (proxy) hipLaunchKernelGGL((splat_tiles_kernel<0, true>), dim3(grid), dim3(256), lds_bytes, s, a); else hipLaunchKernelGGL((splat_tiles_kernel<0, false>), dim3(grid), dim3(256), lds_bytes, s, a);
        if (dom_end) hipEventRecord(dom_end, s);
        return;
    }
    if (split_tiles) { if (proxy) hipLaunchKernelGGL((splat_tiles_kernel<4, true>), dim3(ntiles), dim3(256), lds_bytes, s, a); else hipLaunchKernelGGL((splat_tiles_kernel<4, false>), dim3(ntiles), dim3(256), lds_bytes, s, a); }
    else { if (proxy) hipLaunchKernelGGL((splat_tiles_kernel<1, true>), dim3((ntiles + 3) / 4), dim3(256), lds_bytes, s, a); else hipLaunchKernelGGL((splat_tiles_kernel<1, false>), dim3((ntiles + 3) / 4), dim3(256), lds_bytes, s, a); }
    if (dom_end) hipEventRecord(dom_end, s);
}

// shaders/final.frag:19-35
__global__ __launch_bounds__(256) void resolve_kernel(StripDev st, const float4 *vpl, const float4 *pm, const float4 *light,
                                                      float vs, float ps, float ls, int mask_emitter, int gamma, float *out_rgb) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)st.W * st.local_rows;
    if (i >= n) return;
    float4 v = vpl[i], q = pm[i], l = light[i];
    float lx = l.x * ls;
    float stepv = mask_emitter ? ((0.0f < lx) ? 0.0f : 1.0f) : 1.0f;   // step(lightColor.x, 0.0)
    float r = stepv * (v.x * vs + q.x * ps) + l.x * ls;
    float g = stepv * (v.y * vs + q.y * ps) + l.y * ls;
    float b = stepv * (v.z * vs + q.z * ps) + l.z * ls;
    if (gamma) { r = powf(r, 1.0f / 2.2f); g = powf(g, 1.0f / 2.2f); b = powf(b, 1.0f / 2.2f); }
    out_rgb[3 * i + 0] = r; out_rgb[3 * i + 1] = g; out_rgb[3 * i + 2] = b;
}
void launch_resolve(const StripDev &st, const float4 *vpl, const float4 *pm, const float4 *light,
                    float vs, float ps, float ls, int mask_emitter, int gamma, float *out_rgb, hipStream_t s) {
    size_t n = (size_t)st.W * st.local_rows;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, vpl, pm, light, vs, ps, ls, mask_emitter, gamma, out_rgb);
}
// De-interleave of the all-gathered row strips (evplp_group_resolve): gathered = [n ranks][chunk_rows][W][3] (the rows an exchange moves: <= local_rows), rank r's local row l
// is image row StripDev{rank r}.global_row(l); frame = [H][W][3].  One thread per float of the frame.
// bands: rank r owns the rows [bands.first[r], bands.first[r + 1]) (evplp_group with contiguous bands); null = interleaved strips
// owner: the dealt blocks of the group (evplp_group_rebalance), owner[b] = rank << 16 | local block of image block b; null = round-robin
__global__ __launch_bounds__(256) void assemble_strips_kernel(StripDev st, int nranks, BandTable bands, int use_bands, const uint32_t *owner, int chunk_rows, const float *gathered, float *frame) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row_floats = (size_t)st.W * 3, n = row_floats * (size_t)st.H;
    if (i >= n) return;
    const int y = (int)(i / row_floats); const size_t x = i - (size_t)y * row_floats;
    int r, l;
    if (use_bands) { r = 0; while (r + 1 < nranks && y >= bands.first[r + 1]) r++; l = y - bands.first[r]; }
    else {
        const int blk = y / st.strip_rows; int lb;
        if (owner) { const uint32_t o = owner[blk]; r = (int)(o >> 16); lb = (int)(o & 0xffffu); } else { r = blk % nranks; lb = blk / nranks; }
        l = lb * st.strip_rows + (y - blk * st.strip_rows);
    }
    frame[i] = gathered[((size_t)r * chunk_rows + l) * row_floats + x];
}
void launch_assemble_strips(const StripDev &st, int nranks, const BandTable *bands, const uint32_t *owner, int chunk_rows, const float *gathered, float *frame, hipStream_t s) {
    const size_t n = (size_t)st.W * 3 * st.H;
    BandTable none; std::memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(assemble_strips_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, nranks, bands ? *bands : none, bands ? 1 : 0, owner, chunk_rows, gathered, frame);
}
void launch_fill_zero(void *p, size_t bytes, hipStream_t s) { hipMemsetAsync(p, 0, bytes, s); }

} // namespace evplp
