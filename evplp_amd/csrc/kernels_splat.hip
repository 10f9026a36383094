// Image-space photon splat and the final composite.
//   runPhotonSplat + shaders/photonsplatinstanced.{vert,geom,frag}
//       (rt/rtcomphoton/rtcomphoton.h:789-837, frag:146-240)        -> prepare / scan / fill / tiles
//   shaders/final.frag:19-35, rtcomphoton.h:756-787                  -> resolve_kernel
//
// The reference rasterises one icosphere proxy per record and lets the ROP blend every fragment
// (one RMW of HBM per fragment).  Here (ideal kernel of SURVEY A.4: photon i adds to pixel p iff
// |X_p - P_i|^2 <= r^2):
//   0. splat_tile_box : world-space bounding box of the G-buffer positions of every 8x8-pixel tile.
//   1. splat_bin : one lane per record (records staged through LDS).  Everything of the fragment
//      shader that does not depend on the pixel (w12, the MIS/clamp weight, 1/(pi r^2 N) scaling) is
//      folded into a 64-byte compact photon; the photon id goes into the bin of every 8x8-pixel tile of the
//      conservative screen rectangle of its radius-r sphere whose position box the sphere reaches.  Bins are
//      fixed slabs of bin_stride slots per tile (slot = one returning atomic on the tile's cursor): no counting
//      pass, no scan, one scattered atomic per entry instead of two.  A bin that wants more than its slab
//      raises the overflow flag; the host doubles the slabs and runs the pass again (context.cpp settle_splat).
//      (deterministic mode: + rank sort so every pixel accumulates in ascending record order, like the oracle.)
//   2. splat_summary : total entries and fullest bin (sizes the slabs, picks the tile kernel variant).
//   3. splat_tiles   : one wavefront per tile (four when bins are very full), lane = pixel with its G-buffer texel in registers;
//      the bin streams through LDS 64 photons at a time (each lane fetches one compact photon,
//      all lanes then read it back as an LDS broadcast); RGB accumulates in registers and is
//      written once per pixel with coalesced 16-byte stores -- no atomics, no per-fragment RMW.
#include "device_common.hpp"
#include "kernels.h"

namespace evplp {

// GLSL flavours of the BRDF helpers (photonsplatinstanced.frag:42-98; they differ from the CUDA ones)
EV_DEV V3 g_lambert_eval(V3 w10, V3 w12, V3 n, V3 rd) {
    if (dot(w10, n) <= 0.0f || dot(w12, n) <= 0.0f) return v3(0.f, 0.f, 0.f);
    return rd * EV_INV_PI;
}
EV_DEV V3 g_phong_eval(V3 outv, V3 inv_, V3 n, V3 rs, float e) {
    V3 r = reflect(-inv_, n);
    float d = dot(outv, r);
    if (d <= 0.00001f) return v3(0.f, 0.f, 0.f);
    return rs * (e + 2.0f) * powf(d, e) * EV_INV_PI * 0.5f;
}
EV_DEV float g_lambert_pdf_w(V3 n1, V3 v12) { return fmaxf(dot(n1, normalize(v12)), 0.f) * EV_INV_PI; }
EV_DEV float g_phong_pdf_w(V3 n1, V3 wi12, V3 inv_, V3 rs, float e) {
    V3 r = reflect(-inv_, n1);
    float d = fmaxf(dot(wi12, r), 0.f);
    if (d <= 0.00001f || rs.x <= 0.00001f) return 0.0f;
    return (e + 1.0f) * 0.5f * EV_INV_PI * powf(d, e);
}

constexpr int kRecF4 = sizeof(evplp_record) / 16;   // 6 float4 per record
struct Rec { V3 pos, n, flux, fdir, rd, rs; float psel, e; uint32_t flags; };
EV_DEV Rec load_rec(const float4 *q) {
    float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5];
    Rec v; v.pos = v3(a); v.flags = __float_as_uint(a.w); v.n = v3(b); v.psel = b.w; v.flux = v3(c);
    v.fdir = v3(d); v.rd = v3(e); v.rs = v3(f); v.e = f.w;
    return v;
}

// compact photon: [0] pos.xyz, cpn   [1] w12.xyz, d2   [2] wflux.xyz, alive   [3] brdf2.xyz, n1.w12 (unused)
// Everything of one photon that does not depend on the pixel (compact record) + its conservative rectangle of 8x8-px
// tiles, packed (x0 | x1 << 16, y0 | y1 << 16); x0 > x1 = nothing to splat.
EV_DEV uint2 splat_prepare_one(const SplatArgs &a, uint32_t i, const float4 *s_ph, const float4 *s_prev, V3 &photon_pos) {
    const uint2 none = make_uint2(1u, 0u);
    Rec ph = load_rec(s_ph);
    if (!(ph.flags & EVPLP_USABLE_PHOTON)) return none;  // vert:31, geom:20
    Rec prev = load_rec(s_prev);                    // frag:163
    photon_pos = ph.pos;

    const float r = a.fp.photon_radius;
    V3 v12 = prev.pos - ph.pos;                                           // frag:170
    float d2 = dot(v12, v12);
    V3 w12 = normalize(v12);
    float mix_w = g_lambert_pdf_w(prev.n, -w12) * prev.psel;              // frag:184-187
    mix_w += g_phong_pdf_w(prev.n, -w12, prev.fdir, prev.rs, prev.e) * (1.0f - prev.psel);
    float mix_a = mix_w * fmaxf(dot(ph.n, w12), 0.0f) / d2;               // frag:189
    bool alive = mix_w > 0.0f;                                            // frag:191
    float k = EV_INV_PI * (1.0f / (r * r));                               // InvPi * uInvPhotonRadius2
    float inv_n = 1.0f / (float)a.fp.num_light_paths;                     // uInvNumLightPaths
    float wgt = 1.0f;
    const uint32_t mode = a.fp.mis_mode;
    if (mode == 1u) wgt = mix_a / (mix_a + a.fp.pdf_mc);
    else if (mode == 2u) wgt = mix_a > a.fp.pdf_mc ? 1.0f : 0.0f;
    else if (mode == 3u) { float a2 = mix_a * mix_a, b2 = a.fp.pdf_mc * a.fp.pdf_mc; wgt = a2 / (a2 + b2); }
    V3 wflux = ph.flux * (k * inv_n * wgt);
    float cpn = fmaxf(-dot(prev.n, w12), 0.0f);
    V3 brdf2 = g_lambert_eval(-w12, prev.fdir, prev.n, prev.rd) + g_phong_eval(-w12, prev.fdir, prev.n, prev.rs, prev.e);  // frag:182
    float4 *c = a.compact + (size_t)i * kCompactF4;
    c[0] = make_float4(ph.pos.x, ph.pos.y, ph.pos.z, cpn);
    c[1] = make_float4(w12.x, w12.y, w12.z, d2);
    c[2] = make_float4(wflux.x, wflux.y, wflux.z, alive ? 1.0f : 0.0f);
    c[3] = make_float4(brdf2.x, brdf2.y, brdf2.z, 0.f);

    // conservative screen rectangle of every visible point within r of the photon, through the
    // (jittered) camera of this iteration: uMVP of runPhotonSplat is the jittered matrix (:982)
    V3 q = ph.pos - v3(a.cam.eye);
    float vx = dot(q, v3(a.cam.s)), vy = dot(q, v3(a.cam.u)), vz = dot(q, v3(a.cam.f));
    // visible surface points have view depth in [near, far] = [0.1, 100] (rtcommon.h:586): clip the
    // sphere's depth range to it -- a photon closer than r to the camera plane then needs no
    // whole-screen fallback (those few photons used to produce most of the bin entries)
    const float zlo = fmaxf(vz - r, 0.1f), zhi = fminf(vz + r, 100.0f);
    int x0, x1, y0, y1;
    if (zlo > zhi) return none;
    {
        float sx = 1.0f / (a.cam.aspect * a.cam.tan_half), sy = 1.0f / a.cam.tan_half;
        float il = 1.0f / zlo, ih = 1.0f / zhi;
        float nx0 = fminf((vx - r) * il, (vx - r) * ih) * sx + a.fp.jitter[0];
        float nx1 = fmaxf((vx + r) * il, (vx + r) * ih) * sx + a.fp.jitter[0];
        float ny0 = fminf((vy - r) * il, (vy - r) * ih) * sy + a.fp.jitter[1];
        float ny1 = fmaxf((vy + r) * il, (vy + r) * ih) * sy + a.fp.jitter[1];
        float fx0 = (nx0 * 0.5f + 0.5f) * (float)a.st.W - 0.5f, fx1 = (nx1 * 0.5f + 0.5f) * (float)a.st.W - 0.5f;
        float fy0 = (ny0 * 0.5f + 0.5f) * (float)a.st.H - 0.5f, fy1 = (ny1 * 0.5f + 0.5f) * (float)a.st.H - 0.5f;
        // +-1 pixel guard against rounding in the projection
        fx0 = fminf(fmaxf(fx0 - 1.0f, -1.0f), (float)a.st.W); fx1 = fminf(fmaxf(fx1 + 1.0f, -1.0f), (float)a.st.W);
        fy0 = fminf(fmaxf(fy0 - 1.0f, -1.0f), (float)a.st.H); fy1 = fminf(fmaxf(fy1 + 1.0f, -1.0f), (float)a.st.H);
        x0 = max((int)ceilf(fx0), 0); x1 = min((int)floorf(fx1), a.st.W - 1);
        y0 = max((int)ceilf(fy0), 0); y1 = min((int)floorf(fy1), a.st.H - 1);
    }
    if (x0 > x1 || y0 > y1) return none;
    int tx0 = x0 >> 3, tx1 = x1 >> 3, ty0 = y0 >> 3, ty1 = y1 >> 3;   // GLOBAL tile rows
    return make_uint2((uint32_t)tx0 | ((uint32_t)tx1 << 16), (uint32_t)ty0 | ((uint32_t)ty1 << 16));
}

// World-space bounding box of every 8x8-px tile's G-buffer positions (one wave per tile).  A pixel can only receive a
// photon whose centre lies within r of its position: splat_bin drops the (photon, tile) entries whose sphere does not reach
// the tile's box.  Screen-space bins otherwise collect every photon along the tile's frustum -- floor under the table,
// wall behind the chairs -- several times more than ever pass the radius test.  Every in-image pixel counts, background
// included (its position is the clear colour, which is what the radius test of frag:152-154 sees too).
__global__ __launch_bounds__(256) void splat_tile_box_kernel(SplatArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
    const int x = tx * 8 + (lane & 7), ly = lty * 8 + (lane >> 3);
    const bool in_image = x < a.st.W && ly < a.st.local_rows && a.st.global_row(min(ly, a.st.local_rows - 1)) < a.st.H;
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    if (in_image) {
        float4 gp = a.g_pos[(size_t)ly * a.st.W + x];
        lo[0] = hi[0] = gp.x; lo[1] = hi[1] = gp.y; lo[2] = hi[2] = gp.z;
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
    if (lane == 0) { a.tile_box[2 * tile] = make_float4(lo[0], lo[1], lo[2], 0.f); a.tile_box[2 * tile + 1] = make_float4(hi[0], hi[1], hi[2], 0.f); }
}

// The 96-byte AoS records are read ONCE, as a coalesced 16 B/lane stream, into LDS (slot k = record base - 1 + k:
// every photon also needs its predecessor on the light path, frag:163); a lane then picks its two records from LDS.
// One lane per record reading its own 6 float4 at a 96-byte stride touched every line six times and every record
// twice.
__global__ __launch_bounds__(256) void splat_bin_kernel(SplatArgs a, uint32_t *items) {
    __shared__ float4 s_rec[257 * kRecF4];
    const uint32_t base = blockIdx.x * 256u;
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.records);
        const uint32_t first = base == 0u ? 0u : base - 1u, last = min(base + 256u, a.num_records);   // records [first, last)
        const uint32_t n4 = (last - first) * kRecF4, slot0 = (first + 1u - base) * kRecF4;
        for (uint32_t k = threadIdx.x; k < n4; k += 256u) s_rec[slot0 + k] = src[(size_t)first * kRecF4 + k];
    }
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    uint2 rc = make_uint2(1u, 0u);
    V3 c0 = v3(0.f, 0.f, 0.f);
    if (i < a.num_records && i != 0u) rc = splat_prepare_one(a, i, &s_rec[(threadIdx.x + 1u) * kRecF4], &s_rec[threadIdx.x * kRecF4], c0);
    const int tx0 = rc.x & 0xffff, tx1 = rc.x >> 16, ty0 = rc.y & 0xffff, ty1 = rc.y >> 16;
    if (tx0 > tx1) return;
    // squared radius with a guard for the rounding of the distance computation (relative 1e-5 and an absolute term scaled by
    // the coordinates): a tile whose box is farther than that from the photon cannot hold a pixel within r of it
    const float pmag = fmaxf(fmaxf(fabsf(c0.x), fabsf(c0.y)), fmaxf(fabsf(c0.z), 1.0f));
    const float reach = a.fp.photon_radius * (1.0f + 1.0e-5f) + 4.0e-6f * pmag;
    const float reach2 = reach * reach;
    // bin entries of this photon on THIS rank's row strips; rectangles of more than 64 tiles (huge radii) are not depth-culled
    const int tiles_per_block = a.st.strip_rows >> 3;
    const int nx = tx1 - tx0 + 1, ny = ty1 - ty0 + 1;
    const bool small = nx * ny <= 64;
    for (int ty = ty0; ty <= ty1; ty++) {
        int blk = ty / tiles_per_block;
        if (blk % a.st.strip_count != a.st.strip_rank) continue;        // row strip of another GPU
        int lty = (blk / a.st.strip_count) * tiles_per_block + (ty - blk * tiles_per_block);
        for (int tx = tx0; tx <= tx1; tx++) {
            const int tile = lty * a.tiles_x + tx;
            if (small) {
                const float4 blo = a.tile_box[2 * tile], bhi = a.tile_box[2 * tile + 1];
                const float dx = fmaxf(fmaxf(blo.x - c0.x, c0.x - bhi.x), 0.0f), dy = fmaxf(fmaxf(blo.y - c0.y, c0.y - bhi.y), 0.0f),
                            dz = fmaxf(fmaxf(blo.z - c0.z, c0.z - bhi.z), 0.0f);
                if (dx * dx + dy * dy + dz * dz > reach2) continue;
            }
            const uint32_t slot = atomicAdd(&a.tile_cursor[tile], 1u);
            if (slot < a.bin_stride) items[(size_t)tile * a.bin_stride + slot] = i;
            else atomicMax(a.overflow, slot + 1u);
        }
    }
}

// total entries and fullest bin (one workgroup; the cursors of <= a few 100k tiles)
__global__ __launch_bounds__(1024) void splat_summary_kernel(const uint32_t *cursor, uint32_t n, uint32_t *summary) {
    __shared__ uint32_t s_sum, s_max;
    if (threadIdx.x == 0) { s_sum = 0u; s_max = 0u; }
    __syncthreads();
    uint32_t sum = 0u, mx = 0u;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) { const uint32_t v = cursor[i]; sum += v; mx = max(mx, v); }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off); mx = max(mx, (uint32_t)__shfl_down((int)mx, off)); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_sum, sum); atomicMax(&s_max, mx); }
    __syncthreads();
    if (threadIdx.x == 0) { summary[0] = s_sum; summary[1] = s_max; }
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

template <int WAVES>   // waves per tile: 1 (four tiles per workgroup) or 4 (one tile per workgroup, for launches with very full bins)
__global__ __launch_bounds__(256) void splat_tiles_kernel(SplatArgs a) {
    // One workgroup = one tile; its four waves share the bin (wave w takes the 64-photon batches w, w + 4, ...) and
    // their per-pixel sums are folded in wave order: the fullest bins (tiles that see a floor at grazing angle) set
    // the duration of the launch.
    __shared__ float4 lds[4][64 * kCompactF4];
    __shared__ float4 red[3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = WAVES == 4 ? wave : 0;                       // this wave's share of the bin
    const int tile = WAVES == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (tile >= a.tiles_x * a.tiles_y || *a.overflow != 0u) return;                    // (WAVES == 1 only: whole waves leave, no barrier below)
    const int tx = tile % a.tiles_x, lty = tile / a.tiles_x;
    const int x = tx * 8 + (lane & 7), ly = lty * 8 + (lane >> 3);
    const bool in_image = x < a.st.W && ly < a.st.local_rows && a.st.global_row(min(ly, a.st.local_rows - 1)) < a.st.H;
    const size_t p = (size_t)min(ly, a.st.local_rows - 1) * a.st.W + min(x, a.st.W - 1);
    const uint32_t b = (uint32_t)tile * a.bin_stride, e = b + min(a.tile_cursor[tile], a.bin_stride);
    if (b >= e) { if (lane == 0 && part == 0) a.tile_pairs[tile] = 0u; return; }

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
    uint32_t pairs = 0;
    float4 *stage = lds[wave];
    // no specular lobe anywhere in the tile: PhongEval is rho_s * (...) = exactly 0, skip its powf (wave-uniform)
    const bool tile_glossy = __ballot(sps.x != 0.0f || sps.y != 0.0f || sps.z != 0.0f) != 0ull;

    for (uint32_t base = b + 64u * (uint32_t)part; base < e; base += 64u * WAVES) {
        uint32_t n = min(64u, e - base);
        if ((uint32_t)lane < n) {
            uint32_t id = a.bin_items[base + lane];
            const float4 *c = a.compact + (size_t)id * kCompactF4;
            float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            stage[lane * kCompactF4 + 0] = c0; stage[lane * kCompactF4 + 1] = c1;
            stage[lane * kCompactF4 + 2] = c2; stage[lane * kCompactF4 + 3] = c3;
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = 0; j < n; j++) {
            float4 c0 = stage[j * kCompactF4 + 0];
            V3 dv = v3(c0) - X;
            bool inside = in_image && !(dot(dv, dv) > r2);                // frag:153-154
            if (__ballot(inside) == 0ull) continue;
            float4 c1 = stage[j * kCompactF4 + 1], c2 = stage[j * kCompactF4 + 2];
            if (inside) {
                pairs++;
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
                            float g = cc / c1.w;
                            if (mode == 4u) col = (brdf1 * v3(c2)) * fmaxf(g - clampv, 0.0f) / g;
                            else {
                                V3 brdf2 = v3(stage[j * kCompactF4 + 3]);
                                V3 num = (brdf1 * brdf2) * g;
                                num = v3(fmaxf(num.x - clampv, 0.f), fmaxf(num.y - clampv, 0.f), fmaxf(num.z - clampv, 0.f));
                                V3 den = brdf2 * g;
                                V3 pre = v3(c2);
                                // zero denominator contributes 0 (the reference produces NaN here)
                                col = v3(den.x != 0.f ? pre.x * num.x / den.x : 0.f, den.y != 0.f ? pre.y * num.y / den.y : 0.f,
                                         den.z != 0.f ? pre.z * num.z / den.z : 0.f);
                            }
                        }
                    }
                    sum = sum + col;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int off = 32; off > 0; off >>= 1) pairs += __shfl_down(pairs, off);
    if (WAVES == 4) {
        if (wave != 0) red[wave - 1][lane] = make_float4(sum.x, sum.y, sum.z, lane == 0 ? __uint_as_float(pairs) : 0.f);
        __syncthreads();
    }
    if (part == 0) {
        if (WAVES == 4) for (int w = 0; w < 3; w++) { float4 q = red[w][lane]; sum = sum + v3(q); if (lane == 0) pairs += __float_as_uint(q.w); }
        if (in_image) {
            float4 o = a.out[p];
            a.out[p] = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w);   // additive blend ONE, ONE (:793)
        }
        if (lane == 0) a.tile_pairs[tile] = pairs;
    }
}

// Phase A: tile depth ranges, compact photons + bins, summary.
void launch_splat_bin(const SplatArgs &a, hipStream_t s) {
    const uint32_t ntiles = (uint32_t)(a.tiles_x * a.tiles_y);
    hipMemsetAsync(a.tile_cursor, 0, sizeof(uint32_t) * ntiles, s);
    hipLaunchKernelGGL(splat_tile_box_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, s, a);
    const uint32_t nb = (a.num_records + 255) / 256;
    hipLaunchKernelGGL(splat_bin_kernel, dim3(nb), dim3(256), 0, s, a, a.deterministic ? a.bin_items_tmp : a.bin_items);
    hipLaunchKernelGGL(splat_summary_kernel, dim3(1), dim3(1024), 0, s, a.tile_cursor, ntiles, a.summary);
}
// Phase B: (deterministic: sort the bins) and accumulate the tiles.  Both do nothing when a bin overflowed.
void launch_splat_tiles(const SplatArgs &a, bool split_tiles, hipStream_t s, hipEvent_t dom_begin, hipEvent_t dom_end) {
    const uint32_t ntiles = (uint32_t)(a.tiles_x * a.tiles_y);
    if (a.deterministic) hipLaunchKernelGGL(splat_sort_kernel, dim3(ntiles), dim3(256), 0, s, a.tile_cursor, a.bin_stride, a.bin_items_tmp, a.bin_items, a.overflow);
    if (dom_begin) hipEventRecord(dom_begin, s);
    if (split_tiles) hipLaunchKernelGGL(splat_tiles_kernel<4>, dim3(ntiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(splat_tiles_kernel<1>, dim3((ntiles + 3) / 4), dim3(256), 0, s, a);
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
void launch_fill_zero(void *p, size_t bytes, hipStream_t s) { hipMemsetAsync(p, 0, bytes, s); }

} // namespace evplp
