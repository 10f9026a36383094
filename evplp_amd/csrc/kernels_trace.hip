// Feeder kernels: primary visibility (G-buffer) and light sub-path tracing.
//   primary_kernel      <- shaders/deferred.{vert,geom,frag} + light.{vert,frag} drawn by
//                          runDeferredProgram / runLightProgram (rt/rtcomphoton/rtcomphoton.h:710-754, 839-855)
//   light_trace_kernel  <- tracePhotons + rtMaterialClosestHit (rt/lighttracing.cu:192-250, 113-182)
// Primary rays of an 8x8 tile share the eye and walk the tree as a packet (closest_wave); light sub-paths are
// incoherent: one ray per lane, closest hit, per-lane stack in LDS laid out [entry][lane] so that a push/pop of
// the whole wave is one conflict-free ds access.
#include "device_common.hpp"
#include "kernels.h"

namespace evplp {

#ifndef EVPLP_PRIMARY_BLOCKS
#define EVPLP_PRIMARY_BLOCKS 1
#endif
#ifndef EVPLP_PRIMARY_BLOCK_LOG2
#define EVPLP_PRIMARY_BLOCK_LOG2 2     // 4 x 4 tiles
#endif
#if EVPLP_PRIMARY_TIMES      // developer build (tools/primary_times.py): start / end of every tile's wavefront, s_memrealtime ticks (100 MHz)
__device__ unsigned long long g_primary_times[2 * 65536];
extern "C" int evplp_debug_primary_times(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_primary_times), sizeof(unsigned long long) * (size_t)n); }
#endif
__global__ __launch_bounds__(64) void primary_kernel(PrimaryArgs a) {
    // ray set-up and the hit point are written without fused multiply-adds, in the oracle's operation order: together
    // with the exact closest hit the G-buffer POSITIONS are then bit-identical to the CPU restatement, and so is every
    // threshold test downstream that reads them (the photon radius test |X_p - X|^2 <= r^2, frag:152-154)
#pragma clang fp contract(off)
#if EVPLP_PRIMARY_TIMES
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x;
    const int tiles_x = (a.st.W + 7) >> 3;
#if EVPLP_PRIMARY_BLOCKS
    // Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own.  Tiles in row-major order gave every XCD every
    // eighth tile of every row: all eight L2s held the same nodes (hit rate 54 %) and a packet walk is a chain of dependent node
    // fetches.  The tiles are dealt in square blocks instead (4 x 4): the tiles of a block follow each other on ONE XCD.
    constexpr int L = EVPLP_PRIMARY_BLOCK_LOG2, B = 1 << L;
    const int tiles_y = (a.st.local_rows + 7) >> 3, nbx = (tiles_x + B - 1) >> L;
    const int bj = (int)blockIdx.x >> 3, blk = (bj >> (2 * L)) * 8 + ((int)blockIdx.x & 7), within = bj & (B * B - 1);
    const int tx = (blk % nbx) * B + (within & (B - 1)), ty = (blk / nbx) * B + (within >> L);
    if (tx >= tiles_x || ty >= tiles_y) return;                     // padding of the block grid (wave-uniform)
    const int tile = ty * tiles_x + tx;
#else
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
#endif
    const int x = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const int y = a.st.global_row(min(ly, a.st.local_rows - 1));
    const bool in_image = x < a.st.W && ly < a.st.local_rows && y < a.st.H;   // no early return: the walk is wave-collective
    const size_t p = (size_t)min(ly, a.st.local_rows - 1) * a.st.W + min(x, a.st.W - 1);

    V3 eye = v3(a.cam.eye), S = v3(a.cam.s), U = v3(a.cam.u), F = v3(a.cam.f);
    float cx = ((float)x + 0.5f) / (float)a.st.W * 2.0f - 1.0f;
    float cy = ((float)y + 0.5f) / (float)a.st.H * 2.0f - 1.0f;
    // jittered matrix for the scene, original matrix for the light mesh (rtcomphoton.h:720-727)
    float jx = (cx - a.jitter[0]) * a.cam.aspect * a.cam.tan_half, jy = (cy - a.jitter[1]) * a.cam.tan_half;
    float ox = cx * a.cam.aspect * a.cam.tan_half, oy = cy * a.cam.tan_half;
    // (component-wise: the V3 operators are compiled with contraction allowed and would fuse after inlining)
    V3 dj = v3((S.x * jx + U.x * jy) + F.x, (S.y * jx + U.y * jy) + F.y, (S.z * jx + U.z * jy) + F.z);
    V3 d0 = v3((S.x * ox + U.x * oy) + F.x, (S.y * ox + U.y * oy) + F.y, (S.z * ox + U.z * oy) + F.z);

    float t = 0.f, b = 0.f, g = 0.f, tl = 0.f, bl = 0.f, gl = 0.f;
    // the 64 primary rays of a tile share the eye: packet walk (closest_wave), no per-lane stack.
    // view depth == t because the camera-space z of the direction is -1: near/far = [0.1, 100] (rtcommon.h:586)
    // the tile group's entry cut from the eye (primary_cut_kernel, once per camera: kernels.h PrimaryCutArgs), or the root
    const char *cut = nullptr;
    if (a.cuts) cut = a.cuts + (size_t)((ty >> a.cut_gh_log2) * a.cut_groups_x + (tx >> a.cut_gw_log2)) * (size_t)kCutSlotBytes;
    int32_t tri = closest_wave(a.sc, eye, dj, 0.1f, 100.0f, 1, in_image, t, b, g, cut);
    // the light mesh only matters in front of (or at) the scene hit (depth LEQUAL): bound its walk by that depth --
    // both directions have camera-space z = -1, so t is the view depth on either ray
    const bool light_unoccluded = (a.clear_light & EVPLP_LIGHT_UNOCCLUDED) != 0;      // wave-uniform
    const float light_far = (tri >= 0 && !light_unoccluded) ? fminf(t * 1.000001f + 1.0e-30f, 100.0f) : 100.0f;
    // ... and only tiles with a ray through the (padded) bounds of the light mesh walk at all: the emitters cover a small
    // part of most views and this second walk otherwise costs as much as the first
    bool light_maybe = in_image;
    {
        float t0 = 0.1f, t1 = light_far;
        const float o[3] = { eye.x, eye.y, eye.z }, d[3] = { d0.x, d0.y, d0.z };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (d[k] == 0.0f) { if (o[k] < a.sc.light_lo[k] || o[k] > a.sc.light_hi[k]) light_maybe = false; }
            else {
                const float inv = 1.0f / d[k], ta = (a.sc.light_lo[k] - o[k]) * inv, tb = (a.sc.light_hi[k] - o[k]) * inv;
                t0 = fmaxf(t0, fminf(ta, tb) * 0.99999f - 1.0e-6f); t1 = fminf(t1, fmaxf(ta, tb) * 1.00001f + 1.0e-6f);
            }
        }
        if (t0 > t1) light_maybe = false;
    }
    const bool walk_light = a.sc.light_count > 0 && __ballot(light_maybe) != 0ull;        // wave-uniform
    int32_t ltri = walk_light ? closest_wave(a.sc, eye, d0, 0.1f, light_far, 2, in_image, tl, bl, gl, cut) : -1;
    bool use_light = ltri >= 0 && (tri < 0 || tl <= t);  // depth LEQUAL, light mesh drawn last
    const bool light_visible = light_unoccluded ? ltri >= 0 : use_light;              // the emitter IMAGE (rtcomphoton.h:985-995)
    if (use_light) { tri = ltri; b = bl; g = gl; }

    float4 pos = make_float4(0.f, 0.f, 0.f, 1.f);  // clear colour (0,0,0,1) rtcomphoton.h:885
    float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f), dif = nrm, phg = nrm;
    if (tri >= 0) {
        const TriAttr &ta = a.sc.attrs[tri];
        V3 p0 = v3(ta.v), p1 = v3(ta.v + 3), p2 = v3(ta.v + 6);
        const float w0 = 1.0f - b - g;
        V3 P = v3((p1.x * b + p2.x * g) + p0.x * w0, (p1.y * b + p2.y * g) + p0.y * w0, (p1.z * b + p2.z * g) + p0.z * w0);
        V3 N = normalize_exact(cross_exact(p1 - p0, p2 - p0));  // deferred.geom:16-18 flat winding normal; the oracle's roundings: the
                                                                // sign of n1 . v12 decides which pairs trace a shadow ray (lighttracing.cu:284-288)
        V3 kd, ks; float ns;
        material_at(a.sc, ta, b, g, kd, ks, ns);
        pos = make_float4(P.x, P.y, P.z, 1.0f);
        nrm = make_float4(N.x, N.y, N.z, 0.f);
        dif = make_float4(kd.x, kd.y, kd.z, 0.f);
        phg = make_float4(ks.x, ks.y, ks.z, ns);
    }
    if (a.tile_box) {
        // world-space box of the tile's G-buffer positions, background pixels (the clear colour) included: the photon splat
        // bins a photon only into tiles whose box its sphere reaches (kernels_splat.hip; splat_tile_box_kernel computes the
        // same for G-buffers that did not come from this pass)
        float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        if (in_image) { lo[0] = hi[0] = pos.x; lo[1] = hi[1] = pos.y; lo[2] = hi[2] = pos.z; }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
        if (lane == 0) { a.tile_box[2 * tile] = make_float4(lo[0], lo[1], lo[2], 0.f); a.tile_box[2 * tile + 1] = make_float4(hi[0], hi[1], hi[2], 0.f); }
    }
#if EVPLP_PRIMARY_TIMES
    if (lane == 0 && tile < 65536) { g_primary_times[2 * tile] = t_start; g_primary_times[2 * tile + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (!in_image) return;
    a.g_pos[p] = pos; a.g_nrm[p] = nrm; a.g_dif[p] = dif; a.g_phg[p] = phg;
    if (!(a.clear_light & EVPLP_LIGHT_SKIP)) {
        if (light_visible) a.g_light[p] = make_float4(a.sc.light_unscaled[0], a.sc.light_unscaled[1], a.sc.light_unscaled[2], 0.f);
        else if (a.clear_light & EVPLP_LIGHT_CLEAR) a.g_light[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// lighttracing.cu:93-96
EV_DEV float russian_prob_lt(V3 f) { return fminf(fmaxf(f.x, fmaxf(f.y, f.z)), 0.98f); }

EV_DEV void store_record(evplp_record *r, V3 pos, uint32_t flags, V3 n, float psel, V3 flux, V3 fdir, V3 rd, V3 rs, float e) {
    float4 *q = reinterpret_cast<float4 *>(r);
    q[0] = make_float4(pos.x, pos.y, pos.z, __uint_as_float(flags));
    q[1] = make_float4(n.x, n.y, n.z, psel);
    q[2] = make_float4(flux.x, flux.y, flux.z, 0.f);
    q[3] = make_float4(fdir.x, fdir.y, fdir.z, 0.f);
    q[4] = make_float4(rd.x, rd.y, rd.z, 0.f);
    q[5] = make_float4(rs.x, rs.y, rs.z, e);
}

// (round 4, measured and removed: light tracing as a PERSISTENT, self-refilling wavefront -- lanes in IDLE / TRAV / HIT states, a wave
// refilling its idle lanes from a range of paths of its own, leaving the walk when 8 / 16 / 32 lanes wait to be shaded.  Records
// byte-identical, all GPU tests green, lane utilisation up as intended -- and 300 000 paths took 0.77 / 0.55 / 0.48 / 0.47 ms at 1 / 2 / 3 /
// 4 resident waves per SIMD (124 registers: no fifth) against 0.44 ms for one path per lane at 5 waves below: the time falls with the
// number of rays in flight, not with the number of busy lanes.  The kernel is bound by the dependent 128-byte node gathers (8 KB per
// wave and step out of L2 / Infinity Cache), not by vector-instruction issue; profiles/r04_light_trace_persistent_sweep.txt.)
// four-wide nodes; 5 waves per SIMD (96 registers, no spills) with the first 20 stack entries in LDS and the rest of the worst case in
// global memory (closest_lane4): the worst-case LDS stack alone allowed 3 waves per SIMD.  Config #4's light tracing (300 000 paths):
// round 2 binary nodes 627 us, four-wide 553; round 3 (peeled tree) 511 us at 3 waves per SIMD
#ifndef EVPLP_LT_WAVES
// 300 000 paths, config #4 (kernel alone / the overlapped iteration, ms): 5 waves per SIMD (96 registers) 0.424 / 0.608, 4 waves (128
// registers, no spill) 0.46 / 0.595 -- alone the kernel loses its full residency (4 688 wavefronts, 4 096 slots), beside the G-buffer pass
// and the splat it leaves them the registers they need; the iteration is what a frame pays
#define EVPLP_LT_WAVES 4
#endif
#ifndef EVPLP_LT_PAIRS
#define EVPLP_LT_PAIRS 1         // leaf triangles two at a time (closest_lane4 PAIRS)
#endif
#ifndef EVPLP_LT_SPEC
#define EVPLP_LT_SPEC 1          // speculative while-while: leaves a lane may postpone (closest_lane4 SPEC); 300 000 paths: 0.433 / 0.423 / 0.440 ms for 0 / 1 / 2
#endif
#if EVPLP_LT_TIMES           // developer build (tools/lt_times.py): start / end clock (100 MHz) of every wavefront
__device__ unsigned long long g_lt_times[2 * 16384];
extern "C" int evplp_debug_lt_times(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lt_times), sizeof(unsigned long long) * (size_t)n); }
#endif
__global__ __launch_bounds__(64, EVPLP_LT_WAVES) void light_trace_kernel(LightTraceArgs a) {
    extern __shared__ int32_t lds_stack[];   // [bvh_depth + 2][64 lanes]
#if EVPLP_LT_TIMES
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x;
    const uint32_t local = blockIdx.x * 64u + lane;
    if (local >= a.path_count) return;
    const uint32_t id = a.path_begin + local;
    const uint32_t P = a.photons_per_path;
    evplp_record *rec = a.records + (size_t)id * P;
    int32_t *stack = lds_stack + lane;

    const V3 zero = v3(0.f, 0.f, 0.f);
    uint32_t filled = 1;     // slots written so far; the rest are cleared at the end (:197-200 clears them all first: same final state)

    Rng rng; rng_init(rng, id, a.rng_seed, 0u);
    V3 position, normal; float pdf;
    V3 flux = light_sample(a.sc, position, normal, pdf, rng);
    V3 direction; float phong_pdf;
    V3 att = phong_sample(direction, phong_pdf, normal, normal, v3(1.f, 1.f, 1.f), a.sc.light_intensity[3], rng);
    // record 0: the on-light vertex is a VPL (:215-225)
    store_record(&rec[0], position, EVPLP_USABLE_VPL, normal, 0.0f, flux, normal, zero, v3(1.f, 1.f, 1.f), a.sc.light_intensity[3]);

    V3 pflux = flux * att;
    V3 next_pos = position, next_dir = direction;
    for (uint32_t i = 1; i < P; i++) {
        uint32_t flag = (i != P - 1) ? (EVPLP_USABLE_VPL | EVPLP_USABLE_PHOTON) : EVPLP_USABLE_PHOTON;
        float t, b, g;
        int32_t tri = closest_lane4<64, kLtLdsStack, EVPLP_LT_SPEC, EVPLP_LT_PAIRS != 0>(a.sc, next_pos, next_dir, 0.0001f, 3.0e38f, 0, t, b, g, stack, a.stack_overflow + local, a.overflow_stride);
        if (tri < 0) break;  // no miss program in the reference; a miss ends the path here
        const TriAttr &ta = a.sc.attrs[tri];
        V3 p0 = v3(ta.v), p1 = v3(ta.v + 3), p2 = v3(ta.v + 6);
        // geometry of the hit with the oracle's roundings (record positions and normals feed the cosine and visibility tests)
        V3 gn = normalize_exact(cross_exact(p0 - p2, p1 - p0));   // triangleintersect.cu:31
        V3 wgn = normalize_exact(gn);                     // :115
        V3 ffn = wgn * copysignf(1.0f, dot_exact(-next_dir, wgn));   // :116 faceforward(wgn, -next_dir, wgn)
        V3 hit_pos = madd_exact(next_pos, next_dir, t);   // :120
        const Material &m = a.sc.materials[ta.material];
        if (dot_exact(gn, next_dir) > 0.f || m.light[0] > 0.01f) break;  // :124-128
        V3 kd, ks; float ns;
        material_at(a.sc, ta, b, g, kd, ks, ns);
        float max_l = max_color(kd), max_p = max_color(ks);
        if (max_l + max_p <= 0.000001f) break;            // :143-147
        float psel = max_l / (max_p + max_l);
        float choose = fminf(rng_uniform(rng), 0.999999f);
        float russian = russian_prob_lt(pflux);           // :164
        V3 stored_flux = pflux;
        pflux = pflux / russian;
        bool done = rng_uniform(rng) >= russian;          // :166
        V3 dir = zero; float pdfw;
        if (!done) {
            if (choose < psel) {
                V3 w = lambert_sample(dir, pdfw, ffn, kd, rng);
                pflux = pflux * (w / psel);
                flag |= EVPLP_LAMBERT_ONLY;
            } else {
                V3 w = phong_sample(dir, pdfw, -next_dir, gn, ks, ns, rng);  // un-flipped normal (:176)
                pflux = pflux * (w / (1.0f - psel));
                flag |= EVPLP_PHONG_ONLY;
            }
        }
        store_record(&rec[i], hit_pos, flag, ffn, psel, stored_flux, -next_dir, kd, ks, ns);
        filled = i + 1;
        if (done) break;
        next_pos = hit_pos; next_dir = dir;
    }
    for (uint32_t i = filled; i < P; i++) store_record(&rec[i], zero, 0u, zero, 0.f, zero, zero, zero, zero, 0.f);
#if EVPLP_LT_TIMES
    if (lane == 0 && blockIdx.x < 16384u) { g_lt_times[2 * blockIdx.x] = t_start; g_lt_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// Stable compaction of the usable VPL records (flags & IsUsableVpl, rt/lighttracing.cu:372) of
// the first numVplLightPaths paths into a dense list, preserving record order so per-pixel sums
// run in the reference's loop order.  One workgroup; the list is at most a few 10k records.
__global__ __launch_bounds__(1024) void compact_vpl_kernel(const evplp_record *records, uint32_t nrec,
                                                           evplp_record *out, uint32_t *src_index, uint32_t *count_out) {
    __shared__ uint32_t wave_counts[16];
    __shared__ uint32_t base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (uint32_t start = 0; start < nrec; start += 1024) {
        uint32_t i = start + tid;
        bool usable = i < nrec && (records[i].flags & EVPLP_USABLE_VPL) != 0;
        unsigned long long m = __ballot(usable);
        uint32_t prefix = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_counts[wave] = __popcll(m);
        __syncthreads();
        uint32_t off = base;
        for (int w = 0; w < wave; w++) off += wave_counts[w];
        if (usable) {
            const float4 *src = reinterpret_cast<const float4 *>(&records[i]);
            float4 *dst = reinterpret_cast<float4 *>(&out[off + prefix]);
#pragma unroll
            for (int k = 0; k < 6; k++) dst[k] = src[k];
            if (src_index) src_index[off + prefix] = i;
        }
        __syncthreads();
        if (tid == 0) { uint32_t tot = 0; for (int w = 0; w < 16; w++) tot += wave_counts[w]; base += tot; }
        __syncthreads();
    }
    if (tid == 0) *count_out = base;
}

void launch_primary(const PrimaryArgs &a, hipStream_t s) {
    int tiles_x = (a.st.W + 7) / 8, tiles_y = (a.st.local_rows + 7) / 8;
#if EVPLP_PRIMARY_BLOCKS
    constexpr int B = 1 << EVPLP_PRIMARY_BLOCK_LOG2;
    const int blocks = ((tiles_x + B - 1) / B) * ((tiles_y + B - 1) / B);
    if (blocks == 0) return;
    hipLaunchKernelGGL(primary_kernel, dim3((unsigned)((blocks + 7) / 8 * 8 * B * B)), dim3(64), 0, s, a);
#else
    hipLaunchKernelGGL(primary_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, s, a);
#endif
}
void launch_light_trace(const LightTraceArgs &a, hipStream_t s) {
    if (a.path_count == 0) return;
    hipLaunchKernelGGL(light_trace_kernel, dim3((a.path_count + 63) / 64), dim3(64), (size_t)kLtLdsStack * 64 * sizeof(int32_t), s, a);
}
void launch_compact_vpl(const evplp_record *records, uint32_t nrec, evplp_record *out, uint32_t *src_index,
                        uint32_t *count_out, hipStream_t s) {
    hipLaunchKernelGGL(compact_vpl_kernel, dim3(1), dim3(1024), 0, s, records, nrec, out, src_index, count_out);
}

} // namespace evplp
