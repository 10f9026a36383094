// Device-side building blocks shared by all kernels: exact geometric predicates, BRDF model,
// RNG, texture fetch and the two BVH traversal flavours.
//
// Floating-point contract.  Device code is compiled with hipcc's default -ffp-contract=fast.
// Functions in the EXACT section carry `#pragma clang fp contract(off)` and spell out every
// + - * / as a single IEEE operation in the same order as the oracle, so visibility (any-hit)
// and closest-hit results are bit-identical to the CPU restatement.  Shading arithmetic may be
// contracted; it is compared under a stated tolerance.
#pragma once
#include "evplp_types.h"

namespace evplp {

#define EV_PI 3.14159265358979323846f
#define EV_INV_PI 0.318309886183790671537767526745028724068919291480912897495f /* rt/rtmath.cuh:11 */
#define EV_DEV __device__ __forceinline__

struct V3 { float x, y, z; };
EV_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
EV_DEV V3 v3(const float *p) { return v3(p[0], p[1], p[2]); }
EV_DEV V3 v3(float4 a) { return v3(a.x, a.y, a.z); }
EV_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
EV_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
EV_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
EV_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
EV_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
EV_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
EV_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
EV_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// optixu normalize: v * (1 / sqrtf(dot(v, v)))
EV_DEV V3 normalize(V3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
// optixu reflect(i, n) = i - 2 n dot(n, i)
EV_DEV V3 reflect(V3 i, V3 n) { float d = dot(n, i); return i - (n * 2.0f) * d; }
EV_DEV V3 faceforward(V3 n, V3 i, V3 nref) { return n * copysignf(1.0f, dot(i, nref)); }
EV_DEV float max_color(V3 c) { return fmaxf(fmaxf(c.x, c.y), c.z); }

// ---------------------------------------------------------------------------------- EXACT
// optix::intersect_triangle_branchless (OptiX SDK 4.1.1 optixu_math_namespace.h) as called by
// meshFineIntersect, rt/triangleintersect.cu:17-41.  Operands pre-computed by build_bvh.
EV_DEV bool tri_test(const TriPre &tp, V3 o, V3 d, float tmin, float tmax, float &t, float &beta, float &gamma) {
#pragma clang fp contract(off)
    // written out in scalars: the contract(off) pragma is lexical and must cover every operation
    float nx = tp.n[0], ny = tp.n[1], nz = tp.n[2];
    float den = nx * d.x + ny * d.y + nz * d.z;
    float inv = 1.0f / den;
    float qx = (tp.p0[0] - o.x) * inv, qy = (tp.p0[1] - o.y) * inv, qz = (tp.p0[2] - o.z) * inv;
    float ix = d.y * qz - d.z * qy, iy = d.z * qx - d.x * qz, iz = d.x * qy - d.y * qx;
    beta = ix * tp.e1[0] + iy * tp.e1[1] + iz * tp.e1[2];
    gamma = ix * tp.e0[0] + iy * tp.e0[1] + iz * tp.e0[2];
    t = nx * qx + ny * qy + nz * qz;
    return (t < tmax) & (t > tmin) & (beta >= 0.0f) & (gamma >= 0.0f) & (beta + gamma <= 1.0f);
}

// ------------------------------------------------------------------------------------ RNG
// Build-defined generator shared bit-for-bit with the oracle: PCG32 XSH-RR seeded through
// splitmix64; one stream per (index, sequence, substream).  Stands in for
// curand_init(seed=index, sequence=rngSeed, 0) (rt/lighttracing.cu:203,711).
struct Rng { uint64_t state, inc; };
EV_DEV uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
EV_DEV uint32_t rng_u32(Rng &r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ull + r.inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}
EV_DEV void rng_init(Rng &r, uint32_t index, uint32_t sequence, uint32_t substream) {
    uint64_t key = ((uint64_t)sequence << 32) | (uint64_t)index;
    uint64_t s0 = splitmix64(key + (uint64_t)substream * 0xD1B54A32D192ED03ull);
    r.inc = splitmix64(s0) | 1ull;
    r.state = s0 + r.inc;
    (void)rng_u32(r);
}
// (0,1] like curand_uniform; exact in fp32
EV_DEV float rng_uniform(Rng &r) { return (float)((rng_u32(r) >> 8) + 1u) * (1.0f / 16777216.0f); }

// -------------------------------------------------------------------------------- textures
// tex2D, RT_FILTER_LINEAR / RT_WRAP_REPEAT / normalised coordinates (rt/rtcommon.h:223-245)
EV_DEV int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
EV_DEV float4 tex2d(const SceneDev &sc, int id, float u, float v) {
    TexDesc t = sc.textures[id];
    const float4 *px = sc.tex_pool + t.offset;
    if (t.w == 1 && t.h == 1) return px[0];
    float xb = u * (float)t.w - 0.5f, yb = v * (float)t.h - 0.5f;
    float xf = floorf(xb), yf = floorf(yb);
    float a = xb - xf, b = yb - yf;
    int x0 = wrapi((int)xf, t.w), x1 = wrapi((int)xf + 1, t.w);
    int y0 = wrapi((int)yf, t.h), y1 = wrapi((int)yf + 1, t.h);
    float4 p00 = px[(size_t)y0 * t.w + x0], p10 = px[(size_t)y0 * t.w + x1];
    float4 p01 = px[(size_t)y1 * t.w + x0], p11 = px[(size_t)y1 * t.w + x1];
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 r;
    r.x = w00 * p00.x + w10 * p10.x + w01 * p01.x + w11 * p11.x;
    r.y = w00 * p00.y + w10 * p10.y + w01 * p01.y + w11 * p11.y;
    r.z = w00 * p00.z + w10 * p10.z + w01 * p01.z + w11 * p11.z;
    r.w = w00 * p00.w + w10 * p10.w + w01 * p01.w + w11 * p11.w;
    return r;
}
// material fetch at a hit (rt/lighttracing.cu:131-133, shaders/deferred.frag:18-21)
EV_DEV void material_at(const SceneDev &sc, const TriAttr &ta, float beta, float gamma, V3 &kd, V3 &ks, float &ns) {
    const Material &m = sc.materials[ta.material];
    kd = v3(m.kd); ks = v3(m.ks); ns = m.ns;
    if (m.tex_kd >= 0 || m.tex_ks >= 0 || m.tex_ns >= 0) {
        float w0 = 1.0f - beta - gamma;  // rt/triangleintersect.cu:36
        float u = ta.uv[2] * beta + ta.uv[4] * gamma + ta.uv[0] * w0;
        float v = ta.uv[3] * beta + ta.uv[5] * gamma + ta.uv[1] * w0;
        if (m.tex_kd >= 0) { float4 c = tex2d(sc, m.tex_kd, u, v); kd = v3(c.x, c.y, c.z); }
        if (m.tex_ks >= 0) { float4 c = tex2d(sc, m.tex_ks, u, v); ks = v3(c.x, c.y, c.z); }
        if (m.tex_ns >= 0) { float4 c = tex2d(sc, m.tex_ns, u, v); ns = c.x; }
    }
}

// ------------------------------------------------------------------- rt/rtmaterial.cuh model
// :112-118 PhongEvalF
EV_DEV float phong_eval_f(V3 out, V3 in, V3 n, float e) {
    V3 r = reflect(-in, n);
    float d = fmaxf(dot(out, r), 0.0f);
    if (d <= 0.000001f) return 0.0f;
    return (e + 2.0f) * powf(d, e) * EV_INV_PI * 0.5f;
}
// :104-110 PhongEval
EV_DEV V3 phong_eval(V3 out, V3 in, V3 n, V3 rho_s, float e) {
    V3 r = reflect(-in, n);
    float d = fmaxf(dot(out, r), 0.0f);
    if (d <= 0.000001f || rho_s.x <= 0.000001f) return v3(0.f, 0.f, 0.f);
    return rho_s * (e + 2.0f) * powf(d, e) * EV_INV_PI * 0.5f;
}
// :46-54 LambertPdfA
EV_DEV float lambert_pdf_a(V3 n1, V3 n2, V3 v12) {
    float c1 = fmaxf(dot(n1, v12), 0.f), c2 = fmaxf(-dot(n2, v12), 0.f), d2 = dot(v12, v12);
    return c1 * c2 / (d2 * d2) * EV_INV_PI;
}
// :40-44 LambertPdfW (no 1/pi: reference quirk)
EV_DEV float lambert_pdf_w(V3 n1, V3 v12) {
    return fmaxf(dot(n1, normalize(v12)), 0.f);
}
// :78-85 PhongPdfW
EV_DEV float phong_pdf_w(V3 n1, V3 v12, V3 in, V3 rho_s, float e) {
    V3 wi12 = normalize(v12);
    V3 r = normalize(reflect(-in, n1));
    float c = fmaxf(dot(wi12, r), 0.f);
    if (c <= 0.000001f || rho_s.x <= 0.000001f) return 0.0f;
    return (e + 1.0f) * 0.5f * EV_INV_PI * powf(c, e);
}
// :87-102 PhongPdfA
EV_DEV float phong_pdf_a(V3 n1, V3 n2, V3 v12, V3 in, V3 rho_s, float e) {
    V3 wi12 = normalize(v12);
    V3 r = normalize(reflect(-in, n1));
    float c = fmaxf(dot(wi12, r), 0.f);
    if (c <= 0.000001f || rho_s.x <= 0.000001f) return 0.0f;
    float pdfw = (e + 1.0f) * 0.5f * EV_INV_PI * powf(c, e);
    float cos2 = fmaxf(-dot(n2, wi12), 0.0f);
    return pdfw * cos2 / dot(v12, v12);
}
// :30-38 GeometryTerm
EV_DEV float geometry_term(V3 n1, V3 n2, V3 v12) {
    float c1 = fmaxf(dot(n1, v12), 0.f), c2 = fmaxf(-dot(n2, v12), 0.f), d2 = dot(v12, v12);
    return c1 * c2 / (d2 * d2);
}
// optixu Onb
struct Onb { V3 t, b, n; };
EV_DEV Onb onb_make(V3 n) {
    Onb o; o.n = n;
    if (fabsf(n.x) > fabsf(n.z)) o.b = v3(-n.y, n.x, 0.0f);
    else o.b = v3(0.0f, -n.z, n.y);
    o.b = normalize(o.b);
    o.t = cross(o.b, o.n);
    return o;
}
EV_DEV V3 onb_inverse(const Onb &o, V3 p) { return o.t * p.x + o.b * p.y + o.n * p.z; }
// :56-66 LambertSample (draw order: first draw -> u1)
EV_DEV V3 lambert_sample(V3 &out, float &pdfw, V3 normal, V3 rho_d, Rng &rng) {
    float u1 = rng_uniform(rng);
    float u2 = rng_uniform(rng);
    float r = sqrtf(u1);
    float phi = 2.0f * EV_PI * u2;
    V3 p; p.x = r * cosf(phi); p.y = r * sinf(phi);
    p.z = sqrtf(fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    Onb o = onb_make(normal);
    out = onb_inverse(o, p);
    pdfw = fmaxf(dot(out, normal), 0.f) * EV_INV_PI;
    return rho_d;
}
// :120-154 PhongSample
EV_DEV V3 phong_sample(V3 &out, float &pdfw, V3 in, V3 normal, V3 rho_s, float e, Rng &rng) {
    V3 r = reflect(-in, normal);
    float sx = rng_uniform(rng);
    float sy = rng_uniform(rng);
    float cos_t = powf(sx, 1.f / (e + 1.f));
    float sin_t = sqrtf(1.0f - cos_t * cos_t);
    float phi = 2.f * EV_PI * sy;
    float cp = cosf(phi), sp = sinf(phi);
    V3 p = v3(sin_t * cp, sin_t * sp, cos_t);
    Onb o = onb_make(r);
    out = onb_inverse(o, p);
    float unsafe_cos = dot(out, normal);
    float cos_n = fmaxf(unsafe_cos, 0.f);
    float cos_r = fmaxf(dot(out, r), 0.f);
    if (unsafe_cos > 0.0f) pdfw = (e + 1.0f) * 0.5f * powf(cos_r, e) * EV_INV_PI;
    else pdfw = 0.0f;
    return rho_s * ((e + 2.0f) / (e + 1.0f) * cos_n);
}

// rt/rtlightsource.cuh:24-80 LightSample (+ rt/rtmath.cuh:22-27)
EV_DEV V3 light_sample(const SceneDev &sc, V3 &position, V3 &normal, float &pdf, Rng &rng) {
    float r = rng_uniform(rng);
    uint32_t count = (uint32_t)sc.light_count, first = 0;
    while (count > 0) {
        uint32_t it = first, step = count / 2; it += step;
        if (sc.light_cdf[it] < r) { first = ++it; count -= step + 1; } else count = step;
    }
    if (first >= (uint32_t)sc.light_count) first = (uint32_t)sc.light_count - 1;
    const TriAttr &ta = sc.attrs[sc.light_first + (int32_t)first];
    float x = rng_uniform(rng);
    float y = rng_uniform(rng);
    float sq = sqrtf(x), beta = sq * (1.0f - y), gamma = sq * y;
    V3 p1 = v3(ta.v), p2 = v3(ta.v + 3), p3 = v3(ta.v + 6);
    position = p1 * beta + p2 * gamma + p3 * (1.0f - gamma - beta);
    normal = normalize(cross(p2 - p1, p3 - p1));
    pdf = 1.f / sc.light_area;
    return v3(sc.light_intensity) * sc.light_area;
}

// ------------------------------------------------------------------------------ traversal
// Conservative slab test of the parametric segment o + t d, t in [tmin, tmax], against a padded
// box.  inv = 1/d (approximate is fine), noi = -(o * inv).  6 FMA + min/max.
EV_DEV bool slab_hit(const float *lo, const float *hi, V3 inv, V3 noi, float tmin, float tmax) {
    float t0x = __builtin_fmaf(lo[0], inv.x, noi.x), t1x = __builtin_fmaf(hi[0], inv.x, noi.x);
    float t0y = __builtin_fmaf(lo[1], inv.y, noi.y), t1y = __builtin_fmaf(hi[1], inv.y, noi.y);
    float t0z = __builtin_fmaf(lo[2], inv.z, noi.z), t1z = __builtin_fmaf(hi[2], inv.z, noi.z);
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tn <= tf;
}
EV_DEV float slab_near(const float *lo, const float *hi, V3 inv, V3 noi, float tmin, float tmax, bool &hit) {
    float t0x = __builtin_fmaf(lo[0], inv.x, noi.x), t1x = __builtin_fmaf(hi[0], inv.x, noi.x);
    float t0y = __builtin_fmaf(lo[1], inv.y, noi.y), t1y = __builtin_fmaf(hi[1], inv.y, noi.y);
    float t0z = __builtin_fmaf(lo[2], inv.z, noi.z), t1z = __builtin_fmaf(hi[2], inv.z, noi.z);
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    hit = tn <= tf;
    return tn;
}
EV_DEV float safe_rcp(float d) {
    float a = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}

// One wide scalar fetch of a wave-uniform 64-byte block (s_load_dwordx16): the whole BVH node (or a
// third of a 4-triangle leaf block) arrives with ONE exposed latency instead of one per field.
typedef int v16i __attribute__((ext_vector_type(16)));
EV_DEV float f_of(int x) { return __int_as_float(x); }

// exact triangle test on raw dwords d[0..11] = p0, e0, e1, n (same arithmetic as tri_test)
EV_DEV bool tri_test_raw(float p0x, float p0y, float p0z, float e0x, float e0y, float e0z, float e1x, float e1y, float e1z,
                         float nx, float ny, float nz, V3 o, V3 d, float tmin, float tmax) {
#pragma clang fp contract(off)
    float den = nx * d.x + ny * d.y + nz * d.z;
    float inv = 1.0f / den;
    float qx = (p0x - o.x) * inv, qy = (p0y - o.y) * inv, qz = (p0z - o.z) * inv;
    float ix = d.y * qz - d.z * qy, iy = d.z * qx - d.x * qz, iz = d.x * qy - d.y * qx;
    float beta = ix * e1x + iy * e1y + iz * e1z;
    float gamma = ix * e0x + iy * e0y + iz * e0z;
    float t = nx * qx + ny * qy + nz * qz;
    return (t < tmax) & (t > tmin) & (beta >= 0.0f) & (gamma >= 0.0f) & (beta + gamma <= 1.0f);
}

// Stack-in-a-VGPR helpers: entry k of the wave's stack is lane k of one register.  A push is a
// compare + select against the lane id (this clang has no v_writelane builtin), a pop is v_readlane
// with a scalar lane index; neither touches memory.
EV_DEV int lane_write(int value, int slot, int old) { return ((int)(threadIdx.x & 63u) == slot) ? value : old; }
EV_DEV int lane_read(int v, int slot) { return __builtin_amdgcn_readlane(v, slot); }

// Any-hit traversal of ONE WAVE whose 64 rays share the origin `o` (a VPL): the node index, the
// stack and all node/triangle fetches are wave-uniform (scalar loads); lanes only differ in
// direction.  `alive` lanes still need an answer; a lane that finds an occluder drops out of the
// ballots, and the walk ends when no lane is alive or the stack is empty.
// The per-wavefront stack lives in the 64 lanes of ONE VGPR (select-by-lane-id push, v_readlane pop,
// scalar stack pointer: no memory latency on either); entries beyond 64 spill to the wave's LDS stack.  Replaces rtTrace(..., ray type 1) + rtMaterialAnyHit,
// rt/lighttracing.cu:184-188,290-294.  Returns true for lanes whose segment is occluded.
EV_DEV bool occluded_wave(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, bool alive,
                          int32_t *wave_stack, uint32_t &nodes_visited) {
    V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    V3 noi = v3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    bool hit = false;
    int sp = 0;
    int vstack = 0;
    int32_t cur = 0;  // root is always an inner node
    if (__ballot(alive) == 0ull) return false;
    for (;;) {
        if (cur >= 0) {
            const v16i *np = reinterpret_cast<const v16i *>(sc.nodes + __builtin_amdgcn_readfirstlane(cur));
            const v16i n = *np;
            nodes_visited++;
            const float lo0[3] = { f_of(n[0]), f_of(n[1]), f_of(n[2]) }, hi0[3] = { f_of(n[3]), f_of(n[4]), f_of(n[5]) };
            const float lo1[3] = { f_of(n[6]), f_of(n[7]), f_of(n[8]) }, hi1[3] = { f_of(n[9]), f_of(n[10]), f_of(n[11]) };
            const bool h0 = slab_hit(lo0, hi0, inv, noi, tmin, tmax) & alive;
            const bool h1 = slab_hit(lo1, hi1, inv, noi, tmin, tmax) & alive;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            const int32_t c0 = n[12], c1 = n[13];
            if (m0 && m1) {
                // descend into the child wanted by more lanes first
                const bool first0 = __popcll(m0) >= __popcll(m1);
                const int32_t oth = first0 ? c1 : c0;
                if (sp < 64) vstack = lane_write(oth, sp, vstack); else wave_stack[sp - 64] = oth;
                sp++;
                cur = first0 ? c0 : c1;
                continue;
            } else if (m0) { cur = c0; continue; }
            else if (m1) { cur = c1; continue; }
        } else if (cur != kNoChild) {
            const int32_t id = __builtin_amdgcn_readfirstlane(~cur);
            const int32_t first = id >> 2, cnt = (id & 3) + 1;
            // a leaf is a block of up to 4 triangles (192 B); fetch all of it before testing
            const v16i *tp = reinterpret_cast<const v16i *>(sc.tris + first);
            const v16i a = tp[0], b = tp[1], c = tp[2];
            bool h = tri_test_raw(f_of(a[0]), f_of(a[1]), f_of(a[2]), f_of(a[3]), f_of(a[4]), f_of(a[5]), f_of(a[6]), f_of(a[7]), f_of(a[8]),
                                  f_of(a[9]), f_of(a[10]), f_of(a[11]), o, d, tmin, tmax);
            if (cnt > 1) h |= tri_test_raw(f_of(a[12]), f_of(a[13]), f_of(a[14]), f_of(a[15]), f_of(b[0]), f_of(b[1]), f_of(b[2]), f_of(b[3]), f_of(b[4]),
                                           f_of(b[5]), f_of(b[6]), f_of(b[7]), o, d, tmin, tmax);
            if (cnt > 2) h |= tri_test_raw(f_of(b[8]), f_of(b[9]), f_of(b[10]), f_of(b[11]), f_of(b[12]), f_of(b[13]), f_of(b[14]), f_of(b[15]), f_of(c[0]),
                                           f_of(c[1]), f_of(c[2]), f_of(c[3]), o, d, tmin, tmax);
            if (cnt > 3) h |= tri_test_raw(f_of(c[4]), f_of(c[5]), f_of(c[6]), f_of(c[7]), f_of(c[8]), f_of(c[9]), f_of(c[10]), f_of(c[11]), f_of(c[12]),
                                           f_of(c[13]), f_of(c[14]), f_of(c[15]), o, d, tmin, tmax);
            hit = hit | (alive & h);
            alive = alive & !hit;
            if (__ballot(alive) == 0ull) return hit;
        }
        if (sp == 0) return hit;
        sp--;
        cur = sp < 64 ? lane_read(vstack, sp) : __builtin_amdgcn_readfirstlane(wave_stack[sp - 64]);
    }
}

// Per-lane closest-hit traversal with a private stack (incoherent rays: primary visibility and
// light sub-paths).  Replaces rtTrace(..., ray type 0) + meshFineIntersect.  The closest hit is
// order-independent: ties in t keep the lowest ORIGINAL triangle index (same rule as the oracle).
// filter: 0 all, 1 skip light mesh, 2 light mesh only.  Returns original triangle index or -1.
template <int STACK_STRIDE>
EV_DEV int32_t closest_lane(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, int filter,
                            float &t_out, float &beta_out, float &gamma_out, int32_t *stack /* stack[k*STACK_STRIDE] */) {
    V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    V3 noi = v3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    int32_t best = -1; float bt = tmax, bb = 0.f, bg = 0.f;
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const BvhNode &n = sc.nodes[cur];
            bool h0, h1;
            float n0 = slab_near(n.lo0, n.hi0, inv, noi, tmin, bt, h0);
            float n1 = slab_near(n.lo1, n.hi1, inv, noi, tmin, bt, h1);
            if (h0 && h1) {
                bool first0 = n0 <= n1;
                stack[sp * STACK_STRIDE] = first0 ? n.c1 : n.c0; sp++;
                cur = first0 ? n.c0 : n.c1;
                continue;
            } else if (h0) { cur = n.c0; continue; }
            else if (h1) { cur = n.c1; continue; }
        } else if (cur != kNoChild) {
            int32_t id = ~cur;
            int32_t first = id >> 2, cnt = (id & 3) + 1;
            for (int32_t k = 0; k < cnt; k++) {
                int32_t orig = sc.tri_index[first + k];
                bool is_light = orig >= sc.light_first && orig < sc.light_first + sc.light_count;
                if ((filter == 1 && is_light) || (filter == 2 && !is_light)) continue;
                float t, b, g;
                if (tri_test(sc.tris[first + k], o, d, tmin, 3.0e38f, t, b, g)) {
                    if (t < bt || (t == bt && best >= 0 && orig < best)) { bt = t; bb = b; bg = g; best = orig; }
                }
            }
        }
        if (sp == 0) break;
        --sp; cur = stack[sp * STACK_STRIDE];
    }
    if (best >= 0) { t_out = bt; beta_out = bb; gamma_out = bg; }
    return best;
}

} // namespace evplp
