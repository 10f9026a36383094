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
#include "ev_math.h"

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
// (a.x b.x + a.y b.y) + a.z b.z with every product and sum rounded on its own (the oracle's dot under -ffp-contract=off): used
// where a SIGN decides something discrete -- the cosine test of lighttracing.cu:284-288 -- so that the set of pairs that trace a
// shadow ray is identical on CPU and GPU, not just the radiance within a tolerance
EV_DEV V3 cross_exact(V3 a, V3 b) {
#pragma clang fp contract(off)
    const float x0 = a.y * b.z, x1 = a.z * b.y, y0 = a.z * b.x, y1 = a.x * b.z, z0 = a.x * b.y, z1 = a.y * b.x;
    return v3(x0 - x1, y0 - y1, z0 - z1);
}
EV_DEV float dot_exact(V3 a, V3 b);
// optixu normalize with the oracle's roundings: v * (1 / sqrt((x x + y y) + z z)), IEEE sqrt and division
EV_DEV V3 normalize_exact(V3 v) {
#pragma clang fp contract(off)
    const float d = dot_exact(v, v);
    const float inv = 1.0f / __builtin_sqrtf(d);
    return v3(v.x * inv, v.y * inv, v.z * inv);
}
// a + b * t, product and sum rounded separately
EV_DEV V3 madd_exact(V3 a, V3 b, float t) {
#pragma clang fp contract(off)
    const float x = b.x * t, y = b.y * t, z = b.z * t;
    return v3(a.x + x, a.y + y, a.z + z);
}
EV_DEV float dot_exact(V3 a, V3 b) {
#pragma clang fp contract(off)
    const float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z;
    const float xy = x + y;
    return xy + z;
}

// ---------------------------------------------------------------------------------- EXACT
// optix::intersect_triangle_branchless (OptiX SDK 4.1.1 optixu_math_namespace.h) as called by
// meshFineIntersect, rt/triangleintersect.cu:17-41.  Operands pre-computed by build_bvh.
// The reference is compiled by nvcc with -fmad=true, i.e. with compiler-chosen fused multiply-adds;
// this build FIXES the placement (below, identical in oracle/evplp_oracle.c tri_test) so that the
// predicate is bit-identical on CPU and GPU: every dot product is  fma(z, z', fma(y, y', x*x'))  and
// every cross component is  fma(a, b, -(c*d)).  1/den is an IEEE-correct division.
typedef float v2f __attribute__((ext_vector_type(2)));
// 1 / x with the bits of the IEEE division for every normal x below 2^126 -- v_rcp_f32 (a 1-ulp instruction), one Newton-Raphson
// step and ONE residual correction, without the range scaling of v_div_scale / v_div_fixup (5 instructions instead of 11; the
// division's own expansion has a second correction, which rounds 1-4 carried: with libm's exact fmaf on the CPU, over every mantissa and
// seeds at and one ulp either side of the correctly rounded reciprocal, the second correction never changes a bit -- 75 M cases).
// evplp_selftest(0) compares it with 1.0f / x on all 2^32 bit patterns ON THE DEVICE (tests/test_gpu_parity.py runs it): the
// two differ only for zero / denormal / infinite x and for |x| >= 2^126.  In the triangle predicates x = n . d: a zero or
// denormal x makes 1 / x infinite or larger than 2^126 under IEEE and NaN or infinite here -- either way t is infinite, NaN or
// beyond every segment's range and the predicate is false; |x| >= 2^126 needs coordinates beyond 1e18.  So the predicates
// return the oracle's booleans for every input, and its t / beta / gamma wherever they report a hit.
EV_DEV float rcp_exact(float x) {
#pragma clang fp contract(off)
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float err = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(err, r, r);
}
EV_DEV bool tri_test(const TriPair &tp, int h, V3 o, V3 d, float tmin, float tmax, float &t, float &beta, float &gamma) {
#pragma clang fp contract(off)
    const float nx = tp.n[0][h], ny = tp.n[1][h], nz = tp.n[2][h];
    float den = __builtin_fmaf(nz, d.z, __builtin_fmaf(ny, d.y, nx * d.x));
    float inv = rcp_exact(den);
    float qx = (tp.p0[0][h] - o.x) * inv, qy = (tp.p0[1][h] - o.y) * inv, qz = (tp.p0[2][h] - o.z) * inv;
    float ix = __builtin_fmaf(d.y, qz, -(d.z * qy)), iy = __builtin_fmaf(d.z, qx, -(d.x * qz)), iz = __builtin_fmaf(d.x, qy, -(d.y * qx));
    beta = __builtin_fmaf(iz, tp.e1[2][h], __builtin_fmaf(iy, tp.e1[1][h], ix * tp.e1[0][h]));
    gamma = __builtin_fmaf(iz, tp.e0[2][h], __builtin_fmaf(iy, tp.e0[1][h], ix * tp.e0[0][h]));
    t = __builtin_fmaf(nz, qz, __builtin_fmaf(ny, qy, nx * qx));
    return (t < tmax) & (t > tmin) & (beta >= 0.0f) & (gamma >= 0.0f) & (beta + gamma <= 1.0f);
}

// tri_test on the one-triangle-per-48-bytes copy (three 16-byte loads); same operations, same order.
EV_DEV bool tri_test_flat(const TriFlat *tf, V3 o, V3 d, float tmin, float tmax, float &t, float &beta, float &gamma) {
#pragma clang fp contract(off)
    const float4 *q4 = reinterpret_cast<const float4 *>(tf);
    const float4 a = q4[0], b = q4[1], c = q4[2];
    const float p0x = a.x, p0y = a.y, p0z = a.z, e0x = a.w, e0y = b.x, e0z = b.y, e1x = b.z, e1y = b.w, e1z = c.x, nx = c.y, ny = c.z, nz = c.w;
    float den = __builtin_fmaf(nz, d.z, __builtin_fmaf(ny, d.y, nx * d.x));
    float inv = rcp_exact(den);
    float qx = (p0x - o.x) * inv, qy = (p0y - o.y) * inv, qz = (p0z - o.z) * inv;
    float ix = __builtin_fmaf(d.y, qz, -(d.z * qy)), iy = __builtin_fmaf(d.z, qx, -(d.x * qz)), iz = __builtin_fmaf(d.x, qy, -(d.y * qx));
    beta = __builtin_fmaf(iz, e1z, __builtin_fmaf(iy, e1y, ix * e1x));
    gamma = __builtin_fmaf(iz, e0z, __builtin_fmaf(iy, e0y, ix * e0x));
    t = __builtin_fmaf(nz, qz, __builtin_fmaf(ny, qy, nx * qx));
    return (t < tmax) & (t > tmin) & (beta >= 0.0f) & (gamma >= 0.0f) & (beta + gamma <= 1.0f);
}

// ... on a triangle already in registers (the three float4 of its TriFlat)
EV_DEV bool tri_test_regs(float4 a, float4 b, float4 c, V3 o, V3 d, float tmin, float tmax, float &t, float &beta, float &gamma) {
#pragma clang fp contract(off)
    const float p0x = a.x, p0y = a.y, p0z = a.z, e0x = a.w, e0y = b.x, e0z = b.y, e1x = b.z, e1y = b.w, e1z = c.x, nx = c.y, ny = c.z, nz = c.w;
    float den = __builtin_fmaf(nz, d.z, __builtin_fmaf(ny, d.y, nx * d.x));
    float inv = rcp_exact(den);
    float qx = (p0x - o.x) * inv, qy = (p0y - o.y) * inv, qz = (p0z - o.z) * inv;
    float ix = __builtin_fmaf(d.y, qz, -(d.z * qy)), iy = __builtin_fmaf(d.z, qx, -(d.x * qz)), iz = __builtin_fmaf(d.x, qy, -(d.y * qx));
    beta = __builtin_fmaf(iz, e1z, __builtin_fmaf(iy, e1y, ix * e1x));
    gamma = __builtin_fmaf(iz, e0z, __builtin_fmaf(iy, e0y, ix * e0x));
    t = __builtin_fmaf(nz, qz, __builtin_fmaf(ny, qy, nx * qx));
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
// ((k + 1) 2^-24 as one fused multiply-add on (float)k: k < 2^24, so every step is exact -- the same value, one instruction fewer per draw)
EV_DEV float rng_uniform(Rng &r) { return __builtin_fmaf((float)(rng_u32(r) >> 8), 1.0f / 16777216.0f, 1.0f / 16777216.0f); }

// -------------------------------------------------------------------------------- textures
// tex2D, RT_FILTER_LINEAR / RT_WRAP_REPEAT / normalised coordinates (rt/rtcommon.h:223-245)
EV_DEV int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
// Every operation rounded on its own, in the oracle's order (contract off): texels feed discrete decisions downstream (lobe
// selection and Russian roulette of the light paths, lighttracing.cu:141-166), so they are reproduced bit for bit.
EV_DEV float4 tex2d(const SceneDev &sc, int id, float u, float v) {
#pragma clang fp contract(off)
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
    r.x = ((w00 * p00.x + w10 * p10.x) + w01 * p01.x) + w11 * p11.x;
    r.y = ((w00 * p00.y + w10 * p10.y) + w01 * p01.y) + w11 * p11.y;
    r.z = ((w00 * p00.z + w10 * p10.z) + w01 * p01.z) + w11 * p11.z;
    r.w = ((w00 * p00.w + w10 * p10.w) + w01 * p01.w) + w11 * p11.w;
    return r;
}
// material fetch at a hit (rt/lighttracing.cu:131-133, shaders/deferred.frag:18-21)
EV_DEV void material_at(const SceneDev &sc, const TriAttr &ta, float beta, float gamma, V3 &kd, V3 &ks, float &ns) {
#pragma clang fp contract(off)
    const Material &m = sc.materials[ta.material];
    kd = v3(m.kd); ks = v3(m.ks); ns = m.ns;
    if (m.tex_kd >= 0 || m.tex_ks >= 0 || m.tex_ns >= 0) {
        float w0 = 1.0f - beta - gamma;  // rt/triangleintersect.cu:36
        float u = (ta.uv[2] * beta + ta.uv[4] * gamma) + ta.uv[0] * w0;
        float v = (ta.uv[3] * beta + ta.uv[5] * gamma) + ta.uv[1] * w0;
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
    float sp, cp; evm_sincosf(phi, &sp, &cp);          // shared with the oracle bit for bit (ev_math.h)
    V3 p; p.x = r * cp; p.y = r * sp;
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
    float cos_t = evm_powf(sx, 1.f / (e + 1.f));
    float sin_t = sqrtf(1.0f - cos_t * cos_t);
    float phi = 2.f * EV_PI * sy;
    float sp, cp; evm_sincosf(phi, &sp, &cp);
    V3 p = v3(sin_t * cp, sin_t * sp, cos_t);
    Onb o = onb_make(r);
    out = onb_inverse(o, p);
    float unsafe_cos = dot(out, normal);
    float cos_n = fmaxf(unsafe_cos, 0.f);
    float cos_r = fmaxf(dot(out, r), 0.f);
    if (unsafe_cos > 0.0f) pdfw = (e + 1.0f) * 0.5f * evm_powf(cos_r, e) * EV_INV_PI;
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
// third of a leaf block) arrives with ONE exposed latency instead of one per field.
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
// Scalar loads spelled out.  Left to the compiler a wave-uniform load becomes s_load only while it can prove the memory is not
// written -- which it gives up as soon as the pointer has been anywhere it cannot see through; and pointers it CAN see through
// (kernel arguments) it re-loads from the argument block inside the traversal loop when SGPRs are short: a dependent scalar
// load in front of every node fetch.  So the hot wave-uniform fetches are inline s_load_dwordx16 / x8 (the wait is part of the
// statement: nothing is in flight when it ends), and their base pointers go through `pinned` once per kernel.
EV_DEV v16i sload16(const void *base, uint32_t byte_offset) {
    v16i r;
    asm("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(base), "s"(byte_offset));
    return r;
}
EV_DEV v8i sload8(const void *base, uint32_t byte_offset) {
    v8i r;
    asm("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(base), "s"(byte_offset));
    return r;
}
// two / three consecutive 64-byte blocks with one wait
EV_DEV void sload16x2(const void *base, uint32_t byte_offset, v16i &a, v16i &b) {
    asm("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %3 offset:0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(base), "s"(byte_offset));
}
EV_DEV void sload16x3(const void *base, uint32_t byte_offset, v16i &a, v16i &b, v16i &c) {
    asm("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %4 offset:0x40\n\ts_load_dwordx16 %2, %3, %4 offset:0x80\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(base), "s"(byte_offset));
}
template <class T> EV_DEV const T *pinned(const T *p) { asm("" : "+s"(p)); return p; }
// Asynchronous copy global -> LDS (global_load_lds_dwordx4: every active lane moves 16 bytes from ITS source address to
// lds_dst + 16 * lane; no register in between, counted by vmcnt).  The compiler neither sees the LDS write nor counts the load:
// wait_vmcnt0() before the destination is read.  (M0 carries the LDS base and is compiler-reserved: written in the same statement.)
#ifndef EVPLP_RING_NT
#define EVPLP_RING_NT 1
#endif
EV_DEV void lds_dma16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    // (nt: the cut slots are read once per tile and never again -- a stream that should not displace the tree in the XCD's L2: gather at
    // config #2 52.5 -> 51.9 ms, box scene 20.7 -> 20.0.  Measured with it and not kept: sc1 (drop-from-L2) stores of the slots and of the
    // partial sums, 54.3 / 23.5 ms; one emptiness bit per (tile group, VPL) so that empty cuts are neither written nor fetched, 53.8 / 21.7 ms --
    // profiles/r05_cut_stream_experiments.txt)
#if EVPLP_RING_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}
EV_DEV void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> EV_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }   // all but the N youngest vector-memory operations
template <class T> EV_DEV uint32_t lds_offset(const T *p) { return (uint32_t)(uintptr_t)p; }                 // (the low half of a generic pointer into LDS is the LDS address)
// (Measured and removed: prefetching both children's nodes into the scalar cache with one-dword s_loads into SGPRs kept outside
// the allocatable set -- cfg2 85.9 ms with, 78.5 ms without on the furnished scene, 36.5 / 32.2 on the box scene.  The scalar
// pipe of a CU is as busy as its vector pipe in this walk; two more SMEM instructions per visit cost more than they hide.)
EV_DEV float f_of(int x) { return __int_as_float(x); }
EV_DEV v2f pk(int a, int b) { v2f r; r.x = __int_as_float(a); r.y = __int_as_float(b); return r; }
EV_DEV v2f bc(float a) { v2f r; r.x = a; r.y = a; return r; }
EV_DEV v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
EV_DEV v2f pk_min(v2f a, v2f b) { return __builtin_elementwise_min(a, b); }
EV_DEV v2f pk_max(v2f a, v2f b) { return __builtin_elementwise_max(a, b); }

// rcp_exact of both halves: two v_rcp_f32, then the refinement as four packed fmas (a v_pk_fma_f32 rounds each half like v_fma_f32;
// six until round 5, see rcp_exact)
EV_DEV v2f rcp_exact2(v2f x) {
#pragma clang fp contract(off)
    v2f r; r.x = __builtin_amdgcn_rcpf(x.x); r.y = __builtin_amdgcn_rcpf(x.y);
    const v2f one = bc(1.0f);
    const v2f e = pk_fma(-x, r, one);
    r = pk_fma(e, r, r);
    const v2f err = pk_fma(-x, r, one);
    return pk_fma(err, r, r);
}
// Exact test of a PAIR of triangles with packed fp32 (half 0 = triangle A, half 1 = B): the same
// operations in the same order as tri_test, two triangles per instruction.  r[0..23] = the 24 dwords
// of a TriPair.  Returns the two hit flags.
EV_DEV unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
struct Hit2 { bool a, b; };
EV_DEV Hit2 tri_pair_test(v2f p0x, v2f p0y, v2f p0z, v2f e0x, v2f e0y, v2f e0z, v2f e1x, v2f e1y, v2f e1z, v2f nx, v2f ny, v2f nz,
                          V3 o, V3 d, float tmin, float tmax) {
#pragma clang fp contract(off)
    const v2f dx = bc(d.x), dy = bc(d.y), dz = bc(d.z);
    v2f den = pk_fma(nz, dz, pk_fma(ny, dy, nx * dx));
    const v2f inv = rcp_exact2(den);
    v2f qx = (p0x - bc(o.x)) * inv, qy = (p0y - bc(o.y)) * inv, qz = (p0z - bc(o.z)) * inv;
    v2f ix = pk_fma(dy, qz, -(dz * qy)), iy = pk_fma(dz, qx, -(dx * qz)), iz = pk_fma(dx, qy, -(dy * qx));
    v2f beta = pk_fma(iz, e1z, pk_fma(iy, e1y, ix * e1x));
    v2f gamma = pk_fma(iz, e0z, pk_fma(iy, e0y, ix * e0x));
    v2f t = pk_fma(nz, qz, pk_fma(ny, qy, nx * qx));
    // beta >= 0 & gamma >= 0 & beta + gamma <= 1 as ONE compare: min3(beta, gamma, 1 - (beta + gamma)) >= 0.  (1 - s >= 0 exactly
    // when s <= 1: the subtraction is exact for s in [1/2, 2] and keeps its sign elsewhere.  A NaN among beta / gamma would be skipped
    // by v_min3, but they are NaN only when q is not finite, and then t fails its range test -- or when a cross term d * q overflows
    // to inf - inf with q finite, or the denominator is a denormal (the exact reciprocal then returns inf where IEEE division stays
    // finite): the boolean parity with the oracle therefore holds for scenes whose coordinates and edge lengths stay within about
    // 1e-15 ... 1e15 units -- products of three of them inside the float range -- which is the range evplp_build_accel's inputs
    // are expected in; it is not unconditional.)
    const v2f rest = bc(1.0f) - (beta + gamma);
    Hit2 h;
    h.a = (t.x < tmax) & (t.x > tmin) & (__builtin_fminf(__builtin_fminf(beta.x, gamma.x), rest.x) >= 0.0f);
    h.b = (t.y < tmax) & (t.y > tmin) & (__builtin_fminf(__builtin_fminf(beta.y, gamma.y), rest.y) >= 0.0f);
    return h;
}

// Any-hit of a PAIR of triangles for a whole wave (exact predicate, tri_pair_test).
// Measured and removed: a plane-distance pre-test in front of it (t = N / den with N = n . (p0 - o) common to all lanes; a pair
// none of whose live lanes can have t in range skips the division, cross products and barycentrics).  It rejects the coplanar
// neighbours of the surfaces a segment starts and ends on, but costs ~30 instructions per pair and only pays when EVERY live
// lane is rejected: 39 % of the pairs on the box scene, 9 % on the furnished one -- 50.7 / 104.1 ms with it against
// 49.6 / 97.0 ms without (cfg2).  The smaller box padding (bvh_build.cpp) already keeps segments out of those leaves.
struct PairOps { v2f p0x, p0y, p0z, e0x, e0y, e0z, e1x, e1y, e1z, nx, ny, nz; };
EV_DEV bool tri_pair_any(const PairOps &T, V3 o, V3 d, float tmin, float tmax, uint32_t *exact_runs = nullptr) {
    if (exact_runs) (*exact_runs)++;
    Hit2 h = tri_pair_test(T.p0x, T.p0y, T.p0z, T.e0x, T.e0y, T.e0z, T.e1x, T.e1y, T.e1z, T.nx, T.ny, T.nz, o, d, tmin, tmax);
    return h.a | h.b;
}
// the (up to) two pairs of one 192-byte leaf block, fetched with three s_load_dwordx16
struct LeafOps { PairOps A, B; uint32_t cnt; };
EV_DEV LeafOps fetch_leaf(const char *leaf_base, uint32_t leafref) {
    const uint32_t id = ~leafref;
    LeafOps L; L.cnt = (id & 3u) + 1u;
    const uint32_t off = (id >> 2) * 192u;
    v16i a, b;
    if (L.cnt > 2u) {
        v16i c;
        sload16x3(leaf_base, off, a, b, c);
        L.B.p0x = pk(b[8], b[9]); L.B.p0y = pk(b[10], b[11]); L.B.p0z = pk(b[12], b[13]); L.B.e0x = pk(b[14], b[15]); L.B.e0y = pk(c[0], c[1]); L.B.e0z = pk(c[2], c[3]);
        L.B.e1x = pk(c[4], c[5]); L.B.e1y = pk(c[6], c[7]); L.B.e1z = pk(c[8], c[9]); L.B.nx = pk(c[10], c[11]); L.B.ny = pk(c[12], c[13]); L.B.nz = pk(c[14], c[15]);
    } else sload16x2(leaf_base, off, a, b);     // (pair B is never read, the callers test it only when cnt > 2)
    L.A.p0x = pk(a[0], a[1]); L.A.p0y = pk(a[2], a[3]); L.A.p0z = pk(a[4], a[5]); L.A.e0x = pk(a[6], a[7]); L.A.e0y = pk(a[8], a[9]); L.A.e0z = pk(a[10], a[11]);
    L.A.e1x = pk(a[12], a[13]); L.A.e1y = pk(a[14], a[15]); L.A.e1z = pk(b[0], b[1]); L.A.nx = pk(b[2], b[3]); L.A.ny = pk(b[4], b[5]); L.A.nz = pk(b[6], b[7]);
    return L;
}

// ... in two steps (the any-hit walk): the first pair's 24 dwords, then -- after the first pair has been tested -- the second pair's.
// All 48 dwords at once are what pushed the walk's scalar registers into VGPR lanes once the entry cuts took six more of them
// (16 v_writelane + 24 v_readlane per two-pair leaf); the second fetch waits on its own, a latency the other waves of the SIMD cover.
EV_DEV PairOps fetch_leaf_a(const char *leaf_base, uint32_t off, v16i &b) {
    v16i a;
    sload16x2(leaf_base, off, a, b);
    PairOps A;
    A.p0x = pk(a[0], a[1]); A.p0y = pk(a[2], a[3]); A.p0z = pk(a[4], a[5]); A.e0x = pk(a[6], a[7]); A.e0y = pk(a[8], a[9]); A.e0z = pk(a[10], a[11]);
    A.e1x = pk(a[12], a[13]); A.e1y = pk(a[14], a[15]); A.e1z = pk(b[0], b[1]); A.nx = pk(b[2], b[3]); A.ny = pk(b[4], b[5]); A.nz = pk(b[6], b[7]);
    return A;
}
EV_DEV PairOps fetch_leaf_b(const char *leaf_base, uint32_t off, const v16i &b) {
    const v16i c = sload16(leaf_base, off + 128u);
    PairOps B;
    B.p0x = pk(b[8], b[9]); B.p0y = pk(b[10], b[11]); B.p0z = pk(b[12], b[13]); B.e0x = pk(b[14], b[15]); B.e0y = pk(c[0], c[1]); B.e0z = pk(c[2], c[3]);
    B.e1x = pk(c[4], c[5]); B.e1y = pk(c[6], c[7]); B.e1z = pk(c[8], c[9]); B.nx = pk(c[10], c[11]); B.ny = pk(c[12], c[13]); B.nz = pk(c[14], c[15]);
    return B;
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
// scalar stack pointer: no memory latency on either).  It holds at most one entry per tree level and
// evplp_build_accel rejects trees deeper than 62 levels, so 64 entries always suffice.  Replaces rtTrace(..., ray type 1) + rtMaterialAnyHit,
// rt/lighttracing.cu:184-188,290-294.  Returns true for lanes whose segment is occluded.
// [0, 1] clamp that the backend folds into the clamp modifier of the instruction producing x
EV_DEV float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

#ifndef EVPLP_TRAVERSAL_STATS
#define EVPLP_TRAVERSAL_STATS 0    // 1: count node visits / leaf blocks / triangle pairs per walk (diagnostic build, tools/traversal_stats.py)
#endif
struct WalkStats { uint32_t nodes, leaves, pairs, exact; int32_t hit_leaf; uint32_t syn; };

// One node visit of the packet walk, hand-scheduled: the slab tests of both children and the decision where to go next (push the
// other child when both are entered, pop the lane stack when none is).  Written out because the scalar pipe of a CU is as busy as
// its vector pipe in this loop (a prefetch experiment that added six scalar instructions per visit cost 9 %) and the compiler
// spends ~17 scalar instructions and 4-5 branches per visit on the decision alone (lane-mask booleans, s_cselect_b64 / s_and exec /
// s_cbranch_vcc chains); this spends 6-9 and 2-3.  Inline-asm operands cannot name the halves of a register tuple: the node arrives
// as seven 64-bit scalar operands (sub-registers of the 16-dword tuple the load fills), and the six packed temporaries live in FIXED
// registers v[VT:VT+11], which the kernels that use this keep free with amdgpu_num_vgpr(VT).  The nine per-lane ray constants ride
// in five register pairs -- {1/dx, 1/dy} {1/dz, |1/dx|} {|1/dy|, |1/dz|} {-ox/dx, -oy/dy} {-oz/dz, -} -- and op_sel / op_sel_hi
// broadcast the wanted half to both children (the C++ loop keeps every constant in both halves of a pair of its own: 18 registers).
// Same arithmetic and the same descent order (the child more lanes enter first) as the C++ loop below, which stays as the reference
// implementation (EVPLP_WALK_ASM=0, and the counters build).  Afterwards cur is the next node, a leaf reference, or kNoChild when
// nothing was entered and the stack was empty.
#ifndef EVPLP_WALK_ASM
#define EVPLP_WALK_ASM 1
#endif
#define EV_WALK_VISIT_ASM(T0, T1, T2, T3, T4, T5, T0L, T0H, T1L, T1H, T2L, T2H, T3L, T3H, T4L, T4H, T5L, T5H) EV_WALK_VISIT_ASM_("s", T0, T1, T2, T3, T4, T5, T0L, T0H, T1L, T1H, T2L, T2H, T3L, T3H, T4L, T4H, T5L, T5H)
// NC: where the node's six box operands live -- "s" (a node fetched with s_load) or "v" (a synthetic node of an entry cut, read from LDS
// with one address for all lanes: the same value in every lane of a VGPR serves as well)
#define EV_WALK_VISIT_ASM_(NC, T0, T1, T2, T3, T4, T5, T0L, T0H, T1L, T1H, T2L, T2H, T3L, T3H, T4L, T4H, T5L, T5H)                               \
    asm volatile(                                                                                                                            \
        "v_pk_fma_f32 " T0 ", %[cx], %[pa], %[pd] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                                                                                     \
        "v_pk_fma_f32 " T1 ", %[cy], %[pa], %[pd] op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"                                                                                     \
        "v_pk_fma_f32 " T2 ", %[cz], %[pb], %[pe] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                                                                                     \
        "v_pk_fma_f32 " T3 ", %[hx], %[pb], " T0 " op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"                                                        \
        "v_pk_fma_f32 " T4 ", %[hy], %[pc], " T1 " op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"                                                        \
        "v_pk_fma_f32 " T5 ", %[hz], %[pc], " T2 " op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"                                                        \
        "v_pk_fma_f32 " T0 ", %[hx], %[pb], " T0 " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                                                                                     \
        "v_pk_fma_f32 " T1 ", %[hy], %[pc], " T1 " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"                                                                                     \
        "v_pk_fma_f32 " T2 ", %[hz], %[pc], " T2 " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                                                                                     \
        "v_max3_f32 " T3L ", " T3L ", " T4L ", " T5L " clamp\n\t"                                                                            \
        "v_min3_f32 " T0L ", " T0L ", " T1L ", " T2L " clamp\n\t"                                                                            \
        "v_max3_f32 " T3H ", " T3H ", " T4H ", " T5H " clamp\n\t"                                                                            \
        "v_min3_f32 " T0H ", " T0H ", " T1H ", " T2H " clamp\n\t"                                                                            \
        "v_cmp_lt_f32 vcc, " T3L ", " T0L "\n\t"                                                                                             \
        "v_cmp_lt_f32 %[m1], " T3H ", " T0H "\n\t"                                                                                           \
        "s_or_b64 %[t64], vcc, %[m1]\n\t"                                                                                                    \
        "s_cbranch_scc0 L_pop%=\n\t"                                                                                                         \
        "s_cmp_eq_u64 vcc, 0\n\t"                                                                                                            \
        "s_cbranch_scc1 L_c1%=\n\t"                                                                                                          \
        "s_cmp_eq_u64 %[m1], 0\n\t"                                                                                                          \
        "s_cbranch_scc1 L_c0%=\n\t"                                                                                                          \
        "s_bcnt1_i32_b64 %[p0], vcc\n\t"                                                                                                     \
        "s_bcnt1_i32_b64 %[p1], %[m1]\n\t"                                                                                                   \
        "s_cmp_ge_i32 %[p0], %[p1]\n\t"                                                                                                      \
        "s_cselect_b32 %[p0], %[c1], %[c0]\n\t"                                                                                              \
        "s_cselect_b32 %[cur], %[c0], %[c1]\n\t"                                                                                             \
        "s_mov_b32 m0, %[sp]\n\t"                                                                                                            \
        "s_add_i32 %[sp], %[sp], 1\n\t"                                                                                                      \
        "v_writelane_b32 %[vstack], %[p0], m0\n\t"                                                                                           \
        "s_branch L_end%=\n"                                                                                                                 \
        "L_c0%=:\n\t"                                                                                                                        \
        "s_mov_b32 %[cur], %[c0]\n\t"                                                                                                        \
        "s_branch L_end%=\n"                                                                                                                 \
        "L_c1%=:\n\t"                                                                                                                        \
        "s_mov_b32 %[cur], %[c1]\n\t"                                                                                                        \
        "s_branch L_end%=\n"                                                                                                                 \
        "L_pop%=:\n\t"                                                                                                                       \
        "s_brev_b32 %[cur], 1\n\t"                                                                                                           \
        "s_cmp_eq_u32 %[sp], 0\n\t"                                                                                                          \
        "s_cbranch_scc1 L_end%=\n\t"                                                                                                         \
        "s_sub_i32 %[sp], %[sp], 1\n\t"                                                                                                      \
        "s_nop 0\n\t"                                                                                                                        \
        "v_readlane_b32 %[cur], %[vstack], %[sp]\n"                                                                                          \
        "L_end%=:\n"                                                                                                                         \
        : [cur] "+s"(cur), [sp] "+s"(sp), [vstack] "+v"(vstack), [m1] "=&s"(m1_), [t64] "=&s"(t64_), [p0] "=&s"(p0_), [p1] "=&s"(p1_)          \
        : [cx] NC(cx_), [cy] NC(cy_), [cz] NC(cz_), [hx] NC(hx_), [hy] NC(hy_), [hz] NC(hz_), [c0] "s"(c0_), [c1] "s"(c1_),              \
          [pa] "v"(pa_), [pb] "v"(pb_), [pc] "v"(pc_),                                     \
          [pd] "v"(pd_), [pe] "v"(pe_), [lane] "v"(lane_id)                                                                 \
        : "vcc", "scc", "m0", T0L, T0H, T1L, T1H, T2L, T2H, T3L, T3H, T4L, T4H, T5L, T5H)

// VT = first of the twelve reserved temporaries: 52 for the 64-register VPL gather, 116 for the 128-register VSL gather; 0 = the C++ loop.
// CUT: the walk starts from an entry cut (kernels.h CutArgs: `cut` points at the synthetic nodes of this (tile group, VPL), node 0
// carries their count in its first padding word) instead of from the root: every synthetic node is visited like a node -- its two
// children are cut entries with their boxes -- and whatever it lets in is walked to the end before the next one is fetched.
template <int VT = 0, bool CUT = false>
EV_DEV bool occluded_wave(const char *node_base, const char *leaf_base, V3 o, V3 d, float tmin, float tmax, bool alive_lane, WalkStats *ws = nullptr, const char *cut = nullptr, uint32_t cut_off = 0u,
                          const float4 *cut_lds = nullptr) {
    // All control state is wave-uniform (SGPRs): `alive` / `hitm` are 64-bit lane masks, `cur` the
    // node reference, `sp` the stack pointer.  Per-lane registers hold only the ray (1/d, -o/d) and its
    // far bound `tfar`: a lane that is inactive or already occluded carries tfar = -1, so its slab tests
    // fail by themselves and the ballots need no masking with `alive`.
    // The slab test runs in the segment's own parameter u = (t - tmin) / (tmax - tmin): entry / exit distances are
    // clamped to [0, 1] by the clamp modifier of the v_max3 / v_min3 that form them (no separate max with tmin /
    // min with tmax), and a box is entered iff entry < exit (a box wholly before the start or beyond the end clamps
    // both to the same end point; a box that holds part of a triangle inside the range is padded, so its interval
    // is far wider than an ulp).  A lane without a live ray carries +inf as its origin term: every entry and exit
    // distance is +inf, both clamp to 1, and the lane never enters a box -- the ballots need no masking.
    unsigned long long alive = ballot64(alive_lane), hitm = 0ull;
    if (alive == 0ull) return false;
    // `cut` + `cut_off` = the slot of this (tile group, VPL); `cut_off` then walks over its synthetic nodes, `cut_end` is where they end.
    // The count is looked at BEFORE the ray is set up: every second walk of the bench scene starts from an empty cut, and the three
    // exact reciprocals and the origin terms below are ~50 vector instructions.
    uint32_t cut_end = 0u;
    constexpr bool kCutLds = CUT && VT != 0 && EVPLP_WALK_ASM && !EVPLP_TRAVERSAL_STATS;   // the slot was copied to LDS ahead of the walk (gather kernels)
    if constexpr (kCutLds) {
        const int32_t nsyn = __builtin_amdgcn_readfirstlane(__float_as_int(cut_lds[3].z));
        if (nsyn == 0) return false;                  // nothing between the VPL and the tile group
        cut_off = 0u; cut_end = (uint32_t)nsyn;       // (here: node indices)
    } else if constexpr (CUT) {
        // (the count alone: a synthetic node is fetched right where it is visited -- held across the walk of the previous one the
        // compiler parks its sixteen dwords in VGPR lanes, 32 cross-lane moves per synthetic node)
        int32_t nsyn;
        asm volatile("s_load_dword %0, %1, %2 offset:0x38\n\ts_waitcnt lgkmcnt(0)" : "=s"(nsyn) : "s"(cut), "s"(cut_off) : "memory");
        if (nsyn == 0) return false;                  // nothing between the VPL and the tile group
        cut_end = cut_off + ((uint32_t)nsyn << 6);
    }
    const V3 inv0 = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const float ku = 1.0f / (tmax - tmin);
    const V3 inv = inv0 * ku;
    const v2f ivx = bc(inv.x), ivy = bc(inv.y), ivz = bc(inv.z);
    const v2f avx = bc(fabsf(inv.x)), avy = bc(fabsf(inv.y)), avz = bc(fabsf(inv.z));
    const float dead = __builtin_inff();
    v2f nox = bc(alive_lane ? (-(o.x * inv0.x) - tmin) * ku : dead), noy = bc(alive_lane ? (-(o.y * inv0.y) - tmin) * ku : dead),
        noz = bc(alive_lane ? (-(o.z * inv0.z) - tmin) * ku : dead);
    int sp = 0;
    int vstack = 0;
    int32_t cur = 0;  // root is always an inner node
#if EVPLP_WALK_ASM && !EVPLP_TRAVERSAL_STATS
    if constexpr (VT != 0) {
        static_assert(VT == 52 || VT == 116, "reserved temporaries: v[52:63] or v[116:127]");
        const int lane_id = (int)(threadIdx.x & 63u);
        v2f pa_, pb_, pc_, pd_, pe_;
        pa_.x = ivx.x; pa_.y = ivy.x; pb_.x = ivz.x; pb_.y = avx.x; pc_.x = avy.x; pc_.y = avz.x;
        pd_.x = nox.x; pd_.y = noy.x; pe_.x = noz.x; pe_.y = noz.x;
#define EV_VISIT(N)                                                                                                                          \
        {                                                                                                                                    \
            const v2f cx_ = pk(N[0], N[1]), cy_ = pk(N[2], N[3]), cz_ = pk(N[4], N[5]), hx_ = pk(N[6], N[7]), hy_ = pk(N[8], N[9]), hz_ = pk(N[10], N[11]); \
            const int32_t c0_ = N[12], c1_ = N[13];                                                                                          \
            unsigned long long m1_, t64_; int32_t p0_, p1_;                                                                                  \
            if constexpr (VT == 52) EV_WALK_VISIT_ASM("v[52:53]", "v[54:55]", "v[56:57]", "v[58:59]", "v[60:61]", "v[62:63]", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); \
            else EV_WALK_VISIT_ASM("v[116:117]", "v[118:119]", "v[120:121]", "v[122:123]", "v[124:125]", "v[126:127]", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"); \
        }
        for (;;) {
            if constexpr (CUT) {
                if (cut_off >= cut_end) break;
                // the synthetic node from the slot's LDS copy: one address for all lanes (a broadcast read), then into scalar registers like a
                // fetched node (v_readfirstlane; as VGPR operands of the visit its twelve dwords cost the kernel seven spilled registers)
                v16i syn;
                {
                    const float4 *q = cut_lds + 4u * cut_off;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const float4 qq = q[w];
                        if (w < 3 || true) { syn[4 * w] = __builtin_amdgcn_readfirstlane(__float_as_int(qq.x)); syn[4 * w + 1] = __builtin_amdgcn_readfirstlane(__float_as_int(qq.y)); }
                        if (w < 3) { syn[4 * w + 2] = __builtin_amdgcn_readfirstlane(__float_as_int(qq.z)); syn[4 * w + 3] = __builtin_amdgcn_readfirstlane(__float_as_int(qq.w)); }
                    }
                    syn[14] = 0; syn[15] = 0;
                }
                cut_off++;
                EV_VISIT(syn)
            }
            for (;;) {
                while (cur >= 0) {
                    const v16i n = sload16(node_base, (uint32_t)cur << 6);
                    EV_VISIT(n)
                }
                if (cur == kNoChild) break;
                const uint32_t cnt = (((uint32_t)~cur) & 3u) + 1u, loff = (((uint32_t)~cur) >> 2) * 192u;
                bool any;
                if constexpr (CUT) {
                    v16i lb;
                    { const PairOps A = fetch_leaf_a(leaf_base, loff, lb); any = tri_pair_any(A, o, d, tmin, tmax); }
                    if (cnt > 2u) { const PairOps B = fetch_leaf_b(leaf_base, loff, lb); any = any | tri_pair_any(B, o, d, tmin, tmax); }
                } else {
                    const LeafOps L = fetch_leaf(leaf_base, (uint32_t)cur);
                    any = tri_pair_any(L.A, o, d, tmin, tmax);
                    if (cnt > 2u) any = any | tri_pair_any(L.B, o, d, tmin, tmax);
                }
                const unsigned long long hm = ballot64(any) & alive;
                if (hm != 0ull) {
                    hitm |= hm;
                    alive &= ~hm;
                    if (alive == 0ull) return ((hitm >> (threadIdx.x & 63u)) & 1ull) != 0ull;
                    if (any) { pd_ = bc(dead); pe_ = bc(dead); }     // newly occluded lanes stop driving the walk
                }
                if (sp == 0) break;
                sp--;
                cur = lane_read(vstack, sp);
            }
            if constexpr (!CUT) break;
        }
#undef EV_VISIT
        return ((hitm >> (threadIdx.x & 63u)) & 1ull) != 0ull;
    }
#endif
    // the reference implementation of the same walk in C++ (EVPLP_WALK_ASM=0, and the counters build)
    auto visit_cpp = [&](const v16i &n) {
        // both children at once (half 0 = child 0, half 1 = child 1), conservative slab test in
        // centre / half-size form: A = ctr/d - o/d, B = hal/|d|, entry = A - B, exit = A + B
        const v2f ax = pk_fma(pk(n[0], n[1]), ivx, nox), ay = pk_fma(pk(n[2], n[3]), ivy, noy), az = pk_fma(pk(n[4], n[5]), ivz, noz);
        const v2f hx = pk(n[6], n[7]), hy = pk(n[8], n[9]), hz = pk(n[10], n[11]);
        const v2f enx = pk_fma(hx, -avx, ax), eny = pk_fma(hy, -avy, ay), enz = pk_fma(hz, -avz, az);
        const v2f exx = pk_fma(hx, avx, ax), exy = pk_fma(hy, avy, ay), exz = pk_fma(hz, avz, az);
        const float tn0 = clamp01(fmaxf(fmaxf(enx.x, eny.x), enz.x)), tf0 = clamp01(fminf(fminf(exx.x, exy.x), exz.x));
        const float tn1 = clamp01(fmaxf(fmaxf(enx.y, eny.y), enz.y)), tf1 = clamp01(fminf(fminf(exx.y, exy.y), exz.y));
        const unsigned long long m0 = ballot64(tn0 < tf0), m1 = ballot64(tn1 < tf1);
        const int32_t c0 = n[12], c1 = n[13];
        // 32-bit scalar compares on purpose: this compiler turns compares of 64-bit masks into lane-mask
        // booleans (s_cselect_b64 / s_and exec / s_cbranch_vcc, 4-5 instructions per branch)
        const uint32_t a0 = (uint32_t)m0 | (uint32_t)(m0 >> 32), a1 = (uint32_t)m1 | (uint32_t)(m1 >> 32);
        if ((a0 | a1) == 0u) { if (sp == 0) cur = kNoChild; else { sp--; cur = lane_read(vstack, sp); } return; }
        if (a0 == 0u) { cur = c1; return; }
        if (a1 == 0u) { cur = c0; return; }
        // both hit: descend into the child wanted by more lanes first, keep the other one on the stack
        // (scalar popcounts through inline asm: given __builtin_popcountll this compiler widens the counts to 64 bits and
        // compares them with a VECTOR instruction, v_cmp_lt_u64 on a v_mov'd copy)
        int p0, p1;
        asm("s_bcnt1_i32_b64 %0, %1" : "=s"(p0) : "s"(m0) : "scc");
        asm("s_bcnt1_i32_b64 %0, %1" : "=s"(p1) : "s"(m1) : "scc");
        const bool first0 = p0 >= p1;
        const int32_t oth = first0 ? c1 : c0;
        vstack = lane_write(oth, sp, vstack);
        sp++;
        cur = first0 ? c0 : c1;
    };
    for (;;) {
        if constexpr (CUT) {
            if (cut_off >= cut_end) break;
            const v16i syn = sload16(cut, cut_off);
            cut_off += 64u;
#if EVPLP_TRAVERSAL_STATS
            if (ws) { ws->nodes++; ws->syn++; }
#endif
            visit_cpp(syn);
        }
        for (;;) {
            while (cur >= 0) {
                const v16i n = sload16(node_base, (uint32_t)cur << 6);
#if EVPLP_TRAVERSAL_STATS
                if (ws) ws->nodes++;
#endif
                visit_cpp(n);
            }
            if (cur == kNoChild) break;
            {
                const uint32_t id = (uint32_t)~cur;
                const uint32_t cnt = (id & 3u) + 1u;
                // a leaf block is two triangle pairs (192 B); fetch all of it before testing
                const LeafOps L = fetch_leaf(leaf_base, (uint32_t)cur);
#if EVPLP_TRAVERSAL_STATS
                if (ws) { ws->leaves++; ws->pairs += cnt > 2u ? 2u : 1u; }
                uint32_t *ex = ws ? &ws->exact : nullptr;
#else
                uint32_t *ex = nullptr;
#endif
                bool any = tri_pair_any(L.A, o, d, tmin, tmax, ex);    // an empty slot B is all zeros: never a hit
                if (cnt > 2u) any = any | tri_pair_any(L.B, o, d, tmin, tmax, ex);
                const unsigned long long hm = ballot64(any) & alive;
                if (hm != 0ull) {
#if EVPLP_TRAVERSAL_STATS
                    if (ws) ws->hit_leaf = cur;
#endif
                    hitm |= hm;
                    alive &= ~hm;
                    if (alive == 0ull) return ((hitm >> (threadIdx.x & 63u)) & 1ull) != 0ull;
                    if (any) { nox = bc(dead); noy = bc(dead); noz = bc(dead); }   // newly occluded lanes stop driving the walk
                }
            }
            if (sp == 0) break;
            sp--;
            cur = lane_read(vstack, sp);
        }
        if constexpr (!CUT) break;
    }
    return ((hitm >> (threadIdx.x & 63u)) & 1ull) != 0ull;
}

// tri_pair_test without the range test folded in: t / beta / gamma of both triangles (bit-identical to tri_test)
// and the barycentric acceptance flags; the caller compares t with its own bounds.
struct Tri2 { v2f t, beta, gamma; bool in_a, in_b; };
EV_DEV Tri2 tri_pair_eval(v2f p0x, v2f p0y, v2f p0z, v2f e0x, v2f e0y, v2f e0z, v2f e1x, v2f e1y, v2f e1z, v2f nx, v2f ny, v2f nz, V3 o, V3 d) {
#pragma clang fp contract(off)
    const v2f dx = bc(d.x), dy = bc(d.y), dz = bc(d.z);
    v2f den = pk_fma(nz, dz, pk_fma(ny, dy, nx * dx));
    const v2f inv = rcp_exact2(den);
    v2f qx = (p0x - bc(o.x)) * inv, qy = (p0y - bc(o.y)) * inv, qz = (p0z - bc(o.z)) * inv;
    v2f ix = pk_fma(dy, qz, -(dz * qy)), iy = pk_fma(dz, qx, -(dx * qz)), iz = pk_fma(dx, qy, -(dy * qx));
    Tri2 r;
    r.beta = pk_fma(iz, e1z, pk_fma(iy, e1y, ix * e1x));
    r.gamma = pk_fma(iz, e0z, pk_fma(iy, e0y, ix * e0x));
    r.t = pk_fma(nz, qz, pk_fma(ny, qy, nx * qx));
    v2f bg = r.beta + r.gamma;
    r.in_a = (r.beta.x >= 0.0f) & (r.gamma.x >= 0.0f) & (bg.x <= 1.0f);
    r.in_b = (r.beta.y >= 0.0f) & (r.gamma.y >= 0.0f) & (bg.y <= 1.0f);
    return r;
}

// Closest-hit traversal of ONE WAVE whose rays share their origin (the camera): the packet walk of occluded_wave with a
// per-lane far bound that shrinks to the lane's best hit, nearer child first by majority vote.  Same result as
// closest_lane (order-independent: ties in t keep the lowest ORIGINAL triangle index).  Replaces the rasteriser's
// depth test for the G-buffer (rt/rtcomphoton/rtcomphoton.h:710-754).
EV_DEV int32_t closest_wave(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, int filter, bool active_lane,
                            float &t_out, float &beta_out, float &gamma_out, const char *cut = nullptr) {
    // cut != nullptr: the walk starts from the entry cut of the tile group (kernels.h PrimaryCutArgs: synthetic nodes, nearest first,
    // node 0 carries their count) instead of from the root
    const V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const v2f ivx = bc(inv.x), ivy = bc(inv.y), ivz = bc(inv.z);
    const v2f avx = bc(fabsf(inv.x)), avy = bc(fabsf(inv.y)), avz = bc(fabsf(inv.z));
    const v2f nox = bc(-(o.x * inv.x)), noy = bc(-(o.y * inv.y)), noz = bc(-(o.z * inv.z));
    int32_t best = -1; float bb = 0.f, bg = 0.f;
    float bt = active_lane ? tmax : -1.0f;          // far bound = best hit so far; inactive lanes never pass a slab test
    if (ballot64(active_lane) == 0ull) return -1;
    int sp = 0, vstack = 0;
    int32_t cur = 0;
    const char *node_base = reinterpret_cast<const char *>(sc.nodes);
    const char *leaf_base = reinterpret_cast<const char *>(sc.leaves);
    // one node visit: both children's slab tests against [tmin, best hit so far]; afterwards cur = the next node, a leaf, or kNoChild
    auto visit = [&](const v16i &n) {
        const v2f ax = pk_fma(pk(n[0], n[1]), ivx, nox), ay = pk_fma(pk(n[2], n[3]), ivy, noy), az = pk_fma(pk(n[4], n[5]), ivz, noz);
        const v2f hx = pk(n[6], n[7]), hy = pk(n[8], n[9]), hz = pk(n[10], n[11]);
        const v2f enx = pk_fma(hx, -avx, ax), eny = pk_fma(hy, -avy, ay), enz = pk_fma(hz, -avz, az);
        const v2f exx = pk_fma(hx, avx, ax), exy = pk_fma(hy, avy, ay), exz = pk_fma(hz, avz, az);
        const float tn0 = fmaxf(fmaxf(enx.x, eny.x), fmaxf(enz.x, tmin)), tf0 = fminf(fminf(exx.x, exy.x), fminf(exz.x, bt));
        const float tn1 = fmaxf(fmaxf(enx.y, eny.y), fmaxf(enz.y, tmin)), tf1 = fminf(fminf(exx.y, exy.y), fminf(exz.y, bt));
        const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
        const unsigned long long m0 = ballot64(h0), m1 = ballot64(h1);
        const int32_t c0 = n[12], c1 = n[13];
        const uint32_t a0 = (uint32_t)m0 | (uint32_t)(m0 >> 32), a1 = (uint32_t)m1 | (uint32_t)(m1 >> 32);
        if ((a0 | a1) == 0u) { if (sp == 0) cur = kNoChild; else { sp--; cur = lane_read(vstack, sp); } return; }
        if (a0 == 0u) { cur = c1; return; }
        if (a1 == 0u) { cur = c0; return; }
        // both wanted: the child that more lanes enter first goes first, the other one waits on the stack
        const unsigned long long near0 = ballot64(h0 && (!h1 || tn0 <= tn1)), near1 = ballot64(h1 && (!h0 || tn1 < tn0));
        const uint32_t p0 = (uint32_t)(__builtin_popcount((uint32_t)near0) + __builtin_popcount((uint32_t)(near0 >> 32)));
        const uint32_t p1 = (uint32_t)(__builtin_popcount((uint32_t)near1) + __builtin_popcount((uint32_t)(near1 >> 32)));
        const bool first0 = p0 >= p1;
        vstack = lane_write(first0 ? c1 : c0, sp, vstack);
        sp++;
        cur = first0 ? c0 : c1;
    };
    uint32_t cut_off = 0u, cut_end = 0u;
    if (cut) {
        const v16i head = *reinterpret_cast<const v16i *>(cut);
        cut_end = (uint32_t)head[14] << 6;
        if (cut_end == 0u) return -1;
    }
    for (;;) {
        if (cut) {
            if (cut_off >= cut_end) break;
            const v16i syn = *reinterpret_cast<const v16i *>(cut + cut_off);
            cut_off += 64u;
            visit(syn);
        }
        for (;;) {
            while (cur >= 0) {
                const v16i n = *reinterpret_cast<const v16i *>(node_base + ((uint32_t)cur << 6));
                visit(n);
            }
            if (cur == kNoChild) break;
            {
                const uint32_t id = (uint32_t)~cur;
                const uint32_t block = id >> 2, cnt = (id & 3u) + 1u;
                const v16i *tp = reinterpret_cast<const v16i *>(leaf_base + block * 192u);
                const int32_t *orig = sc.tri_index + block * 4u;
                const v16i a = tp[0], b = tp[1];
                auto take = [&](float t, float be, float ga, bool inside, int32_t tri) {
                    const bool is_light = tri >= sc.light_first && tri < sc.light_first + sc.light_count;    // wave-uniform
                    if ((filter == 1 && is_light) || (filter == 2 && !is_light)) return;
                    if (inside && t > tmin && t < 3.0e38f && (t < bt || (t == bt && best >= 0 && tri < best))) { bt = t; bb = be; bg = ga; best = tri; }
                };
                {
                    Tri2 r = tri_pair_eval(pk(a[0], a[1]), pk(a[2], a[3]), pk(a[4], a[5]), pk(a[6], a[7]), pk(a[8], a[9]), pk(a[10], a[11]),
                                           pk(a[12], a[13]), pk(a[14], a[15]), pk(b[0], b[1]), pk(b[2], b[3]), pk(b[4], b[5]), pk(b[6], b[7]), o, d);
                    take(r.t.x, r.beta.x, r.gamma.x, r.in_a, orig[0]);
                    if (cnt > 1u) take(r.t.y, r.beta.y, r.gamma.y, r.in_b, orig[1]);
                }
                if (cnt > 2u) {
                    const v16i c = tp[2];
                    Tri2 r = tri_pair_eval(pk(b[8], b[9]), pk(b[10], b[11]), pk(b[12], b[13]), pk(b[14], b[15]), pk(c[0], c[1]), pk(c[2], c[3]),
                                           pk(c[4], c[5]), pk(c[6], c[7]), pk(c[8], c[9]), pk(c[10], c[11]), pk(c[12], c[13]), pk(c[14], c[15]), o, d);
                    take(r.t.x, r.beta.x, r.gamma.x, r.in_a, orig[2]);
                    if (cnt > 3u) take(r.t.y, r.beta.y, r.gamma.y, r.in_b, orig[3]);
                }
            }
            if (sp == 0) break;
            sp--;
            cur = lane_read(vstack, sp);
        }
        if (!cut) break;
    }
    if (best >= 0) { t_out = bt; beta_out = bb; gamma_out = bg; }
    return best;
}

// Per-lane closest-hit traversal with a private stack (incoherent rays: primary visibility and
// light sub-paths).  Replaces rtTrace(..., ray type 0) + meshFineIntersect.  The closest hit is
// order-independent: ties in t keep the lowest ORIGINAL triangle index (same rule as the oracle).
// filter: 0 all, 1 skip light mesh, 2 light mesh only.  Returns original triangle index or -1.
// Four-wide variants (BvhNode4, evplp_types.h): the same boxes, the same leaves, half the dependent node fetches per ray.  A ray of
// an incoherent wave spends its time waiting for the next node (300 k light paths are 4.6 waves per SIMD, nothing to switch to).
EV_DEV void node4_slabs(const BvhNode4 &n, V3 inv, V3 noi, float tmin, float tmax, float tn[4], bool h[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float lo[3] = { n.lo[0][q], n.lo[1][q], n.lo[2][q] }, hi[3] = { n.hi[0][q], n.hi[1][q], n.hi[2][q] };
        tn[q] = slab_near(lo, hi, inv, noi, tmin, tmax, h[q]);
    }
}
// Stack of the four-wide walk: the first LDS_ENTRIES entries in LDS ([entry][lane]), the rest -- reached only by rays that defer more
// children than any ray of the test scenes ever did -- in a per-thread column of global memory (ovf[k * ovf_stride]).  The worst case
// for a tree (3 pushes per four-wide level: 46 entries for the 31-level tree of the bench scene, 12 KB of LDS per wave, 3 waves per
// SIMD) would otherwise set the occupancy of every launch.  LDS_ENTRIES = 0: everything in `stack` (sized for the worst case).
// SPEC ("speculative while-while", Aila & Laine 2009): a lane that reaches a leaf while other lanes of the wave still descend keeps the
// leaf for later and goes on with its walk; the wave then tests up to two leaves per lane in one go.  Fewer, fuller trips through both
// loops; a postponed leaf shrinks the ray later than it could have, so a few more nodes are visited.  Same hits (order-independent).
template <int STACK_STRIDE, int LDS_ENTRIES = 0, int SPEC = 0, bool PAIRS = false>      // SPEC: leaves a lane may postpone (0, 1 or 2); PAIRS: two triangles per trip
EV_DEV int32_t closest_lane4(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, int filter,
                             float &t_out, float &beta_out, float &gamma_out, int32_t *stack /* stack[k*STACK_STRIDE] */,
                             int32_t *ovf = nullptr, uint32_t ovf_stride = 0) {
    auto push = [&](int sp_, int32_t v) {
        if (LDS_ENTRIES == 0 || sp_ < LDS_ENTRIES) stack[sp_ * STACK_STRIDE] = v; else ovf[(size_t)(sp_ - LDS_ENTRIES) * ovf_stride] = v;
    };
    auto top = [&](int sp_) -> int32_t {
        return (LDS_ENTRIES == 0 || sp_ < LDS_ENTRIES) ? stack[sp_ * STACK_STRIDE] : ovf[(size_t)(sp_ - LDS_ENTRIES) * ovf_stride];
    };
    V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    V3 noi = v3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    int32_t best = -1; float bt = tmax, bb = 0.f, bg = 0.f;
    int sp = 0;
    int32_t cur = 0, postponed = kNoChild, postponed2 = kNoChild;
    auto test_leaf = [&](int32_t leaf) {
        int32_t id = ~leaf;
        int32_t block = id >> 2, cnt = (id & 3) + 1;
        if constexpr (PAIRS) {
            // two triangles per trip, their six loads (and the four indices) in flight together: a leaf of a latency-bound walk otherwise
            // waits once per triangle.  An unused slot of the block is all zeros (den = 0: never a hit).
            const float4 *q = reinterpret_cast<const float4 *>(sc.tri_flat + block * 4);
            const int4 og = *reinterpret_cast<const int4 *>(sc.tri_index + block * 4);
            auto take = [&](bool hit, float t, float b, float g, int32_t orig) {
                const bool is_light = orig >= sc.light_first && orig < sc.light_first + sc.light_count;
                if ((filter == 1 && is_light) || (filter == 2 && !is_light)) return;
                if (hit && (t < bt || (t == bt && best >= 0 && orig < best))) { bt = t; bb = b; bg = g; best = orig; }
            };
            {
                const float4 a0 = q[0], b0 = q[1], c0 = q[2], a1 = q[3], b1 = q[4], c1 = q[5];
                float t0, be0, ga0, t1, be1, ga1;
                const bool h0 = tri_test_regs(a0, b0, c0, o, d, tmin, 3.0e38f, t0, be0, ga0);
                const bool h1 = tri_test_regs(a1, b1, c1, o, d, tmin, 3.0e38f, t1, be1, ga1);
                take(h0, t0, be0, ga0, og.x);
                if (cnt > 1) take(h1, t1, be1, ga1, og.y);
            }
            if (cnt > 2) {
                const float4 a0 = q[6], b0 = q[7], c0 = q[8], a1 = q[9], b1 = q[10], c1 = q[11];
                float t0, be0, ga0, t1, be1, ga1;
                const bool h0 = tri_test_regs(a0, b0, c0, o, d, tmin, 3.0e38f, t0, be0, ga0);
                const bool h1 = tri_test_regs(a1, b1, c1, o, d, tmin, 3.0e38f, t1, be1, ga1);
                take(h0, t0, be0, ga0, og.z);
                if (cnt > 3) take(h1, t1, be1, ga1, og.w);
            }
            return;
        }
        for (int32_t k = 0; k < cnt; k++) {
            int32_t orig = sc.tri_index[block * 4 + k];
            bool is_light = orig >= sc.light_first && orig < sc.light_first + sc.light_count;
            if ((filter == 1 && is_light) || (filter == 2 && !is_light)) continue;
            float t, b, g;
            if (tri_test_flat(sc.tri_flat + block * 4 + k, o, d, tmin, 3.0e38f, t, b, g)) {
                if (t < bt || (t == bt && best >= 0 && orig < best)) { bt = t; bb = b; bg = g; best = orig; }
            }
        }
    };
    for (;;) {
        while (cur >= 0) {
            const BvhNode4 &n = sc.nodes4[cur];
            float tn[4]; bool h[4];
            node4_slabs(n, inv, noi, tmin, bt, tn, h);
            int32_t c[4] = { n.child[0], n.child[1], n.child[2], n.child[3] };
#pragma unroll
            for (int q = 0; q < 4; q++) if (!h[q] || c[q] == kNoChild) { tn[q] = 3.0e38f; c[q] = kNoChild; }
            // nearest first: sorting network on (entry distance, child)
#define EV_CSWAP(i_, j_) { const bool sw = tn[j_] < tn[i_]; const float tt = sw ? tn[j_] : tn[i_]; tn[j_] = sw ? tn[i_] : tn[j_]; tn[i_] = tt; \
                           const int32_t cc = sw ? c[j_] : c[i_]; c[j_] = sw ? c[i_] : c[j_]; c[i_] = cc; }
            EV_CSWAP(0, 1) EV_CSWAP(2, 3) EV_CSWAP(0, 2) EV_CSWAP(1, 3) EV_CSWAP(1, 2)
#undef EV_CSWAP
            if (c[3] != kNoChild) { push(sp, c[3]); sp++; }
            if (c[2] != kNoChild) { push(sp, c[2]); sp++; }
            if (c[1] != kNoChild) { push(sp, c[1]); sp++; }
            if (c[0] != kNoChild) cur = c[0];
            else if (sp == 0) cur = kNoChild;                    // nothing left to visit: out of the loop (and of the walk, below)
            else { --sp; cur = top(sp); }
            if constexpr (SPEC >= 1) {
                if (cur < 0 && cur != kNoChild && (postponed == kNoChild || (SPEC >= 2 && postponed2 == kNoChild))) {
                    if (postponed == kNoChild) postponed = cur; else postponed2 = cur;
                    if (sp == 0) cur = kNoChild; else { --sp; cur = top(sp); }
                }
            }
        }
        if (SPEC >= 1 && postponed != kNoChild) { test_leaf(postponed); postponed = kNoChild; }
        if (SPEC >= 2 && postponed2 != kNoChild) { test_leaf(postponed2); postponed2 = kNoChild; }
        if (cur != kNoChild) test_leaf(cur);
        if (sp == 0) break;
        --sp; cur = top(sp);
    }
    if (best >= 0) { t_out = bt; beta_out = bb; gamma_out = bg; }
    return best;
}
template <int STACK_STRIDE, int SPEC = 0>
EV_DEV bool occluded_lane4(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, int32_t *stack) {
    V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    V3 noi = v3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    int sp = 0;
    int32_t cur = 0, postponed = kNoChild;
    auto leaf_hit = [&](int32_t leaf) {
        int32_t id = ~leaf;
        int32_t block = id >> 2, cnt = (id & 3) + 1;
        bool any = false;
        for (int32_t k = 0; k < cnt; k++) {
            float t, b, g;
            any = any | tri_test_flat(sc.tri_flat + block * 4 + k, o, d, tmin, tmax, t, b, g);
        }
        return any;
    };
    for (;;) {
        while (cur >= 0) {
            const BvhNode4 &n = sc.nodes4[cur];
            float tn[4]; bool h[4];
            node4_slabs(n, inv, noi, tmin, tmax, tn, h);
            int32_t next = kNoChild;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int32_t c = n.child[q];
                if (h[q] && c != kNoChild) { if (next != kNoChild) { stack[sp * STACK_STRIDE] = next; sp++; } next = c; }
            }
            if (next != kNoChild) cur = next;
            else if (sp == 0) cur = kNoChild;
            else { --sp; cur = stack[sp * STACK_STRIDE]; }
            if constexpr (SPEC >= 1) {
                if (cur < 0 && cur != kNoChild && postponed == kNoChild) {
                    postponed = cur;
                    if (sp == 0) cur = kNoChild; else { --sp; cur = stack[sp * STACK_STRIDE]; }
                }
            }
        }
        if (SPEC >= 1 && postponed != kNoChild) { if (leaf_hit(postponed)) return true; postponed = kNoChild; }
        if (cur != kNoChild) { if (leaf_hit(cur)) return true; }
        if (sp == 0) break;
        --sp; cur = stack[sp * STACK_STRIDE];
    }
    return false;
}

template <int STACK_STRIDE, int SPEC = 0>      // SPEC: leaves a lane may postpone (closest_lane4)
EV_DEV int32_t closest_lane(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, int filter,
                            float &t_out, float &beta_out, float &gamma_out, int32_t *stack /* stack[k*STACK_STRIDE] */) {
    V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    V3 noi = v3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    int32_t best = -1; float bt = tmax, bb = 0.f, bg = 0.f;
    int sp = 0;
    int32_t cur = 0, postponed = kNoChild, postponed2 = kNoChild;
    auto test_leaf = [&](int32_t leaf) {
        int32_t id = ~leaf;
        int32_t block = id >> 2, cnt = (id & 3) + 1;
        for (int32_t k = 0; k < cnt; k++) {
            int32_t orig = sc.tri_index[block * 4 + k];
            bool is_light = orig >= sc.light_first && orig < sc.light_first + sc.light_count;
            if ((filter == 1 && is_light) || (filter == 2 && !is_light)) continue;
            float t, b, g;
            if (tri_test_flat(sc.tri_flat + block * 4 + k, o, d, tmin, 3.0e38f, t, b, g)) {
                if (t < bt || (t == bt && best >= 0 && orig < best)) { bt = t; bb = b; bg = g; best = orig; }
            }
        }
    };
    // "while-while" (Aila & Laine 2009): every lane first descends to its next leaf, then the wave tests leaves together; with one
    // node-or-leaf step per iteration a wave of incoherent rays runs both branches, half empty, every time
    for (;;) {
        while (cur >= 0) {
            const BvhNode &n = sc.nodes[cur];
            bool h0, h1;
            const float lo0[3] = { n.ctr[0][0] - n.hal[0][0], n.ctr[1][0] - n.hal[1][0], n.ctr[2][0] - n.hal[2][0] };
            const float hi0[3] = { n.ctr[0][0] + n.hal[0][0], n.ctr[1][0] + n.hal[1][0], n.ctr[2][0] + n.hal[2][0] };
            const float lo1[3] = { n.ctr[0][1] - n.hal[0][1], n.ctr[1][1] - n.hal[1][1], n.ctr[2][1] - n.hal[2][1] };
            const float hi1[3] = { n.ctr[0][1] + n.hal[0][1], n.ctr[1][1] + n.hal[1][1], n.ctr[2][1] + n.hal[2][1] };
            float n0 = slab_near(lo0, hi0, inv, noi, tmin, bt, h0);
            float n1 = slab_near(lo1, hi1, inv, noi, tmin, bt, h1);
            if (h0 && h1) {
                bool first0 = n0 <= n1;
                stack[sp * STACK_STRIDE] = first0 ? n.c1 : n.c0; sp++;
                cur = first0 ? n.c0 : n.c1;
            } else if (h0) cur = n.c0;
            else if (h1) cur = n.c1;
            else if (sp == 0) cur = kNoChild;                    // nothing left to visit
            else { --sp; cur = stack[sp * STACK_STRIDE]; }
            if constexpr (SPEC >= 1) {
                if (cur < 0 && cur != kNoChild && (postponed == kNoChild || (SPEC >= 2 && postponed2 == kNoChild))) {
                    if (postponed == kNoChild) postponed = cur; else postponed2 = cur;
                    if (sp == 0) cur = kNoChild; else { --sp; cur = stack[sp * STACK_STRIDE]; }
                }
            }
        }
        if (SPEC >= 1 && postponed != kNoChild) { test_leaf(postponed); postponed = kNoChild; }
        if (SPEC >= 2 && postponed2 != kNoChild) { test_leaf(postponed2); postponed2 = kNoChild; }
        if (cur != kNoChild) test_leaf(cur);
        if (sp == 0) break;
        --sp; cur = stack[sp * STACK_STRIDE];
    }
    if (best >= 0) { t_out = bt; beta_out = bb; gamma_out = bg; }
    return best;
}

// Per-lane any-hit walk for incoherent shadow rays (path tracer next-event estimation, light-subpath windows):
// true iff some triangle has t in (tmin, tmax) -- order independent, so exact against the oracle's evo_occluded.
template <int STACK_STRIDE, int SPEC = 0>
EV_DEV bool occluded_lane(const SceneDev &sc, V3 o, V3 d, float tmin, float tmax, int32_t *stack) {
    V3 inv = v3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    V3 noi = v3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    int sp = 0;
    int32_t cur = 0, postponed = kNoChild;
    auto leaf_hit = [&](int32_t leaf) {
        int32_t id = ~leaf;
        int32_t block = id >> 2, cnt = (id & 3) + 1;
        bool any = false;
        for (int32_t k = 0; k < cnt; k++) {
            float t, b, g;
            any = any | tri_test_flat(sc.tri_flat + block * 4 + k, o, d, tmin, tmax, t, b, g);
        }
        return any;
    };
    for (;;) {                                                           // while-while, as closest_lane
        while (cur >= 0) {
            const BvhNode &n = sc.nodes[cur];
            const float lo0[3] = { n.ctr[0][0] - n.hal[0][0], n.ctr[1][0] - n.hal[1][0], n.ctr[2][0] - n.hal[2][0] };
            const float hi0[3] = { n.ctr[0][0] + n.hal[0][0], n.ctr[1][0] + n.hal[1][0], n.ctr[2][0] + n.hal[2][0] };
            const float lo1[3] = { n.ctr[0][1] - n.hal[0][1], n.ctr[1][1] - n.hal[1][1], n.ctr[2][1] - n.hal[2][1] };
            const float hi1[3] = { n.ctr[0][1] + n.hal[0][1], n.ctr[1][1] + n.hal[1][1], n.ctr[2][1] + n.hal[2][1] };
            bool h0 = slab_hit(lo0, hi0, inv, noi, tmin, tmax), h1 = slab_hit(lo1, hi1, inv, noi, tmin, tmax);
            if (h0 && h1) { stack[sp * STACK_STRIDE] = n.c1; sp++; cur = n.c0; }
            else if (h0) cur = n.c0;
            else if (h1) cur = n.c1;
            else if (sp == 0) cur = kNoChild;
            else { --sp; cur = stack[sp * STACK_STRIDE]; }
            if constexpr (SPEC >= 1) {
                if (cur < 0 && cur != kNoChild && postponed == kNoChild) {
                    postponed = cur;
                    if (sp == 0) cur = kNoChild; else { --sp; cur = stack[sp * STACK_STRIDE]; }
                }
            }
        }
        if (SPEC >= 1 && postponed != kNoChild) { if (leaf_hit(postponed)) return true; postponed = kNoChild; }
        if (cur != kNoChild) { if (leaf_hit(cur)) return true; }
        if (sp == 0) break;
        --sp; cur = stack[sp * STACK_STRIDE];
    }
    return false;
}

} // namespace evplp
