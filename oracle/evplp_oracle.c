/*
 * evplp_oracle.c -- CPU restatement of the evplp hot path.  TEST INFRASTRUCTURE ONLY
 * (see evplp_oracle.h for the rules and the parity status: "parity unpinned" for the
 * device arithmetic; output surface + camera pinned through oracle/_ref).
 *
 * All reference citations are relative to /root/reference/reflectcuts (rt/ =
 * realtimetechniques/).  Arithmetic is fp32 throughout, compiled with
 * -ffp-contract=off so that every + - * / sqrt is a single IEEE operation: the
 * geometric predicates (triangle test, shadow-ray set-up) are then bit-identical
 * to the HIP kernels, which are written to the same operation order.
 */
#include "evplp_oracle.h"
/* sin / cos / pow of the direction sampling: one implementation shared with the kernels, so that light-path records can be compared
 * bit for bit (see the header; checked against libm in tests/test_oracle_selfcheck.py) */
#include "../evplp_amd/csrc/ev_math.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EVO_PI 3.14159265358979323846f            /* M_PIf */
#define EVO_INV_PI 0.318309886183790671537767526745028724068919291480912897495f /* rt/rtmath.cuh:11 */

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;
static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* optixu: dot = a.x*b.x + a.y*b.y + a.z*b.z */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* optixu: normalize(v) = v * (1.0f / sqrtf(dot(v, v))) */
static inline v3 normalize(v3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return muls(v, inv); }
/* optixu: reflect(i, n) = i - 2.0f * n * dot(n, i) */
static inline v3 reflect(v3 i, v3 n) { float d = dot(n, i); return sub(i, muls(muls(n, 2.0f), d)); }
/* optixu: faceforward(n, i, nref) = n * copysignf(1.0f, dot(i, nref)) */
static inline v3 faceforward(v3 n, v3 i, v3 nref) { return muls(n, copysignf(1.0f, dot(i, nref))); }
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float minf(float a, float b) { return a < b ? a : b; }

/* optixu Onb: binormal from the larger of |n.x|,|n.z|; inverse_transform */
typedef struct { v3 t, b, n; } onb_t;
static inline onb_t onb_make(v3 n) {
    onb_t o; o.n = n;
    if (fabsf(n.x) > fabsf(n.z)) o.b = V3(-n.y, n.x, 0.0f);
    else o.b = V3(0.0f, -n.z, n.y);
    o.b = normalize(o.b);
    o.t = cross(o.b, o.n);
    return o;
}
static inline v3 onb_inverse(const onb_t *o, v3 p) {
    return add(add(muls(o->t, p.x), muls(o->b, p.y)), muls(o->n, p.z));
}
/* optixu cosine_sample_hemisphere(u1,u2,p) */
static inline v3 cosine_sample_hemisphere(float u1, float u2) {
    float r = sqrtf(u1);
    float phi = 2.0f * EVO_PI * u2;
    float sp, cp; evm_sincosf(phi, &sp, &cp);
    v3 p; p.x = r * cp; p.y = r * sp;
    p.z = sqrtf(maxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    return p;
}

/* ---------------------------------------------------------------------- RNG */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* Stands in for curand_init(seed=index, sequence=rngSeed, offset=0) (lighttracing.cu:203,711):
 * one independent stream per (index, sequence, substream). */
void evo_rng_init(evo_rng *r, uint32_t index, uint32_t sequence, uint32_t substream) {
    uint64_t key = ((uint64_t)sequence << 32) | (uint64_t)index;
    uint64_t s0 = splitmix64(key + (uint64_t)substream * 0xD1B54A32D192ED03ull);
    r->inc = splitmix64(s0) | 1ull;
    r->state = s0 + r->inc;
    r->s0 = r->s1 = 0u; r->vsl_draw = -1; r->reserved = 0u;
    (void)evo_rng_u32(r);
}
uint32_t evo_rng_u32(evo_rng *r) {
    uint64_t old = r->state;
    r->state = old * 6364136223846793005ull + r->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
/* The generator of the VSL estimators (vslSplat's three samplers, :395-594): the reference draws them from the pixel's cuRAND XORWOW
 * stream -- a xorshift generator whose seeding is not public -- so the build defines its own here as well, and (round 5) picks one of
 * that family for them instead of the PCG stream above: xoroshiro64 (Blackman & Vigna, "Scrambled linear pseudorandom number
 * generators", 2018: two 32-bit words, rotations 26 / 13, shift 9, period 2^64 - 1), one stream per (pixel, record), ONE STEP PER
 * SAMPLE of an estimator.  The 64 bits of the state after the step are the sample's random numbers: the top 24 bits of each word
 * are its two uniforms ((0, 1] like curand_uniform), the 16 bits those leave over its lobe choice (chooseMaterial, :472 / :548) --
 * over the period every 64-bit state occurs once, so the three are exactly equidistributed and independent of each other.  The draws
 * the reference makes and does not use (:414, :506, :579) exist there only to advance its stream; here they consume nothing.
 * Seed: the pixel's and the record's splitmix64 keys, multiplied into each other (the product makes the state a non-linear function
 * of the two, so the F2-linear generator carries no XOR relation between the streams of (p, r), (p, r'), (p', r), (p', r')).
 * evo_rng_uniform serves the samplers' draws in the order the reference makes them: #0 lobe choice, #1 / #2 the two uniforms. */
void evo_vsl_rng_init(evo_rng *r, uint32_t index, uint32_t sequence, uint32_t substream) {
    uint64_t pk = splitmix64(((uint64_t)sequence << 32) | (uint64_t)index);
    uint64_t rk = splitmix64((uint64_t)substream * 0xD1B54A32D192ED03ull);
    uint32_t a = (uint32_t)pk ^ (uint32_t)rk, b = (uint32_t)(pk >> 32) ^ (uint32_t)(rk >> 32);
    uint64_t t = (uint64_t)a * (uint64_t)(b | 1u);
    r->state = r->inc = 0ull;
    r->s0 = (uint32_t)t ^ b;
    r->s1 = ((uint32_t)(t >> 32) ^ a) | 0x80000000u;      /* (never the all-zero state) */
    r->vsl_draw = 3; r->reserved = 0u;
}
void evo_vsl_rng_step(evo_rng *r) {
    uint32_t t = r->s1 ^ r->s0;
    r->s0 = ((r->s0 << 26) | (r->s0 >> 6)) ^ t ^ (t << 9);
    r->s1 = (t << 13) | (t >> 19);
    r->vsl_draw = 0;
}
/* curand_uniform: (0,1].  ((x>>8)+1) * 2^-24 is exact in fp32. */
float evo_rng_uniform(evo_rng *r) {
    if (r->vsl_draw >= 0) {
        int k = r->vsl_draw++;
        if (k == 0) return ((float)(((r->s0 & 0xffu) << 8) | (r->s1 & 0xffu)) + 0.5f) * (1.0f / 65536.0f);   /* (0, 1): 16 bits */
        return (float)(((k == 1 ? r->s0 : r->s1) >> 8) + 1u) * (1.0f / 16777216.0f);                        /* (a draw past #2 is one the reference does not use) */
    }
    return (float)((evo_rng_u32(r) >> 8) + 1u) * (1.0f / 16777216.0f);
}

/* -------------------------------------------------------------------- scene */
typedef struct { float lo[3], hi[3]; int32_t left, right; int32_t first, count; } bnode;

struct evo_scene {
    int32_t ntri;
    float *verts;   /* 9 per tri */
    float *uvs;     /* 6 per tri */
    int32_t *mat;
    int32_t nmat; evo_material *mats;
    int32_t ntex; evo_texture *tex;
    int32_t light_first, light_count;
    float light_intensity[4];  /* (I*pi, w) */
    float light_unscaled[4];   /* (I, w) */
    float *light_cdf; float light_area;
    /* private BVH over all triangles */
    bnode *nodes; int32_t nnodes; int32_t *order;
    float pad;
};

static int g_threads = 0;
void evo_set_threads(int n) { g_threads = n; }
int evo_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* shapes/trianglemesh.cpp:13-19 Triangle::ComputeArea */
static float tri_area(const float *v) {
    v3 a = ld3(v), b = ld3(v + 3), c = ld3(v + 6);
    v3 cr = cross(sub(b, a), sub(c, a));
    return sqrtf(dot(cr, cr)) / 2.0f;
}

static void tri_bounds(const float *v, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; k++) {
        lo[k] = minf(minf(v[k], v[3 + k]), v[6 + k]);
        hi[k] = maxf(maxf(v[k], v[3 + k]), v[6 + k]);
    }
}
/* private BVH of the oracle: top-down binned SAH (12 bins), leaves of <= 4 triangles.  Any correct
 * BVH gives the same query results (the triangle test decides); this one only makes the checker and the
 * CPU baseline reasonably fast. */
static int32_t build_rec(evo_scene *s, int32_t first, int32_t count) {
    int32_t id = s->nnodes++;
    bnode *n = &s->nodes[id];
    float clo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, chi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int k = 0; k < 3; k++) { n->lo[k] = 3.0e38f; n->hi[k] = -3.0e38f; }
    for (int32_t i = 0; i < count; i++) {
        float lo[3], hi[3];
        tri_bounds(s->verts + 9 * (size_t)s->order[first + i], lo, hi);
        for (int k = 0; k < 3; k++) {
            n->lo[k] = minf(n->lo[k], lo[k]); n->hi[k] = maxf(n->hi[k], hi[k]);
            float c = 0.5f * (lo[k] + hi[k]); clo[k] = minf(clo[k], c); chi[k] = maxf(chi[k], c);
        }
    }
    for (int k = 0; k < 3; k++) { n->lo[k] -= s->pad; n->hi[k] += s->pad; }
    n->first = first; n->count = count; n->left = n->right = -1;
    if (count <= 4) return id;
    enum { NB = 12 };
    float best = 3.0e38f; int baxis = -1, bbin = -1;
    for (int axis = 0; axis < 3; axis++) {
        float ext = chi[axis] - clo[axis];
        if (!(ext > 0.f)) continue;
        float blo[NB][3], bhi[NB][3]; int cnt[NB];
        for (int b = 0; b < NB; b++) { cnt[b] = 0; for (int k = 0; k < 3; k++) { blo[b][k] = 3.0e38f; bhi[b][k] = -3.0e38f; } }
        float scale = NB / ext;
        for (int32_t i = 0; i < count; i++) {
            float lo[3], hi[3];
            tri_bounds(s->verts + 9 * (size_t)s->order[first + i], lo, hi);
            int b = (int)((0.5f * (lo[axis] + hi[axis]) - clo[axis]) * scale); if (b >= NB) b = NB - 1;
            cnt[b]++;
            for (int k = 0; k < 3; k++) { blo[b][k] = minf(blo[b][k], lo[k]); bhi[b][k] = maxf(bhi[b][k], hi[k]); }
        }
        float ra[NB]; int rc[NB]; float alo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, ahi[3] = { -3.0e38f, -3.0e38f, -3.0e38f }; int c = 0;
        for (int b = NB - 1; b > 0; b--) {
            for (int k = 0; k < 3; k++) { alo[k] = minf(alo[k], blo[b][k]); ahi[k] = maxf(ahi[k], bhi[b][k]); }
            c += cnt[b]; rc[b] = c;
            float dx = ahi[0] - alo[0], dy = ahi[1] - alo[1], dz = ahi[2] - alo[2];
            ra[b] = c ? dx * dy + dy * dz + dz * dx : 0.f;
        }
        for (int k = 0; k < 3; k++) { alo[k] = 3.0e38f; ahi[k] = -3.0e38f; }
        c = 0;
        for (int b = 0; b < NB - 1; b++) {
            for (int k = 0; k < 3; k++) { alo[k] = minf(alo[k], blo[b][k]); ahi[k] = maxf(ahi[k], bhi[b][k]); }
            c += cnt[b];
            if (!c || !rc[b + 1]) continue;
            float dx = ahi[0] - alo[0], dy = ahi[1] - alo[1], dz = ahi[2] - alo[2];
            float cost = (dx * dy + dy * dz + dz * dx) * (float)c + ra[b + 1] * (float)rc[b + 1];
            if (cost < best) { best = cost; baxis = axis; bbin = b; }
        }
    }
    int32_t mid = first;
    if (baxis >= 0) {
        float scale = NB / (chi[baxis] - clo[baxis]);
        int32_t i = first, j = first + count - 1;
        while (i <= j) {
            float lo[3], hi[3];
            tri_bounds(s->verts + 9 * (size_t)s->order[i], lo, hi);
            int b = (int)((0.5f * (lo[baxis] + hi[baxis]) - clo[baxis]) * scale); if (b >= NB) b = NB - 1;
            if (b <= bbin) i++; else { int32_t t = s->order[i]; s->order[i] = s->order[j]; s->order[j] = t; j--; }
        }
        mid = i;
    }
    if (mid == first || mid == first + count) {   /* coincident centroids: split by index */
        mid = first + count / 2;
    }
    int32_t l = build_rec(s, first, mid - first);
    int32_t r = build_rec(s, mid, first + count - mid);
    n = &s->nodes[id];
    n->left = l; n->right = r; n->count = 0;
    return id;
}

evo_scene *evo_scene_create(int32_t ntri, const float *verts, const float *uvs, const int32_t *mat,
                            int32_t nmat, const evo_material *mats, int32_t ntex, const evo_texture *tex,
                            int32_t light_first, int32_t light_count, const float light_intensity[4]) {
    evo_scene *s = (evo_scene *)calloc(1, sizeof(*s));
    s->ntri = ntri;
    s->verts = (float *)malloc(sizeof(float) * 9 * (size_t)ntri);
    memcpy(s->verts, verts, sizeof(float) * 9 * (size_t)ntri);
    s->uvs = (float *)calloc(6 * (size_t)ntri, sizeof(float));
    if (uvs) memcpy(s->uvs, uvs, sizeof(float) * 6 * (size_t)ntri);
    s->mat = (int32_t *)malloc(sizeof(int32_t) * (size_t)ntri);
    memcpy(s->mat, mat, sizeof(int32_t) * (size_t)ntri);
    s->nmat = nmat; s->mats = (evo_material *)malloc(sizeof(evo_material) * (size_t)nmat);
    memcpy(s->mats, mats, sizeof(evo_material) * (size_t)nmat);
    s->ntex = ntex; s->tex = (evo_texture *)calloc((size_t)(ntex > 0 ? ntex : 1), sizeof(evo_texture));
    for (int i = 0; i < ntex; i++) {
        size_t n = (size_t)tex[i].w * tex[i].h * 4;
        float *d = (float *)malloc(n * sizeof(float)); memcpy(d, tex[i].rgba, n * sizeof(float));
        s->tex[i].w = tex[i].w; s->tex[i].h = tex[i].h; s->tex[i].rgba = d;
    }
    s->light_first = light_first; s->light_count = light_count;
    /* rt/rtcommon.h:780-782: xyz scaled by pi for sampling, w (emission Phong exponent) kept */
    memcpy(s->light_unscaled, light_intensity, sizeof(float) * 4);
    for (int k = 0; k < 3; k++) s->light_intensity[k] = light_intensity[k] * EVO_PI;
    s->light_intensity[3] = light_intensity[3];
    /* rt/rtcommon.h:784-790: the light mesh gets a black material carrying mLightIntensity = I*pi */
    for (int32_t i = 0; i < light_count; i++) memcpy(s->mats[s->mat[light_first + i]].light, s->light_intensity, sizeof(float) * 4);
    /* rt/rtcommon.h:501-531 createOptixCdf: running float sum, then normalise */
    s->light_cdf = (float *)malloc(sizeof(float) * (size_t)(light_count > 0 ? light_count : 1));
    float sum = 0.f;
    for (int32_t i = 0; i < light_count; i++) { sum += tri_area(s->verts + 9 * (size_t)(light_first + i)); s->light_cdf[i] = sum; }
    for (int32_t i = 0; i < light_count; i++) s->light_cdf[i] /= sum;
    s->light_area = sum;
    /* private BVH; boxes padded so the (inexact) slab test can never cull a triangle the
     * exact triangle test would accept */
    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    for (size_t i = 0; i < (size_t)ntri * 3; i++) for (int k = 0; k < 3; k++) {
        lo[k] = minf(lo[k], s->verts[3 * i + k]); hi[k] = maxf(hi[k], s->verts[3 * i + k]);
    }
    float diag = sqrtf((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    s->pad = 1e-5f * diag;
    s->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ntri > 0 ? ntri : 1));
    for (int32_t i = 0; i < ntri; i++) s->order[i] = i;
    s->nodes = (bnode *)malloc(sizeof(bnode) * (size_t)(2 * ntri + 2));
    s->nnodes = 0;
    if (ntri > 0) build_rec(s, 0, ntri);
    return s;
}
void evo_scene_destroy(evo_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->ntex; i++) free((void *)s->tex[i].rgba);
    free(s->tex); free(s->verts); free(s->uvs); free(s->mat); free(s->mats);
    free(s->light_cdf); free(s->nodes); free(s->order); free(s);
}
float evo_scene_light_area(const evo_scene *s) { return s->light_area; }
/* rt/rtcommon.h:759-768 totalArea(): all meshes, light mesh included (it is in mMeshes) */
float evo_scene_total_area(const evo_scene *s) {
    float sum = 0.f;
    for (int32_t i = 0; i < s->ntri; i++) sum += tri_area(s->verts + 9 * (size_t)i);
    return sum;
}
/* rt/rtcommon.h:805-814 + math/aabb.h:27-31 */
float evo_scene_bounding_sphere_radius(const evo_scene *s) {
    float lo[3] = { 3.4028235e38f, 3.4028235e38f, 3.4028235e38f }, hi[3] = { -3.4028235e38f, -3.4028235e38f, -3.4028235e38f };
    for (size_t i = 0; i < (size_t)s->ntri * 3; i++) for (int k = 0; k < 3; k++) {
        lo[k] = minf(lo[k], s->verts[3 * i + k]); hi[k] = maxf(hi[k], s->verts[3 * i + k]);
    }
    v3 d = V3(maxf(hi[0] - lo[0], 0.f), maxf(hi[1] - lo[1], 0.f), maxf(hi[2] - lo[2], 0.f));
    return sqrtf(dot(d, d)) / 2.0f;
}

/* ------------------------------------------------------------ ray queries */
/* optix::intersect_triangle_branchless (OptiX SDK 4.1.1 optixu_math_namespace.h), as called
 * from rt/triangleintersect.cu:27.  e0 = p1-p0, e1 = p0-p2, n = cross(e1,e0). */
static inline int tri_test(v3 p0, v3 p1, v3 p2, v3 o, v3 d, float tmin, float tmax,
                           float *t, float *beta, float *gamma, v3 *nout) {
    /* The reference is built by nvcc with -fmad=true: its mul+add pairs are fused wherever the compiler
     * chose to.  This restatement FIXES the fusion placement (shared with the HIP kernels, device_common.hpp
     * tri_test / tri_pair_test): operand set-up e0, e1, n unfused; every dot product
     * fma(z, z', fma(y, y', x*x')); every cross component fma(a, b, -(c*d)); IEEE division. */
    v3 e0 = sub(p1, p0);
    v3 e1 = sub(p0, p2);
    v3 n = cross(e1, e0);
    float den = fmaf(n.z, d.z, fmaf(n.y, d.y, n.x * d.x));
    float inv = 1.0f / den;
    v3 q = V3((p0.x - o.x) * inv, (p0.y - o.y) * inv, (p0.z - o.z) * inv);
    float ix = fmaf(d.y, q.z, -(d.z * q.y)), iy = fmaf(d.z, q.x, -(d.x * q.z)), iz = fmaf(d.x, q.y, -(d.y * q.x));
    *beta = fmaf(iz, e1.z, fmaf(iy, e1.y, ix * e1.x));
    *gamma = fmaf(iz, e0.z, fmaf(iy, e0.y, ix * e0.x));
    *t = fmaf(n.z, q.z, fmaf(n.y, q.y, n.x * q.x));
    if (nout) *nout = n;
    return (*t < tmax) & (*t > tmin) & (*beta >= 0.0f) & (*gamma >= 0.0f) & (*beta + *gamma <= 1.0f);
}
int evo_tri_test(const float p0[3], const float p1[3], const float p2[3], const float o[3], const float d[3],
                 float tmin, float tmax, float *t, float *beta, float *gamma) {
    return tri_test(ld3(p0), ld3(p1), ld3(p2), ld3(o), ld3(d), tmin, tmax, t, beta, gamma, NULL);
}
/* rt/triangleintersect.cu:62-81 meshBound: degenerate triangles have no bounds -> never hit */
static inline int tri_valid(const float *v) {
    v3 a = ld3(v), b = ld3(v + 3), c = ld3(v + 6);
    v3 cr = cross(sub(b, a), sub(c, a));
    float area = sqrtf(dot(cr, cr));
    return area > 0.0f && !isinf(area);
}
static inline int slab(const bnode *n, v3 o, v3 inv, float tmin, float tmax) {
    float t0 = (n->lo[0] - o.x) * inv.x, t1 = (n->hi[0] - o.x) * inv.x;
    float tn = minf(t0, t1), tf = maxf(t0, t1);
    t0 = (n->lo[1] - o.y) * inv.y; t1 = (n->hi[1] - o.y) * inv.y;
    tn = maxf(tn, minf(t0, t1)); tf = minf(tf, maxf(t0, t1));
    t0 = (n->lo[2] - o.z) * inv.z; t1 = (n->hi[2] - o.z) * inv.z;
    tn = maxf(tn, minf(t0, t1)); tf = minf(tf, maxf(t0, t1));
    tn = maxf(tn, tmin); tf = minf(tf, tmax);
    /* NaN-safe: (0 * inf) slabs compare false -> treat as overlapping */
    return !(tn > tf * 1.0000004f + 1e-30f);
}
static inline int tri_filter_ok(const evo_scene *s, int32_t tri, int filter) {
    int is_light = tri >= s->light_first && tri < s->light_first + s->light_count;
    return filter == 0 || (filter == 1 && !is_light) || (filter == 2 && is_light);
}
int evo_occluded(const evo_scene *s, const float o_[3], const float d_[3], float tmin, float tmax) {
    if (s->ntri == 0) return 0;
    v3 o = ld3(o_), d = ld3(d_);
    v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const bnode *n = &s->nodes[stack[--sp]];
        if (!slab(n, o, inv, tmin, tmax)) continue;
        if (n->left < 0) {
            for (int32_t i = 0; i < n->count; i++) {
                const float *v = s->verts + 9 * (size_t)s->order[n->first + i];
                float t, b, g;
                if (tri_valid(v) && tri_test(ld3(v), ld3(v + 3), ld3(v + 6), o, d, tmin, tmax, &t, &b, &g, NULL)) return 1;
            }
        } else { stack[sp++] = n->left; stack[sp++] = n->right; }
    }
    return 0;
}
int evo_occluded_brute(const evo_scene *s, const float o_[3], const float d_[3], float tmin, float tmax) {
    v3 o = ld3(o_), d = ld3(d_);
    for (int32_t i = 0; i < s->ntri; i++) {
        const float *v = s->verts + 9 * (size_t)i; float t, b, g;
        if (tri_valid(v) && tri_test(ld3(v), ld3(v + 3), ld3(v + 6), o, d, tmin, tmax, &t, &b, &g, NULL)) return 1;
    }
    return 0;
}
/* closest hit = rtPotentialIntersection(t) shrinking tmax; ties keep the lowest triangle index */
int evo_closest(const evo_scene *s, const float o_[3], const float d_[3], float tmin, float tmax,
                int filter, float *t_out, float *beta_out, float *gamma_out) {
    if (s->ntri == 0) return -1;
    v3 o = ld3(o_), d = ld3(d_);
    v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int32_t best = -1; float bt = tmax, bb = 0, bg = 0;
    int32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const bnode *n = &s->nodes[stack[--sp]];
        if (!slab(n, o, inv, tmin, bt)) continue;
        if (n->left < 0) {
            for (int32_t i = 0; i < n->count; i++) {
                int32_t tri = s->order[n->first + i];
                if (!tri_filter_ok(s, tri, filter)) continue;
                const float *v = s->verts + 9 * (size_t)tri; float t, b, g;
                if (!tri_valid(v)) continue;
                /* accept t == bt only for a lower triangle index (order-independent result) */
                if (tri_test(ld3(v), ld3(v + 3), ld3(v + 6), o, d, tmin, 3.0e38f, &t, &b, &g, NULL)) {
                    if (t < bt || (t == bt && best >= 0 && tri < best)) { bt = t; bb = b; bg = g; best = tri; }
                }
            }
        } else {
            /* near child first (pushed last): split-axis heuristic by box centre along the ray direction */
            const bnode *L = &s->nodes[n->left], *R = &s->nodes[n->right];
            float cl = (L->lo[0] + L->hi[0]) * d.x + (L->lo[1] + L->hi[1]) * d.y + (L->lo[2] + L->hi[2]) * d.z;
            float cr = (R->lo[0] + R->hi[0]) * d.x + (R->lo[1] + R->hi[1]) * d.y + (R->lo[2] + R->hi[2]) * d.z;
            if (cl <= cr) { stack[sp++] = n->right; stack[sp++] = n->left; } else { stack[sp++] = n->left; stack[sp++] = n->right; }
        }
    }
    if (best >= 0) { *t_out = bt; *beta_out = bb; *gamma_out = bg; }
    return best;
}

/* ---------------------------------------------------------------- textures */
/* tex2D on an RT_FILTER_LINEAR / RT_WRAP_REPEAT / normalised-coordinate sampler
 * (rt/rtcommon.h:223-245): CUDA linear filtering, x_B = x*N - 0.5, repeat addressing. */
static inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static void tex2d(const evo_texture *t, float u, float v, float out[4]) {
    if (t->w == 1 && t->h == 1) { memcpy(out, t->rgba, sizeof(float) * 4); return; }
    float xb = u * (float)t->w - 0.5f, yb = v * (float)t->h - 0.5f;
    float xf = floorf(xb), yf = floorf(yb);
    float a = xb - xf, b = yb - yf;
    int x0 = wrapi((int)xf, t->w), x1 = wrapi((int)xf + 1, t->w);
    int y0 = wrapi((int)yf, t->h), y1 = wrapi((int)yf + 1, t->h);
    const float *p00 = t->rgba + 4 * ((size_t)y0 * t->w + x0), *p10 = t->rgba + 4 * ((size_t)y0 * t->w + x1);
    const float *p01 = t->rgba + 4 * ((size_t)y1 * t->w + x0), *p11 = t->rgba + 4 * ((size_t)y1 * t->w + x1);
    for (int k = 0; k < 4; k++)
        out[k] = (1.0f - a) * (1.0f - b) * p00[k] + a * (1.0f - b) * p10[k] + (1.0f - a) * b * p01[k] + a * b * p11[k];
}
static void material_at(const evo_scene *s, int32_t tri, float beta, float gamma, v3 *kd, v3 *ks, float *ns) {
    const evo_material *m = &s->mats[s->mat[tri]];
    *kd = ld3(m->kd); *ks = ld3(m->ks); *ns = m->ns;
    if (m->tex_kd >= 0 || m->tex_ks >= 0 || m->tex_ns >= 0) {
        /* rt/triangleintersect.cu:33-36: texcoord = t1*beta + t2*gamma + t0*(1-beta-gamma) */
        const float *uv = s->uvs + 6 * (size_t)tri;
        float w0 = 1.0f - beta - gamma;
        float u = uv[2] * beta + uv[4] * gamma + uv[0] * w0;
        float v = uv[3] * beta + uv[5] * gamma + uv[1] * w0;
        float c[4];
        if (m->tex_kd >= 0) { tex2d(&s->tex[m->tex_kd], u, v, c); *kd = V3(c[0], c[1], c[2]); }
        if (m->tex_ks >= 0) { tex2d(&s->tex[m->tex_ks], u, v, c); *ks = V3(c[0], c[1], c[2]); }
        if (m->tex_ns >= 0) { tex2d(&s->tex[m->tex_ns], u, v, c); *ns = c[0]; }
    }
}

/* ------------------------------------------------------ rt/rtmaterial.cuh */
static inline float max_color(v3 c) { return maxf(maxf(c.x, c.y), c.z); } /* :25-28 */
/* :30-38 */
static inline float geometry_term(v3 n1, v3 n2, v3 v12) {
    float c1 = maxf(dot(n1, v12), 0.f), c2 = maxf(-dot(n2, v12), 0.f), d2 = dot(v12, v12);
    return c1 * c2 / (d2 * d2);
}
/* :40-44 (no 1/pi in the CUDA version -- SURVEY A.6 quirk, reproduced) */
static inline float lambert_pdf_w(v3 n1, v3 v12) { return maxf(dot(n1, normalize(v12)), 0.f); }
/* :46-54 */
static inline float lambert_pdf_a(v3 n1, v3 n2, v3 v12) {
    float c1 = maxf(dot(n1, v12), 0.f), c2 = maxf(-dot(n2, v12), 0.f), d2 = dot(v12, v12);
    return c1 * c2 / (d2 * d2) * EVO_INV_PI;
}
/* :78-85 */
static inline float phong_pdf_w(v3 n1, v3 v12, v3 in, v3 rho_s, float e) {
    v3 wi12 = normalize(v12);
    v3 r = normalize(reflect(neg(in), n1));
    float c = maxf(dot(wi12, r), 0.f);
    if (c <= 0.000001f || rho_s.x <= 0.000001f) return 0.0f;
    return (e + 1.0f) * 0.5f * EVO_INV_PI * powf(c, e);
}
/* :87-102 */
static inline float phong_pdf_a(v3 n1, v3 n2, v3 v12, v3 in, v3 rho_s, float e) {
    v3 wi12 = normalize(v12);
    v3 r = normalize(reflect(neg(in), n1));
    float c = maxf(dot(wi12, r), 0.f);
    if (c <= 0.000001f || rho_s.x <= 0.000001f) return 0.0f;
    float pdfw = (e + 1.0f) * 0.5f * EVO_INV_PI * powf(c, e);
    float cos2 = maxf(-dot(n2, wi12), 0.0f);
    float dist2 = dot(v12, v12);
    return pdfw * cos2 / dist2;
}
/* :104-110 */
static inline v3 phong_eval(v3 out, v3 in, v3 n, v3 rho_s, float e) {
    v3 r = reflect(neg(in), n);
    float d = maxf(dot(out, r), 0.0f);
    if (d <= 0.000001f || rho_s.x <= 0.000001f) return V3(0, 0, 0);
    return muls(muls(muls(muls(rho_s, e + 2.0f), powf(d, e)), EVO_INV_PI), 0.5f);
}
/* :112-118 */
static inline float phong_eval_f(v3 out, v3 in, v3 n, float e) {
    v3 r = reflect(neg(in), n);
    float d = maxf(dot(out, r), 0.0f);
    if (d <= 0.000001f) return 0.0f;
    return (e + 2.0f) * powf(d, e) * EVO_INV_PI * 0.5f;
}
/* :56-66.  Draw order pinned left-to-right: first draw -> u1 */
static inline v3 lambert_sample(v3 *out, float *pdfw, v3 normal, v3 rho_d, evo_rng *rng) {
    float u1 = evo_rng_uniform(rng);
    float u2 = evo_rng_uniform(rng);
    v3 p = cosine_sample_hemisphere(u1, u2);
    onb_t o = onb_make(normal);
    *out = onb_inverse(&o, p);
    *pdfw = maxf(dot(*out, normal), 0.f) * EVO_INV_PI;
    return rho_d;
}
/* :120-154 */
static inline v3 phong_sample(v3 *out, float *pdfw, v3 in, v3 normal, v3 rho_s, float e, evo_rng *rng) {
    v3 r = reflect(neg(in), normal);
    float sx = evo_rng_uniform(rng);
    float sy = evo_rng_uniform(rng);
    float cos_t = evm_powf(sx, 1.f / (e + 1.f));
    float sin_t = sqrtf(1.0f - cos_t * cos_t);
    float phi = 2.f * EVO_PI * sy;
    float sp, cp; evm_sincosf(phi, &sp, &cp);
    v3 p = V3(sin_t * cp, sin_t * sp, cos_t);
    onb_t o = onb_make(r);
    *out = onb_inverse(&o, p);
    float unsafe_cos = dot(*out, normal);
    float cos_n = maxf(unsafe_cos, 0.f);
    float cos_r = maxf(dot(*out, r), 0.f);
    if (unsafe_cos > 0.0f) *pdfw = (e + 1.0f) * 0.5f * evm_powf(cos_r, e) * EVO_INV_PI;
    else *pdfw = 0.0f;
    return muls(rho_s, (e + 2.0f) / (e + 1.0f) * cos_n);
}
float evo_tri_area(const float v9[9]) { return tri_area(v9); }
void evo_math_sincos(float x, float *s, float *c) { evm_sincosf(x, s, c); }
float evo_math_pow(float x, float y) { return evm_powf(x, y); }
/* ... over arrays (tests/test_gpu_parity.py: the device's results of the same header against these, bit for bit) */
void evo_math_sincos_array(const float *x, int n, float *s, float *c) { for (int i = 0; i < n; i++) evm_sincosf(x[i], &s[i], &c[i]); }
void evo_math_pow_array(const float *x, const float *y, int n, float *out) { for (int i = 0; i < n; i++) out[i] = evm_powf(x[i], y[i]); }
float evo_phong_eval_f(const float out[3], const float in[3], const float n[3], float e) { return phong_eval_f(ld3(out), ld3(in), ld3(n), e); }
float evo_lambert_pdf_a(const float n1[3], const float n2[3], const float v12[3]) { return lambert_pdf_a(ld3(n1), ld3(n2), ld3(v12)); }
float evo_phong_pdf_a(const float n1[3], const float n2[3], const float v12[3], const float in[3], const float rs[3], float e) {
    return phong_pdf_a(ld3(n1), ld3(n2), ld3(v12), ld3(in), ld3(rs), e);
}
float evo_phong_pdf_w(const float n1[3], const float v12[3], const float in[3], const float rs[3], float e) {
    return phong_pdf_w(ld3(n1), ld3(v12), ld3(in), ld3(rs), e);
}

/* ---------------------------------------------------- rt/rtlightsource.cuh */
/* :24-80 LightSample.  Returns I*pi*Area; draws: 1 (triangle) + 2 (barycentric, left-to-right) */
static v3 light_sample(const evo_scene *s, v3 *position, v3 *normal, float *pdf, evo_rng *rng) {
    float r = evo_rng_uniform(rng);
    uint32_t count = (uint32_t)s->light_count, first = 0;
    while (count > 0) {            /* lower_bound on the normalised CDF */
        uint32_t it = first, step = count / 2; it += step;
        if (s->light_cdf[it] < r) { first = ++it; count -= step + 1; } else count = step;
    }
    if (first >= (uint32_t)s->light_count) first = (uint32_t)s->light_count - 1; /* guard: OOB read in the reference */
    const float *v = s->verts + 9 * (size_t)(s->light_first + (int32_t)first);
    float x = evo_rng_uniform(rng);
    float y = evo_rng_uniform(rng);
    /* rt/rtmath.cuh:22-27 SquareToBarycentric */
    float sq = sqrtf(x), beta = sq * (1.0f - y), gamma = sq * y;
    v3 p1 = ld3(v), p2 = ld3(v + 3), p3 = ld3(v + 6);
    *position = add(add(muls(p1, beta), muls(p2, gamma)), muls(p3, 1.0f - gamma - beta));
    *normal = normalize(cross(sub(p2, p1), sub(p3, p1)));
    *pdf = 1.f / s->light_area;
    return muls(V3(s->light_intensity[0], s->light_intensity[1], s->light_intensity[2]), s->light_area);
}

/* ----------------------------------------------------------------- camera */
typedef struct { v3 eye, s, u, f; float tan_half, aspect; } cam_basis;
/* glm::lookAt (RH) + glm::perspective as used by rt/rtcommon.h:586-591; SURVEY A.8 */
static cam_basis cam_make(const evo_camera *c) {
    cam_basis b; b.eye = ld3(c->origin);
    b.f = normalize(sub(ld3(c->lookat), b.eye));
    b.s = normalize(cross(b.f, ld3(c->up)));
    b.u = cross(b.s, b.f);
    b.tan_half = tanf(c->fovy / 2.0f); b.aspect = c->aspect;
    return b;
}
static inline v3 cam_dir(const cam_basis *b, float ndcx, float ndcy) {
    float dx = ndcx * b->aspect * b->tan_half, dy = ndcy * b->tan_half;
    return add(add(muls(b->s, dx), muls(b->u, dy)), b->f);
}

/* deferred.geom:15-29 / deferred.frag:16-22 restated as one primary ray per pixel centre
 * (SURVEY A.3, A.8): the scene is seen through the jittered matrix, the light mesh through
 * the un-jittered one (rtcomphoton.h:720-727); depth LEQUAL, light mesh drawn last. */
void evo_primary(const evo_scene *s, const evo_camera *cam, int32_t W, int32_t H, const float jitter[2],
                 int32_t row_begin, int32_t row_end,
                 float *g_pos, float *g_nrm, float *g_dif, float *g_phg, float *g_light, int32_t light_unoccluded) {
    cam_basis cb = cam_make(cam);
#pragma omp parallel for schedule(dynamic, 4) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        for (int32_t x = 0; x < W; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            float cx = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
            float cy = ((float)y + 0.5f) / (float)H * 2.0f - 1.0f;
            v3 dj = cam_dir(&cb, cx - jitter[0], cy - jitter[1]);
            v3 d0 = cam_dir(&cb, cx, cy);
            float o[3]; st3(o, cb.eye);
            float dd[3], t, b, g, tl, bl, gl;
            st3(dd, dj);
            int32_t tri = evo_closest(s, o, dd, 0.1f, 100.0f, 1, &t, &b, &g);
            st3(dd, d0);
            int32_t ltri = evo_closest(s, o, dd, 0.1f, 100.0f, 2, &tl, &bl, &gl);
            float pos[4] = { 0, 0, 0, 1 }, nrm[4] = { 0, 0, 0, 0 }, dif[4] = { 0, 0, 0, 0 }, phg[4] = { 0, 0, 0, 0 }, lig[4] = { 0, 0, 0, 0 };
            int use_light = ltri >= 0 && (tri < 0 || tl <= t);
            if (use_light) { tri = ltri; b = bl; g = gl; }
            if (tri >= 0) {
                const float *v = s->verts + 9 * (size_t)tri;
                v3 p0 = ld3(v), p1 = ld3(v + 3), p2 = ld3(v + 6);
                /* interpolated world position of the rasterised fragment */
                v3 P = add(add(muls(p1, b), muls(p2, g)), muls(p0, 1.0f - b - g));
                v3 N = normalize(cross(sub(p1, p0), sub(p2, p0)));
                v3 kd, ks; float ns; material_at(s, tri, b, g, &kd, &ks, &ns);
                st3(pos, P); st3(nrm, N); st3(dif, kd); st3(phg, ks); phg[3] = ns;
            }
            /* the emitter image: depth-tested against the scene unless the frame mode cleared the shared depth buffer before the
             * light pass (cleareveryframe, rtcomphoton.h:989-994) */
            if (light_unoccluded ? ltri >= 0 : use_light) {
                /* light.frag:7-10 with uLightIntensity = unscaled I (rtcomphoton.h:845) */
                lig[0] = s->light_unscaled[0]; lig[1] = s->light_unscaled[1]; lig[2] = s->light_unscaled[2];
            }
            memcpy(g_pos + p, pos, 16); memcpy(g_nrm + p, nrm, 16); memcpy(g_dif + p, dif, 16); memcpy(g_phg + p, phg, 16);
            if (g_light) memcpy(g_light + p, lig, 16);
        }
    }
}

/* ---------------------------------------------------------- light tracing */
/* lighttracing.cu:93-96 */
static inline float russian_prob_lt(v3 f) { return minf(maxf(f.x, maxf(f.y, f.z)), 0.98f); }

/* lighttracing.cu:192-250 tracePhotons with rtMaterialClosestHit (:113-182) inlined */
void evo_trace_light_paths(const evo_scene *s, uint32_t rng_seed, uint32_t path_begin, uint32_t path_count,
                           uint32_t P, evo_record *records) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(evo_get_threads())
    for (int64_t pi = 0; pi < (int64_t)path_count; pi++) {
        uint32_t id = path_begin + (uint32_t)pi;
        evo_record *rec = records + (size_t)id * P;
        for (uint32_t i = 0; i < P; i++) memset(&rec[i], 0, sizeof(evo_record)); /* flags = 0 (:197-200); rest zeroed for determinism */
        evo_rng rng; evo_rng_init(&rng, id, rng_seed, 0);
        v3 position, normal; float pdf;
        v3 flux = light_sample(s, &position, &normal, &pdf, &rng);
        v3 direction; float phong_pdf;
        v3 att = phong_sample(&direction, &phong_pdf, normal, normal, V3(1, 1, 1), s->light_intensity[3], &rng);
        st3(rec[0].pos, position); st3(rec[0].normal, normal); st3(rec[0].flux, flux);
        rec[0].flags = EVO_USABLE_VPL; rec[0].p_select_lambert = 0.0f;
        st3(rec[0].rho_d, V3(0, 0, 0)); st3(rec[0].rho_s, V3(1, 1, 1)); rec[0].phong_exp = s->light_intensity[3];
        st3(rec[0].flux_dir, normal);
        v3 pflux = mulv(flux, att);
        v3 next_pos = position, next_dir = direction;
        for (uint32_t i = 1; i < P; i++) {
            uint32_t flag = (i != P - 1) ? (EVO_USABLE_VPL | EVO_USABLE_PHOTON) : EVO_USABLE_PHOTON;
            float o[3], d[3], t, b, g; st3(o, next_pos); st3(d, next_dir);
            int32_t tri = evo_closest(s, o, d, 0.0001f, 3.0e38f, 0, &t, &b, &g);
            if (tri < 0) break; /* no miss program: prd untouched, loop re-traces the same ray; terminate instead */
            const float *v = s->verts + 9 * (size_t)tri;
            v3 p0 = ld3(v), p1 = ld3(v + 3), p2 = ld3(v + 6);
            /* triangleintersect.cu:31: geometryNormal = normalize(n), n = cross(p0-p2, p1-p0) */
            v3 gn = normalize(cross(sub(p0, p2), sub(p1, p0)));
            v3 wgn = normalize(gn);                         /* :115 rtTransformNormal = identity */
            v3 ffn = faceforward(wgn, neg(next_dir), wgn);  /* :116 */
            v3 hit_pos = add(next_pos, muls(next_dir, t));  /* :120 */
            const evo_material *m = &s->mats[s->mat[tri]];
            if (dot(gn, next_dir) > 0.f || m->light[0] > 0.01f) break; /* :124-128 */
            v3 kd, ks; float ns; material_at(s, tri, b, g, &kd, &ks, &ns);
            float max_l = max_color(kd), max_p = max_color(ks);
            if (max_l + max_p <= 0.000001f) break;          /* :143-147 */
            evo_record *r = &rec[i];
            st3(r->flux_dir, neg(next_dir)); st3(r->pos, hit_pos); st3(r->normal, ffn); st3(r->flux, pflux);
            st3(r->rho_d, kd); st3(r->rho_s, ks); r->phong_exp = ns; r->flags = flag;
            float psel = max_l / (max_p + max_l);
            float choose = minf(evo_rng_uniform(&rng), 0.999999f);
            r->p_select_lambert = psel;
            float russian = russian_prob_lt(pflux);         /* :164 */
            pflux = divs(pflux, russian);
            if (evo_rng_uniform(&rng) >= russian) break;    /* :166-167 */
            v3 dir; float pdfw;
            if (choose < psel) {
                v3 w = lambert_sample(&dir, &pdfw, ffn, kd, &rng);
                pflux = mulv(pflux, divs(w, psel));
                r->flags = flag | EVO_LAMBERT_ONLY;
            } else {
                v3 w = phong_sample(&dir, &pdfw, neg(next_dir), gn, ks, ns, &rng); /* un-flipped normal :176 */
                pflux = mulv(pflux, divs(w, 1.0f - psel));
                r->flags = flag | EVO_PHONG_ONLY;
            }
            next_pos = hit_pos; next_dir = dir;
        }
    }
}

/* -------------------------------------------------------------- VPL gather */
/* lighttracing.cu:275-346 after the visibility test */
static v3 vpl_shade(const evo_frame_params *fp, v3 wi10, v3 p1, v3 n1, v3 rd1, v3 rs1, float e1, const evo_record *rec, v3 v12, float c1c2) {
    v3 pn = ld3(rec->normal), pfd = ld3(rec->flux_dir), pflux = ld3(rec->flux);
    float dist2 = dot(v12, v12);
    float dist = sqrtf(dist2);
    v3 wi12 = divs(v12, dist);
    v3 brdf2 = add(muls(ld3(rec->rho_d), EVO_INV_PI), muls(ld3(rec->rho_s), phong_eval_f(neg(wi12), pfd, pn, rec->phong_exp)));
    v3 brdf1 = add(muls(rd1, EVO_INV_PI), muls(rs1, phong_eval_f(wi10, wi12, n1, e1)));
    float g21 = c1c2 / (dist2 * dist2);
    (void)p1;
    uint32_t mode = fp->mis_mode;
    if (mode == 0) return muls(mulv(mulv(pflux, brdf1), brdf2), g21);
    if (mode >= 1 && mode <= 3) {
        float pdf_de = lambert_pdf_a(pn, n1, neg(v12)) * rec->p_select_lambert;
        pdf_de += phong_pdf_a(pn, n1, neg(v12), pfd, ld3(rec->rho_s), rec->phong_exp) * (1.0f - rec->p_select_lambert);
        float w;
        if (mode == 1) w = fp->pdf_mc / (fp->pdf_mc + pdf_de);
        else if (mode == 2) w = fp->pdf_mc > pdf_de ? 1.0f : 0.0f;
        else { float a2 = fp->pdf_mc * fp->pdf_mc, b2 = pdf_de * pdf_de; w = a2 / (a2 + b2); }
        return muls(mulv(mulv(muls(pflux, w), brdf1), brdf2), g21);
    }
    if (mode == 4) return mulv(mulv(muls(pflux, minf(g21, fp->clamping_value)), brdf1), brdf2);
    /* mode 5 */
    /* g21 * brdf1 * brdf2 evaluated left to right */
    v3 x = mulv(muls(brdf1, g21), brdf2);
    x = V3(minf(x.x, fp->clamping_value), minf(x.y, fp->clamping_value), minf(x.z, fp->clamping_value));
    return mulv(pflux, x);
}
void evo_vpl_splat_pair(const evo_frame_params *fp, const float wi10[3], const float p1_[3], const float n1_[3],
                        const float rd[3], const float rs[3], float e, const evo_record *rec, int visible, float out[3]) {
    v3 p1 = ld3(p1_), n1 = ld3(n1_);
    v3 v12 = sub(ld3(rec->pos), p1);
    float c1 = maxf(dot(n1, v12), 0.0f), c2 = maxf(-dot(ld3(rec->normal), v12), 0.0f);
    float c1c2 = c1 * c2;
    v3 r = V3(0, 0, 0);
    if (!(c1c2 <= 0.0f) && visible) r = vpl_shade(fp, ld3(wi10), p1, n1, ld3(rd), ld3(rs), e, rec, v12, c1c2);
    st3(out, r);
}

/* lvclighttracing.cu:348-384 splatColor of the "lvcphotonfam" variant: a per-pixel random window of
 * num_vpl_light_paths consecutive light paths (mod num_light_paths) over the uncompacted record buffer */
void evo_gather_lvc(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                    const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                    const evo_record *records, float *out, uint64_t *pairs_out) {
    (void)H;
    uint64_t pairs = 0;
    v3 cam = ld3(fp->camera_pos);
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : pairs) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        for (int32_t x = 0; x < W; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            v3 p1 = ld3(g_pos + p);
            if (g_pos[p + 3] == 0.0f) continue;
            v3 n1 = ld3(g_nrm + p), rd = ld3(g_dif + p), rs = ld3(g_phg + p); float e = g_phg[p + 3];
            v3 wi01 = normalize(sub(cam, p1));
            v3 result = V3(0, 0, 0);
            evo_rng rng; evo_rng_init(&rng, (uint32_t)y * (uint32_t)W + (uint32_t)x, fp->rng_seed, 0x4c564300u); /* :369-370 */
            uint32_t offset = (uint32_t)(minf(evo_rng_uniform(&rng), 0.999999f) * (float)fp->num_light_paths);   /* :372 */
            for (uint32_t i = 0; i < fp->num_vpl_light_paths; i++) {
                uint32_t path = (i + offset) % fp->num_light_paths;
                for (uint32_t j = 0; j < fp->photons_per_path; j++) {
                    const evo_record *rec = &records[(size_t)path * fp->photons_per_path + j];
                    if (!(rec->flags & EVO_USABLE_VPL)) continue;
                    pairs++;
                    v3 pv = ld3(rec->pos);
                    v3 v12 = sub(pv, p1);
                    float c1 = maxf(dot(n1, v12), 0.0f), c2 = maxf(-dot(ld3(rec->normal), v12), 0.0f);
                    float c1c2 = c1 * c2;
                    if (c1c2 <= 0.000f) continue;
                    float o[3], d[3]; st3(o, pv); st3(d, neg(v12));
                    if (evo_occluded(s, o, d, 0.0001f, 1.0f - 0.0001f)) continue;
                    result = add(result, vpl_shade(fp, wi01, p1, n1, rd, rs, e, rec, v12, c1c2));
                }
            }
            float inv = (float)fp->num_vpl_light_paths;
            float acc = (float)fp->do_accumulate;
            out[p + 0] = result.x / inv + acc * out[p + 0];
            out[p + 1] = result.y / inv + acc * out[p + 1];
            out[p + 2] = result.z / inv + acc * out[p + 2];
            out[p + 3] = 0.0f + acc * out[p + 3];
        }
    }
    if (pairs_out) *pairs_out = pairs;
}

/* lighttracing.cu:348-379 splatColor */
void evo_gather_vpl(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                    const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                    const evo_record *records, float *out, uint64_t *pairs_out) {
    (void)H;
    uint32_t nrec = fp->photons_per_path * fp->num_vpl_light_paths;
    uint64_t pairs = 0;
    v3 cam = ld3(fp->camera_pos);
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : pairs) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        for (int32_t x = 0; x < W; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            v3 p1 = ld3(g_pos + p);
            float stencil = g_pos[p + 3];
            if (stencil == 0.0f) continue;
            v3 n1 = ld3(g_nrm + p), rd = ld3(g_dif + p), rs = ld3(g_phg + p); float e = g_phg[p + 3];
            v3 wi01 = normalize(sub(cam, p1));
            v3 result = V3(0, 0, 0);
            for (uint32_t i = 0; i < nrec; i++) {
                const evo_record *rec = &records[i];
                if (!(rec->flags & EVO_USABLE_VPL)) continue;
                pairs++;
                v3 pv = ld3(rec->pos);
                v3 v12 = sub(pv, p1);
                float c1 = maxf(dot(n1, v12), 0.0f), c2 = maxf(-dot(ld3(rec->normal), v12), 0.0f);
                float c1c2 = c1 * c2;
                if (c1c2 <= 0.000f) continue;
                /* Ray(photon.mPosition, -v12, 1, 0.0001, 1 - 0.0001) :292 */
                float o[3], d[3]; st3(o, pv); st3(d, neg(v12));
                if (evo_occluded(s, o, d, 0.0001f, 1.0f - 0.0001f)) continue;
                result = add(result, vpl_shade(fp, wi01, p1, n1, rd, rs, e, rec, v12, c1c2));
            }
            float inv = (float)fp->num_vpl_light_paths;
            float acc = (float)fp->do_accumulate;
            out[p + 0] = result.x / inv + acc * out[p + 0];
            out[p + 1] = result.y / inv + acc * out[p + 1];
            out[p + 2] = result.z / inv + acc * out[p + 2];
            out[p + 3] = 0.0f + acc * out[p + 3];
        }
    }
    if (pairs_out) *pairs_out = pairs;
}

/* Counting twin of evo_gather_vpl (test infrastructure for the product's statistics): shadow rays traced = pairs that pass the
 * cosine test of lighttracing.cu:288, and pairs left after the any-hit test of :290-294.  rows may be any subset of image rows. */
void evo_gather_vpl_counts(const evo_scene *s, const evo_frame_params *fp, int32_t W, const int32_t *rows, int32_t nrows,
                           const float *g_pos, const float *g_nrm, const evo_record *records, uint64_t *rays_out, uint64_t *unoccluded_out) {
    uint32_t nrec = fp->photons_per_path * fp->num_vpl_light_paths;
    uint64_t rays = 0, lit = 0;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : rays, lit) num_threads(evo_get_threads())
    for (int32_t r = 0; r < nrows; r++) {
        int32_t y = rows[r];
        for (int32_t x = 0; x < W; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            if (g_pos[p + 3] == 0.0f) continue;
            v3 p1 = ld3(g_pos + p), n1 = ld3(g_nrm + p);
            for (uint32_t i = 0; i < nrec; i++) {
                const evo_record *rec = &records[i];
                if (!(rec->flags & EVO_USABLE_VPL)) continue;
                v3 pv = ld3(rec->pos);
                v3 v12 = sub(pv, p1);
                float c1 = maxf(dot(n1, v12), 0.0f), c2 = maxf(-dot(ld3(rec->normal), v12), 0.0f);
                float c1c2 = c1 * c2;
                if (c1c2 <= 0.000f) continue;
                rays++;
                float o[3], d[3]; st3(o, pv); st3(d, neg(v12));
                if (evo_occluded(s, o, d, 0.0001f, 1.0f - 0.0001f)) continue;
                lit++;
            }
        }
    }
    if (rays_out) *rays_out = rays;
    if (unoccluded_out) *unoccluded_out = lit;
}

/* -------------------------------------------------------------- VSL gather */
/* lighttracing.cu:382-390 */
static inline v3 square_to_solid_angle(float sx, float sy, float half_angle_max) {
    float phi = 2.0f * EVO_PI * sx;
    float z = 1.0f - sy * (1.0f - cosf(half_angle_max));
    float l = sqrtf(1.0f - z * z);
    return V3(cosf(phi) * l, sinf(phi) * l, z);
}
typedef struct {
    v3 wi10, n1, rd1, rs1; float e1;
    const evo_record *rec; v3 pn, pfd, pflux, prd, prs; float pe;
    float half_cone, cos_half_cone, solid_angle, inv_solid_angle; v3 nd12;
    float vsl_inv_pi_r2;
} vsl_ctx;
/* the three estimators share this MIS denominator block (:433-443, 508-518, 581-591); note the
 * quirk of SURVEY A.6: pdfBrdf2 uses the PIXEL's pSelectLambert and no (1-pSel) on its Phong term */
static inline void vsl_pdfs(const vsl_ctx *c, v3 wi12, float psel, float *pdf1, float *pdf2) {
    *pdf1 = lambert_pdf_w(c->n1, wi12) * psel + phong_pdf_w(c->n1, wi12, c->wi10, c->rs1, c->e1) * (1.0f - psel);
    *pdf2 = lambert_pdf_w(c->pn, neg(wi12)) * psel + phong_pdf_w(c->pn, neg(wi12), c->pfd, c->prs, c->pe);
}
/* :395-446 */
static v3 vsl_sample_cone(const vsl_ctx *c, float *w, evo_rng *rng) {
    float ml = max_color(c->rd1), mp = max_color(c->rs1);
    if (ml + mp <= 0.000001f) return V3(0, 0, 0);
    float psel = ml / (mp + ml);
    evo_vsl_rng_step(rng);
    (void)minf(evo_rng_uniform(rng), 0.999999f); /* chooseMaterial: drawn, unused (:414) */
    float a = evo_rng_uniform(rng);
    float b = evo_rng_uniform(rng);
    v3 wi12 = normalize(square_to_solid_angle(a, b, c->half_cone));
    onb_t o = onb_make(c->nd12);
    wi12 = onb_inverse(&o, wi12);
    wi12 = normalize(wi12);
    float c1c2 = fmaxf(dot(c->n1, wi12), 0.0f) * fmaxf(-dot(c->pn, wi12), 0.0f);
    if (c1c2 <= 0.000000001f) return V3(0, 0, 0);
    v3 brdf2 = add(muls(c->prd, EVO_INV_PI), muls(c->prs, phong_eval_f(neg(wi12), c->pfd, c->pn, c->pe)));
    v3 brdf1 = add(muls(c->rd1, EVO_INV_PI), muls(c->rs1, phong_eval_f(c->wi10, wi12, c->n1, c->e1)));
    float pdf1, pdf2; vsl_pdfs(c, wi12, psel, &pdf1, &pdf2);
    float pdf_cone = c->inv_solid_angle;
    *w = pdf_cone / (pdf1 + pdf2 + pdf_cone);
    return muls(mulv(mulv(muls(muls(c->pflux, c->vsl_inv_pi_r2), c1c2), brdf1), brdf2), c->solid_angle);
}
/* :448-521 */
static v3 vsl_sample_brdf1(const vsl_ctx *c, float *w, evo_rng *rng) {
    v3 wi12, brdf1; float pdfw;
    float ml = max_color(c->rd1), mp = max_color(c->rs1);
    if (ml + mp <= 0.000001f) return V3(0, 0, 0);
    float psel = ml / (mp + ml);
    evo_vsl_rng_step(rng);
    float choose = minf(evo_rng_uniform(rng), 0.999999f);
    if (choose < psel) brdf1 = divs(lambert_sample(&wi12, &pdfw, c->n1, c->rd1, rng), psel);
    else brdf1 = divs(phong_sample(&wi12, &pdfw, c->wi10, c->n1, c->rs1, c->e1, rng), 1.0f - psel);
    if (dot(wi12, c->nd12) <= c->cos_half_cone) return V3(0, 0, 0);
    float cos1 = fmaxf(dot(c->n1, wi12), 0.0f);
    if (cos1 <= 0.000000001f) return V3(0, 0, 0);
    float cos2 = fmaxf(-dot(c->pn, wi12), 0.0f);
    v3 brdf2 = add(muls(c->prd, EVO_INV_PI), muls(c->prs, phong_eval_f(neg(wi12), c->pfd, c->pn, c->pe)));
    (void)minf(evo_rng_uniform(rng), 0.999999f); /* second chooseMaterial draw (:506) */
    float pdf1, pdf2; vsl_pdfs(c, wi12, psel, &pdf1, &pdf2);
    *w = pdf1 / (pdf1 + pdf2 + c->inv_solid_angle);
    return mulv(mulv(muls(muls(c->pflux, c->vsl_inv_pi_r2), cos2), brdf1), brdf2);
}
/* :523-594 */
static v3 vsl_sample_brdf2(const vsl_ctx *c, float *w, evo_rng *rng) {
    v3 wi21, brdf2; float pdfw;
    {
        float ml = max_color(c->prd), mp = max_color(c->prs);
        if (ml + mp <= 0.000001f) return V3(0, 0, 0);
        float psel = ml / (mp + ml);
        evo_vsl_rng_step(rng);
        float choose = minf(evo_rng_uniform(rng), 0.999999f);
        if (choose < psel) brdf2 = divs(lambert_sample(&wi21, &pdfw, c->pn, c->prd, rng), psel);
        else brdf2 = divs(phong_sample(&wi21, &pdfw, c->pfd, c->pn, c->prs, c->pe, rng), 1.0f - psel);
    }
    if (-dot(wi21, c->nd12) <= c->cos_half_cone) return V3(0, 0, 0);
    v3 brdf1 = add(muls(c->rd1, EVO_INV_PI), muls(c->rs1, phong_eval_f(c->wi10, neg(wi21), c->n1, c->e1)));
    float cos2 = fmaxf(dot(c->pn, wi21), 0.0f);
    if (cos2 <= 0.00000001f) return V3(0, 0, 0);
    float cos1 = fmaxf(-dot(c->n1, wi21), 0.0f);
    float ml = max_color(c->rd1), mp = max_color(c->rs1);
    if (ml + mp <= 0.000001f) return V3(0, 0, 0);
    float psel = ml / (mp + ml);
    (void)minf(evo_rng_uniform(rng), 0.999999f); /* :579 */
    float pdf1, pdf2; vsl_pdfs(c, neg(wi21), psel, &pdf1, &pdf2);
    *w = pdf2 / (pdf1 + pdf2 + c->inv_solid_angle);
    return mulv(mulv(muls(muls(c->pflux, c->vsl_inv_pi_r2), cos1), brdf1), brdf2);
}
/* :596-686 vslSplat, after the shadow ray (:612-614).  `only` and `samples_override` exist for tests/test_oracle_selfcheck.py alone
 * (evo_vsl_splat_pair): only = 0 is the reference's MIS-combined sum; only = 1 / 2 / 3 keeps the cone / pixel-BRDF / VSL-BRDF estimator
 * ALONE with weight 1 -- each is then by itself an estimator of the same integral; samples_override > 0 replaces numSamples (:632). */
static v3 vsl_splat_lit(const evo_frame_params *fp, v3 wi10, v3 p1, v3 n1, v3 rd1, v3 rs1, float e1,
                        const evo_record *rec, evo_rng *rng, int only, int samples_override) {
    v3 pv = ld3(rec->pos);
    v3 v12 = sub(pv, p1);
    float dist2 = dot(v12, v12);
    float dist = sqrtf(dist2);
    v3 nv12 = divs(v12, dist);
    vsl_ctx c;
    c.pn = ld3(rec->normal);
    float c1c2 = fmaxf(dot(n1, nv12), 0.0f) * fmaxf(-dot(c.pn, nv12), 0.0f);
    if (c1c2 <= 0.000000001f) return V3(0, 0, 0);
    float rdratio = fp->vsl_radius / dist;
    c.half_cone = (rdratio >= 1.0f) ? EVO_PI / 2.0f : asinf(rdratio);
    c.cos_half_cone = cosf(c.half_cone);
    c.solid_angle = EVO_PI * 2.0f * (1.0f - c.cos_half_cone);
    c.inv_solid_angle = 1.0f / c.solid_angle;
    c.wi10 = wi10; c.n1 = n1; c.rd1 = rd1; c.rs1 = rs1; c.e1 = e1; c.rec = rec;
    c.pfd = ld3(rec->flux_dir); c.pflux = ld3(rec->flux); c.prd = ld3(rec->rho_d); c.prs = ld3(rec->rho_s); c.pe = rec->phong_exp;
    c.nd12 = nv12; c.vsl_inv_pi_r2 = fp->vsl_inv_pi_radius2;
    int num_samples = (int)(c.half_cone / EVO_PI * 2.0f * 100.0f) + 1;
    if (samples_override > 0) num_samples = samples_override;
    v3 result = V3(0, 0, 0);
    for (int i = 0; i < num_samples; i++) {
        float wc = 0.0f, w1 = 0.0f, w2 = 0.0f;
        v3 rc = vsl_sample_cone(&c, &wc, rng);
        v3 r1 = vsl_sample_brdf1(&c, &w1, rng);
        v3 r2 = vsl_sample_brdf2(&c, &w2, rng);
        if (only) { wc = only == 1 ? 1.0f : 0.0f; w1 = only == 2 ? 1.0f : 0.0f; w2 = only == 3 ? 1.0f : 0.0f; }
        result = add(result, muls(rc, wc));
        result = add(result, muls(r1, w1));
        result = add(result, muls(r2, w2));
    }
    return divs(result, (float)num_samples);
}
static v3 vsl_splat(const evo_scene *s, const evo_frame_params *fp, v3 wi10, v3 p1, v3 n1, v3 rd1, v3 rs1, float e1,
                    const evo_record *rec, evo_rng *rng) {
    v3 pv = ld3(rec->pos);
    v3 v12 = sub(pv, p1);
    float o[3], d[3]; st3(o, pv); st3(d, neg(v12));
    if (evo_occluded(s, o, d, 0.0001f, 1.0f - 0.0001f)) return V3(0, 0, 0);
    return vsl_splat_lit(fp, wi10, p1, n1, rd1, rs1, e1, rec, rng, 0, 0);
}
void evo_vsl_splat_pair(const evo_frame_params *fp, const float wi10[3], const float p1[3], const float n1[3],
                        const float rd[3], const float rs[3], float e, const evo_record *rec, int visible,
                        uint32_t rng_index, uint32_t rng_sequence, uint32_t rng_substream, int only, int samples_override, float out[3]) {
    v3 r = V3(0, 0, 0);
    if (visible) {
        evo_rng rng; evo_vsl_rng_init(&rng, rng_index, rng_sequence, rng_substream);
        r = vsl_splat_lit(fp, ld3(wi10), ld3(p1), ld3(n1), ld3(rd), ld3(rs), e, rec, &rng, only, samples_override);
    }
    st3(out, r);
}
/* :689-722 splatSplotch.  RNG: one substream per (pixel, record) instead of one cuRAND
 * stream per pixel, so any decomposition of the record loop reproduces the same numbers. */
void evo_gather_vsl_window(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                           int32_t x_begin, int32_t x_end, const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                           const evo_record *records, float *out, uint64_t *pairs_out);
void evo_gather_vsl(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                    const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                    const evo_record *records, float *out, uint64_t *pairs_out) {
    evo_gather_vsl_window(s, fp, W, H, row_begin, row_end, 0, W, g_pos, g_nrm, g_dif, g_phg, records, out, pairs_out);
}
/* the same on the pixels [x_begin, x_end) of the rows only: the estimators of one 2048-pixel row against 12 k VSLs take ~25 s, so the
 * full-size tests sample windows spread over the frame instead of whole rows */
void evo_gather_vsl_window(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                           int32_t x_begin, int32_t x_end, const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                           const evo_record *records, float *out, uint64_t *pairs_out) {
    (void)H;
    uint32_t nrec = fp->photons_per_path * fp->num_vpl_light_paths;
    uint64_t pairs = 0;
    v3 cam = ld3(fp->camera_pos);
#pragma omp parallel for collapse(2) schedule(dynamic, 8) reduction(+ : pairs) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        for (int32_t x = x_begin; x < x_end; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            v3 p1 = ld3(g_pos + p), n1 = ld3(g_nrm + p), rd = ld3(g_dif + p), rs = ld3(g_phg + p); float e = g_phg[p + 3];
            v3 wi10 = normalize(sub(cam, p1));
            v3 result = V3(0, 0, 0);
            uint32_t pixel_id = (uint32_t)y * (uint32_t)W + (uint32_t)x;
            for (uint32_t i = 0; i < nrec; i++) {
                const evo_record *rec = &records[i];
                if (!(rec->flags & EVO_USABLE_VPL)) continue;
                pairs++;
                evo_rng rng; evo_vsl_rng_init(&rng, pixel_id, fp->rng_seed, 1u + i);
                result = add(result, vsl_splat(s, fp, wi10, p1, n1, rd, rs, e, rec, &rng));
            }
            float inv = (float)fp->num_vpl_light_paths, acc = (float)fp->do_accumulate;
            out[p + 0] = result.x / inv + acc * out[p + 0];
            out[p + 1] = result.y / inv + acc * out[p + 1];
            out[p + 2] = result.z / inv + acc * out[p + 2];
            out[p + 3] = 0.0f + acc * out[p + 3];
        }
    }
    if (pairs_out) *pairs_out = pairs;
}

/* -------------------------------------------------------------- photon splat */
/* GLSL helpers of shaders/photonsplatinstanced.frag (they differ from the CUDA ones) */
static inline v3 g_lambert_eval(v3 w10, v3 w12, v3 n, v3 rd) { /* frag:42-50 */
    if (dot(w10, n) <= 0.0f || dot(w12, n) <= 0.0f) return V3(0, 0, 0);
    return muls(rd, EVO_INV_PI);
}
static inline v3 g_phong_eval(v3 outv, v3 inv_, v3 n, v3 rs, float e) { /* frag:52-58 */
    v3 r = reflect(neg(inv_), n);
    float d = dot(outv, r);
    if (d <= 0.00001f) return V3(0, 0, 0);
    return muls(muls(muls(muls(rs, e + 2.0f), powf(d, e)), EVO_INV_PI), 0.5f);
}
static inline float g_lambert_pdf_w(v3 n1, v3 v12) { return maxf(dot(n1, normalize(v12)), 0.f) * EVO_INV_PI; } /* frag:65-69 */
static inline float g_phong_pdf_w(v3 n1, v3 wi12, v3 inv_, v3 rs, float e) { /* frag:79-85 */
    v3 r = reflect(neg(inv_), n1);
    float d = maxf(dot(wi12, r), 0.f);
    if (d <= 0.00001f || rs.x <= 0.00001f) return 0.0f;
    return (e + 1.0f) * 0.5f * EVO_INV_PI * powf(d, e);
}
/* frag:146-240 main().  Returns 0 when the fragment is discarded. */
int evo_photon_frag(const evo_frame_params *fp, const evo_record *ph, const evo_record *prev,
                    const float x_pos[3], const float x_nrm[3], const float x_dif[3], const float x_phg[4], float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    v3 X = ld3(x_pos), ppos = ld3(ph->pos);
    float r2 = fp->photon_radius * fp->photon_radius;
    v3 dv = sub(ppos, X);
    if (dot(dv, dv) > r2) return 0;                                   /* :153-154 */
    v3 sn = ld3(x_nrm), sd = ld3(x_dif), sps = ld3(x_phg); float se = x_phg[3];
    v3 v12 = sub(ld3(prev->pos), ppos);                               /* :170 */
    v3 w12 = normalize(v12);
    v3 n1 = ld3(ph->normal);
    v3 w10 = normalize(sub(ld3(fp->camera_pos), X));
    v3 pfd = ld3(prev->flux_dir), pn = ld3(prev->normal), prs = ld3(prev->rho_s);
    v3 brdf1 = add(g_lambert_eval(w10, w12, sn, sd), g_phong_eval(w10, w12, sn, sps, se)); /* :181 */
    float mix_w = g_lambert_pdf_w(pn, neg(w12)) * prev->p_select_lambert;                  /* :184-187 */
    mix_w += g_phong_pdf_w(pn, neg(w12), pfd, prs, prev->phong_exp) * (1.0f - prev->p_select_lambert);
    float mix_a = mix_w * maxf(dot(n1, w12), 0.0f) / dot(v12, v12);                         /* :189 */
    if (!(mix_w > 0.0f)) return 1;                                    /* colour = 0, not discarded (:235-238) */
    float inv_r2 = 1.0f / (fp->photon_radius * fp->photon_radius);    /* rtcomphoton.h:819 */
    float inv_n = 1.0f / (float)fp->num_light_paths;                  /* rtcomphoton.h:820 */
    float k = EVO_INV_PI * inv_r2;
    v3 flux = ld3(ph->flux);
    v3 base = muls(mulv(muls(brdf1, k), flux), inv_n);                /* brdf1 * (InvPi*invR2) * flux * invN */
    uint32_t mode = fp->mis_mode;
    v3 c;
    if (mode == 0) c = base;
    else if (mode == 1) c = muls(base, mix_a / (mix_a + fp->pdf_mc));
    else if (mode == 2) c = muls(base, mix_a > fp->pdf_mc ? 1.0f : 0.0f);
    else if (mode == 3) { float a2 = mix_a * mix_a, b2 = fp->pdf_mc * fp->pdf_mc; c = muls(base, a2 / (a2 + b2)); }
    else {
        float d2 = dot(v12, v12);
        float cc = maxf(dot(sn, w12), 0.0f) * maxf(-dot(pn, w12), 0.0f);
        if (cc <= 0.0f) return 0;                                     /* discard :218,228 */
        float g = cc / d2;
        if (mode == 4) c = divs(muls(base, maxf(g - fp->clamping_value, 0.0f)), g);
        else {
            v3 brdf2 = add(g_lambert_eval(neg(w12), pfd, pn, ld3(prev->rho_d)), g_phong_eval(neg(w12), pfd, pn, prs, prev->phong_exp)); /* :182 */
            v3 pre = muls(muls(flux, k), inv_n);
            v3 num = muls(mulv(brdf1, brdf2), g);
            num = V3(maxf(num.x - fp->clamping_value, 0.f), maxf(num.y - fp->clamping_value, 0.f), maxf(num.z - fp->clamping_value, 0.f));
            v3 den = muls(brdf2, g);
            /* deliberate deviation: the reference divides by zero (NaN) when a component of brdf2 is 0
             * (SURVEY A.9); a zero denominator contributes 0 here. */
            c = V3(den.x != 0.f ? pre.x * num.x / den.x : 0.f, den.y != 0.f ? pre.y * num.y / den.y : 0.f, den.z != 0.f ? pre.z * num.z / den.z : 0.f);
        }
    }
    st3(out, c);
    return 1;
}
/* runPhotonSplat rtcomphoton.h:789-837 with the ideal kernel of SURVEY A.4: photon i adds to
 * pixel p iff |X_p - P_i|^2 <= r^2 (X_p = visible surface point).  Accumulation order here is
 * record index ascending per pixel. */
void evo_splat_photons(const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                       const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                       const evo_record *records, uint32_t num_records, float *out, uint64_t *pairs_out) {
    (void)H;
    /* uniform grid over photon positions, cell = radius */
    float r = fp->photon_radius;
    uint64_t pairs = 0;
    if (!(r > 0.0f)) { if (pairs_out) *pairs_out = 0; return; }
    uint32_t nph = 0; float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    for (uint32_t i = 0; i < num_records; i++) if (records[i].flags & EVO_USABLE_PHOTON) {
        nph++; for (int k = 0; k < 3; k++) { lo[k] = minf(lo[k], records[i].pos[k]); hi[k] = maxf(hi[k], records[i].pos[k]); }
    }
    if (!nph) { if (pairs_out) *pairs_out = 0; return; }
    float cell = r; int dim[3];
    for (;;) {
        double cells = 1; for (int k = 0; k < 3; k++) { dim[k] = (int)floorf((hi[k] - lo[k]) / cell) + 1; cells *= dim[k]; }
        if (cells <= 64e6) break;
        cell *= 2.0f;
    }
    size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
    uint32_t *start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t));
    uint32_t *items = (uint32_t *)malloc(sizeof(uint32_t) * nph);
#define CELL_OF(P, k) ((int)floorf(((P)[k] - lo[k]) / cell))
    for (uint32_t i = 0; i < num_records; i++) if (records[i].flags & EVO_USABLE_PHOTON) {
        size_t c = ((size_t)CELL_OF(records[i].pos, 2) * dim[1] + CELL_OF(records[i].pos, 1)) * dim[0] + CELL_OF(records[i].pos, 0);
        start[c + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) start[c + 1] += start[c];
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * ncell); memcpy(cur, start, sizeof(uint32_t) * ncell);
    for (uint32_t i = 0; i < num_records; i++) if (records[i].flags & EVO_USABLE_PHOTON) {
        size_t c = ((size_t)CELL_OF(records[i].pos, 2) * dim[1] + CELL_OF(records[i].pos, 1)) * dim[0] + CELL_OF(records[i].pos, 0);
        items[cur[c]++] = i;   /* ascending record index inside each cell */
    }
    free(cur);
    int reach = (int)ceilf(r / cell);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : pairs) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        uint32_t cand[4096];
        for (int32_t x = 0; x < W; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            const float *X = g_pos + p;
            int c0[3], c1[3], skip = 0;
            for (int k = 0; k < 3; k++) {
                int c = (int)floorf((X[k] - lo[k]) / cell);
                c0[k] = c - reach; c1[k] = c + reach;
                if (c1[k] < 0 || c0[k] >= dim[k]) skip = 1;
                if (c0[k] < 0) c0[k] = 0;
                if (c1[k] >= dim[k]) c1[k] = dim[k] - 1;
            }
            if (skip) continue;
            v3 sum = V3(0, 0, 0);
            uint32_t nc = 0; uint32_t *cp = cand; uint32_t cap = 4096;
            for (int cz = c0[2]; cz <= c1[2]; cz++) for (int cy = c0[1]; cy <= c1[1]; cy++) for (int cx = c0[0]; cx <= c1[0]; cx++) {
                size_t c = ((size_t)cz * dim[1] + cy) * dim[0] + cx;
                for (uint32_t j = start[c]; j < start[c + 1]; j++) {
                    if (nc == cap) { uint32_t *np_ = (uint32_t *)malloc(sizeof(uint32_t) * cap * 2); memcpy(np_, cp, sizeof(uint32_t) * nc); if (cp != cand) free(cp); cp = np_; cap *= 2; }
                    cp[nc++] = items[j];
                }
            }
            /* ascending record index = fixed accumulation order */
            for (uint32_t a = 1; a < nc; a++) { uint32_t v = cp[a]; uint32_t b = a; while (b > 0 && cp[b - 1] > v) { cp[b] = cp[b - 1]; b--; } cp[b] = v; }
            for (uint32_t a = 0; a < nc; a++) {
                uint32_t i = cp[a]; float c[3];
                /* prev = i - 1: previous vertex of the same path (record 0 of a path is never a photon) */
                if (i == 0) continue;
                int kept = evo_photon_frag(fp, &records[i], &records[i - 1], X, g_nrm + p, g_dif + p, g_phg + p, c);
                v3 dv = sub(ld3(records[i].pos), ld3(X));
                if (dot(dv, dv) <= r * r) pairs++;
                if (kept) sum = add(sum, V3(c[0], c[1], c[2]));
            }
            if (cp != cand) free(cp);
            out[p + 0] += sum.x; out[p + 1] += sum.y; out[p + 2] += sum.z;
        }
    }
#undef CELL_OF
    free(start); free(items);
    if (pairs_out) *pairs_out = pairs;
}

/* ------------------------------------------------------------------ the reference's splat FOOTPRINT (test infrastructure) */
/* The reference does not test a sphere: it draws an instanced icosphere mesh of radius r around every photon with the depth test
 * on (LEQUAL against the deferred pass's depth, no depth writes) and WITHOUT face culling (rtcomphoton.h:653-655: glEnable
 * (GL_CULL_FACE) is commented out; :789-837; photonsplatinstanced.vert:28-33, .geom:16-32).  A pixel therefore receives the
 * fragment shader's value once per proxy FACE that the pixel's eye ray crosses in front of the visible surface -- zero times when
 * the surface point lies inside the sphere but outside (or in front of) the inscribed polyhedron, twice when both faces lie in front
 * of it -- and the shader itself still discards |X_p - P_i|^2 > r^2 (frag:152-154).  sphere/icosphere.obj is a Git-LFS stub of 2178
 * bytes: the size of Blender's default icosphere (2 subdivisions: 42 vertices on the unit sphere, 80 faces) exported without normals;
 * its orientation is unknown (poles on the y axis here, Blender's z-up to y-up export), which moves WHICH silhouette pixels are
 * missed, not how many.  evo_splat_photons_proxy applies that rule next to the ideal one (evo_splat_photons) so that
 * the difference -- DESIGN.md section 2, deviation (4) -- is a measured number. */
static void icosphere42(double v[42][3], int f[80][3]) {
    /* icosahedron: poles (0, +-1, 0), two rings of five at y = +-1/sqrt(5), the upper ring turned by 36 degrees */
    double base[12][3]; int nb = 0;
    const double h = 1.0 / sqrt(5.0), rr = 2.0 / sqrt(5.0), pi = 3.14159265358979323846;
    base[nb][0] = 0; base[nb][1] = -1; base[nb][2] = 0; nb++;
    for (int k = 0; k < 5; k++) { double a = 2.0 * pi * k / 5.0; base[nb][0] = rr * cos(a); base[nb][1] = -h; base[nb][2] = rr * sin(a); nb++; }
    for (int k = 0; k < 5; k++) { double a = 2.0 * pi * (k + 0.5) / 5.0; base[nb][0] = rr * cos(a); base[nb][1] = h; base[nb][2] = rr * sin(a); nb++; }
    base[nb][0] = 0; base[nb][1] = 1; base[nb][2] = 0; nb++;
    int bf[20][3], nf = 0;
    for (int k = 0; k < 5; k++) { bf[nf][0] = 0; bf[nf][1] = 1 + k; bf[nf][2] = 1 + (k + 1) % 5; nf++; }                       /* bottom cap */
    for (int k = 0; k < 5; k++) { bf[nf][0] = 1 + k; bf[nf][1] = 6 + k; bf[nf][2] = 1 + (k + 1) % 5; nf++; }                   /* belt */
    for (int k = 0; k < 5; k++) { bf[nf][0] = 6 + k; bf[nf][1] = 6 + (k + 1) % 5; bf[nf][2] = 1 + (k + 1) % 5; nf++; }
    for (int k = 0; k < 5; k++) { bf[nf][0] = 11; bf[nf][1] = 6 + (k + 1) % 5; bf[nf][2] = 6 + k; nf++; }                      /* top cap */
    int nv = 12; for (int i = 0; i < 12; i++) for (int k = 0; k < 3; k++) v[i][k] = base[i][k];
    int mid[12][12]; for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) mid[i][j] = -1;
    int out = 0;
    for (int t = 0; t < 20; t++) {
        int m[3];
        for (int e = 0; e < 3; e++) {
            int a = bf[t][e], b = bf[t][(e + 1) % 3];
            if (mid[a][b] < 0) {
                double q[3], l = 0; for (int k = 0; k < 3; k++) { q[k] = 0.5 * (base[a][k] + base[b][k]); l += q[k] * q[k]; }
                l = sqrt(l); for (int k = 0; k < 3; k++) v[nv][k] = q[k] / l;
                mid[a][b] = mid[b][a] = nv++;
            }
            m[e] = mid[a][b];
        }
        int tri[4][3] = { { bf[t][0], m[0], m[2] }, { m[0], bf[t][1], m[1] }, { m[2], m[1], bf[t][2] }, { m[0], m[1], m[2] } };
        for (int q = 0; q < 4; q++) { for (int k = 0; k < 3; k++) f[out][k] = tri[q][k]; out++; }
    }
}
/* the generated mesh as floats, for callers that hand the same mesh to the product (evplp_set_splat_proxy) */
void evo_icosphere42(float *verts, int32_t *tris) {
    double v[42][3]; int f[80][3]; icosphere42(v, f);
    for (int i = 0; i < 42; i++) for (int k = 0; k < 3; k++) verts[3 * i + k] = (float)v[i][k];
    for (int t = 0; t < 80; t++) for (int k = 0; k < 3; k++) tris[3 * t + k] = f[t][k];
}
/* faces of the proxy mesh (vertices in units of the radius r around the centre c) crossed by the ray e + t d with t in [tnear, tfar]:
 * one ray / triangle test per face, in double precision (a ray through an edge or a vertex of the proxy -- measure zero -- may count a
 * face twice, the rasteriser's fill rule would not) */
int evo_proxy_faces_in_front(const float *mv, const int32_t *mf, int32_t nf, const float c[3], float r, const double e[3], const double d[3], double tnear, double tfar) {
    int n = 0;
    for (int t = 0; t < nf; t++) {
        double p0[3], p1[3], p2[3];
        for (int k = 0; k < 3; k++) { p0[k] = c[k] + (double)r * mv[3 * mf[3 * t] + k]; p1[k] = c[k] + (double)r * mv[3 * mf[3 * t + 1] + k]; p2[k] = c[k] + (double)r * mv[3 * mf[3 * t + 2] + k]; }
        double e1[3], e2[3], pv[3], tv[3], qv[3];
        for (int k = 0; k < 3; k++) { e1[k] = p1[k] - p0[k]; e2[k] = p2[k] - p0[k]; tv[k] = e[k] - p0[k]; }
        pv[0] = d[1] * e2[2] - d[2] * e2[1]; pv[1] = d[2] * e2[0] - d[0] * e2[2]; pv[2] = d[0] * e2[1] - d[1] * e2[0];
        const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
        if (det == 0.0) continue;
        const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
        if (u < 0.0 || u > 1.0) continue;
        qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
        const double w = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) / det;
        if (w < 0.0 || u + w > 1.0) continue;
        const double tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
        if (tt >= tnear && tt <= tfar) n++;
    }
    return n;
}
/* Both footprints over the same pixels and photons: out_ideal as evo_splat_photons, out_proxy with the reference's coverage count for
 * the proxy mesh (mesh_verts: float3 per vertex in units of the radius, mesh_tris: int3 per face).
 * stats[0] = (photon, pixel) pairs inside the radius, [1] = of those with no proxy face in front (missed by the reference),
 * [2] = with two (counted twice by the reference), [3] = proxy fragments in total. */
void evo_splat_photons_proxy_mesh(const evo_frame_params *fp, const evo_camera *cam, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                                  const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                                  const evo_record *records, uint32_t num_records, const float *mesh_verts, const int32_t *mesh_tris, int32_t mesh_ntris,
                                  float *out_ideal, float *out_proxy, uint64_t stats[4]) {
    const cam_basis cb = cam_make(cam);
    const float r = fp->photon_radius;
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    /* (brute force over the photons per pixel row would be too slow: a uniform grid as in evo_splat_photons) */
    uint32_t nph = 0; float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    for (uint32_t i = 0; i < num_records; i++) if (records[i].flags & EVO_USABLE_PHOTON) { nph++; for (int k = 0; k < 3; k++) { lo[k] = minf(lo[k], records[i].pos[k]); hi[k] = maxf(hi[k], records[i].pos[k]); } }
    if (!nph || !(r > 0.0f)) { if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0; return; }
    float cell = r; int dim[3];
    for (;;) { double cells = 1; for (int k = 0; k < 3; k++) { dim[k] = (int)floorf((hi[k] - lo[k]) / cell) + 1; cells *= dim[k]; } if (cells <= 64e6) break; cell *= 2.0f; }
    size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
    uint32_t *start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t)), *items = (uint32_t *)malloc(sizeof(uint32_t) * nph);
#define CELL_OF(P, k) ((int)floorf(((P)[k] - lo[k]) / cell))
    for (uint32_t i = 0; i < num_records; i++) if (records[i].flags & EVO_USABLE_PHOTON) start[((size_t)CELL_OF(records[i].pos, 2) * dim[1] + CELL_OF(records[i].pos, 1)) * dim[0] + CELL_OF(records[i].pos, 0) + 1]++;
    for (size_t c = 0; c < ncell; c++) start[c + 1] += start[c];
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * ncell); memcpy(cur, start, sizeof(uint32_t) * ncell);
    for (uint32_t i = 0; i < num_records; i++) if (records[i].flags & EVO_USABLE_PHOTON) items[cur[((size_t)CELL_OF(records[i].pos, 2) * dim[1] + CELL_OF(records[i].pos, 1)) * dim[0] + CELL_OF(records[i].pos, 0)]++] = i;
    free(cur);
    const int reach = (int)ceilf(r / cell);
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : s0, s1, s2, s3) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        uint32_t cand[4096];
        for (int32_t x = 0; x < W; x++) {
            const size_t p = ((size_t)y * W + x) * 4;
            const float *X = g_pos + p;
            int c0[3], c1[3], skip = 0;
            for (int k = 0; k < 3; k++) { int c = (int)floorf((X[k] - lo[k]) / cell); c0[k] = c - reach; c1[k] = c + reach; if (c1[k] < 0 || c0[k] >= dim[k]) skip = 1; if (c0[k] < 0) c0[k] = 0; if (c1[k] >= dim[k]) c1[k] = dim[k] - 1; }
            if (skip) continue;
            /* the pixel's eye ray under the jittered matrix (the one the deferred pass and the splat draw share, rtcomphoton.h:951, 959, 982) */
            const float cx = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f, cy = ((float)y + 0.5f) / (float)H * 2.0f - 1.0f;
            const v3 dj = cam_dir(&cb, cx - fp->jitter[0], cy - fp->jitter[1]);
            const double e[3] = { cb.eye.x, cb.eye.y, cb.eye.z }, d[3] = { dj.x, dj.y, dj.z };
            /* view depth of the visible surface along that ray: the direction has camera-space z = -1 */
            const double tsurf = ((double)X[0] - e[0]) * cb.f.x + ((double)X[1] - e[1]) * cb.f.y + ((double)X[2] - e[2]) * cb.f.z;
            /* candidates in ascending record order: the accumulation order of evo_splat_photons (and of a deterministic product context) */
            uint32_t nc = 0; uint32_t *cp = cand; uint32_t cap = 4096;
            for (int cz = c0[2]; cz <= c1[2]; cz++) for (int cyy = c0[1]; cyy <= c1[1]; cyy++) for (int cxx = c0[0]; cxx <= c1[0]; cxx++) {
                const size_t c = ((size_t)cz * dim[1] + cyy) * dim[0] + cxx;
                for (uint32_t j = start[c]; j < start[c + 1]; j++) {
                    if (nc == cap) { uint32_t *np_ = (uint32_t *)malloc(sizeof(uint32_t) * cap * 2); memcpy(np_, cp, sizeof(uint32_t) * nc); if (cp != cand) free(cp); cp = np_; cap *= 2; }
                    cp[nc++] = items[j];
                }
            }
            for (uint32_t a = 1; a < nc; a++) { uint32_t v = cp[a]; uint32_t b = a; while (b > 0 && cp[b - 1] > v) { cp[b] = cp[b - 1]; b--; } cp[b] = v; }
            v3 si = V3(0, 0, 0), sp = V3(0, 0, 0);
            for (uint32_t a = 0; a < nc; a++) {
                const uint32_t i = cp[a]; float col[3];
                if (i == 0) continue;
                const v3 dv = sub(ld3(records[i].pos), ld3(X));
                if (!(dot(dv, dv) <= r * r)) continue;
                s0++;
                const int kept = evo_photon_frag(fp, &records[i], &records[i - 1], X, g_nrm + p, g_dif + p, g_phg + p, col);
                const int faces = evo_proxy_faces_in_front(mesh_verts, mesh_tris, mesh_ntris, records[i].pos, r, e, d, 0.1, tsurf * (1.0 + 1e-7));
                if (faces == 0) s1++; if (faces >= 2) s2++; s3 += (uint64_t)faces;
                if (kept) {
                    si = add(si, V3(col[0], col[1], col[2]));
                    for (int f = 0; f < faces; f++) sp = add(sp, V3(col[0], col[1], col[2]));     /* one blend per fragment */
                }
            }
            if (cp != cand) free(cp);
            out_ideal[p + 0] += si.x; out_ideal[p + 1] += si.y; out_ideal[p + 2] += si.z;
            out_proxy[p + 0] += sp.x; out_proxy[p + 1] += sp.y; out_proxy[p + 2] += sp.z;
        }
    }
#undef CELL_OF
    free(start); free(items);
    if (stats) { stats[0] = s0; stats[1] = s1; stats[2] = s2; stats[3] = s3; }
}
/* ... with the generated 42-vertex / 80-face icosphere */
void evo_splat_photons_proxy(const evo_frame_params *fp, const evo_camera *cam, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                             const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                             const evo_record *records, uint32_t num_records, float *out_ideal, float *out_proxy, uint64_t stats[4]) {
    float mv[42 * 3]; int32_t mf[80 * 3];
    evo_icosphere42(mv, mf);
    evo_splat_photons_proxy_mesh(fp, cam, W, H, row_begin, row_end, g_pos, g_nrm, g_dif, g_phg, records, num_records, mv, mf, 80, out_ideal, out_proxy, stats);
}

/* ------------------------------------------------------------------ resolve */
/* final.frag:19-35; the saved images (rtcomphoton.h:1121-1132) use mask_emitter = 0 */
void evo_resolve(int32_t W, int32_t H, const float *vpl, const float *pm, const float *light,
                 float vs, float ps, float ls, int mask_emitter, int gamma, float *out_rgb) {
    for (size_t i = 0; i < (size_t)W * H; i++) {
        for (int k = 0; k < 3; k++) {
            float v = vpl ? vpl[4 * i + k] * vs : 0.f, q = pm ? pm[4 * i + k] * ps : 0.f, l = light ? light[4 * i + k] * ls : 0.f;
            float lx = light ? light[4 * i] * ls : 0.f;
            float step = mask_emitter ? ((0.0f < lx) ? 0.0f : 1.0f) : 1.0f;
            float sum = step * (v + q) + l;
            out_rgb[3 * i + k] = gamma ? powf(sum, 1.0f / 2.2f) : sum;
        }
    }
}

/* rtcomphoton.h:1033-1063 (called after numIterations++) */
void evo_progressive_step(int32_t n, float alpha, float clamp_start, uint32_t n_vpl, uint32_t n_light,
                          float *radius, float *clamp, float *pdf_mc, int force_vsl, float *vsl_radius, float *vsl_inv_pi_r2) {
    float ratio = ((float)n + alpha) / (float)(n + 1);
    *radius *= sqrtf(ratio);
    *clamp = clamp_start * powf((float)n, alpha);
    *pdf_mc = (float)n_vpl / (float)n_light * EVO_INV_PI / (*radius * *radius);
    if (force_vsl) {
        *vsl_radius *= sqrtf(ratio);
        if (*vsl_radius <= 0.008f) *vsl_radius = maxf(*vsl_radius, 0.008f);
        *vsl_inv_pi_r2 = EVO_INV_PI / (*vsl_radius * *vsl_radius);
    }
}

/* ------------------------------------------------------- path tracer (baseline) */
/* pathtracing.cu:53-56 */
static inline float russian_prob_pt(v3 t) { return maxf(maxf(t.x, 0.98f), maxf(t.y, t.z)); }
/* pathtracing.cu:93-97 */
static inline float pdf_w2a(v3 n2, v3 v12) { v3 nv = normalize(v12); return maxf(-dot(n2, nv), 0.f) / dot(v12, v12); }

/* pathtracing.cu:240-348 pathTraceSimple + :112-228 closest hit.  Returns radiance for one camera path. */
static v3 path_trace_simple(const evo_scene *s, v3 cam, v3 first_pos, v3 first_n, v3 rd1, v3 rs1, float e1,
                            uint32_t max_bounces, evo_rng *rng) {
    v3 camera_vec = normalize(sub(first_pos, cam));
    v3 result = V3(0, 0, 0);
    v3 position = first_pos, normal = first_n;
    v3 prd_pos = first_pos, att = V3(1, 1, 1), dir = V3(0, 0, 0); float brdf_pdf_w = 0.f;
    float lw = s->light_intensity[3];
    {
        float lpdf; v3 lp, ln;
        v3 lval = light_sample(s, &lp, &ln, &lpdf, rng);
        v3 to_light = sub(lp, position);
        v3 tln = normalize(to_light);
        float o[3], d[3]; st3(o, lp); st3(d, neg(to_light));
        int hit = evo_occluded(s, o, d, 0.0001f, 1.0f - 0.0001f);
        float ml = max_color(rd1), mp = max_color(rs1);
        float psel = ml / (mp + ml);
        if (ml + mp <= 0.000001f) return V3(0, 0, 0);
        float choose = minf(evo_rng_uniform(rng), 0.999999f);
        if (choose < psel) {
            if (!hit) {
                float bpdf = lambert_pdf_a(normal, ln, to_light);
                float w = lpdf / (lpdf + bpdf);
                v3 le = muls(rd1, EVO_INV_PI); /* LambertEval */
                v3 c = muls(divs(muls(mulv(muls(lval, w), le), geometry_term(normal, ln, to_light)), psel), phong_eval_f(ln, neg(tln), ln, lw));
                result = add(result, c);
            }
            v3 wgt = lambert_sample(&dir, &brdf_pdf_w, normal, rd1, rng);
            att = mulv(att, divs(wgt, psel));
        } else {
            if (!hit) {
                float bpdf = phong_pdf_a(normal, ln, to_light, neg(camera_vec), rs1, e1);
                float w = lpdf / (lpdf + bpdf);
                v3 pe = phong_eval(neg(camera_vec), tln, normal, rs1, e1);
                v3 c = muls(divs(muls(mulv(muls(lval, w), pe), geometry_term(normal, ln, to_light)), 1.0f - psel), phong_eval_f(ln, neg(tln), ln, lw));
                result = add(result, c);
            }
            v3 wgt = phong_sample(&dir, &brdf_pdf_w, neg(camera_vec), normal, rs1, e1, rng);
            att = mulv(att, divs(wgt, 1.0f - psel));
        }
    }
    for (uint32_t i = 0; i < max_bounces; i++) {
        int done = (i == max_bounces - 1);
        v3 res = V3(0, 0, 0);
        float o[3], d[3], t, b, g; st3(o, prd_pos); st3(d, dir);
        int32_t tri = evo_closest(s, o, d, 0.00001f, 3.0e38f, 0, &t, &b, &g);
        if (tri < 0) break; /* miss: no miss program; treat as terminated */
        const float *v = s->verts + 9 * (size_t)tri;
        v3 p0 = ld3(v), p1 = ld3(v + 3), p2 = ld3(v + 6);
        v3 gn = normalize(cross(sub(p0, p2), sub(p1, p0)));
        v3 ffn = faceforward(normalize(gn), neg(dir), normalize(gn));
        v3 npos = add(prd_pos, muls(dir, t));
        const evo_material *m = &s->mats[s->mat[tri]];
        if (dot(gn, dir) > 0.f) break;                                    /* :125-130 */
        if (m->light[0] > 0.01f) {                                        /* :133-148 */
            float bpa = brdf_pdf_w * pdf_w2a(ffn, sub(npos, prd_pos));
            float lpa = 1.f / s->light_area;
            float w = bpa / (bpa + lpa);
            v3 li = V3(m->light[0], m->light[1], m->light[2]);
            res = mulv(muls(muls(att, w), phong_eval_f(gn, normalize(sub(prd_pos, npos)), gn, m->light[3])), li);
            result = add(result, res);
            break;
        }
        if (done) break;                                                  /* :151 */
        float lpdf; v3 lp, ln;
        v3 lval = light_sample(s, &lp, &ln, &lpdf, rng);
        v3 to_light = sub(lp, npos);
        v3 tln = normalize(to_light);
        st3(o, lp); st3(d, neg(to_light));
        int hit = evo_occluded(s, o, d, 0.00001f, 0.99999f);
        v3 kd, ks; float ns; material_at(s, tri, b, g, &kd, &ks, &ns);
        float ml = max_color(kd), mp = max_color(ks);
        if (ml + mp <= 0.000001f) break;                                  /* :172-173 */
        float psel = ml / (mp + ml);
        float choose = minf(evo_rng_uniform(rng), 0.999999f);
        v3 back = normalize(sub(prd_pos, npos));
        if (choose < psel) {
            if (!hit) {
                float bpdf = lambert_pdf_a(ffn, ln, to_light);
                float w = lpdf / (lpdf + bpdf);
                v3 le = muls(kd, EVO_INV_PI);
                res = muls(divs(mulv(muls(mulv(muls(lval, w), le), geometry_term(ffn, ln, to_light)), att), psel), phong_eval_f(ln, neg(tln), ln, lw));
            }
            v3 wgt = lambert_sample(&dir, &brdf_pdf_w, gn, kd, rng);       /* geometryNormal :197 */
            att = mulv(att, divs(wgt, psel));
        } else {
            if (!hit) {
                float bpdf = phong_pdf_a(ffn, ln, to_light, back, ks, ns);
                float w = lpdf / (lpdf + bpdf);
                v3 pe = phong_eval(tln, back, ffn, ks, ns);
                res = muls(divs(mulv(muls(mulv(muls(lval, w), pe), geometry_term(ffn, ln, to_light)), att), 1.0f - psel), phong_eval_f(ln, neg(tln), ln, lw));
            }
            v3 wgt = phong_sample(&dir, &brdf_pdf_w, back, gn, ks, ns, rng);
            att = mulv(att, divs(wgt, 1.0f - psel));
        }
        result = add(result, res);
        float russian = russian_prob_pt(att);                             /* :219-225 */
        if (evo_rng_uniform(rng) >= russian) break;
        prd_pos = npos;
        att = divs(att, russian);
    }
    return result;
}
/* pathtracing.cu:350-377 splatColor */
uint64_t evo_path_trace(const evo_scene *s, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces,
                        int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                        const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                        float *out, int do_accumulate) {
    (void)H;
    uint64_t paths = 0;
    v3 cam = ld3(camera_pos);
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : paths) num_threads(evo_get_threads())
    for (int32_t y = row_begin; y < row_end; y++) {
        for (int32_t x = 0; x < W; x++) {
            size_t p = ((size_t)y * W + x) * 4;
            if (g_pos[p + 3] == 0.0f) continue;
            evo_rng rng; evo_rng_init(&rng, (uint32_t)y * (uint32_t)W + (uint32_t)x, rng_seed, 0x50540000u);
            v3 r = path_trace_simple(s, cam, ld3(g_pos + p), ld3(g_nrm + p), ld3(g_dif + p), ld3(g_phg + p), g_phg[p + 3], max_bounces, &rng);
            paths++;
            if (do_accumulate) { out[p] += r.x; out[p + 1] += r.y; out[p + 2] += r.z; }
            else { out[p] = r.x; out[p + 1] = r.y; out[p + 2] = r.z; out[p + 3] = 0.f; }
        }
    }
    return paths;
}

/* ------------------------------------------------------------ output surface */
/* common/floatimage/floatimage.cpp:178-199 SavePFM: "PF\nW H\n-1\n", rows bottom-to-top of a
 * top-down image, RGB fp32 little-endian */
int evo_write_pfm(const char *path, int32_t W, int32_t H, const float *rgb) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "PF\n%d %d\n-1\n", W, H);
    for (int32_t i = 0; i < H; i++) fwrite(rgb + (size_t)W * (H - i - 1) * 3, sizeof(float), (size_t)W * 3, f);
    fclose(f);
    return 0;
}
/* floatimage.cpp:241-258 SavePNG pixel conversion: pow(c, 1/2.2) in fp32, * 255.99 and min in
 * double, truncated to a byte */
void evo_png_bytes(int32_t n, const float *rgb, uint8_t *out) {
    for (int32_t i = 0; i < n; i++) {
        float p = powf(rgb[i], (float)(1 / 2.2));
        double q = (double)p * 255.99; if (q > 255.0) q = 255.0;
        p = (float)q;
        out[i] = (uint8_t)(int)p;
    }
}
/* floatimage.cpp:64-84 / 86-112 (Float = float accumulators) */
double evo_mse(int32_t npix, const float *a, const float *ref) {
    float result = 0;
    for (int32_t i = 0; i < npix; i++) { v3 d = sub(ld3(a + 3 * i), ld3(ref + 3 * i)); result += dot(d, d); }
    return result / (float)npix;
}
double evo_rel_mse(int32_t npix, const float *a, const float *ref) {
    float result = 0;
    for (int32_t i = 0; i < npix; i++) {
        v3 r = ld3(ref + 3 * i), d = sub(ld3(a + 3 * i), r);
        result += dot(d, d) / (dot(r, r) + 0.001f);
    }
    return result / (float)npix;
}
