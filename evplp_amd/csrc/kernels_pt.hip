// Unidirectional path tracer with next-event estimation: the reference's own ground-truth technique
// ("pt" block of the scene JSON, rt/rtpt/rtpt2.h) and the generator of converged images for the
// convergence tests.
//   path_trace_kernel  <- splatColor + pathTraceSimple (rt/pathtracing.cu:350-377, 240-348) with the
//                         closest-hit program inlined as the bounce loop (rt/pathtracing.cu:112-228)
// One lane per pixel of an 8x8 tile; after the first bounce the rays of a wave are incoherent, so both the
// closest-hit and the shadow walk are per-lane with the [entry][lane] LDS stack of the feeder kernels.
#include "device_common.hpp"
#include "kernels.h"

// per-lane walks: binary nodes (EVPLP_PT_WIDE 0) or four-wide nodes (1).  1024^2, furnished scene: binary at 6 waves per SIMD 361 M
// camera paths/s, four-wide at 4 waves (no spills, 1.5x the LDS stack) 362: no gain here, the path tracer has waves to switch to
// while a node is in flight; the four-wide nodes pay off in light tracing (kernels_trace.hip)
#ifndef EVPLP_PT_WIDE
#define EVPLP_PT_WIDE 0
#endif
#if EVPLP_PT_WIDE
#define PT_OCCLUDED occluded_lane4
#define PT_CLOSEST closest_lane4
#ifndef EVPLP_PT_SPEC
#define EVPLP_PT_SPEC 0
#endif
#define PT_SPEC_ARGS , 0, EVPLP_PT_SPEC
#define PT_OCC_SPEC_ARGS
#else
#define PT_OCCLUDED occluded_lane
#define PT_CLOSEST closest_lane
#ifndef EVPLP_PT_SPEC
#define EVPLP_PT_SPEC 0            // speculative while-while (device_common.hpp closest_lane / occluded_lane): 688 against 687 M paths/s -- the path tracer has waves to switch to; off
#endif
#define PT_SPEC_ARGS , EVPLP_PT_SPEC
#define PT_OCC_SPEC_ARGS , EVPLP_PT_SPEC
#endif

namespace evplp {

// pathtracing.cu:53-56
EV_DEV float russian_prob_pt(V3 t) { return fmaxf(fmaxf(t.x, 0.98f), fmaxf(t.y, t.z)); }
// pathtracing.cu:93-97
EV_DEV float pdf_w2a(V3 n2, V3 v12) { V3 nv = normalize(v12); return fmaxf(-dot(n2, nv), 0.f) / dot(v12, v12); }

// one camera path continued from the G-buffer texel p of pixel (x, y); returns the rays it traced
EV_DEV unsigned long long path_trace_pixel(const PathTraceArgs &a, int x, int y, size_t p, float4 gp, int32_t *stack) {

    const V3 first_pos = v3(gp), first_n = v3(a.g_nrm[p]), rd1 = v3(a.g_dif[p]);
    const float4 ph = a.g_phg[p];
    const V3 rs1 = v3(ph); const float e1 = ph.w;
    const V3 cam = v3(a.camera_pos);
    Rng rng; rng_init(rng, (uint32_t)y * (uint32_t)a.st.W + (uint32_t)x, a.rng_seed, 0x50540000u);   // (:369-370)

    unsigned long long rays = 0;
    const V3 camera_vec = normalize(first_pos - cam);
    V3 result = v3(0.f, 0.f, 0.f);
    V3 prd_pos = first_pos, att = v3(1.f, 1.f, 1.f), dir = v3(0.f, 0.f, 0.f);
    float brdf_pdf_w = 0.f;
    const float lw = a.sc.light_intensity[3];
    bool alive = true;
    {   // first vertex from the G-buffer (:246-300)
        float lpdf; V3 lp, ln;
        V3 lval = light_sample(a.sc, lp, ln, lpdf, rng);
        V3 to_light = lp - first_pos;
        V3 tln = normalize(to_light);
        bool hit = PT_OCCLUDED<64 PT_OCC_SPEC_ARGS>(a.sc, lp, -to_light, 0.0001f, 1.0f - 0.0001f, stack); rays++;
        float ml = max_color(rd1), mp = max_color(rs1);
        float psel = ml / (mp + ml);
        if (ml + mp <= 0.000001f) alive = false;
        else {
            float choose = fminf(rng_uniform(rng), 0.999999f);
            if (choose < psel) {
                if (!hit) {
                    float bpdf = lambert_pdf_a(first_n, ln, to_light);
                    float w = lpdf / (lpdf + bpdf);
                    V3 le = rd1 * EV_INV_PI;
                    result = result + (((lval * w) * le) * geometry_term(first_n, ln, to_light)) / psel * phong_eval_f(ln, -tln, ln, lw);
                }
                V3 wgt = lambert_sample(dir, brdf_pdf_w, first_n, rd1, rng);
                att = att * (wgt / psel);
            } else {
                if (!hit) {
                    float bpdf = phong_pdf_a(first_n, ln, to_light, -camera_vec, rs1, e1);
                    float w = lpdf / (lpdf + bpdf);
                    V3 pe = phong_eval(-camera_vec, tln, first_n, rs1, e1);
                    result = result + (((lval * w) * pe) * geometry_term(first_n, ln, to_light)) / (1.0f - psel) * phong_eval_f(ln, -tln, ln, lw);
                }
                V3 wgt = phong_sample(dir, brdf_pdf_w, -camera_vec, first_n, rs1, e1, rng);
                att = att * (wgt / (1.0f - psel));
            }
        }
    }
    for (uint32_t i = 0; alive && i < a.max_bounces; i++) {
        const bool done = (i == a.max_bounces - 1);
        float t, b, g;
        int32_t tri = PT_CLOSEST<64 PT_SPEC_ARGS>(a.sc, prd_pos, dir, 0.00001f, 3.0e38f, 0, t, b, g, stack); rays++;
        if (tri < 0) break;                                               // no miss program: the path ends
        const TriAttr &ta = a.sc.attrs[tri];
        V3 p0 = v3(ta.v), p1 = v3(ta.v + 3), p2 = v3(ta.v + 6);
        V3 gn = normalize(cross(p0 - p2, p1 - p0));                       // triangleintersect.cu:31
        V3 wgn = normalize(gn);
        V3 ffn = faceforward(wgn, -dir, wgn);
        V3 npos = prd_pos + dir * t;
        const Material &m = a.sc.materials[ta.material];
        if (dot(gn, dir) > 0.f) break;                                    // back face (:125-130)
        if (m.light[0] > 0.01f) {                                         // emitter reached by BRDF sampling (:133-148)
            float bpa = brdf_pdf_w * pdf_w2a(ffn, npos - prd_pos);
            float lpa = 1.f / a.sc.light_area;
            float w = bpa / (bpa + lpa);
            V3 li = v3(m.light[0], m.light[1], m.light[2]);
            result = result + ((att * w) * phong_eval_f(gn, normalize(prd_pos - npos), gn, m.light[3])) * li;
            break;
        }
        if (done) break;                                                  // (:151)
        float lpdf; V3 lp, ln;
        V3 lval = light_sample(a.sc, lp, ln, lpdf, rng);
        V3 to_light = lp - npos;
        V3 tln = normalize(to_light);
        bool hit = PT_OCCLUDED<64 PT_OCC_SPEC_ARGS>(a.sc, lp, -to_light, 0.00001f, 0.99999f, stack); rays++;
        V3 kd, ks; float ns;
        material_at(a.sc, ta, b, g, kd, ks, ns);
        float ml = max_color(kd), mp = max_color(ks);
        if (ml + mp <= 0.000001f) break;                                  // (:172-173)
        float psel = ml / (mp + ml);
        float choose = fminf(rng_uniform(rng), 0.999999f);
        V3 back = normalize(prd_pos - npos);
        V3 res = v3(0.f, 0.f, 0.f);
        if (choose < psel) {
            if (!hit) {
                float bpdf = lambert_pdf_a(ffn, ln, to_light);
                float w = lpdf / (lpdf + bpdf);
                V3 le = kd * EV_INV_PI;
                res = ((((lval * w) * le) * geometry_term(ffn, ln, to_light)) * att) / psel * phong_eval_f(ln, -tln, ln, lw);
            }
            V3 wgt = lambert_sample(dir, brdf_pdf_w, gn, kd, rng);        // geometric normal (:197)
            att = att * (wgt / psel);
        } else {
            if (!hit) {
                float bpdf = phong_pdf_a(ffn, ln, to_light, back, ks, ns);
                float w = lpdf / (lpdf + bpdf);
                V3 pe = phong_eval(tln, back, ffn, ks, ns);
                res = ((((lval * w) * pe) * geometry_term(ffn, ln, to_light)) * att) / (1.0f - psel) * phong_eval_f(ln, -tln, ln, lw);
            }
            V3 wgt = phong_sample(dir, brdf_pdf_w, back, gn, ks, ns, rng);
            att = att * (wgt / (1.0f - psel));
        }
        result = result + res;
        float russian = russian_prob_pt(att);                             // (:219-225)
        if (rng_uniform(rng) >= russian) break;
        prd_pos = npos;
        att = att / russian;
    }
    float4 o = a.do_accumulate ? a.out[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    a.out[p] = make_float4(o.x + result.x, o.y + result.y, o.z + result.z, o.w);
    return rays;
}

#ifndef EVPLP_PT_WAVES
#define EVPLP_PT_WAVES 4   // 128 VGPRs, zero scratch (126 needed): 1.61 ms per sample per pixel at 1024^2 against 1.59 ms at 6 waves with 65 spilled registers
#endif
__global__ __launch_bounds__(64, EVPLP_PT_WAVES) void path_trace_kernel(PathTraceArgs a) {
    extern __shared__ int32_t lds_stack[];   // [bvh_depth + 2][64 lanes]
    const int lane = threadIdx.x;
    const int tiles_x = (a.st.W + 7) >> 3;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3);
    const int y = a.st.global_row(min(ly, a.st.local_rows - 1));
    const bool in_image = x < a.st.W && ly < a.st.local_rows && y < a.st.H;
    const size_t p = (size_t)min(ly, a.st.local_rows - 1) * a.st.W + min(x, a.st.W - 1);
    const float4 gp = a.g_pos[p];
    const bool valid = in_image && gp.w != 0.0f;                          // stencil (:357)
    unsigned long long rays = 0, paths = valid ? 1ull : 0ull;
    if (valid) rays = path_trace_pixel(a, x, y, p, gp, lds_stack + lane);
    // statistics: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { rays += __shfl_xor(rays, off); paths += __shfl_xor(paths, off); }
    if (lane == 0 && a.counters && paths) { atomicAdd(&a.counters->rays, rays); atomicAdd(&a.counters->pairs, paths); }
}

void launch_path_trace(const PathTraceArgs &a, hipStream_t s) {
    int tiles_x = (a.st.W + 7) / 8, tiles_y = (a.st.local_rows + 7) / 8;
    if (tiles_x * tiles_y == 0) return;
    hipLaunchKernelGGL(path_trace_kernel, dim3(tiles_x * tiles_y), dim3(64), EVPLP_PT_WIDE ? lane_stack_bytes4(a.sc) : lane_stack_bytes(a.sc), s, a);
}

} // namespace evplp
