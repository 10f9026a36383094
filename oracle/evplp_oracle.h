/*
 * evplp_oracle.h -- CPU restatement of the evplp radiance-accumulation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / baseline.  The product path
 * (evplp_amd/csrc, libevplp_hip.so) never includes, links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the device arithmetic.  The reference
 * (jamornsriwasansak/evplp) ships no tests, golden vectors or images
 * (SURVEY.md section 4), its device code needs OptiX 4.1.1 / cuRAND / OpenGL
 * headers that are absent from this image, and its scene meshes are Git-LFS
 * stubs, so the reference cannot be run or compiled here.  This file restates
 * the reference algorithm function by function (each citing file:line under
 * /root/reference/reflectcuts).  What IS pinned against the reference itself:
 * the PFM/PNG byte streams and MSE/relMSE metrics (oracle/_ref, built from the
 * reference's own floatimage.cpp) and the camera matrices (vendored GLM); see
 * oracle/ref_pin.cpp and tests/test_oracle_pins.py.
 *
 * Third-party arithmetic restated from published definitions (dependency absent
 * from /root/reference): NVIDIA OptiX SDK 4.1.1 optixu/optixu_math_namespace.h
 * (normalize, reflect, faceforward, Onb, cosine_sample_hemisphere,
 * intersect_triangle_branchless), CUDA tex2D bilinear filtering.
 * cuRAND XORWOW streams are NOT reproduced (seeding tables are not public): the
 * build defines its own generators (PCG32 XSH-RR seeded by splitmix64; for the VSL
 * estimators xoroshiro64, one step per sample -- outputs in (0,1] like
 * curand_uniform), shared bit-for-bit by this oracle and the HIP path.
 */
#ifndef EVPLP_ORACLE_H
#define EVPLP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rt/rtcomphoton/rtphotonrecord.h:9-25 -- identical 96-byte layout */
enum { EVO_USABLE_VPL = 1, EVO_USABLE_PHOTON = 2, EVO_LAMBERT_ONLY = 4, EVO_PHONG_ONLY = 8 };

typedef struct evo_record {
    float pos[3];     uint32_t flags;
    float normal[3];  float p_select_lambert;
    float flux[3];    float pad1;
    float flux_dir[3]; float pad2;
    float rho_d[3];   float pad3;
    float rho_s[3];   float phong_exp;
} evo_record;

/* rt/rtcommon.h:278-308 (RtMaterial): three textures (constant = 1x1) + mLightIntensity */
typedef struct evo_material {
    float kd[3]; float ks[3]; float ns;
    float light[4];            /* mLightIntensity: zeros from the caller; set to (I*pi, w) for the light material by evo_scene_create */
    int32_t tex_kd, tex_ks, tex_ns; /* texture ids or -1 */
} evo_material;

typedef struct evo_texture { int32_t w, h; const float *rgba; } evo_texture;

/* rt/rtcommon.h:546-598 (RtStableCamera) */
typedef struct evo_camera {
    float origin[3]; float lookat[3]; float up[3];
    float fovy;   /* radians */
    float aspect;
} evo_camera;

/* rtcomphoton.h:895-930 -- the OptiX variables / GL uniforms of one iteration */
typedef struct evo_frame_params {
    float camera_pos[3];
    uint32_t mis_mode;           /* EMis rtcomphoton.h:64-72 */
    float pdf_mc;
    float clamping_value;
    float photon_radius;
    float vsl_radius;
    float vsl_inv_pi_radius2;
    uint32_t num_light_paths;
    uint32_t num_vpl_light_paths;
    uint32_t photons_per_path;   /* numMaxBounces + 1 */
    uint32_t do_accumulate;
    uint32_t rng_seed;           /* numIterations + rngOffset rtcomphoton.h:965 */
    float jitter[2];             /* NDC translation rtcomphoton.h:949 */
} evo_frame_params;

typedef struct evo_scene evo_scene;

/* Scene = flattened triangle soup of all meshes (rtcommon.h:816-819).  verts: 9 floats
 * per triangle (p0,p1,p2); uvs: 6 floats per triangle; mat: material index per
 * triangle.  The area-light mesh is the triangle range [light_first, light_first+light_count)
 * (rtcommon.h:772-798: exactly one light mesh, appended last). */
evo_scene *evo_scene_create(int32_t ntri, const float *verts, const float *uvs, const int32_t *mat,
                            int32_t nmat, const evo_material *mats,
                            int32_t ntex, const evo_texture *tex,
                            int32_t light_first, int32_t light_count,
                            const float light_intensity[4] /* UNSCALED (I, w) of the JSON; scaled by pi inside, rtcommon.h:780-782 */);
void evo_scene_destroy(evo_scene *s);
float evo_scene_light_area(const evo_scene *s);
float evo_scene_total_area(const evo_scene *s);             /* rtcommon.h:759-768 */
float evo_scene_bounding_sphere_radius(const evo_scene *s); /* rtcommon.h:805-814 */
void evo_set_threads(int n);
int evo_get_threads(void);

/* ---- ray queries (shared triangle test; see evo_tri_test) ---- */
/* restates optix::intersect_triangle_branchless as called from triangleintersect.cu:27 */
int evo_tri_test(const float p0[3], const float p1[3], const float p2[3],
                 const float o[3], const float d[3], float tmin, float tmax,
                 float *t, float *beta, float *gamma);
int evo_occluded(const evo_scene *s, const float o[3], const float d[3], float tmin, float tmax);
int evo_occluded_brute(const evo_scene *s, const float o[3], const float d[3], float tmin, float tmax);
/* filter: 0 = all triangles, 1 = skip light mesh, 2 = light mesh only. returns tri index or -1 */
int evo_closest(const evo_scene *s, const float o[3], const float d[3], float tmin, float tmax,
                int filter, float *t, float *beta, float *gamma);

/* ---- RNG (build-defined; see header comment) ---- */
typedef struct evo_rng { uint64_t state, inc; uint32_t s0, s1; int32_t vsl_draw; uint32_t reserved; } evo_rng;
void evo_rng_init(evo_rng *r, uint32_t index, uint32_t sequence, uint32_t substream);
/* the VSL estimators' own stream (xoroshiro64, one step per sample; evplp_oracle.c): evo_rng_uniform then serves the draws of the sample */
void evo_vsl_rng_init(evo_rng *r, uint32_t index, uint32_t sequence, uint32_t substream);
void evo_vsl_rng_step(evo_rng *r);
uint32_t evo_rng_u32(evo_rng *r);
float evo_rng_uniform(evo_rng *r); /* (0,1] like curand_uniform */

/* ---- BRDF helpers rt/rtmaterial.cuh (exported for unit tests) ---- */
/* Triangle::ComputeArea (shapes/trianglemesh.cpp:13-19) as the light CDF and totalArea use it (rtcommon.h:501-531, 759-768) */
float evo_tri_area(const float v9[9]);
/* the shared direction-sampling math (evplp_amd/csrc/ev_math.h), exported for the accuracy check against libm */
void evo_math_sincos(float x, float *s, float *c);
float evo_math_pow(float x, float y);
float evo_phong_eval_f(const float out[3], const float in[3], const float n[3], float e);
float evo_lambert_pdf_a(const float n1[3], const float n2[3], const float v12[3]);
float evo_phong_pdf_a(const float n1[3], const float n2[3], const float v12[3], const float in[3],
                      const float rho_s[3], float e);
float evo_phong_pdf_w(const float n1[3], const float v12[3], const float in[3], const float rho_s[3], float e);

/* ---- passes.  All images are row-major, y = 0 at the BOTTOM (GL / OptiX launch index
 * convention, final.frag:22-23), 4 floats per pixel unless stated. ---- */

/* deferred.* + light.*: G-buffer planes position(w=1)/normal/diffuse/phong(rgb,e) and the
 * light plane (rgb = unscaled intensity where the emitter is front-most, else 0).
 * Rows [row_begin,row_end) of the full W x H image are written at their global offset. */
void evo_primary(const evo_scene *s, const evo_camera *cam, int32_t W, int32_t H,
                 const float jitter[2], int32_t row_begin, int32_t row_end,
                 float *g_pos, float *g_nrm, float *g_dif, float *g_phg, float *g_light,
                 int32_t light_unoccluded /* 1: the emitter image is not depth-tested (cleareveryframe, rtcomphoton.h:989-994) */);

/* lighttracing.cu:192-250 + 113-182 */
void evo_trace_light_paths(const evo_scene *s, uint32_t rng_seed, uint32_t path_begin, uint32_t path_count,
                           uint32_t photons_per_path, evo_record *records /* whole buffer */);

/* lighttracing.cu:275-346 -- one pair, visibility supplied by caller (visible != 0) */
void evo_vpl_splat_pair(const evo_frame_params *fp, const float wi10[3], const float p1[3], const float n1[3],
                        const float rho_d[3], const float rho_s[3], float e,
                        const evo_record *rec, int visible, float out[3]);
/* lighttracing.cu:596-686 vslSplat for one pair, visibility supplied by the caller, RNG stream (index, sequence, substream).
 * TEST HOOK (tests/test_oracle_selfcheck.py): only = 0 the reference's MIS-combined estimator; 1 / 2 / 3 the cone / pixel-BRDF /
 * VSL-BRDF estimator alone with weight 1; samples_override > 0 replaces numSamples (:632). */
void evo_vsl_splat_pair(const evo_frame_params *fp, const float wi10[3], const float p1[3], const float n1[3],
                        const float rho_d[3], const float rho_s[3], float e, const evo_record *rec, int visible,
                        uint32_t rng_index, uint32_t rng_sequence, uint32_t rng_substream, int only, int samples_override, float out[3]);
/* lvclighttracing.cu:348-384: splatColor with a per-pixel random light-path window */
void evo_gather_lvc(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H,
                    int32_t row_begin, int32_t row_end,
                    const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                    const evo_record *records, float *out, uint64_t *pairs_out);
/* lighttracing.cu:348-379 */
void evo_gather_vpl(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H,
                    int32_t row_begin, int32_t row_end,
                    const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                    const evo_record *records, float *out /* RGBA accumulate */,
                    uint64_t *pairs_out /* optional: evaluated (pixel, usable record) pairs */);
/* counting twin: shadow rays traced / pairs left unoccluded over the given image rows (checks the product's statistics) */
void evo_gather_vpl_counts(const evo_scene *s, const evo_frame_params *fp, int32_t W, const int32_t *rows, int32_t nrows,
                           const float *g_pos, const float *g_nrm, const evo_record *records, uint64_t *rays_out, uint64_t *unoccluded_out);
/* lighttracing.cu:596-722 */
void evo_gather_vsl(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H,
                    int32_t row_begin, int32_t row_end,
                    const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                    const evo_record *records, float *out, uint64_t *pairs_out);
/* ... on the pixels [x_begin, x_end) of the rows only */
void evo_gather_vsl_window(const evo_scene *s, const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                           int32_t x_begin, int32_t x_end, const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                           const evo_record *records, float *out, uint64_t *pairs_out);
/* photonsplatinstanced.frag:146-240 for one (photon, shading point) -- returns 0 if discarded */
int evo_photon_frag(const evo_frame_params *fp, const evo_record *photon, const evo_record *prev,
                    const float x_pos[3], const float x_nrm[3], const float x_dif[3], const float x_phg[4],
                    float out[3]);
/* rtcomphoton.h:789-837 + shaders, ideal sphere semantics.
 * out: RGB in RGBA-strided buffer, additive. */
void evo_splat_photons(const evo_frame_params *fp, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                       const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                       const evo_record *records, uint32_t num_records, float *out, uint64_t *pairs_out);
/* the same pixels and photons under both footprints: the ideal sphere (out_ideal) and the reference's instanced icosphere proxy with
 * depth test and no face culling (out_proxy; rtcomphoton.h:632-655, 789-837, photonsplatinstanced.vert:28-33).  stats: pairs inside the
 * radius, of those missed by the proxy, counted twice by it, proxy fragments.  Test infrastructure: quantifies DESIGN.md deviation (4) */
void evo_icosphere42(float *verts42x3, int32_t *tris80x3);
/* faces of the proxy mesh (vertices in units of r around c) that the ray e + t d crosses with t in [tnear, tfar] */
int evo_proxy_faces_in_front(const float *mesh_verts, const int32_t *mesh_tris, int32_t mesh_ntris, const float c[3], float r, const double e[3], const double d[3], double tnear, double tfar);
/* ... for any proxy mesh (vertices in units of the radius; the product's evplp_set_splat_proxy takes the same arrays) */
void evo_splat_photons_proxy_mesh(const evo_frame_params *fp, const evo_camera *cam, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                                  const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                                  const evo_record *records, uint32_t num_records, const float *mesh_verts, const int32_t *mesh_tris, int32_t mesh_ntris,
                                  float *out_ideal, float *out_proxy, uint64_t stats[4]);
void evo_splat_photons_proxy(const evo_frame_params *fp, const evo_camera *cam, int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                             const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                             const evo_record *records, uint32_t num_records, float *out_ideal, float *out_proxy, uint64_t stats[4]);
/* final.frag:19-35 / rtcomphoton.h:1121-1132.  out_rgb: 3 floats per pixel, y = 0 bottom */
void evo_resolve(int32_t W, int32_t H, const float *vpl, const float *pm, const float *light,
                 float vpl_scale, float pm_scale, float light_scale, int mask_emitter, int gamma,
                 float *out_rgb);
/* rtcomphoton.h:1033-1063 */
void evo_progressive_step(int32_t num_iterations_done, float alpha, float clamp_start,
                          uint32_t n_vpl_paths, uint32_t n_light_paths,
                          float *photon_radius, float *clamping_value, float *pdf_mc,
                          int force_vsl, float *vsl_radius, float *vsl_inv_pi_radius2);
/* pathtracing.cu:240-377 (the CPU baseline).  out: RGBA accumulate (+=). returns camera paths traced */
uint64_t evo_path_trace(const evo_scene *s, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces,
                        int32_t W, int32_t H, int32_t row_begin, int32_t row_end,
                        const float *g_pos, const float *g_nrm, const float *g_dif, const float *g_phg,
                        float *out, int do_accumulate);

/* floatimage.cpp:178-199 / 241-258 / 64-112 restated (checked byte-for-byte against oracle/_ref) */
int evo_write_pfm(const char *path, int32_t W, int32_t H, const float *rgb_top_down);
void evo_png_bytes(int32_t n, const float *rgb, uint8_t *out);
double evo_mse(int32_t npix, const float *a, const float *ref);
double evo_rel_mse(int32_t npix, const float *a, const float *ref);

#ifdef __cplusplus
}
#endif
#endif
