/*
 * evplp.h -- C ABI of libevplp_hip.so: the MI355X (gfx950) implementation of evplp's
 * per-pixel indirect-radiance accumulation path (VPL/VSL gather with shadow rays,
 * image-space photon splat, and the feeders either side of them).
 *
 * This is the boundary a maintainer of jamornsriwasansak/evplp binds instead of OptiX +
 * OpenGL.  Every entry point names the reference interface it replaces (paths relative to
 * reflectcuts/; rt/ = realtimetechniques/).  Plain pointers and sizes only; no C++ types,
 * no exceptions cross the ABI.  Every call returns EVPLP_OK (0) or a negative evplp_status;
 * evplp_last_error() gives the message.  One caller thread per context; one context per GPU.
 * There is NO CPU fallback: without a usable HIP device evplp_create fails with
 * EVPLP_ERR_NO_DEVICE.
 *
 * Image convention: row-major, y = 0 at the BOTTOM row (OpenGL / OptiX launch index,
 * shaders/final.frag:22-23).  With row strips (multi-GPU) a context owns the rows of the
 * blocks  b = y / strip_rows  with  b % strip_count == strip_rank  and stores them compactly:
 * local_row = (b / strip_count) * strip_rows + y % strip_rows.
 */
#ifndef EVPLP_H
#define EVPLP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVPLP_ABI_VERSION 4

typedef enum evplp_status {
    EVPLP_OK = 0,
    EVPLP_ERR_INVALID = -1,     /* bad argument / call order */
    EVPLP_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime error at creation */
    EVPLP_ERR_HIP = -3,         /* HIP runtime error (message has the hipError string) */
    EVPLP_ERR_IO = -4,          /* file could not be read / written */
    EVPLP_ERR_PARSE = -5,       /* JSON / OBJ syntax or missing required key */
    EVPLP_ERR_OOM = -6
} evplp_status;

/* rt/rtcomphoton/rtphotonrecord.h:9-15 PhotonRecordFlag */
enum {
    EVPLP_USABLE_VPL = 1, EVPLP_USABLE_PHOTON = 2, EVPLP_LAMBERT_ONLY = 4, EVPLP_PHONG_ONLY = 8
};

/* rt/rtcomphoton/rtphotonrecord.h:17-25 RtPhotonRecord (GLSL mirror photonsplatinstanced.frag:25-33):
 * identical 96-byte array-of-structures layout, so record buffers are interchangeable. */
typedef struct evplp_record {
    float pos[3];      uint32_t flags;
    float normal[3];   float p_select_lambert;
    float flux[3];     float pad1;
    float flux_dir[3]; float pad2;
    float rho_d[3];    float pad3;
    float rho_s[3];    float phong_exp;
} evplp_record;

/* rt/rtcomphoton/rtcomphoton.h:64-72 EMis */
typedef enum evplp_mis_mode {
    EVPLP_MIS_ONE = 0, EVPLP_MIS_BALANCE = 1, EVPLP_MIS_MAX = 2, EVPLP_MIS_POWER2 = 3,
    EVPLP_MIS_GEOMETRY_CLAMP = 4, EVPLP_MIS_GEOMETRY_BRDF_CLAMP = 5
} evplp_mis_mode;

typedef enum evplp_bvh_builder {
    EVPLP_BVH_LBVH = 0,   /* Morton-code LBVH (Karras topology) */
    EVPLP_BVH_SAH = 1,    /* binned-SAH top-down build into the same flattened node format (default of the host side) */
    EVPLP_BVH_SBVH = 2,   /* binned SAH with spatial splits (triangle references clipped at split planes; ~20 % more leaf slots) */
    EVPLP_BVH_LBVH_GPU = 3 /* the LBVH built on the device (Morton sort, Karras hierarchy, bottom-up refit): for scenes that change */
} evplp_bvh_builder;

/* Creation-time configuration: what RtComPhoton::render fixes before setup()
 * (rtcomphoton.h:107-223) plus the build-only device / strip block. */
typedef struct evplp_config {
    int32_t abi_version;         /* EVPLP_ABI_VERSION */
    int32_t device;              /* HIP device ordinal */
    int32_t res_x, res_y;        /* main.cpp:108,114 */
    int32_t strip_rank;          /* row-strip partition; {0,1,H} = whole image */
    int32_t strip_count;
    int32_t strip_rows;          /* block height in rows, multiple of 8 */
    uint32_t num_light_paths;    /* rtcomphoton.h:114 */
    uint32_t num_vpl_light_paths;/* rtcomphoton.h:115 */
    uint32_t photons_per_path;   /* numMaxBounces + 1, rtcomphoton.h:116-117 */
    int32_t bvh_builder;         /* evplp_bvh_builder */
    int32_t deterministic;       /* 1: photon bins are accumulated in record order (bitwise reproducible) */
    int32_t gather_splits_per_wave; /* VPL gather work-item size: consecutive VPL splits (of 128) one wavefront sums; a power of two
                                  * 1..32, 0 = the library's default.  Results do not depend on it (fixed summation tree). */
    int32_t overlap_light_tracing;  /* 1: evplp_trace_light_paths runs on a second HIP stream of the context, ordered only behind the last
                                  * pass that READ the record buffer (gathers, photon splat) and in front of everything enqueued on the
                                  * context's stream after it -- so a preceding evplp_primary overlaps with it (light paths are
                                  * latency-bound, 4.6 waves per SIMD).  Off by default: a caller with kernels of its own on the
                                  * context's stream that read the records between the last such pass and the next light tracing
                                  * needs them ordered.  The technique loops of evplp_render_json and bench.py switch it on.
                                  * What else it does, and what a caller has to know:
                                  *  - the library DOUBLE-BUFFERS what the pipelined passes write: EVPLP_BUF_RECORDS (when a call traces
                                  *    the whole path set), the four G-buffer planes and the tile boxes.  A call that writes one of them
                                  *    flips EVPLP_BUF_* to the copy nobody reads, so the device address behind an EVPLP_BUF_* id CHANGES
                                  *    between calls (evplp_download / evplp_upload always see the current one);
                                  *  - the second copies are allocated on first use: + num_light_paths x photons_per_path x 96 bytes of
                                  *    records (192 MB at config #3, 115 MB at #4) and + 4 x W x rows x 16 bytes of G-buffer;
                                  *  - taking a device pointer with evplp_buffer_info, or binding memory with evplp_bind_buffer, pins
                                  *    that buffer for good: it is never flipped again (the pointer stays valid) and the passes that
                                  *    write it wait for its readers as they do without this flag;
                                  *  - up to two photon splats may be waiting for the verdict on their bin sizes; with `deterministic`
                                  *    set none is ever left pending behind a younger one (bitwise reproducible accumulation). */
    /* Upper bounds of the two large scratch buffers of the gathers, in bytes; 0 = the library's default.  Both are allocated on the first
     * gather that needs them, grow to what the configuration asks for within the bound, and live until evplp_destroy.
     *  cut_scratch_bytes: the entry cuts of the VPL / VSL gathers, 256 bytes per (group of 2 x 2 tiles, VPL record slot).  Default 8 GB.
     *    A configuration that needs more is gathered in bands of tile rows (same results, every band's launches have a tail of their own:
     *    config #5 in nine bands and five launches instead of one and one: 1 174 against 1 167 ms per iteration, 0.7 %); a bound too small for one row of tile blocks, or an allocation that fails,
     *    makes the walks start at the root (same results, ~40 % slower).
     *  vsl_mask_bytes: the lit-lane masks between the two kernels of the VSL gather (8 bytes per (tile, VSL slot) of a launch).
     *    Default 2 GB; smaller bounds mean more launches.
     * What one context allocates on one GPU (whole image; a rank of an n-way group: ~1/n of the per-pixel rows, x 1.5 with the deal's capacity):
     *                                  config #2 (ir)   #3 (evplp)        #4 (ppm)        #5 (vsl)
     *   G-buffer, accumulators, RGB     0.16 GB          0.16 GB           0.30 GB         0.62 GB      (9 planes x 16 B + 12 B per pixel)
     *   records (x 2 when overlapped)   0.4 MB           192 (384) MB      115 (230) MB    115 (230) MB
     *   gather partial sums             1.07 GB          1.07 GB           -               8.6 GB       (64 / 128 groups x 16 B per pixel)
     *   entry cuts (bounded, above)     4.3 GB           4.3 GB            -               8 GB of 68   (in nine bands)
     *   VSL masks (bounded, above)      -                -                 -               2 GB of 8.6  (in five launches)
     *   photon bins + compact photons   -                0.5 GB            0.4 GB          0.5 GB
     *   scene (331 k triangles)         0.1 GB everywhere (nodes, leaf blocks in two layouts, attributes; textures on top)
     * A caller that has the device to itself sets cut_scratch_bytes = 72 GB, vsl_mask_bytes = 14 GB for config #5 (one band, one launch). */
    uint64_t cut_scratch_bytes;
    uint64_t vsl_mask_bytes;
    /* Band mode, the other way to give a context a share of the image: with band_rows > 0 (and strip_count <= 1) it owns the CONTIGUOUS image
     * rows [band_first_row, band_first_row + band_rows), stored from local row 0; band_first_row and band_rows are multiples of 16 (the last
     * band may end at res_y).  band_capacity_rows >= band_rows sizes its buffers (0 = band_rows): evplp_set_band may later move the band
     * anywhere within that capacity.  evplp_group's "bands" partition deals such bands by measured cost (evplp_group_rebalance). */
    int32_t band_first_row, band_rows, band_capacity_rows;
    /* Row strips (strip_count > 1): rows of strip storage, a multiple of strip_rows; 0 = the equal share, ceil(blocks / strip_count) blocks.
     * More than that leaves room for a deal by cost (evplp_set_blocks), which gives a rank of cheap blocks more of them. */
    int32_t strip_capacity_rows;
} evplp_config;

/* rt/rtcommon.h:278-308 RtMaterial: three RGBA32F textures (a constant is a 1x1 texture,
 * rtcommon.h:80-90) + mLightIntensity.  tex_* = id from evplp_add_texture or -1 for the constant. */
typedef struct evplp_material {
    float kd[3]; float ks[3]; float ns;
    int32_t tex_kd, tex_ks, tex_ns;
} evplp_material;

/* rt/rtcommon.h:546-598 RtStableCamera ("direction" of the JSON is a look-at point) */
typedef struct evplp_camera {
    float origin[3]; float lookat[3]; float up[3];
    float fovy;     /* radians; fovx is converted by the caller as rtcommon.h:559 */
    float aspect;
} evplp_camera;

/* The OptiX variables / GL uniforms RtComPhoton::run() sets per iteration
 * (rtcomphoton.h:895-930, 819-823, 1043-1061): the de-facto device ABI of the reference. */
typedef struct evplp_frame_params {
    float camera_pos[3];         /* cameraPosition / uCameraPosition */
    uint32_t mis_mode;           /* misMode / uMisMode */
    float pdf_mc;                /* pdfMc / uPdfMc */
    float clamping_value;        /* clampingValue / uClampingValue */
    float photon_radius;         /* radius / uPhotonRadius */
    float vsl_radius;            /* vslRadius */
    float vsl_inv_pi_radius2;    /* vslInvPiRadius2 */
    uint32_t num_light_paths;    /* numLightPaths (1/N = uInvNumLightPaths) */
    uint32_t num_vpl_light_paths;/* numVplLightPaths */
    uint32_t photons_per_path;   /* numPhotonsPerLightPath */
    uint32_t do_accumulate;      /* doAccumulate */
    uint32_t rng_seed;           /* rngSeed = numIterations + rngOffset (rtcomphoton.h:965) */
    float jitter[2];             /* NDC translation of the jitter matrix (rtcomphoton.h:949) */
    uint32_t splat_footprint;    /* evplp_splat_footprint: which pixels a photon reaches (evplp_splat_photons only) */
    uint32_t reserved;
} evplp_frame_params;

/* The footprint of a photon in evplp_splat_photons.
 *  EVPLP_FOOTPRINT_IDEAL: pixel p receives photon i once iff |X_p - P_i|^2 <= r^2 (the radius test of photonsplatinstanced.frag:152-154
 *    alone).
 *  EVPLP_FOOTPRINT_PROXY: the reference's coverage rule.  It draws an instanced proxy mesh (sphere/icosphere.obj, rtcomphoton.h:677,
 *    632-644) scaled by the radius around every photon (photonsplatinstanced.vert:28-33), un-culled (glEnable(GL_CULL_FACE) is commented
 *    out, rtcomphoton.h:653-655), depth-tested LEQUAL against the deferred pass without depth writes (:789-837), and runs the fragment
 *    shader once per rasterised FACE fragment (.geom:16-32, .frag:146-240): a pixel inside the radius receives the photon once per
 *    face of the scaled proxy that its eye ray crosses between the near plane and the visible surface -- 0, 1 or 2 times for a convex
 *    mesh.  The mesh is the one given to evplp_set_splat_proxy, or the generated 42-vertex / 80-face icosphere
 *    (evplp_default_splat_proxy) if none was given. */
typedef enum evplp_splat_footprint { EVPLP_FOOTPRINT_IDEAL = 0, EVPLP_FOOTPRINT_PROXY = 1 } evplp_splat_footprint;

/* Device buffers a caller may read back, bind to external memory, or hand to a collective. */
typedef enum evplp_buffer {
    EVPLP_BUF_RECORDS = 0,   /* evplp_record[num_light_paths * photons_per_path]  ("photons", rtcomphoton.h:910) */
    EVPLP_BUF_GBUF_POSITION, /* float4[local_rows * W]  deferredPositionTexture */
    EVPLP_BUF_GBUF_NORMAL,   /* float4[...]             deferredNormalTexture */
    EVPLP_BUF_GBUF_DIFFUSE,  /* float4[...]             deferredDiffuseTexture */
    EVPLP_BUF_GBUF_PHONG,    /* float4[...]             deferredPhongReflectanceTexture (rgb, exponent) */
    EVPLP_BUF_LIGHT,         /* float4[...]             mLightTexture */
    EVPLP_BUF_VPL_ACCUM,     /* float4[...]             outputBuffer (rtcomphoton.h:897) */
    EVPLP_BUF_PHOTON_ACCUM,  /* float4[...] (rgb used)  mPhotonSplatTexture */
    EVPLP_BUF_COUNT
} evplp_buffer;

/* Per-pass statistics of the last call (hipEvent timing on the context stream). */
typedef struct evplp_pass_stats {
    float ms;                /* device time of the pass */
    uint64_t pairs;          /* gather: (pixel, usable record) pairs; splat: (photon, covered pixel) pairs */
    uint64_t rays;           /* rays traced by the pass; photon splat with EVPLP_FOOTPRINT_PROXY: fragments of the proxy mesh (0, 1 or 2 per pair) */
    uint64_t usable;         /* usable VPL / photon records consumed */
    float dominant_kernel_ms;/* device time of the pass's dominant kernel alone (summed over its launches) */
    uint32_t reserved[3];    /* [0], [1]: a 64-bit count -- VSL gather: sample-iterations of the estimators; diagnostic builds: node visits; splat: bin entries, fullest bin */
    uint64_t shaded;         /* gather: pairs that passed the cosine test AND the visibility test (contributions evaluated);
                              * photon splat: (photon, pixel) pairs of ALL splat passes of the context so far (running total) */
    uint32_t launches;       /* launches of the dominant kernel in the pass */
    uint32_t pad;
} evplp_pass_stats;

typedef enum evplp_pass {
    EVPLP_PASS_PRIMARY = 0, EVPLP_PASS_LIGHT_TRACE, EVPLP_PASS_GATHER_VPL, EVPLP_PASS_GATHER_VSL,
    EVPLP_PASS_SPLAT, EVPLP_PASS_RESOLVE, EVPLP_PASS_PATH_TRACE, EVPLP_PASS_GATHER_LVC, EVPLP_PASS_COUNT
} evplp_pass;

typedef struct evplp_context evplp_context;

/* ---- lifetime.  Replaces RtComPhoton::setup()/destroy() (rtcomphoton.h:646-708, 1135-1138). ---- */
int evplp_create(const evplp_config *cfg, evplp_context **out);
void evplp_destroy(evplp_context *ctx);
const char *evplp_last_error(const evplp_context *ctx); /* ctx may be NULL: error of a failed create */
int evplp_abi_version(void);
/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = context's own stream. */
int evplp_set_stream(evplp_context *ctx, void *hip_stream);
int evplp_synchronize(evplp_context *ctx);

/* ---- scene upload.  Replaces the createOptix.. / createOpengl.. uploads of RtMesh, RtMaterial and RtTexture
 * (rt/rtcommon.h:196-245, 357-429) and RtScene::addObject/addAreaLight's results (:644-798).
 * Host pointers; the library copies.  Call order: textures, materials, meshes, area light,
 * camera, then evplp_build_accel. ---- */
int evplp_add_texture(evplp_context *ctx, int32_t w, int32_t h, const float *rgba); /* returns id >= 0 */
int evplp_add_material(evplp_context *ctx, const evplp_material *m);                /* returns index >= 0 */
/* RtMesh SoA (rtcommon.h:460-467): vertices float3[nverts], texcoords float2[nverts] (NULL = zeros,
 * rtcommon.h:701-705), triangle indices int3[ntris], one material.  Returns mesh index >= 0. */
int evplp_add_mesh(evplp_context *ctx, const float *vertices, const float *texcoords, int32_t nverts,
                   const int32_t *indices, int32_t ntris, int32_t material);
/* RtScene::addAreaLight (rtcommon.h:772-798): mesh becomes the single emitter, its material is
 * replaced by the black emitter material; intensity = JSON [r,g,b,w] (xyz scaled by pi inside). */
int evplp_set_arealight(evplp_context *ctx, int32_t mesh, const float intensity[4]);
int evplp_set_camera(evplp_context *ctx, const evplp_camera *cam);
/* LoadScene (main.cpp:42-85) in one call: read the scene JSON (resX/resY, scene[], arealight, camera |
 * stablecamera), its OBJ/MTL files, upload everything and build the acceleration structure. */
int evplp_load_scene_json(evplp_context *ctx, const char *json_path);
int evplp_get_camera(evplp_context *ctx, evplp_camera *out);
/* OptiX "Trbvh" acceleration (rtcomphoton.h:705-707) + area-light CDF (rtcommon.h:501-531). */
int evplp_build_accel(evplp_context *ctx);
/* RtScene::findBoundingSphereRadius (rtcommon.h:805-814), totalArea (:759-768), light area (:529) */
int evplp_scene_metrics(evplp_context *ctx, float *bounding_sphere_radius, float *total_area, float *light_area);

/* ---- the per-iteration passes of RtComPhoton::run() (rtcomphoton.h:936-1068) ---- */
/* [deferredShading] + [lightRender]: runDeferredProgram (:710-754) + runLightProgram (:839-855);
 * jitter = (2u-1)/res NDC translation (:949).  light_flags (evplp_light_flags) restate :985-995:
 *   EVPLP_LIGHT_CLEAR       cleareveryframe: pixels that do not show the emitter are written 0 (glClear of the light framebuffer);
 *   EVPLP_LIGHT_UNOCCLUDED  cleareveryframe also clears the depth buffer the light pass shares with the deferred pass
 *                           (GL_DEPTH_BUFFER_BIT, :992): the emitter image is then NOT depth-tested against the scene;
 *   EVPLP_LIGHT_SKIP        run.lightRender = false: the light image is not touched at all.
 * The G-buffer itself is always depth-correct.
 * jitter: any finite NDC translation.  Up to one pixel (|jx| <= 2 / res_x, |jy| <= 2 / res_y; the reference's is at most half a pixel) the
 * rays start from per-tile-group entry cuts of the tree that are built once per camera; a larger one walks from the root (same result,
 * ~2x the pass's time). */
enum evplp_light_flags { EVPLP_LIGHT_CLEAR = 1, EVPLP_LIGHT_UNOCCLUDED = 2, EVPLP_LIGHT_SKIP = 4 };
int evplp_primary(evplp_context *ctx, const float jitter[2], int32_t light_flags);
/* [lightTracing]: launch(LightTrace, numLightPaths) (:869-881).  Traces paths
 * [path_begin, path_begin+path_count) into the record buffer (multi-GPU: each rank a slice). */
int evplp_trace_light_paths(evplp_context *ctx, uint32_t rng_seed, uint32_t path_begin, uint32_t path_count);
/* [vplSplat]: launch(VplSplat, W, H) with splatColor (rt/lighttracing.cu:348-379) */
int evplp_gather_vpl(evplp_context *ctx, const evplp_frame_params *fp);
/* [vplSplat] with forceVsl: splatSplotch (rt/lighttracing.cu:689-722) */
int evplp_gather_vsl(evplp_context *ctx, const evplp_frame_params *fp);
/* The "lvcphotonfam" variant of [vplSplat]: every pixel gathers the usable records of a window of
 * num_vpl_light_paths consecutive light paths starting at a per-pixel random path (mod num_light_paths):
 * splatColor of rt/lvclighttracing.cu:348-384, driven by RtLvcComPhoton (rt/rtcomphoton/rtlvccomphoton.h).
 * Uses fp->rng_seed for the per-pixel offset. */
int evplp_gather_lvc(evplp_context *ctx, const evplp_frame_params *fp);
/* The "pt" technique's device pass: launch(PathTrace, W, H) with splatColor/pathTraceSimple
 * (rt/pathtracing.cu:350-377, 240-348; host runOptixPtProgram rt/rtpt/rtpt2.h:561-573).  One camera path per
 * visible pixel continued from the G-buffer for at most max_bounces bounces, next-event estimation at every
 * vertex; radiance is ADDED to EVPLP_BUF_VPL_ACCUM when do_accumulate != 0, else replaces it ("outputBuffer"). */
int evplp_path_trace(evplp_context *ctx, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces, int32_t do_accumulate);
/* [photonSplat]: runPhotonSplat (:789-837); clear != 0 = cleareveryframe (:978-981); fp->splat_footprint selects the coverage rule */
int evplp_splat_photons(evplp_context *ctx, const evplp_frame_params *fp, int32_t clear);
/* setupPhotonSplatIcosohedron (rtcomphoton.h:632-644, called with "sphere/icosphere.obj" at :677): the proxy mesh of
 * EVPLP_FOOTPRINT_PROXY, in units of the photon radius around the photon (vertices float3[nverts], triangles int3[ntris]; any winding).
 * The mesh must be closed and convex with the origin strictly inside, and have at most EVPLP_MAX_PROXY_PLANES distinct face planes
 * (coplanar triangles count once: a ray crosses one of them); anything else is refused with EVPLP_ERR_INVALID -- the fragment count
 * of a non-convex proxy is not the entry / exit rule the kernel implements.  vertices = NULL restores the generated icosphere. */
#define EVPLP_MAX_PROXY_PLANES 128
int evplp_set_splat_proxy(evplp_context *ctx, const float *vertices, int32_t nverts, const int32_t *indices, int32_t ntris);
/* [finalize] / dumpImage: runFinalProgram(vplScale, photonScale, lightScale, gamma) (:756-787,
 * final.frag:19-35).  mask_emitter = on-screen composite (1) or saved-image sum (0, :1121-1132).
 * out_rgb: HOST pointer, 3 floats per pixel, local_rows * W pixels, y = 0 bottom. */
int evplp_resolve(evplp_context *ctx, float vpl_scale, float photon_scale, float light_scale,
                  int32_t mask_emitter, int32_t gamma, float *out_rgb);
/* The composite alone (what the reference draws to the screen every iteration, runFinalProgram(param, param, 1, true) :997-1004): the strip's RGB stays in
 * device memory, nothing is copied to the host.  evplp_resolve = this + the download.
 * With overlap_light_tracing the call does NOT wait for the verdict on the bins of a photon splat that is still in flight (the host stays an
 * iteration ahead): in the rare iteration whose bins overflowed, the presented frame lacks that one splat pass -- it runs again and is in
 * the accumulator before the next composite.  Frames that leave the device (evplp_resolve, evplp_download, evplp_group_resolve) always
 * settle first and are exact. */
int evplp_present(evplp_context *ctx, float vpl_scale, float photon_scale, float light_scale, int32_t mask_emitter, int32_t gamma);
int evplp_clear_accumulators(evplp_context *ctx);
/* Band mode only: move the context's band to the rows [first_row, first_row + rows) (multiples of 16; rows <= band_capacity_rows).  The
 * accumulators are cleared and the G-buffer is stale: call between runs, not between the iterations of an accumulating run. */
int evplp_set_band(evplp_context *ctx, int32_t first_row, int32_t rows);

/* Row-strip contexts (strip_count > 1): which blocks of strip_rows image rows this context owns.  By default block b belongs to rank
 * b % strip_count.  evplp_set_blocks replaces that by a table: local block l holds image block image_blocks[l], l < count <= the context's
 * capacity (evplp_config.strip_capacity_rows / strip_rows); image_blocks = NULL restores the default.  Every kernel, statistic and buffer
 * layout follows the table; per-pixel results do not depend on it.  As with evplp_set_band the accumulators are cleared and the G-buffer is
 * stale afterwards: call between runs.  evplp_get_blocks returns the number of blocks owned (and up to `capacity` of them, in local order). */
int evplp_set_blocks(evplp_context *ctx, const int32_t *image_blocks, int32_t count);
int evplp_get_blocks(evplp_context *ctx, int32_t *image_blocks, int32_t capacity);
/* What a deal by cost is made from.  While calibration is on, the VPL / VSL gathers run self-clocking variants of their kernels (same
 * results; every wavefront adds the time it was resident to its block's counter; +1 % of the kernel); switching it on clears the counters.
 * evplp_block_costs: the counters by IMAGE block (cost_per_image_block[b], b < ceil(res_y / strip_rows), 0 for blocks of other ranks), in
 * 100 MHz clock ticks normalised to eight wavefronts per SIMD; returns the number of blocks this context owns. */
int evplp_calibrate_blocks(evplp_context *ctx, int32_t on);
int evplp_block_costs(evplp_context *ctx, uint64_t *cost_per_image_block, int32_t capacity);
/* The deal itself (host only, no GPU): longest-processing-time-first -- blocks in order of falling cost, each to the rank with the least
 * cost so far that still has room (at most capacity_blocks per rank) -- ties by block index, so that every process of a multi-process run
 * computes the same table from the same costs.  owner_rank: nblocks ints.  Returns 0, or EVPLP_ERR_INVALID when nranks * capacity_blocks
 * < nblocks. */
int evplp_deal_blocks(const uint64_t *cost_per_image_block, int32_t nblocks, int32_t nranks, int32_t capacity_blocks, int32_t *owner_rank);
/* The blocks a deal gives `rank`, in the order the rank stores (and launches) them: the most expensive first -- a strip's gather is a small
 * launch and its longest items must not start late in it (cost = NULL: image order).  Returns their number; out_blocks may be NULL. */
int evplp_rank_blocks(const uint64_t *cost_per_image_block, const int32_t *owner_rank, int32_t nblocks, int32_t rank, int32_t *out_blocks, int32_t capacity);

/* ---- buffers / statistics ---- */
int evplp_local_rows(const evplp_context *ctx);
/* device_ptr and bytes may each be null.  Taking the device pointer of EVPLP_BUF_GBUF_POSITION (or binding memory to it)
 * tells the library that the caller may write that plane unseen: the photon splat then rebuilds its per-tile position boxes
 * in every pass instead of taking them from evplp_primary. */
int evplp_buffer_info(evplp_context *ctx, int32_t which, void **device_ptr, size_t *bytes);
/* Use caller-owned device memory (e.g. a torch tensor passed to an RCCL collective). */
int evplp_bind_buffer(evplp_context *ctx, int32_t which, void *device_ptr, size_t bytes);
int evplp_download(evplp_context *ctx, int32_t which, void *host_dst, size_t bytes);
int evplp_upload(evplp_context *ctx, int32_t which, const void *host_src, size_t bytes);
int evplp_pass_stats_get(evplp_context *ctx, int32_t pass, evplp_pass_stats *out);
/* evplp_pass_stats.dominant_kernel_ms of the photon splat needs two HIP events BETWEEN the pass's three dependent launches, and they
 * hold the launches apart (18 us of a 227 us pass): they are recorded only while this is on (off by default; the gathers' dominant
 * kernels are bracketed always -- their events sit beside 50 ms kernels).  Off: the splat reports dominant_kernel_ms = ms. */
int evplp_profile_kernels(evplp_context *ctx, int32_t on);
/* evplp_pass_stats.ms comes from two HIP events around every pass; the command processor retires them between the dispatches, and a
 * loop of sub-millisecond iterations pays for that (config #4: 18 us of a 0.61 ms iteration).  Off (the default is on): a pass records
 * only the events the library itself waits on, and evplp_pass_stats_get reports ms = dominant_kernel_ms = 0 for passes run meanwhile
 * (their counters -- pairs, rays, shaded -- stay valid).  evplp_render_json switches it off for all iterations but the last. */
int evplp_profile_passes(evplp_context *ctx, int32_t on);
/* Raw device-side counters of the last run of `pass` (rays, node visits, pairs, aux, then the traversal histogram that
 * only -DEVPLP_TRAVERSAL_STATS=1 diagnostic builds fill).  Returns the number of 64-bit words written. */
int evplp_debug_counters(evplp_context *ctx, int32_t pass, uint64_t *out, int32_t capacity);
/* Flattened acceleration structure statistics: nodes, leaves, max depth, build ms */
int evplp_accel_info(evplp_context *ctx, int32_t *nodes, int32_t *leaves, int32_t *depth, float *build_ms);
/* The builder evplp_build_accel actually used (an evplp_bvh_builder value): cfg.bvh_builder unless the test override
 * EVPLP_BVH_BUILDER was set when the context was created, or an LBVH came out deeper than the walks' 64-entry stacks and the
 * binned-SAH builder took over.  < 0 before the first build. */
int evplp_accel_builder(const evplp_context *ctx);
/* Worst-case stack entries of the four-wide per-lane walk over the tree that was built (0: the generic bound of the depth applies). */
int evplp_accel_stack_entries(const evplp_context *ctx);
/* Device-side self checks: facts the kernels rely on, verified on the GPU they run on.  which = 0: the 7-instruction exact
 * reciprocal of the triangle predicates against the IEEE division on all 2^32 float bit patterns (under a second): out[0]
 * patterns whose bits differ, [1] of them zero / denormal inputs, [2] infinite / NaN inputs, [3] normal inputs, [4] / [5] the
 * smallest / largest biased exponent among those normal inputs (the kernels need [3] to be confined to exponents >= 253).
 * which = 1: d^e as exp2(e log2 d) on the hardware transcendentals (Phong lobes of the VPL gather and the splat) against the
 * double-precision pow for e = 1, 5, 20, 100, 1000, 10000 over 2^22 values of d in (1e-6, 1]: out[k] = largest relative error
 * where the lobe is >= 1e-4 of its peak, in units of 1e-12.
 * Returns the number of words written or a negative status. */
int evplp_selftest(evplp_context *ctx, int32_t which, uint64_t *out, int32_t capacity);
/* The direction-sampling functions of light tracing (csrc/ev_math.h: evm_sincosf, evm_powf) as the DEVICE computes them, on host arrays of n
 * inputs: which = 0: out0 = sin(x), out1 = cos(x); which = 1: out0 = x^y.  For tests: the header is shared with the CPU oracle by #include,
 * so the byte-equality of the light-path records says nothing about these functions unless they give the same bits on both machines. */
int evplp_debug_ev_math(evplp_context *ctx, int32_t which, const float *x, const float *y, int32_t n, float *out0, float *out1);

/* ---- multi-GPU group (SURVEY 8b "Threading", 8e; the reference has one device, main.cpp:111-115).  One caller thread POSTS to
 * n_ranks contexts -- each driven by a worker thread of its own, bound to its GPU -- that own interleaved row strips of the image (see the top of this file); scene and
 * BVH are replicated; light paths are traced by every rank in full (same seed, identical records, no exchange) or 1/n per rank and shared by
 * an in-place all-gather of the record buffers (evplp_group_config.split_light_paths); every rank gathers / splats its own rows; evplp_group_resolve
 * composites the strips where they are and all-gathers them, so that every GPU holds the frame.  The collectives are RCCL
 * (ncclAllGather over xGMI; librccl is opened at run time).  Ranks that all share ONE device ("virtual ranks": tests, one-GPU
 * boxes) exchange by device copies instead.  Per-pixel results do not depend on the partition.  A pass call posts its arguments
 * (copied) to every rank's worker and returns; the workers run their ranks' calls in order.  Errors are sticky: a rank's first failing
 * call is returned by the next group call -- at the latest by evplp_group_synchronize / evplp_group_resolve, which wait for the
 * workers -- with the message in evplp_group_last_error.  evplp_* calls made directly on a rank's context (evplp_group_context) wait
 * until that rank's worker has nothing queued. ---- */
typedef struct evplp_group evplp_group;
typedef struct evplp_group_config {
    int32_t n_ranks;          /* contexts = row-strip ranks, 1..64 */
    const int32_t *devices;   /* HIP ordinal of every rank; NULL = 0, 1, .. n_ranks-1.  All distinct (RCCL) or all equal (virtual ranks) */
    int32_t strip_rows;       /* height of a row block, multiple of 8; 0 = 16 (keeps the gathers' 2 x 2-tile entry-cut groups whole) */
    int32_t use_rccl;         /* 1: a single-rank group goes through RCCL too (otherwise it needs no exchange at all) */
    int32_t partition;        /* evplp_group_partition: how the image is dealt to the ranks */
    int32_t strip_capacity_pct; /* EVPLP_PARTITION_STRIPS: a rank's strip storage in percent of the equal share; 0 = 150 (room for evplp_group_rebalance's deal by cost), 100 = none */
    /* Light tracing on n ranks: 1 = every rank traces 1/n of the paths and the record buffers are all-gathered in place (num_light_paths must be
     * a multiple of n_ranks, else as -1); -1 = every rank traces ALL paths with the same seed (identical records, no exchange); 0 = whichever
     * the library's cost model expects to be faster (evplp_group_split_model: a light-tracing launch has a latency floor, so a share of the
     * paths is not n times faster to trace, while the exchange moves num_light_paths x photons_per_path x 96 / n bytes over every xGMI link). */
    int32_t split_light_paths;
    int32_t reserved;
} evplp_group_config;
/* The model behind split_light_paths = 0 (host only): light tracing of N paths takes 0.20 ms + 1.2 us per 1 000 paths beyond 131 072 (two
 * wavefronts per SIMD; tools/lt_scale.py on one MI355X); an in-place all-gather of chunks of B bytes takes 0.02 ms + B / 48 GB/s (every chunk
 * crosses one xGMI link; 48 GB/s is an ASSUMED effective rate -- no second device was ever available to this build).  Returns 1 when
 * splitting is expected to save time, 0 when not; out_ms (optional): [0] all paths on every rank, [1] a share + the exchange. */
int evplp_group_split_model(uint32_t num_light_paths, uint32_t photons_per_path, int32_t n_ranks, double out_ms[2]);
/* EVPLP_PARTITION_STRIPS: interleaved blocks of strip_rows rows, block b to rank b % n (balanced by interleaving, at the price of every rank
 * walking the whole tree for a fraction of the rays).  EVPLP_PARTITION_BANDS: one contiguous band of rows per rank (evplp_config band mode;
 * strip_rows is ignored) -- equal heights at first, then dealt by measured cost: evplp_group_rebalance moves the band boundaries so that
 * every rank's last frame would have taken the same time.  Per-pixel results do not depend on either. */
typedef enum evplp_group_partition { EVPLP_PARTITION_STRIPS = 0, EVPLP_PARTITION_BANDS = 1 } evplp_group_partition;
/* cfg: as for evplp_create; device / strip_* are set per rank by the group */
int evplp_group_create(const evplp_config *cfg, const evplp_group_config *gcfg, evplp_group **out);
void evplp_group_destroy(evplp_group *g);
const char *evplp_group_last_error(const evplp_group *g);   /* g may be NULL: error of a failed create */
int evplp_group_size(const evplp_group *g);
evplp_context *evplp_group_context(evplp_group *g, int32_t rank);   /* scene upload by hand, per-rank buffers and statistics */
int evplp_group_load_scene_json(evplp_group *g, const char *json_path);
int evplp_group_clear_accumulators(evplp_group *g);
int evplp_group_primary(evplp_group *g, const float jitter[2], int32_t light_flags);
int evplp_group_trace_light_paths(evplp_group *g, uint32_t rng_seed);
int evplp_group_gather(evplp_group *g, const evplp_frame_params *fp, int32_t kind);   /* 0 evplp_gather_vpl, 1 _vsl, 2 _lvc */
int evplp_group_splat_photons(evplp_group *g, const evplp_frame_params *fp, int32_t clear);
int evplp_group_set_splat_proxy(evplp_group *g, const float *vertices, int32_t nverts, const int32_t *indices, int32_t ntris);
int evplp_group_path_trace(evplp_group *g, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces, int32_t do_accumulate);
int evplp_group_synchronize(evplp_group *g);
/* Calibration for evplp_group_rebalance under EVPLP_PARTITION_STRIPS: evplp_calibrate_blocks on every rank (waits for the workers). */
int evplp_group_calibrate(evplp_group *g, int32_t on);
/* The owner of every image block (nblocks = ceil(res_y / strip_rows) ints, rank numbers); returns nblocks. */
int evplp_group_block_owners(evplp_group *g, int32_t *owner_rank, int32_t capacity);
/* EVPLP_PARTITION_STRIPS: waits for the ranks, collects the per-block costs their gathers clocked since evplp_group_calibrate(g, 1), deals
 * the blocks by cost (evplp_deal_blocks, capacity = strip_capacity_pct of the equal share), gives every rank its table (evplp_set_blocks)
 * and switches the calibration off.  The all-gather of the strips then moves max-blocks-per-rank x strip_rows rows per rank.  Returns
 * EVPLP_ERR_INVALID when no cost was clocked (no calibration, or no gather ran).  As below, accumulators are cleared: calibrate on a frame
 * in front of an accumulating run (the technique loop does: "device": {"deal": "cost"}, or by default when the run is long enough).
 * EVPLP_PARTITION_BANDS: waits for the ranks, takes every rank's device time of the passes it ran since the last rebalance (HIP events of
 * primary rays, gathers, photon splat, path tracer), treats it as spread evenly over the rank's rows, and moves the band boundaries (multiples
 * of 16 rows, within the bands' capacity of twice the equal share) to where every rank would have had the same cost.  The accumulators are
 * cleared and the G-buffers are stale afterwards: call it after one or two calibration frames, before an accumulating run (the technique
 * loops always run on strips: the bands are an option of this API, not of evplp_render_json).  band_first_rows: optional, n_ranks + 1 ints, the boundaries it chose (zeros for strips).  A single rank: nothing to do,
 * EVPLP_OK.  Bands without a timed pass since the last rebalance (evplp_profile_passes off): EVPLP_ERR_INVALID, nothing changes. */
int evplp_group_rebalance(evplp_group *g, int32_t *band_first_rows);
/* Host time of rank `rank`'s worker so far, in ms: out[0] inside its rank's pass calls (enqueueing; waits for a splat's verdict included),
 * out[1] inside exchanges (host barrier + collective / copies), out[2] commands run.  Waits until that worker is idle. */
int evplp_group_host_stats(evplp_group *g, int32_t rank, double out[3]);
/* evplp_profile_passes on every rank's context (waits until the workers are idle) */
int evplp_group_profile_passes(evplp_group *g, int32_t on);
/* evplp_resolve for the whole frame: out_rgb = HOST pointer, res_y * res_x * 3 floats, y = 0 bottom */
/* The per-frame exchange alone: composite every rank's strip on its GPU (final.frag:19-35) and all-gather the strips, so that every
 * GPU holds the frame; nothing is copied to the host.  evplp_group_resolve = this + the strips put into image order on rank 0's device
 * (no host-side assembly) + one copy of the W x H x 3 frame to the caller. */
/* (like evplp_present it does not wait for a pending splat's verdict when the contexts overlap light tracing; evplp_group_resolve does) */
int evplp_group_present(evplp_group *g, float vpl_scale, float photon_scale, float light_scale, int32_t mask_emitter, int32_t gamma);
/* exchange = 0: the composite alone, every rank for itself -- no host barrier, no collective (the reference needs the assembled frame only when
 * it is shown or written: rtcomphoton.h:997-1004, 1079-1102, 1124-1132); exchange != 0 = evplp_group_present.  The technique loop:
 * "device": {"exchangeEvery": k}. */
int evplp_group_present_ex(evplp_group *g, float vpl_scale, float photon_scale, float light_scale, int32_t mask_emitter, int32_t gamma, int32_t exchange);
int evplp_group_resolve(evplp_group *g, float vpl_scale, float photon_scale, float light_scale,
                        int32_t mask_emitter, int32_t gamma, float *out_rgb);

/* ---- host side of the reference interface (no GPU needed for these) ---- */
/* The anti-aliasing jitters of the first `count` iterations of a technique run with this rngOffset: NDC translations (x, y) =
 * (2 u - 1) / resolution with u = IndependentSampler(rngOffset).nextVec2() (rtcomphoton.h:887, 946-952) -- the reference's own
 * sampler headers as they behave under g++ / libstdc++ (tests/golden/jitter.npz); out_ndc_xy: 2 * count floats. */
int evplp_jitter_sequence(uint32_t rng_offset, int32_t count, int32_t res_x, int32_t res_y, float *out_ndc_xy);
/* The proxy EVPLP_FOOTPRINT_PROXY uses when no mesh was given: an icosahedron subdivided once onto the unit sphere (42 vertices, 80
 * faces -- the shape the 2178 bytes of the reference's sphere/icosphere.obj, a Git-LFS stub, imply), poles on the y axis.
 * vertices: 42 x 3 floats, indices: 80 x 3 ints (either may be NULL).  Returns the number of triangles (80). */
int evplp_default_splat_proxy(float *vertices, int32_t *indices);
/* The library's JSON reader as the technique blocks use it, for checking it against the reference's (vendored nlohmann::json
 * 2.1.1; main.cpp:105-121, rtcomphoton.h:107-223 -- tests/golden/json_pins.json): `path` = keys separated by '/', decimal indices
 * into arrays.  want: 0 `int v = json[..]`, 1 float, 2 bool, 3 std::string (into str, cap bytes), 4 size(), 5 kind (0 null, 1 bool,
 * 2 number, 3 string, 4 array, 5 object).  Returns 0 ok, 1 not JSON, 2 key missing / index out of range, 3 conversion error. */
int evplp_json_query(const char *text, const char *path, int32_t want, double *num, char *str, int32_t cap);
/* Progressive schedule, rtcomphoton.h:1033-1063; call after numIterations++ */
void evplp_progressive_step(int32_t num_iterations_done, float alpha, float clamp_start,
                            uint32_t n_vpl_paths, uint32_t n_light_paths,
                            float *photon_radius, float *clamping_value, float *pdf_mc,
                            int32_t force_vsl, float *vsl_radius, float *vsl_inv_pi_radius2);
/* FloatImage::Save by extension (common/floatimage/floatimage.cpp:260-273): .pfm / .hdr / .png.
 * rgb: top-down rows (after FlipY, rtcomphoton.h:1124-1127), 3 floats per pixel. */
int evplp_save_image(const char *path, int32_t w, int32_t h, const float *rgb_top_down);
int evplp_load_pfm(const char *path, int32_t *w, int32_t *h, float *rgb_top_down, size_t capacity_floats);
/* FloatImage::LoadPFM / LoadHDR by extension (floatimage.cpp:146-176, 201-221); rgb may be NULL to query the size */
int evplp_load_image(const char *path, int32_t *w, int32_t *h, float *rgb_top_down, size_t capacity_floats);
/* stbi_load(filepath, &width, &height, &channel, 3) as RtTexture calls it (rt/rtcommon.h:144): JPEG (baseline /
 * progressive) or PNG by content -> 8-bit RGB, rows top to bottom (no flip), bit-identical to the reference's
 * vendored decoder.  *channels = components in the file.  rgb may be NULL to query the size. */
int evplp_decode_image(const char *path, int32_t *w, int32_t *h, int32_t *channels, uint8_t *rgb, size_t capacity_bytes);
double evplp_image_mse(int32_t npix, const float *img, const float *ref);     /* floatimage.cpp:64-84 */
/* FloatImage::ComputeSquareErrorHeatImage (relative = 0) / ComputeRelSquareErrorHeatImage (floatimage.cpp:21-62):
 * per-pixel (relative) squared error / max_error, clamped to 1, through Color::Heat (math/color.h:83-88) */
int evplp_image_error_heat(int32_t npix, const float *img, const float *ref, float max_error, int32_t relative, float *out_rgb);
double evplp_image_rel_mse(int32_t npix, const float *img, const float *ref); /* floatimage.cpp:86-112 */
/* the same over the pixels a mask keeps (any non-zero channel; e.g. evplp_decode_image of scene/conference/conference_mask.png,
 * which blanks the emitters' aliased outlines, scene/conference/README.md); NULL mask = all pixels */
double evplp_image_rel_mse_masked(int32_t npix, const float *img, const float *ref, const uint8_t *mask_rgb8);
/* Writes a procedural closed "conference-like" room (OBJ + MTL + light OBJ + scene JSON in the
 * reference's schema) because every mesh of the reference is a Git-LFS stub.
 * Returns the number of scene triangles written (>= 0) or a negative evplp_status. */
int evplp_synth_scene(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed,
                      int32_t res_x, int32_t res_y);
/* style 0: the room of tessellated boxes above ("easy"); style 1: the same room, light and camera furnished with curved and
 * thin parts (ellipsoid cushions, cylinder legs, rotated clutter, ~2400 small occluders) -- closer to what the real
 * conference model (curved chairs, scene/conference/conference_exported.obj, an LFS stub) asks of an any-hit walk;
 * style 2: style 1 with image textures (map_Kd / map_Ks PNG files next to the OBJ) on the room shell and the table. */
int evplp_synth_scene_ex(const char *out_dir, const char *name, int32_t target_triangles, uint32_t seed,
                         int32_t res_x, int32_t res_y, int32_t style);
/* main() + LoadScene + RtComPhoton::render (main.cpp:87-121, rtcomphoton.h:107-223): parse the
 * scene JSON, load OBJ/MTL, run the `photonfam` technique, write the three images + stat file.
 * json_overrides: optional JSON object text merged over the technique block (may be NULL).
 * Build-only keys of a technique block: "bvhBuilder": "sah" | "sbvh" | "lbvh" | "gpu" (evplp_bvh_builder); "deterministic": bool (photon bins accumulated in record
 * order); "device": {"gpus": N, "virtual": bool, "stripRows": R, "rccl": bool, "deal": "cost" | "roundRobin", "exchangeEvery": k,
 * "stripCapacityPct": p, "splitLightPaths": bool, "cutScratchGB": g, "vslMaskGB": g} -- run on an evplp_group of N row-strip ranks (GPUs device .. device+N-1; "virtual": all ranks on `device`; "rccl": a
 * single rank goes through RCCL too; "deal": row blocks dealt by the cost a calibration frame clocks -- the default when that extra frame pays for
 * itself: from 5 iterations at six ranks and more, 25 at three to five, 100 at two -- or block b to rank b % N; "exchangeEvery": the strips are all-gathered in every k-th iteration's composite, 0 = only for
 * the frames that are written -- the default: the loop is headless; 1 = the reference's per-iteration draw; "splitLightPaths": evplp_group_config.split_light_paths, absent = the cost model; "cutScratchGB" /
 * "vslMaskGB": evplp_config.cut_scratch_bytes / vsl_mask_bytes).
 * "device": {"gpus": N, "partition": "iterations"} (photonfam / lvcphotonfam, frameMode accumulate): the N GPUs share out the ITERATIONS of the
 * progressive run instead of the image -- GPU g renders iterations g, g + N, ... of the whole frame on a context of its own, nothing is
 * exchanged inside the loop, the accumulators are summed (rank order, on the host) when a frame is written.  N times the iterations per second
 * with no replicated work; images agree with one GPU's to fp32 round-off (the sums are associated differently), not bit for bit. */
int evplp_render_json(const char *json_path, const char *json_overrides, int32_t device, char *err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif /* EVPLP_H */
