// C-ABI implementation (include/evplp.h): context, scene upload, pass launchers, statistics.
// There is deliberately no CPU execution path here: every pass is a HIP kernel launch.
#include "context.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdlib>

using namespace evplp;

static thread_local char g_create_error[512] = "";
// default bounds of the gathers' two large scratch buffers (evplp_config.cut_scratch_bytes / vsl_mask_bytes = 0; include/evplp.h has the table)
constexpr size_t kDefaultCutScratchBytes = (size_t)8 << 30, kDefaultVslMaskBytes = (size_t)2 << 30;

void evplp_context::set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(error, sizeof(error), fmt, ap);
    va_end(ap);
}

#define CTX_CHECK(ctx) do { if (!(ctx)) return EVPLP_ERR_INVALID; if ((ctx)->quiesce && std::this_thread::get_id() != (ctx)->worker_tid) (ctx)->quiesce((ctx)->quiesce_arg); } while (0)
#define HIP_TRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (ctx)->set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return EVPLP_ERR_HIP; } } while (0)

static size_t buffer_bytes(const evplp_context *c, int which) {
    const size_t px = (size_t)c->st.W * c->st.local_rows;
    if (which == EVPLP_BUF_RECORDS) return sizeof(evplp_record) * (size_t)c->cfg.num_light_paths * c->cfg.photons_per_path;
    return px * sizeof(float4);
}

static int settle_splat(evplp_context *c);

namespace evplp { void set_context_error(evplp_context *ctx, const char *text) { if (ctx) ctx->set_error("%s", text ? text : ""); } }

extern "C" int evplp_abi_version(void) { return EVPLP_ABI_VERSION; }

extern "C" const char *evplp_last_error(const evplp_context *ctx) { return ctx ? ctx->error : g_create_error; }

extern "C" int evplp_create(const evplp_config *cfg, evplp_context **out) {
    if (!cfg || !out) { snprintf(g_create_error, sizeof(g_create_error), "evplp_create: null argument"); return EVPLP_ERR_INVALID; }
    *out = nullptr;
    if (cfg->abi_version != EVPLP_ABI_VERSION) { snprintf(g_create_error, sizeof(g_create_error), "ABI version mismatch: caller %d, library %d", cfg->abi_version, EVPLP_ABI_VERSION); return EVPLP_ERR_INVALID; }
    if (cfg->res_x <= 0 || cfg->res_y <= 0 || cfg->photons_per_path == 0 || cfg->num_light_paths == 0) {
        snprintf(g_create_error, sizeof(g_create_error), "evplp_create: resolution, num_light_paths and photons_per_path must be positive"); return EVPLP_ERR_INVALID;
    }
    if (cfg->gather_splits_per_wave < 0 || cfg->gather_splits_per_wave > 32 || (cfg->gather_splits_per_wave & (cfg->gather_splits_per_wave - 1)) != 0) {
        snprintf(g_create_error, sizeof(g_create_error), "evplp_create: gather_splits_per_wave must be 0 (automatic) or a power of two <= 32"); return EVPLP_ERR_INVALID;
    }
    if (cfg->strip_capacity_rows < 0) { snprintf(g_create_error, sizeof(g_create_error), "evplp_create: strip_capacity_rows is negative"); return EVPLP_ERR_INVALID; }
    if (cfg->num_vpl_light_paths > cfg->num_light_paths) {
        snprintf(g_create_error, sizeof(g_create_error), "evplp_create: num_vpl_light_paths > num_light_paths (VPLs are the first paths of the same set, lighttracing.cu:368)"); return EVPLP_ERR_INVALID;
    }
    int strip_count = cfg->strip_count > 0 ? cfg->strip_count : 1;
    int strip_rows = strip_count == 1 ? ((cfg->res_y + 7) / 8) * 8 : cfg->strip_rows;
    if (strip_rows <= 0 || (strip_rows % 8) != 0 || cfg->strip_rank < 0 || cfg->strip_rank >= strip_count) {
        snprintf(g_create_error, sizeof(g_create_error), "evplp_create: strip_rows must be a positive multiple of 8 and 0 <= strip_rank < strip_count"); return EVPLP_ERR_INVALID;
    }
    const bool band_mode = cfg->band_rows > 0;
    if (band_mode) {
        const int cap = cfg->band_capacity_rows > 0 ? cfg->band_capacity_rows : cfg->band_rows;
        if (strip_count != 1 || cfg->band_first_row < 0 || (cfg->band_first_row % 16) != 0 || cfg->band_first_row >= cfg->res_y || cap < cfg->band_rows ||
            ((cfg->band_rows % 16) != 0 && cfg->band_first_row + cfg->band_rows < cfg->res_y) || cfg->band_first_row + cfg->band_rows > ((cfg->res_y + 15) / 16) * 16) {
            snprintf(g_create_error, sizeof(g_create_error), "evplp_create: a band starts on a multiple of 16 rows inside the image, is a multiple of 16 rows high unless it ends the image, fits its capacity, and excludes strip_count > 1");
            return EVPLP_ERR_INVALID;
        }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_create_error, sizeof(g_create_error), "no HIP device available (%s); libevplp_hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return EVPLP_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { snprintf(g_create_error, sizeof(g_create_error), "device %d out of range (%d devices)", cfg->device, ndev); return EVPLP_ERR_INVALID; }
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { snprintf(g_create_error, sizeof(g_create_error), "hipSetDevice: %s", hipGetErrorString(e)); return EVPLP_ERR_NO_DEVICE; }

    evplp_context *c = new evplp_context();
    c->cfg = *cfg; c->cfg.strip_count = strip_count; c->cfg.strip_rows = strip_rows;
    if (const char *env = std::getenv("EVPLP_BVH_BUILDER"))
        c->env_bvh_builder = !std::strcmp(env, "sbvh") ? EVPLP_BVH_SBVH : !std::strcmp(env, "lbvh") ? EVPLP_BVH_LBVH : !std::strcmp(env, "gpu") ? EVPLP_BVH_LBVH_GPU : EVPLP_BVH_SAH;
    if (const char *env = std::getenv("EVPLP_CUTS")) c->env_cuts = atoi(env) != 0 ? 1 : 0;
    if (const char *env = std::getenv("EVPLP_ITEM_DEAL")) c->env_item_deal = atoi(env) != 0 ? 1 : 0;
    if (const char *env = std::getenv("EVPLP_CUT_BYTES")) c->env_cut_bytes = (size_t)strtoull(env, nullptr, 10);      // (tests: forces the band path)
    if (const char *env = std::getenv("EVPLP_GATHER_K")) { int v = atoi(env); if (v >= 1 && v <= 32 && (v & (v - 1)) == 0) c->env_gather_k = v; }
    if (const char *env = std::getenv("EVPLP_TILE_BLOCK_LOG2")) c->env_tile_block_log2 = std::max(0, atoi(env));
    if (const char *env = std::getenv("EVPLP_SPLIT_MIN")) c->env_split_min = std::max(0, atoi(env));
    c->st.W = cfg->res_x; c->st.H = cfg->res_y;
    c->st.strip_rank = cfg->strip_rank; c->st.strip_count = strip_count; c->st.strip_rows = strip_rows;
    int nblocks = (cfg->res_y + strip_rows - 1) / strip_rows;
    int owned = (nblocks + strip_count - 1) / strip_count;  // padded: equal on every rank (all-gather chunks)
    // strip_capacity_rows: room for more blocks than the equal share (a deal by cost gives a rank of cheap blocks more of them: evplp_set_blocks)
    if (strip_count > 1 && cfg->strip_capacity_rows > owned * strip_rows) owned = std::min((cfg->strip_capacity_rows + strip_rows - 1) / strip_rows, nblocks);
    c->st.local_rows = owned * strip_rows;
    c->st.cap_blocks = owned; c->image_blocks = nblocks;
    if (band_mode) {
        const int cap = cfg->band_capacity_rows > 0 ? cfg->band_capacity_rows : cfg->band_rows;
        c->st.strip_rows = ((cap + 7) / 8) * 8; c->cfg.strip_rows = c->st.strip_rows;
        c->st.local_rows = c->st.strip_rows;
        c->st.band_first = cfg->band_first_row; c->st.band_rows = std::min(cfg->band_rows, cfg->res_y - cfg->band_first_row);
    }
    // rows of this strip that fall inside the image
    c->rows_in_image = 0;
    for (int l = 0; l < c->st.local_rows; l++) if (c->st.global_row(l) < c->st.H) c->rows_in_image++;

    auto fail = [&](const char *what, hipError_t err) {
        snprintf(g_create_error, sizeof(g_create_error), "%s: %s", what, hipGetErrorString(err));
        evplp_destroy(c); return err == hipErrorOutOfMemory ? EVPLP_ERR_OOM : EVPLP_ERR_HIP;
    };
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    c->stream = c->own_stream;
    if (cfg->overlap_light_tracing) {
        if ((e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
        if ((e = hipEventCreateWithFlags(&c->ev_records_read, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&c->ev_light_done, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&c->ev_back_read, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    }
    for (int i = 0; i < EVPLP_PASS_COUNT; i++) {
        if ((e = hipEventCreate(&c->ev_begin[i])) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreate(&c->ev_end[i])) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreate(&c->ev_dom_begin[i])) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreate(&c->ev_dom_end[i])) != hipSuccess) return fail("hipEventCreate", e);
    }
    for (int b = 0; b < EVPLP_BUF_COUNT; b++) {
        size_t bytes = buffer_bytes(c, b);
        if ((e = hipMalloc(&c->buf[b], bytes)) != hipSuccess) return fail("hipMalloc(buffer)", e);
        if ((e = hipMemset(c->buf[b], 0, bytes)) != hipSuccess) return fail("hipMemset", e);
        c->buf_owned[b] = true;
    }
    const uint32_t nvpl_slots = std::max<uint32_t>(cfg->num_vpl_light_paths * cfg->photons_per_path, 1u);
    if ((e = hipMalloc((void **)&c->d_vpls, sizeof(evplp_record) * nvpl_slots)) != hipSuccess) return fail("hipMalloc(vpls)", e);
    if ((e = hipMalloc((void **)&c->d_vpl_src, sizeof(uint32_t) * nvpl_slots)) != hipSuccess) return fail("hipMalloc(vpl_src)", e);
    if ((e = hipMalloc((void **)&c->d_scalars, 64 * sizeof(uint32_t))) != hipSuccess) return fail("hipMalloc(scalars)", e);
    if ((e = hipMemset(c->d_scalars, 0, 64 * sizeof(uint32_t))) != hipSuccess) return fail("hipMemset", e);
    if ((e = hipMalloc((void **)&c->d_counters, sizeof(PassCounters) * EVPLP_PASS_COUNT)) != hipSuccess) return fail("hipMalloc(counters)", e);
    if ((e = hipMemset(c->d_counters, 0, sizeof(PassCounters) * EVPLP_PASS_COUNT)) != hipSuccess) return fail("hipMemset", e);
    if ((e = hipMalloc((void **)&c->d_rgb, sizeof(float) * 3 * (size_t)c->st.W * c->st.local_rows)) != hipSuccess) return fail("hipMalloc(rgb)", e);
    // splat workspace
    c->tiles_x = (c->st.W + 7) / 8; c->tiles_y = c->st.local_rows / 8;
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    const size_t nrec = (size_t)cfg->num_light_paths * cfg->photons_per_path;
    // photon bins: a slab of bin_stride slots per tile, 8x the mean occupancy to start with (a power of two >= 64), doubled
    // when a bin overflows (tiles that see a floor at grazing angle collect ~10x the mean)
    {
        size_t mean = ntiles ? nrec * 2 / std::max<size_t>(ntiles, 1) : 0, k = 64;
        while (k < 8 * mean && k < (1u << 20)) k <<= 1;
        c->bin_stride = (uint32_t)k;
        if (const char *env = std::getenv("EVPLP_BIN_STRIDE")) c->bin_stride = (uint32_t)std::max(1, atoi(env));   // tests: force the overflow / re-run path
        // coarse buckets of 128 tiles (16 x 8) for the two-level binning (kernels.h); larger ones (up to 32 x 16) for images
        // that would need more than kMaxBuckets of them
        c->bucket_w_log2 = 4; c->bucket_h_log2 = kBucketTilesLog2 - 4;
        auto count_buckets = [&]() {
            c->buckets_x = (c->tiles_x + (1 << c->bucket_w_log2) - 1) >> c->bucket_w_log2;
            c->num_buckets = c->buckets_x * ((c->tiles_y + (1 << c->bucket_h_log2) - 1) >> c->bucket_h_log2);
        };
        count_buckets();
        while (c->num_buckets > kMaxBuckets && c->bucket_w_log2 + c->bucket_h_log2 < kMaxBucketTilesLog2) {
            if (c->bucket_h_log2 < c->bucket_w_log2) c->bucket_h_log2++; else c->bucket_w_log2++;
            count_buckets();
        }
        if (c->num_buckets > kMaxBuckets) {
            snprintf(g_create_error, sizeof(g_create_error), "evplp_create: more than %d x %d image tiles in one context (use row strips)", kMaxBuckets, 1 << kMaxBucketTilesLog2);
            evplp_destroy(c); return EVPLP_ERR_INVALID;
        }
        c->num_bin_groups = (int32_t)((nrec + kBinGroup - 1) / kBinGroup);
    }
    if ((e = hipMalloc((void **)&c->d_tile_box, sizeof(float4) * 2 * std::max<size_t>(ntiles, 1))) != hipSuccess) return fail("hipMalloc(tile_box)", e);
    if ((e = hipMalloc((void **)&c->d_summary, sizeof(uint32_t) * (kSummaryFinal + kSummaryStride))) != hipSuccess) return fail("hipMalloc(summary)", e);
    if (const char *env = std::getenv("EVPLP_TILE_MIXED")) c->env_tile_mixed = atoi(env) != 0 ? 1 : 0;
    if (const char *env = std::getenv("EVPLP_TILE_HEAVY")) c->heavy_threshold = (uint32_t)std::max(atoi(env), 1);
    c->mixed_trigger = 2u * c->heavy_threshold;
    if (const char *env = std::getenv("EVPLP_TILE_MIXED_TRIGGER")) c->mixed_trigger = (uint32_t)std::max(atoi(env), 0);
    c->heavy_cap = (uint32_t)std::min<size_t>(std::max<size_t>((size_t)c->tiles_x * c->tiles_y, 1), 2048);
    if (const char *env = std::getenv("EVPLP_TILE_HEAVY_CAP")) c->heavy_cap = (uint32_t)std::max(atoi(env), 1);
    if ((e = hipMalloc((void **)&c->d_heavy_list, sizeof(uint32_t) * c->heavy_cap)) != hipSuccess) return fail("hipMalloc(heavy list)", e);
    if ((e = hipMalloc((void **)&c->d_tile_flags, std::max<size_t>((size_t)c->tiles_x * c->tiles_y, 1))) != hipSuccess) return fail("hipMalloc(tile flags)", e);
    if ((e = hipMemset(c->d_summary, 0, sizeof(uint32_t) * (kSummaryFinal + kSummaryStride))) != hipSuccess) return fail("hipMemset(summary)", e);
    if ((e = hipMalloc((void **)&c->d_tile_pairs, sizeof(uint32_t) * std::max<size_t>(ntiles, 1))) != hipSuccess) return fail("hipMalloc(tile_pairs)", e);
    if ((e = hipMalloc((void **)&c->d_tile_frags, sizeof(uint32_t) * std::max<size_t>(ntiles, 1))) != hipSuccess) return fail("hipMalloc(tile_frags)", e);
    if ((e = hipMalloc((void **)&c->d_tile_cursor, sizeof(uint32_t) * (ntiles + 1))) != hipSuccess) return fail("hipMalloc(tile_cursor)", e);
    const size_t ngroups = std::max(c->num_bin_groups, 1);
    if ((e = hipMalloc((void **)&c->d_seg, sizeof(uint32_t) * ngroups * kSegCap)) != hipSuccess) return fail("hipMalloc(seg)", e);
    if ((e = hipMalloc((void **)&c->d_seg_off, sizeof(uint16_t) * ngroups * (c->num_buckets + 1))) != hipSuccess) return fail("hipMalloc(seg_off)", e);
    if ((e = hipMalloc((void **)&c->d_big_list, sizeof(uint32_t) * ngroups * kBinGroup)) != hipSuccess) return fail("hipMalloc(big_list)", e);
    if ((e = hipMalloc((void **)&c->d_big_count, sizeof(uint32_t) * ngroups)) != hipSuccess) return fail("hipMalloc(big_count)", e);
    if ((e = hipMalloc((void **)&c->d_bin_items, sizeof(uint32_t) * std::max<size_t>(ntiles * c->bin_stride, 1))) != hipSuccess) return fail("hipMalloc(bin_items)", e);
    if (cfg->deterministic && (e = hipMalloc((void **)&c->d_bin_items_tmp, sizeof(uint32_t) * std::max<size_t>(ntiles * c->bin_stride, 1))) != hipSuccess) return fail("hipMalloc(bin_items_tmp)", e);
    if ((e = hipMalloc((void **)&c->d_compact, sizeof(float4) * kCompactF4 * nrec)) != hipSuccess) return fail("hipMalloc(compact)", e);
    if ((e = hipHostMalloc((void **)&c->h_summary, 8 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc(summary)", e);
    std::memset(c->h_summary, 0, 8 * sizeof(uint32_t));
    for (int i = 0; i < 2; i++) {
        if ((e = hipEventCreate(&c->pend[i].ev)) != hipSuccess) return fail("hipEventCreate", e);
        c->pend[i].h = c->h_summary + 4 * i;
    }
    *out = c;
    return EVPLP_OK;
}

static void free_scene_device(evplp_context *c) {
    hipFree((void *)c->sc.nodes); hipFree((void *)c->sc.nodes4); hipFree((void *)c->sc.leaves); hipFree((void *)c->sc.tri_flat); hipFree((void *)c->sc.tri_index); hipFree((void *)c->sc.attrs);
    hipFree((void *)c->sc.materials); hipFree((void *)c->sc.textures); hipFree((void *)c->sc.tex_pool); hipFree((void *)c->sc.light_cdf);
    std::memset(&c->sc, 0, sizeof(c->sc));
}

extern "C" void evplp_destroy(evplp_context *c) {
    if (!c) return;
    hipSetDevice(c->cfg.device);
    c->npend = 0;
    if (c->aux_stream) hipStreamSynchronize(c->aux_stream);
    if (c->own_stream) hipStreamSynchronize(c->own_stream);
    for (int b = 0; b < EVPLP_BUF_COUNT; b++) if (c->buf_owned[b]) hipFree(c->buf[b]);
    free_scene_device(c);
    hipFree(c->d_vpls); hipFree(c->d_vpl_src); hipFree(c->d_scalars); hipFree(c->d_counters); hipFree(c->d_rgb); hipFree(c->d_partial); hipFree(c->d_vsl_masks); hipFree(c->d_cuts); hipFree(c->d_primary_cuts); hipFree(c->d_lt_overflow);
    hipFree(c->d_tile_cursor); hipFree(c->d_bin_items); hipFree(c->d_bin_items_tmp); hipFree(c->d_seg); hipFree(c->d_seg_off); hipFree(c->d_big_list); hipFree(c->d_big_count);
    hipFree(c->d_compact); hipFree(c->d_tile_box); hipFree(c->d_tile_pairs); hipFree(c->d_summary); hipFree(c->d_heavy_list); hipFree(c->d_tile_flags);
    hipFree(c->d_proxy_slabs); hipFree(c->d_proxy_hm); hipFree(c->d_tile_frags); hipFree(c->d_blocks); hipFree(c->d_block_cost);
    for (int i = 0; i < EVPLP_PASS_COUNT; i++) {
        if (c->ev_begin[i]) hipEventDestroy(c->ev_begin[i]);
        if (c->ev_end[i]) hipEventDestroy(c->ev_end[i]);
        if (c->ev_dom_begin[i]) hipEventDestroy(c->ev_dom_begin[i]);
        if (c->ev_dom_end[i]) hipEventDestroy(c->ev_dom_end[i]);
    }
    for (int i = 0; i < 2; i++) if (c->pend[i].ev) hipEventDestroy(c->pend[i].ev);
    if (c->h_summary) hipHostFree(c->h_summary);
    if (c->aux_stream) hipStreamDestroy(c->aux_stream);
    if (c->ev_records_read) hipEventDestroy(c->ev_records_read);
    if (c->ev_light_done) hipEventDestroy(c->ev_light_done);
    if (c->ev_back_read) hipEventDestroy(c->ev_back_read);
    hipFree(c->records_back);
    for (int k = 0; k < 4; k++) hipFree(c->gbuf_back[k]);
    hipFree(c->d_tile_box_back);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int evplp_set_stream(evplp_context *c, void *s) {
    CTX_CHECK(c);
    if (c->aux_stream) HIP_TRY(c, hipStreamSynchronize(c->aux_stream));    // (its pending wait was enqueued on the old stream)
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return EVPLP_OK;
}
extern "C" int evplp_synchronize(evplp_context *c) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EVPLP_OK;
}

// ------------------------------------------------------------------------------ scene upload
extern "C" int evplp_add_texture(evplp_context *c, int32_t w, int32_t h, const float *rgba) {
    CTX_CHECK(c);
    if (w <= 0 || h <= 0 || !rgba) { c->set_error("evplp_add_texture: bad arguments"); return EVPLP_ERR_INVALID; }
    HostTexture t; t.w = w; t.h = h; t.rgba.assign(rgba, rgba + (size_t)w * h * 4);
    c->textures.push_back(std::move(t));
    return (int)c->textures.size() - 1;
}
extern "C" int evplp_add_material(evplp_context *c, const evplp_material *m) {
    CTX_CHECK(c);
    if (!m) { c->set_error("evplp_add_material: null"); return EVPLP_ERR_INVALID; }
    int nt = (int)c->textures.size();
    if (m->tex_kd >= nt || m->tex_ks >= nt || m->tex_ns >= nt) { c->set_error("evplp_add_material: texture id out of range"); return EVPLP_ERR_INVALID; }
    Material d; std::memset(&d, 0, sizeof(d));
    std::memcpy(d.kd, m->kd, 12); std::memcpy(d.ks, m->ks, 12); d.ns = m->ns;
    d.tex_kd = m->tex_kd < 0 ? -1 : m->tex_kd; d.tex_ks = m->tex_ks < 0 ? -1 : m->tex_ks; d.tex_ns = m->tex_ns < 0 ? -1 : m->tex_ns;
    c->materials.push_back(d);
    return (int)c->materials.size() - 1;
}
extern "C" int evplp_add_mesh(evplp_context *c, const float *vertices, const float *texcoords, int32_t nverts,
                              const int32_t *indices, int32_t ntris, int32_t material) {
    CTX_CHECK(c);
    if (!vertices || !indices || nverts <= 0 || ntris < 0) { c->set_error("evplp_add_mesh: bad arguments"); return EVPLP_ERR_INVALID; }
    if (material < 0 || material >= (int)c->materials.size()) { c->set_error("evplp_add_mesh: material %d out of range", material); return EVPLP_ERR_INVALID; }
    for (int64_t i = 0; i < (int64_t)ntris * 3; i++) if (indices[i] < 0 || indices[i] >= nverts) { c->set_error("evplp_add_mesh: vertex index out of range"); return EVPLP_ERR_INVALID; }
    HostMesh m; m.material = material;
    m.verts.assign(vertices, vertices + (size_t)nverts * 3);
    if (texcoords) m.uvs.assign(texcoords, texcoords + (size_t)nverts * 2); else m.uvs.assign((size_t)nverts * 2, 0.0f);  // rtcommon.h:701-705
    m.idx.assign(indices, indices + (size_t)ntris * 3);
    c->meshes.push_back(std::move(m));
    c->accel_built = false;
    return (int)c->meshes.size() - 1;
}
extern "C" int evplp_set_arealight(evplp_context *c, int32_t mesh, const float intensity[4]) {
    CTX_CHECK(c);
    if (mesh < 0 || mesh >= (int)c->meshes.size() || !intensity) { c->set_error("evplp_set_arealight: bad arguments"); return EVPLP_ERR_INVALID; }
    if (c->light_mesh >= 0) { c->set_error("only one area light is supported (rt/rtcommon.h:770-774)"); return EVPLP_ERR_INVALID; }
    // rtcommon.h:780-790: xyz scaled by pi, black emitter material carrying mLightIntensity
    Material d; std::memset(&d, 0, sizeof(d));
    d.tex_kd = d.tex_ks = d.tex_ns = -1;
    const float pi = 3.14159265358979323846f;
    for (int k = 0; k < 3; k++) { c->light_unscaled[k] = intensity[k]; c->light_scaled[k] = intensity[k] * pi; }
    c->light_unscaled[3] = c->light_scaled[3] = intensity[3];
    std::memcpy(d.light, c->light_scaled, 16);
    c->materials.push_back(d);
    c->meshes[mesh].material = (int)c->materials.size() - 1;
    c->light_mesh = mesh;
    c->accel_built = false;
    return EVPLP_OK;
}
extern "C" int evplp_set_camera(evplp_context *c, const evplp_camera *cam) {
    CTX_CHECK(c);
    if (!cam) { c->set_error("evplp_set_camera: null"); return EVPLP_ERR_INVALID; }
    // glm::lookAt (RH) basis + glm::perspective parameters (rt/rtcommon.h:586-591)
    auto norm = [](float *v) { float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= inv; v[1] *= inv; v[2] *= inv; };
    float f[3] = { cam->lookat[0] - cam->origin[0], cam->lookat[1] - cam->origin[1], cam->lookat[2] - cam->origin[2] };
    norm(f);
    float s[3] = { f[1] * cam->up[2] - f[2] * cam->up[1], f[2] * cam->up[0] - f[0] * cam->up[2], f[0] * cam->up[1] - f[1] * cam->up[0] };
    norm(s);
    float u[3] = { s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0] };
    std::memset(&c->cam, 0, sizeof(c->cam));
    std::memcpy(c->cam.eye, cam->origin, 12); std::memcpy(c->cam.s, s, 12); std::memcpy(c->cam.u, u, 12); std::memcpy(c->cam.f, f, 12);
    c->cam.tan_half = std::tan(cam->fovy / 2.0f); c->cam.aspect = cam->aspect;
    c->cam_in = *cam;
    c->camera_set = true; c->primary_cuts_valid = false;
    return EVPLP_OK;
}

extern "C" int evplp_get_camera(evplp_context *c, evplp_camera *out) {
    CTX_CHECK(c);
    if (!out || !c->camera_set) { c->set_error("evplp_get_camera: camera not set"); return EVPLP_ERR_INVALID; }
    *out = c->cam_in;
    return EVPLP_OK;
}

template <class T> static int upload_array(evplp_context *c, const T *host, size_t n, const T **dev) {
    void *p = nullptr;
    size_t bytes = sizeof(T) * std::max<size_t>(n, 1);
    HIP_TRY(c, hipMalloc(&p, bytes));
    if (n) HIP_TRY(c, hipMemcpy(p, host, sizeof(T) * n, hipMemcpyHostToDevice));
    *dev = (const T *)p;
    return EVPLP_OK;
}

extern "C" int evplp_build_accel(evplp_context *c) {
    CTX_CHECK(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->meshes.empty()) { c->set_error("evplp_build_accel: no meshes"); return EVPLP_ERR_INVALID; }
    if (c->light_mesh < 0) { c->set_error("evplp_build_accel: no area light set"); return EVPLP_ERR_INVALID; }
    free_scene_device(c);
    // flatten to a triangle soup in mesh order
    std::vector<float> verts; std::vector<TriAttr> attrs;
    int32_t light_first = 0, light_count = 0;
    for (size_t mi = 0; mi < c->meshes.size(); mi++) {
        const HostMesh &m = c->meshes[mi];
        if ((int)mi == c->light_mesh) { light_first = (int32_t)attrs.size(); light_count = (int32_t)(m.idx.size() / 3); }
        for (size_t t = 0; t < m.idx.size() / 3; t++) {
            TriAttr a; std::memset(&a, 0, sizeof(a));
            for (int k = 0; k < 3; k++) {
                int32_t vi = m.idx[3 * t + k];
                for (int j = 0; j < 3; j++) a.v[3 * k + j] = m.verts[3 * (size_t)vi + j];
                a.uv[2 * k] = m.uvs[2 * (size_t)vi]; a.uv[2 * k + 1] = m.uvs[2 * (size_t)vi + 1];
            }
            a.material = m.material;
            attrs.push_back(a);
            verts.insert(verts.end(), a.v, a.v + 9);
        }
    }
    if (light_count <= 0) { c->set_error("evplp_build_accel: the area-light mesh has no triangles"); return EVPLP_ERR_INVALID; }
    BvhBuild bb;
    int builder = c->env_bvh_builder >= 0 ? c->env_bvh_builder : c->cfg.bvh_builder;   // (override read by evplp_create)
    if (builder == EVPLP_BVH_LBVH_GPU) {
        BvhDeviceBuild gb;
        const int e = build_bvh_gpu(verts.data(), (int32_t)attrs.size(), bvh_pad_scale(), c->stream, &gb);
        if (e != 0) { c->set_error("evplp_build_accel: device LBVH build: %s", hipGetErrorString((hipError_t)e)); return EVPLP_ERR_HIP; }
        if (gb.depth > kMaxDepth - 2) {
            // long radix-tree chains (clustered or duplicate Morton codes) can exceed the walks' 64-entry stacks: build the
            // binned-SAH tree on the host instead of failing the scene
            hipFree(gb.nodes); hipFree(gb.leaves); hipFree(gb.tri_flat); hipFree(gb.tri_index);
            builder = EVPLP_BVH_SAH;
        } else {
            c->sc.nodes = gb.nodes; c->sc.leaves = gb.leaves; c->sc.tri_flat = gb.tri_flat; c->sc.tri_index = gb.tri_index;
            bb.nnodes = gb.nnodes; bb.nleaves = gb.nleaves; bb.depth = gb.depth; bb.build_ms = gb.build_ms; bb.ntris = gb.ntris;
        }
    }
    if (builder != EVPLP_BVH_LBVH_GPU) {
        build_bvh(verts.data(), (int32_t)attrs.size(), builder, &bb);
        if (bb.depth > kMaxDepth - 2 && builder == EVPLP_BVH_LBVH) { free_bvh(&bb); bb = BvhBuild(); builder = EVPLP_BVH_SAH; build_bvh(verts.data(), (int32_t)attrs.size(), builder, &bb); }
    }
    c->accel_builder_used = builder;
    c->accel_nodes = bb.nnodes; c->accel_leaves = bb.nleaves; c->accel_depth = bb.depth; c->accel_build_ms = bb.build_ms;
    if (bb.depth > kMaxDepth - 2) { free_bvh(&bb); free_scene_device(c); c->set_error("BVH depth %d exceeds the traversal stack (%d)", bb.depth, kMaxDepth); return EVPLP_ERR_INVALID; }
    // area-light CDF, rt/rtcommon.h:501-531 (running float sum, then normalised); Triangle::ComputeArea
    auto tri_area = [](const float *v) {
        float a[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] }, b[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
        float cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
        return std::sqrt(cx * cx + cy * cy + cz * cz) / 2.0f;
    };
    std::vector<float> cdf((size_t)light_count);
    float sum = 0.f;
    for (int32_t i = 0; i < light_count; i++) { sum += tri_area(attrs[(size_t)light_first + i].v); cdf[i] = sum; }
    for (int32_t i = 0; i < light_count; i++) cdf[i] /= sum;
    // scene metrics (rtcommon.h:759-768, 805-814): per-mesh float sums, bbox over all vertices
    float total = 0.f; float lo[3] = { 3.4028235e38f, 3.4028235e38f, 3.4028235e38f }, hi[3] = { -3.4028235e38f, -3.4028235e38f, -3.4028235e38f };
    size_t tcur = 0;
    for (const HostMesh &m : c->meshes) {
        float ms = 0.f;
        for (size_t t = 0; t < m.idx.size() / 3; t++) ms += tri_area(attrs[tcur++].v);
        total += ms;
        for (size_t v = 0; v < m.verts.size() / 3; v++) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], m.verts[3 * v + k]); hi[k] = std::max(hi[k], m.verts[3 * v + k]); }
    }
    float dg[3] = { std::max(hi[0] - lo[0], 0.f), std::max(hi[1] - lo[1], 0.f), std::max(hi[2] - lo[2], 0.f) };
    c->bounding_radius = std::sqrt(dg[0] * dg[0] + dg[1] * dg[1] + dg[2] * dg[2]) / 2.0f;
    c->total_area = total; c->light_area = sum;

    // textures -> one float4 pool
    std::vector<TexDesc> tdesc; std::vector<float4> pool;
    for (const HostTexture &t : c->textures) {
        TexDesc d; d.w = t.w; d.h = t.h; d.offset = (uint32_t)pool.size(); d.pad = 0;
        for (size_t i = 0; i < (size_t)t.w * t.h; i++) pool.push_back(make_float4(t.rgba[4 * i], t.rgba[4 * i + 1], t.rgba[4 * i + 2], t.rgba[4 * i + 3]));
        tdesc.push_back(d);
    }
    int rc;
    if (builder != EVPLP_BVH_LBVH_GPU) {
        if ((rc = upload_array(c, bb.nodes, (size_t)bb.nnodes, &c->sc.nodes))) { free_bvh(&bb); return rc; }
        if ((rc = upload_array(c, bb.leaves, (size_t)std::max(bb.nleaves, 1), &c->sc.leaves))) { free_bvh(&bb); return rc; }
        if ((rc = upload_array(c, bb.tri_index, (size_t)std::max(bb.nleaves, 1) * 4, &c->sc.tri_index))) { free_bvh(&bb); return rc; }
        if ((rc = upload_array(c, bb.tri_flat, (size_t)std::max(bb.nleaves, 1) * 4, &c->sc.tri_flat))) { free_bvh(&bb); return rc; }
    }
    c->sc.ntris = bb.ntris; c->sc.bvh_depth = bb.depth;
    {
        // Worst-case stack of the four-wide per-lane walk (closest_lane4: a visit pushes every entered grandchild but the nearest) over
        // the tree that was built: need(i) = (grandchildren of i) - 1 + max over inner grandchildren g of need(g).  Inner nodes are stored in
        // pre-order (children after parents), so one backward sweep does it.  The LDS stack is what limits the occupancy of light tracing.
        std::vector<BvhNode> host_nodes;
        const BvhNode *hn = bb.nodes;
        if (!hn) { host_nodes.resize((size_t)bb.nnodes); HIP_TRY(c, hipMemcpy(host_nodes.data(), c->sc.nodes, sizeof(BvhNode) * (size_t)bb.nnodes, hipMemcpyDeviceToHost)); hn = host_nodes.data(); }
        std::vector<int32_t> need((size_t)bb.nnodes, 0);
        bool ordered = true;
        for (int32_t i = bb.nnodes - 1; i >= 0; i--) {
            int32_t g[4]; int m = 0;
            const int32_t ch[2] = { hn[i].c0, hn[i].c1 };
            for (int s2 = 0; s2 < 2; s2++) {
                if (ch[s2] >= 0) { if (ch[s2] <= i) ordered = false; g[m++] = hn[ch[s2]].c0; g[m++] = hn[ch[s2]].c1; }
                else if (ch[s2] != kNoChild) g[m++] = ch[s2];
            }
            int32_t valid = 0, deepest = 0;
            for (int q = 0; q < m; q++) if (g[q] != kNoChild) { valid++; if (g[q] >= 0) { if (g[q] <= i) ordered = false; else deepest = std::max(deepest, need[(size_t)g[q]]); } }
            need[(size_t)i] = std::max(valid - 1, 0) + deepest;
        }
        c->sc.stack4_entries = ordered && bb.nnodes > 0 ? need[0] + 2 : 0;      // (0: not pre-ordered -- the generic bound of the depth applies)
    }
    { BvhNode4 *n4 = nullptr; const int e4 = build_nodes4(c->sc.nodes, bb.nnodes, c->stream, &n4);
      if (e4 != 0) { free_bvh(&bb); c->set_error("evplp_build_accel: four-wide nodes: %s", hipGetErrorString((hipError_t)e4)); return EVPLP_ERR_HIP; }
      c->sc.nodes4 = n4; }
    free_bvh(&bb);
    if ((rc = upload_array(c, attrs.data(), attrs.size(), &c->sc.attrs))) return rc;
    if ((rc = upload_array(c, c->materials.data(), c->materials.size(), &c->sc.materials))) return rc;
    if ((rc = upload_array(c, tdesc.data(), tdesc.size(), &c->sc.textures))) return rc;
    if ((rc = upload_array(c, pool.data(), pool.size(), &c->sc.tex_pool))) return rc;
    if ((rc = upload_array(c, cdf.data(), cdf.size(), &c->sc.light_cdf))) return rc;
    c->sc.light_first = light_first; c->sc.light_count = light_count; c->sc.light_area = sum;
    {   // bounds of the light mesh, padded by far more than the rounding of a slab test
        float llo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, lhi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        for (int32_t i = 0; i < light_count; i++) for (int k = 0; k < 9; k++) {
            const float v = attrs[(size_t)light_first + i].v[k];
            llo[k % 3] = std::min(llo[k % 3], v); lhi[k % 3] = std::max(lhi[k % 3], v);
        }
        const float pad = 1e-4f * (2.0f * c->bounding_radius) + 1e-30f;
        for (int k = 0; k < 3; k++) { c->sc.light_lo[k] = llo[k] - pad; c->sc.light_hi[k] = lhi[k] + pad; }
    }
    std::memcpy(c->sc.light_intensity, c->light_scaled, 16); std::memcpy(c->sc.light_unscaled, c->light_unscaled, 16);
    c->accel_built = true; c->primary_cuts_valid = false;
    return EVPLP_OK;
}

extern "C" int evplp_scene_metrics(evplp_context *c, float *r, float *total, float *light) {
    CTX_CHECK(c);
    if (!c->accel_built) { c->set_error("evplp_scene_metrics: call evplp_build_accel first"); return EVPLP_ERR_INVALID; }
    if (r) *r = c->bounding_radius; if (total) *total = c->total_area; if (light) *light = c->light_area;
    return EVPLP_OK;
}
extern "C" int evplp_accel_stack_entries(const evplp_context *c) { return (c && c->accel_built) ? c->sc.stack4_entries : -1; }
extern "C" int evplp_accel_info(evplp_context *c, int32_t *nodes, int32_t *leaves, int32_t *depth, float *build_ms) {
    CTX_CHECK(c);
    if (nodes) *nodes = c->accel_nodes; if (leaves) *leaves = c->accel_leaves; if (depth) *depth = c->accel_depth; if (build_ms) *build_ms = c->accel_build_ms;
    return EVPLP_OK;
}

extern "C" int evplp_accel_builder(const evplp_context *c) {
    if (!c || !c->accel_built) return -1;
    return c->accel_builder_used;
}

// ---------------------------------------------------------------------------------- passes
// Look at the bin summary of the last photon splat (see context.hpp).  Called at the start of every entry point that
// enqueues work, reads results or changes buffers.  Overflow is rare (the bins carry 25 % slack over the last pass and the
// radius only shrinks in a progressive run): then the bins grow and fill + tiles of that pass run again -- they wrote nothing.
// fullest bin from which the tile kernel runs four waves per tile (evplp_splat_photons explains; EVPLP_SPLIT_MIN: developer override)
static uint32_t split_threshold(const evplp_context *c, uint32_t footprint) {
    if (c->env_split_min > 0) return (uint32_t)c->env_split_min;
    // (the proxy variant keeps its four waves busy only in much fuller bins: at config #3, fullest bin ~1400, one wave per tile takes 136 us
    // and four take 178; the ideal kernel 87 / 82)
    return footprint == (uint32_t)EVPLP_FOOTPRINT_PROXY ? 1536u : 768u;
}
static int settle_one(evplp_context *c) {                                // the oldest pending pass
    if (c->npend == 0) return EVPLP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipEventSynchronize(c->pend[0].ev));
    const uint32_t total = c->pend[0].h[0], biggest = c->pend[0].h[1], overflow = c->pend[0].h[2];
    SplatArgs a = c->pend[0].args;
    std::swap(c->pend[0], c->pend[1]); c->npend--;                       // (the slot keeps its event and its pinned words)
    c->last_bin_entries = total; c->last_bin_max = biggest;
    if (!overflow) return EVPLP_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t want = 64; while (want < (uint64_t)overflow + overflow / 4) want <<= 1;     // overflow = slots the fullest bin wanted
    const size_t ntiles = (size_t)c->tiles_x * c->tiles_y;
    if (want * ntiles > 0xfffffff0ull) { c->set_error("photon bins: %llu slots per tile needed", (unsigned long long)overflow); return EVPLP_ERR_OOM; }
    if (want > c->bin_stride) {                                          // (a younger pending pass may have grown them already)
        hipFree(c->d_bin_items); c->d_bin_items = nullptr;
        if (c->d_bin_items_tmp) { hipFree(c->d_bin_items_tmp); c->d_bin_items_tmp = nullptr; }
        hipError_t e1 = hipMalloc((void **)&c->d_bin_items, sizeof(uint32_t) * ntiles * want);
        hipError_t e2 = c->cfg.deterministic ? hipMalloc((void **)&c->d_bin_items_tmp, sizeof(uint32_t) * ntiles * want) : hipSuccess;
        if (e1 != hipSuccess || e2 != hipSuccess) { c->set_error("photon bins: cannot allocate %zu x %llu slots", ntiles, (unsigned long long)want); return EVPLP_ERR_OOM; }
        c->bin_stride = (uint32_t)want;
    }
    a.bin_items = c->d_bin_items; a.bin_items_tmp = c->d_bin_items_tmp; a.bin_stride = c->bin_stride;
    for (int i = 0; i < c->npend; i++) {                                 // a younger pass ran with the old slabs: if IT has to run again, then with these
        c->pend[i].args.bin_items = c->d_bin_items; c->pend[i].args.bin_items_tmp = c->d_bin_items_tmp; c->pend[i].args.bin_stride = c->bin_stride;
    }
    HIP_TRY(c, hipEventRecord(c->ev_begin[EVPLP_PASS_SPLAT], c->stream));  // (the pass statistics then describe this re-run as a whole, not a mix of two passes)
    c->pass_timed[EVPLP_PASS_SPLAT] = true;
    launch_splat_bin(a, c->stream);                                       // (clears the overflow flag, the cursors and the summary)
    const bool split_tiles = c->cfg.deterministic ? true : biggest >= split_threshold(c, a.fp.splat_footprint);
    launch_splat_tiles(a, split_tiles, c->stream, c->ev_dom_begin[EVPLP_PASS_SPLAT], c->ev_dom_end[EVPLP_PASS_SPLAT]);
    HIP_TRY(c, hipEventRecord(c->ev_end[EVPLP_PASS_SPLAT], c->stream));
    // (the records this pass read may meanwhile be the BACK buffer of the overlapped light tracing: its readers' event then)
    if (c->aux_stream) HIP_TRY(c, hipEventRecord((const void *)a.records == c->buf[EVPLP_BUF_RECORDS] ? c->ev_records_read : c->ev_back_read, c->stream));
    HIP_TRY(c, hipGetLastError());
    return EVPLP_OK;
}
static int settle_splat(evplp_context *c) {
    while (c->npend > 0) { int rc = settle_one(c); if (rc) return rc; }
    return EVPLP_OK;
}
// the pending passes that read `records` or the G-buffer whose position plane is `g_pos` (and every older one) must be settled
// before those are overwritten
static int settle_readers_of(evplp_context *c, const void *records, const void *g_pos) {
    int upto = 0;
    for (int i = 0; i < c->npend; i++)
        if ((records && (const void *)c->pend[i].args.records == records) || (g_pos && (const void *)c->pend[i].args.g_pos == g_pos)) upto = i + 1;
    for (int i = 0; i < upto; i++) { int rc = settle_one(c); if (rc) return rc; }
    return EVPLP_OK;
}
static int pass_ready(evplp_context *c, const char *name, bool need_camera, bool settle = true) {
    if (!c->accel_built) { c->set_error("%s: scene not built (evplp_build_accel)", name); return EVPLP_ERR_INVALID; }
    if (need_camera && !c->camera_set) { c->set_error("%s: camera not set", name); return EVPLP_ERR_INVALID; }
    hipError_t e = hipSetDevice(c->cfg.device);
    if (e != hipSuccess) { c->set_error("hipSetDevice: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; }
    return settle ? settle_splat(c) : EVPLP_OK;
}
// device counters are written by the gathers and the path tracer (and by the counters build of the splat)
#ifndef EVPLP_TRAVERSAL_STATS
#define EVPLP_TRAVERSAL_STATS 0
#endif
static bool pass_uses_counters(int pass) {
    return pass == EVPLP_PASS_GATHER_VPL || pass == EVPLP_PASS_GATHER_VSL || pass == EVPLP_PASS_GATHER_LVC || pass == EVPLP_PASS_PATH_TRACE || EVPLP_TRAVERSAL_STATS || EVPLP_DEBUG_NAN;
}
// Every recorded event is a marker the command processor has to retire between two dispatches: the two per pass cost config #4's
// 0.6 ms iteration 18 us (measured, round 5).  evplp_profile_passes(ctx, 0) leaves only the events the library itself waits on.
static int pass_begin(evplp_context *c, int pass) {
    if (pass_uses_counters(pass)) HIP_TRY(c, hipMemsetAsync(&c->d_counters[pass], 0, sizeof(PassCounters), c->stream));
    if (c->profile_passes || pass == EVPLP_PASS_PRIMARY) HIP_TRY(c, hipEventRecord(c->ev_begin[pass], c->stream));    // (the overlapped light tracing starts beside the G-buffer pass)
    c->pass_ran[pass] = true; c->pass_has_dom[pass] = false; c->pass_timed[pass] = c->profile_passes;
    return EVPLP_OK;
}
static int pass_end(evplp_context *c, int pass) {
    if (c->pass_timed[pass]) HIP_TRY(c, hipEventRecord(c->ev_end[pass], c->stream));
    if (c->aux_stream && (pass == EVPLP_PASS_GATHER_VPL || pass == EVPLP_PASS_GATHER_VSL || pass == EVPLP_PASS_GATHER_LVC || pass == EVPLP_PASS_SPLAT))
        HIP_TRY(c, hipEventRecord(c->ev_records_read, c->stream));
    HIP_TRY(c, hipGetLastError());
    return EVPLP_OK;
}

extern "C" int evplp_primary(evplp_context *c, const float jitter[2], int32_t clear_light) {
    CTX_CHECK(c);
    // A pending photon splat may have to run again from the G-buffer it was given (settle_splat).  With overlap_light_tracing the
    // G-buffer is double-buffered like the records: this pass writes the set that splat does not read and the host does not wait here
    // (every reader of that set is in front of this pass on the same stream).
    static const int kPlanes[4] = { EVPLP_BUF_GBUF_POSITION, EVPLP_BUF_GBUF_NORMAL, EVPLP_BUF_GBUF_DIFFUSE, EVPLP_BUF_GBUF_PHONG };
    bool flip = c->aux_stream && c->npend > 0 && !c->gbuf_exposed && !c->gbuf_pos_exposed;
    for (int k = 0; k < 4 && flip; k++) flip = c->buf_owned[kPlanes[k]];
    int rc = pass_ready(c, "evplp_primary", true, !flip); if (rc) return rc;
    if (flip && c->gbuf_back[0] && (rc = settle_readers_of(c, nullptr, c->gbuf_back[0]))) return rc;      // (a splat two passes old: long finished)
    if (flip) {
        for (int k = 0; k < 4; k++) if (!c->gbuf_back[k]) {
            hipError_t me = hipMalloc(&c->gbuf_back[k], buffer_bytes(c, kPlanes[k]));
            if (me != hipSuccess) { c->set_error("evplp_primary: second G-buffer: %s", hipGetErrorString(me)); return EVPLP_ERR_OOM; }
        }
        if (!c->d_tile_box_back) {
            hipError_t me = hipMalloc((void **)&c->d_tile_box_back, sizeof(float4) * 2 * std::max<size_t>((size_t)c->tiles_x * c->tiles_y, 1));
            if (me != hipSuccess) { c->set_error("evplp_primary: second tile-box table: %s", hipGetErrorString(me)); return EVPLP_ERR_OOM; }
        }
        for (int k = 0; k < 4; k++) std::swap(c->buf[kPlanes[k]], c->gbuf_back[k]);
        std::swap(c->d_tile_box, c->d_tile_box_back);
    }
    PrimaryArgs a; std::memset(&a, 0, sizeof(a));
    a.sc = c->sc; a.st = c->st; a.cam = c->cam;
    a.jitter[0] = jitter ? jitter[0] : 0.f; a.jitter[1] = jitter ? jitter[1] : 0.f; a.clear_light = clear_light;
    a.g_pos = (float4 *)c->buf[EVPLP_BUF_GBUF_POSITION]; a.g_nrm = (float4 *)c->buf[EVPLP_BUF_GBUF_NORMAL];
    a.g_dif = (float4 *)c->buf[EVPLP_BUF_GBUF_DIFFUSE]; a.g_phg = (float4 *)c->buf[EVPLP_BUF_GBUF_PHONG];
    a.g_light = (float4 *)c->buf[EVPLP_BUF_LIGHT];
    a.tile_box = c->d_tile_box;
    if (!std::isfinite(a.jitter[0]) || !std::isfinite(a.jitter[1])) { c->set_error("evplp_primary: jitter is not finite"); return EVPLP_ERR_INVALID; }
    // The eye's cuts are built for a pyramid opened by ONE PIXEL (2 / W, 2 / H in NDC) around every tile group (primary_cut_kernel): they
    // hold for the reference's jitter, (2u - 1) / resolution -- half a pixel at most (rtcomphoton.h:949) -- and for anything up to a whole
    // pixel.  A larger translation (the ABI takes any float) moves rays out of their group's pyramid: that call walks from the root.
    const bool jitter_within_cuts = std::fabs(a.jitter[0]) <= 1.99f / (float)c->st.W && std::fabs(a.jitter[1]) <= 1.99f / (float)c->st.H;
    if (c->env_cuts != 0 && c->tiles_x * c->tiles_y > 0 && jitter_within_cuts) {
        // the eye's entry cuts: once per camera / tree (they hold for every jitter up to a pixel), 256 B per group of 2 x 2 tiles (2 x 1 where a
        // strip's tile rows are not neighbours in the image)
        PrimaryCutArgs pc; std::memset(&pc, 0, sizeof(pc));
        pc.nodes = c->sc.nodes; pc.st = c->st; pc.cam = c->cam; pc.tiles_x = c->tiles_x; pc.tiles_y = c->tiles_y;
        pc.gw_log2 = 1; pc.gh_log2 = (c->st.strip_count == 1 || c->st.strip_rows >= 16) ? 1 : 0;
        pc.groups_x = (c->tiles_x + (1 << pc.gw_log2) - 1) >> pc.gw_log2; pc.groups_y = (c->tiles_y + (1 << pc.gh_log2) - 1) >> pc.gh_log2;
        if (!c->d_primary_cuts) {
            hipError_t me = hipMalloc((void **)&c->d_primary_cuts, (size_t)pc.groups_x * pc.groups_y * (size_t)kCutSlotBytes);
            if (me != hipSuccess) { (void)hipGetLastError(); c->d_primary_cuts = nullptr; }
            c->primary_cuts_valid = false;
        }
        if (c->d_primary_cuts) {
            pc.cuts = c->d_primary_cuts;
            if (!c->primary_cuts_valid) { launch_primary_cuts(pc, c->stream); c->primary_cuts_valid = true; }
            a.cuts = c->d_primary_cuts; a.cut_gw_log2 = pc.gw_log2; a.cut_gh_log2 = pc.gh_log2; a.cut_groups_x = pc.groups_x;
        }
    }
    if ((rc = pass_begin(c, EVPLP_PASS_PRIMARY))) return rc;
    launch_primary(a, c->stream);
    c->tile_box_valid = !c->gbuf_pos_exposed;
    c->stats_host[EVPLP_PASS_PRIMARY].rays = 2ull * (uint64_t)c->st.W * c->rows_in_image;
    return pass_end(c, EVPLP_PASS_PRIMARY);
}

extern "C" int evplp_trace_light_paths(evplp_context *c, uint32_t rng_seed, uint32_t path_begin, uint32_t path_count) {
    CTX_CHECK(c);
    // A photon splat that is still in flight may have to run again (bins too small: settle_splat), from the records and the G-buffer
    // it was given.  Waiting for its verdict here stalls the host once per iteration -- unless these light paths go to the OTHER
    // record buffer (double buffering below): then the pending splat keeps its inputs and the next evplp_primary settles it.
    const bool to_back_buffer = c->aux_stream && path_begin == 0 && path_count == c->cfg.num_light_paths && c->buf_owned[EVPLP_BUF_RECORDS] && !c->records_exposed;
    int rc = pass_ready(c, "evplp_trace_light_paths", false, !to_back_buffer); if (rc) return rc;
    if (to_back_buffer && c->records_back && (rc = settle_readers_of(c, c->records_back, nullptr))) return rc;   // (a splat two passes old)
    if ((uint64_t)path_begin + path_count > c->cfg.num_light_paths) { c->set_error("evplp_trace_light_paths: path range exceeds num_light_paths"); return EVPLP_ERR_INVALID; }
    LightTraceArgs a; std::memset(&a, 0, sizeof(a));
    a.sc = c->sc; a.rng_seed = rng_seed; a.path_begin = path_begin; a.path_count = path_count; a.photons_per_path = c->cfg.photons_per_path;
    a.records = (evplp_record *)c->buf[EVPLP_BUF_RECORDS];
    {   // overflow columns of the walk stack (kernels.h): sized for the largest launch so far
        const size_t threads = ((size_t)path_count + 63) / 64 * 64, need = threads * (size_t)lt_overflow_entries(c->sc) * sizeof(int32_t);
        if (need > c->lt_overflow_bytes) {
            HIP_TRY(c, hipStreamSynchronize(c->stream)); if (c->aux_stream) HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
            hipFree(c->d_lt_overflow); c->d_lt_overflow = nullptr; c->lt_overflow_bytes = 0;
            hipError_t me = hipMalloc((void **)&c->d_lt_overflow, need);
            if (me != hipSuccess) { c->set_error("evplp_trace_light_paths: stack overflow area: %s", hipGetErrorString(me)); return EVPLP_ERR_OOM; }
            c->lt_overflow_bytes = need;
        }
        a.stack_overflow = c->d_lt_overflow; a.overflow_stride = (uint32_t)threads;
    }
    if (!c->aux_stream) {
        if ((rc = pass_begin(c, EVPLP_PASS_LIGHT_TRACE))) return rc;
        launch_light_trace(a, c->stream);
        return pass_end(c, EVPLP_PASS_LIGHT_TRACE);
    }
    // overlapped: behind the last reader of the records, beside whatever the main stream is doing now (the G-buffer pass of
    // this iteration), in front of everything the main stream is given from here on
    if (to_back_buffer) {
        // double buffer: write the records nobody reads any more, and make them EVPLP_BUF_RECORDS for every later call
        if (!c->records_back) {
            hipError_t me = hipMalloc(&c->records_back, buffer_bytes(c, EVPLP_BUF_RECORDS));
            if (me != hipSuccess) { c->set_error("evplp_trace_light_paths: second record buffer: %s", hipGetErrorString(me)); return EVPLP_ERR_OOM; }
        }
        std::swap(c->buf[EVPLP_BUF_RECORDS], c->records_back);
        std::swap(c->ev_records_read, c->ev_back_read);                  // (the event of the readers of the buffer that is now in front)
        a.records = (evplp_record *)c->buf[EVPLP_BUF_RECORDS];
    }
    HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_records_read, 0));
    // ... and not before the G-buffer pass most recently given to the main stream starts: a caller that calls evplp_primary first wants
    // the light paths beside IT, not beside the long gather that may still be running in front of it (they would share its CUs)
    if (c->pass_ran[EVPLP_PASS_PRIMARY]) HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_begin[EVPLP_PASS_PRIMARY], 0));
    if (c->profile_passes) HIP_TRY(c, hipEventRecord(c->ev_begin[EVPLP_PASS_LIGHT_TRACE], c->aux_stream));
    c->pass_ran[EVPLP_PASS_LIGHT_TRACE] = true; c->pass_has_dom[EVPLP_PASS_LIGHT_TRACE] = false; c->pass_timed[EVPLP_PASS_LIGHT_TRACE] = c->profile_passes;
    launch_light_trace(a, c->aux_stream);
    if (c->profile_passes) HIP_TRY(c, hipEventRecord(c->ev_end[EVPLP_PASS_LIGHT_TRACE], c->aux_stream));
    HIP_TRY(c, hipEventRecord(c->ev_light_done, c->aux_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_light_done, 0));
    HIP_TRY(c, hipGetLastError());
    return EVPLP_OK;
}

static bool small_gather_launch(const evplp_context *c) { return (size_t)c->tiles_x * (size_t)((c->rows_in_image + 7) / 8) <= 8192; }
static int fill_gather_args(evplp_context *c, const evplp_frame_params *fp, GatherArgs &a, int pass) {
    std::memset(&a, 0, sizeof(a));
    a.sc = c->sc; a.st = c->st; a.fp = *fp; a.pdf_mc2 = fp->pdf_mc * fp->pdf_mc;
    a.g_pos = (const float4 *)c->buf[EVPLP_BUF_GBUF_POSITION]; a.g_nrm = (const float4 *)c->buf[EVPLP_BUF_GBUF_NORMAL];
    a.g_dif = (const float4 *)c->buf[EVPLP_BUF_GBUF_DIFFUSE]; a.g_phg = (const float4 *)c->buf[EVPLP_BUF_GBUF_PHONG];
    a.vpls = c->d_vpls; a.vpl_src_index = c->d_vpl_src; a.nvpl = &c->d_scalars[0];
    a.out = (float4 *)c->buf[EVPLP_BUF_VPL_ACCUM];
    a.partial_stride = (size_t)c->st.W * c->st.local_rows;
    a.counters = &c->d_counters[pass];
    a.splits_per_wave = 1;
    a.block_cost = c->calibrate ? c->d_block_cost : nullptr;
    // (small_gather_launch: at most 8 192 OWNED tiles -- half a 1024 x 1024 image, a rank of a two-way partition; the capacity rows a dealt
    // partition keeps in reserve hold nothing and leave at once)
    // tiles to XCDs while an XCD's share is >= 1024 tiles (its sum of tile costs then averages out: 1.9 % spread at 1024 x 1024), items over all
    // XCDs below that (a strip of an n-way partition, small images); EVPLP_ITEM_DEAL=0 / 1 forces either (developer A/B)
    a.item_deal = c->env_item_deal >= 0 ? c->env_item_deal : (small_gather_launch(c) ? 1 : 0);
    // tile blocks: as many tile rows as a row strip keeps adjacent, at most 8
    int sh = 8;
    if (c->st.strip_count > 1) { sh = 1; while (sh * 2 <= std::min(8, c->st.strip_rows / 8) && (c->st.strip_rows / 8) % (sh * 2) == 0) sh *= 2; }
    a.block_h_log2 = sh == 8 ? 3 : sh == 4 ? 2 : sh == 2 ? 1 : 0;
    if (c->env_tile_block_log2 >= 0) a.block_h_log2 = std::min(c->env_tile_block_log2, a.block_h_log2);   // developer knob: 0 = rows of 8 tiles
    return EVPLP_OK;
}
// gather workspace (lazy: path-tracing / photon-only contexts never pay for it): per-item partial sums for `groups` groups
static int ensure_gather_workspace(evplp_context *c, GatherArgs &a, size_t groups) {
    const size_t px = (size_t)c->st.W * c->st.local_rows;
    if (c->partial_groups < groups) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_partial); c->d_partial = nullptr; c->partial_groups = 0;
        hipError_t e = hipMalloc((void **)&c->d_partial, sizeof(float4) * groups * px);
        if (e != hipSuccess) { c->set_error("gather: cannot allocate %zu bytes of partial sums: %s", sizeof(float4) * groups * px, hipGetErrorString(e)); return EVPLP_ERR_OOM; }
        c->partial_groups = groups;
    }
    a.partial = c->d_partial;
    return EVPLP_OK;
}
static int check_fp(evplp_context *c, const evplp_frame_params *fp, const char *name) {
    if (!fp) { c->set_error("%s: null frame params", name); return EVPLP_ERR_INVALID; }
    if (fp->photons_per_path != c->cfg.photons_per_path || fp->num_light_paths != c->cfg.num_light_paths || fp->num_vpl_light_paths > c->cfg.num_vpl_light_paths) {
        c->set_error("%s: frame params disagree with the configuration (paths / photons per path)", name); return EVPLP_ERR_INVALID;
    }
    if (fp->mis_mode > 5u) { c->set_error("%s: mis_mode %u out of range", name, fp->mis_mode); return EVPLP_ERR_INVALID; }
    return EVPLP_OK;
}
static size_t nvpl_slots_of(const evplp_context *c) { return std::max<size_t>((size_t)c->cfg.num_vpl_light_paths * c->cfg.photons_per_path, 1); }
static int run_gather(evplp_context *c, const evplp_frame_params *fp, bool vsl) {
    const int pass = vsl ? EVPLP_PASS_GATHER_VSL : EVPLP_PASS_GATHER_VPL;
    const char *name = vsl ? "evplp_gather_vsl" : "evplp_gather_vpl";
    int rc = pass_ready(c, name, false); if (rc) return rc;
    if ((rc = check_fp(c, fp, name))) return rc;
    if (fp->num_vpl_light_paths == 0) { c->set_error("%s: num_vpl_light_paths is 0 (the reference disables the pass, rtcomphoton.h:200-203)", name); return EVPLP_ERR_INVALID; }
    GatherArgs a; fill_gather_args(c, fp, a, pass);
    // work-item size: k consecutive splits per wavefront (fixed summation tree: the result does not depend on k).  The per-item
    // statistics need (VPLs per split) * k < 65536.
    int k = 1;
    {
        // (a row strip of an n-way partition is a SMALL launch, and small launches want short items: one rank's gather of an eight-way
        // partition of config #2 takes 9.3 / 12.1 / 17.5 ms for k = 1 / 2 / 4 where an eighth of the one-GPU kernel is 6.2 ms --
        // profiles/r05_strip_projection.json: projected 5.2x instead of 4.1x at eight ranks.  The result does not depend on k.)
        // (round 6: keyed on the size of the launch -- at most 8 192 owned tiles, the threshold of item_deal -- instead of on strip_count > 1, which
        // missed the bands of EVPLP_PARTITION_BANDS; at eight dealt ranks k = 2 projects x5.1 where k = 1 projects x6.5: an item twice as long
        // is a tail twice as long)
        const bool small_launch = small_gather_launch(c);
        k = c->cfg.gather_splits_per_wave > 0 ? c->cfg.gather_splits_per_wave : (small_launch ? 1 : kDefaultSplitsPerWave);
        if (c->env_gather_k > 0) k = c->env_gather_k;
        const size_t max_vpls = std::max<size_t>((size_t)c->cfg.num_vpl_light_paths * c->cfg.photons_per_path, 1);
        while (k > 1 && (max_vpls / kVplSplit + 1) * (size_t)k >= 65536) k >>= 1;
        if (vsl) while (k > 1 && (max_vpls / kVplSplit + 1) * (size_t)k > 4095) k >>= 1;      // (the estimator kernel's packed per-lane counters)
        if (vsl && (max_vpls / kVplSplit + 1) > 4095) {      // k = 1 and still more than 4095 VSLs per item: the packed counters would wrap
            c->set_error("%s: %zu VSL record slots exceed the %d this build can gather in one pass", name, max_vpls, 4094 * kVplSplit); return EVPLP_ERR_INVALID;
        }
    }
    a.splits_per_wave = k;
    if ((rc = ensure_gather_workspace(c, a, (size_t)(kVplSplit / k)))) return rc;
    // entry cuts (kernels.h CutArgs): one slot per (tile group, VPL slot).  Groups are 2 x 2 tiles where the strip's tile rows are
    // neighbours in the image (one GPU, or strips of 16 rows and more), 2 x 1 otherwise.  The scratch is bounded (evplp_config.
    // cut_scratch_bytes; below): a configuration whose slots need more is gathered band by band -- rows of tile blocks -- cuts first, then the walks.
    CutArgs ca; std::memset(&ca, 0, sizeof(ca));
    bool use_cuts = c->env_cuts != 0;
    const int sh = 1 << a.block_h_log2, nby = (c->tiles_y + sh - 1) / sh;       // rows of tile blocks
    int band_rows = nby;
    if (use_cuts && nby > 0) {
        ca.nodes = c->sc.nodes; ca.tile_box = c->d_tile_box; ca.tiles_x = c->tiles_x; ca.tiles_y = c->tiles_y;
        ca.gw_log2 = 1; ca.gh_log2 = (sh >= 2 && (c->st.strip_count == 1 || c->st.strip_rows >= 16)) ? 1 : 0;
        ca.groups_x = (c->tiles_x + (1 << ca.gw_log2) - 1) >> ca.gw_log2;
        ca.vpls = c->d_vpls; ca.nvpl = &c->d_scalars[0]; ca.vpl_stride = (uint32_t)nvpl_slots_of(c);
        const size_t per_block_row = (size_t)(sh >> ca.gh_log2) * ca.groups_x * ca.vpl_stride * (size_t)kCutSlotBytes;
        // The bound of the scratch: evplp_config.cut_scratch_bytes, or 8 GB (round 6; until round 5 a quarter of the device's memory -- 72 GB of
        // 288 -- which let config #5's 68 GB of slots sit in one band -- 0.7 % faster than the nine it takes now, 1 167 against 1 174 ms -- and took a quarter of the GPU from whoever else
        // lives on it: a library embedded in a host application asks for that much only when told to).  Configurations within the bound --
        // config #2 / #3: 4.3 GB -- allocate what they need; larger ones are gathered band by band.  EVPLP_CUT_BYTES: test override.
        if (c->cut_cap == 0) {
            c->cut_cap = c->env_cut_bytes ? c->env_cut_bytes : (size_t)c->cfg.cut_scratch_bytes;
            if (c->cut_cap == 0) c->cut_cap = kDefaultCutScratchBytes;
        }
        band_rows = (int)std::min<size_t>((size_t)nby, c->cut_cap / std::max<size_t>(per_block_row, 1));
        if (band_rows < 1) { use_cuts = false; band_rows = nby; }            // (the bound does not hold one row of tile blocks: walks from the root)
        const size_t need = use_cuts ? per_block_row * (size_t)band_rows : 0;
        if (use_cuts && need > c->cut_bytes) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            hipFree(c->d_cuts); c->d_cuts = nullptr; c->cut_bytes = 0;
            if (hipMalloc((void **)&c->d_cuts, need) == hipSuccess) c->cut_bytes = need;
            else { (void)hipGetLastError(); use_cuts = false; band_rows = nby; }     // (no memory for the scratch: the walks start at the root)
        }
        ca.cuts = c->d_cuts;
    } else use_cuts = false;
    if ((rc = pass_begin(c, pass))) return rc;
    const uint32_t nrec = fp->photons_per_path * fp->num_vpl_light_paths;   // lighttracing.cu:368
    launch_compact_vpl((const evplp_record *)c->buf[EVPLP_BUF_RECORDS], nrec, c->d_vpls, c->d_vpl_src, &c->d_scalars[0], c->stream);
    if (use_cuts && !c->tile_box_valid) {      // the G-buffer did not come from evplp_primary (or the caller may have written it): boxes of the tiles as they are now
        launch_tile_boxes(c->st, (const float4 *)c->buf[EVPLP_BUF_GBUF_POSITION], c->d_tile_box, c->tiles_x, c->tiles_y, c->stream);
        c->tile_box_valid = !c->gbuf_pos_exposed;
    }
    int vsl_groups = 0, vsl_per_launch = 0;
    if (vsl) {
        // lit masks between the walk and the estimator kernel: 8 bytes per (tile, VSL slot) of a launch; the groups of a tile are
        // covered in as many launches as keep the buffer within its cap (below)
        a.masks_per_split = (int32_t)((nvpl_slots_of(c) + kVplSplit - 1) / kVplSplit);
        a.band_first = 0; a.band_rows = band_rows < nby ? band_rows : 0;
        const size_t tiles = (size_t)gather_launch_tiles(a), per_item = (size_t)k * (size_t)a.masks_per_split * sizeof(unsigned long long);
        vsl_groups = kVplSplit / k;
        vsl_per_launch = vsl_groups;
        // the bound of the mask buffer: evplp_config.vsl_mask_bytes, or 2 GB (round 6; until round 5 a twentieth of the device's memory, which
        // held config #5's 8.6 GB in one launch)
        if (c->mask_cap == 0) {
            c->mask_cap = (size_t)c->cfg.vsl_mask_bytes;
            if (const char *me = std::getenv("EVPLP_MASK_BYTES")) c->mask_cap = (size_t)strtoull(me, nullptr, 10);      // (developer switch)
            if (c->mask_cap == 0) c->mask_cap = kDefaultVslMaskBytes;
        }
        const size_t mask_cap = c->mask_cap;
        while (vsl_per_launch > 1 && tiles * (size_t)vsl_per_launch * per_item > mask_cap) vsl_per_launch = (vsl_per_launch + 1) / 2;
        const size_t mask_bytes = tiles * (size_t)vsl_per_launch * per_item;
        if (c->vsl_mask_bytes < mask_bytes) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            hipFree(c->d_vsl_masks); c->d_vsl_masks = nullptr; c->vsl_mask_bytes = 0;
            hipError_t e = hipMalloc((void **)&c->d_vsl_masks, mask_bytes);
            if (e != hipSuccess) { c->set_error("gather_vsl: cannot allocate %zu bytes of lit masks: %s", mask_bytes, hipGetErrorString(e)); return EVPLP_ERR_OOM; }
            c->vsl_mask_bytes = mask_bytes;
        }
        a.vsl_masks = (unsigned long long *)c->d_vsl_masks;
    }
    // (the events bracket the cuts AND the walks: the cut kernel is part of the gather's work)
    HIP_TRY(c, hipEventRecord(c->ev_dom_begin[pass], c->stream));
    for (int b0 = 0; b0 < nby; b0 += band_rows) {
        const int rows = std::min(band_rows, nby - b0);
        a.band_first = b0; a.band_rows = band_rows < nby ? rows : 0;
        if (use_cuts) {
            const int gshift = ca.gh_log2, groups_total_y = (c->tiles_y + (1 << gshift) - 1) >> gshift;
            ca.group_row_first = (b0 * sh) >> gshift;
            ca.groups_y = std::min(((b0 + rows) * sh + (1 << gshift) - 1) >> gshift, groups_total_y) - ca.group_row_first;
            launch_gather_cuts(ca, c->stream);
            a.cuts = ca.cuts; a.cut_vpl_stride = ca.vpl_stride; a.cut_gw_log2 = ca.gw_log2; a.cut_gh_log2 = ca.gh_log2; a.cut_groups_x = ca.groups_x;
            a.cut_group_row_first = ca.group_row_first;
        }
        if (vsl) {
            for (int g0 = 0; g0 < vsl_groups; g0 += vsl_per_launch) {
                a.group_first = g0; a.group_count = std::min(vsl_per_launch, vsl_groups - g0);
                launch_gather_vsl(a, c->stream);
            }
            a.group_first = 0; a.group_count = 0;
        } else launch_gather_vpl_items(a, c->stream);
    }
    a.band_first = 0; a.band_rows = 0;
    HIP_TRY(c, hipEventRecord(c->ev_dom_end[pass], c->stream));
    launch_gather_reduce(a, vsl ? 0 : 1, c->stream);
    c->pass_has_dom[pass] = true;
    return pass_end(c, pass);
}
extern "C" int evplp_gather_vpl(evplp_context *c, const evplp_frame_params *fp) { CTX_CHECK(c); return run_gather(c, fp, false); }
extern "C" int evplp_gather_vsl(evplp_context *c, const evplp_frame_params *fp) { CTX_CHECK(c); return run_gather(c, fp, true); }

// lvclighttracing.cu:348-384: the window covers num_vpl_light_paths paths of ALL num_light_paths * P record slots
extern "C" int evplp_gather_lvc(evplp_context *c, const evplp_frame_params *fp) {
    CTX_CHECK(c);
    const int pass = EVPLP_PASS_GATHER_LVC;
    int rc = pass_ready(c, "evplp_gather_lvc", false); if (rc) return rc;
    if ((rc = check_fp(c, fp, "evplp_gather_lvc"))) return rc;
    if (fp->num_vpl_light_paths == 0) { c->set_error("evplp_gather_lvc: num_vpl_light_paths is 0"); return EVPLP_ERR_INVALID; }
    GatherArgs a; fill_gather_args(c, fp, a, pass);
    if ((rc = pass_begin(c, pass))) return rc;
    launch_gather_lvc(a, (const evplp_record *)c->buf[EVPLP_BUF_RECORDS], c->stream);
    return pass_end(c, pass);
}

extern "C" int evplp_path_trace(evplp_context *c, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces, int32_t do_accumulate) {
    CTX_CHECK(c);
    int rc = pass_ready(c, "evplp_path_trace", false); if (rc) return rc;
    if (!camera_pos) { c->set_error("evplp_path_trace: null camera position"); return EVPLP_ERR_INVALID; }
    PathTraceArgs a; std::memset(&a, 0, sizeof(a));
    a.sc = c->sc; a.st = c->st;
    a.g_pos = (const float4 *)c->buf[EVPLP_BUF_GBUF_POSITION]; a.g_nrm = (const float4 *)c->buf[EVPLP_BUF_GBUF_NORMAL];
    a.g_dif = (const float4 *)c->buf[EVPLP_BUF_GBUF_DIFFUSE]; a.g_phg = (const float4 *)c->buf[EVPLP_BUF_GBUF_PHONG];
    for (int k = 0; k < 3; k++) a.camera_pos[k] = camera_pos[k];
    a.rng_seed = rng_seed; a.max_bounces = max_bounces; a.do_accumulate = do_accumulate ? 1u : 0u;
    a.out = (float4 *)c->buf[EVPLP_BUF_VPL_ACCUM];
    a.counters = &c->d_counters[EVPLP_PASS_PATH_TRACE];
    if ((rc = pass_begin(c, EVPLP_PASS_PATH_TRACE))) return rc;
    launch_path_trace(a, c->stream);
    return pass_end(c, EVPLP_PASS_PATH_TRACE);
}

// setupPhotonSplatIcosohedron (rtcomphoton.h:632-644): the proxy mesh of EVPLP_FOOTPRINT_PROXY -> slabs on the device
static int upload_proxy(evplp_context *c, const float *vertices, int32_t nverts, const int32_t *indices, int32_t ntris, const char *name) {
    ProxyHost ph; std::string why;
    if (!build_proxy_slabs(vertices, nverts, indices, ntris, &ph, &why)) { c->set_error("%s: %s", name, why.c_str()); return EVPLP_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));                          // (a splat in flight reads the old slabs)
    if (!c->d_proxy_slabs) {
        HIP_TRY(c, hipMalloc((void **)&c->d_proxy_slabs, sizeof(float4) * kMaxProxySlabs));
        HIP_TRY(c, hipMalloc((void **)&c->d_proxy_hm, sizeof(float) * kMaxProxySlabs));
    }
    HIP_TRY(c, hipMemcpy(c->d_proxy_slabs, ph.slabs.data(), sizeof(float4) * ph.slabs.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_proxy_hm, ph.hm.data(), sizeof(float) * ph.hm.size(), hipMemcpyHostToDevice));
    c->proxy_count = (int32_t)ph.slabs.size(); c->proxy_rin = ph.rin; c->proxy_rout = ph.rout;
    return EVPLP_OK;
}
static int default_proxy(evplp_context *c, const char *name) {
    std::vector<float> v; std::vector<int32_t> t;
    default_splat_proxy(v, t);
    return upload_proxy(c, v.data(), (int32_t)(v.size() / 3), t.data(), (int32_t)(t.size() / 3), name);
}
extern "C" int evplp_set_splat_proxy(evplp_context *c, const float *vertices, int32_t nverts, const int32_t *indices, int32_t ntris) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (!vertices) return default_proxy(c, "evplp_set_splat_proxy");
    return upload_proxy(c, vertices, nverts, indices, ntris, "evplp_set_splat_proxy");
}
extern "C" int evplp_default_splat_proxy(float *vertices, int32_t *indices) {
    std::vector<float> v; std::vector<int32_t> t;
    default_splat_proxy(v, t);
    if (vertices) std::memcpy(vertices, v.data(), sizeof(float) * v.size());
    if (indices) std::memcpy(indices, t.data(), sizeof(int32_t) * t.size());
    return (int)(t.size() / 3);
}

extern "C" int evplp_splat_photons(evplp_context *c, const evplp_frame_params *fp, int32_t clear) {
    CTX_CHECK(c);
    // (overlapped mode, accumulating: the previous pass may stay pending -- only look whether its verdict has arrived; a clearing pass
    // settles it first, its re-run would otherwise land in the cleared buffer)
    // Deterministic mode never leaves a pass pending behind a younger one: the re-run of an overflowed pass would otherwise be
    // enqueued before or after the younger pass depending on when its verdict arrives, and the float sums into the photon
    // accumulator would differ between runs.
    const bool relaxed = c->aux_stream && !clear && !c->cfg.deterministic;
    int rc = pass_ready(c, "evplp_splat_photons", true, !relaxed); if (rc) return rc;
    if (relaxed) {
        while (c->npend >= 2) if ((rc = settle_one(c))) return rc;
        while (c->npend > 0 && hipEventQuery(c->pend[0].ev) == hipSuccess) if ((rc = settle_one(c))) return rc;
    }
    if ((rc = check_fp(c, fp, "evplp_splat_photons"))) return rc;
    if (!(fp->photon_radius > 0.0f)) { c->set_error("evplp_splat_photons: photon_radius must be > 0"); return EVPLP_ERR_INVALID; }
    if (fp->splat_footprint > (uint32_t)EVPLP_FOOTPRINT_PROXY) { c->set_error("evplp_splat_photons: splat_footprint %u out of range", fp->splat_footprint); return EVPLP_ERR_INVALID; }
    if (fp->splat_footprint == (uint32_t)EVPLP_FOOTPRINT_PROXY && c->proxy_count == 0 && (rc = default_proxy(c, "evplp_splat_photons"))) return rc;
    SplatArgs a; std::memset(&a, 0, sizeof(a));
    a.st = c->st; a.cam = c->cam; a.fp = *fp;
    a.g_pos = (const float4 *)c->buf[EVPLP_BUF_GBUF_POSITION]; a.g_nrm = (const float4 *)c->buf[EVPLP_BUF_GBUF_NORMAL];
    a.g_dif = (const float4 *)c->buf[EVPLP_BUF_GBUF_DIFFUSE]; a.g_phg = (const float4 *)c->buf[EVPLP_BUF_GBUF_PHONG];
    a.records = (const evplp_record *)c->buf[EVPLP_BUF_RECORDS];
    a.num_records = c->cfg.num_light_paths * c->cfg.photons_per_path;   // instances = numLightPaths * P (:832)
    a.out = (float4 *)c->buf[EVPLP_BUF_PHOTON_ACCUM];
    a.tile_box = c->d_tile_box; a.tile_pairs = c->d_tile_pairs; a.tile_cursor = c->d_tile_cursor;
    a.bin_items = c->d_bin_items; a.bin_items_tmp = c->d_bin_items_tmp; a.bin_stride = c->bin_stride;
    a.compact = c->d_compact; a.overflow = &c->d_scalars[8]; a.summary = c->d_summary;
    a.seg = c->d_seg; a.seg_off = c->d_seg_off; a.big_list = c->d_big_list; a.big_count = c->d_big_count; a.num_bin_groups = c->num_bin_groups;
    a.bucket_w_log2 = c->bucket_w_log2; a.bucket_h_log2 = c->bucket_h_log2; a.buckets_x = c->buckets_x; a.num_buckets = c->num_buckets;
    a.tiles_x = c->tiles_x; a.tiles_y = c->tiles_y; a.deterministic = c->cfg.deterministic; a.boxes_valid = c->tile_box_valid ? 1 : 0;
    a.counters = &c->d_counters[EVPLP_PASS_SPLAT];
    a.proxy_slabs = c->d_proxy_slabs; a.proxy_hm = c->d_proxy_hm; a.proxy_count = c->proxy_count; a.proxy_rin = c->proxy_rin; a.proxy_rout = c->proxy_rout;
    a.tile_frags = c->d_tile_frags; c->last_splat_proxy = fp->splat_footprint == (uint32_t)EVPLP_FOOTPRINT_PROXY;
    // MIXED tile launch when the previous pass had bins at least twice the heavy threshold (a heuristic, like the choice between the pure
    // variants it replaces): below that the heavy list stays nearly empty and its bookkeeping costs more than four waves save
    if (c->env_tile_mixed && c->last_bin_max >= c->mixed_trigger) { a.heavy_list = c->d_heavy_list; a.tile_flags = c->d_tile_flags; a.heavy_cap = c->heavy_cap; a.heavy_threshold = c->heavy_threshold; }
    if ((rc = pass_begin(c, EVPLP_PASS_SPLAT))) return rc;
    if (clear) HIP_TRY(c, hipMemsetAsync(c->buf[EVPLP_BUF_PHOTON_ACCUM], 0, buffer_bytes(c, EVPLP_BUF_PHOTON_ACCUM), c->stream));
    launch_splat_bin(a, c->stream);                                       // (clears the overflow flag, the cursors and the summary)
    // Tile kernel variant.  (Round 5: when the previous pass had bins of >= mixed_trigger entries the launch is MIXED -- a.heavy_list above,
    // kernels_splat.hip -- and what follows only decides for deterministic contexts and EVPLP_TILE_MIXED=0.)
    // One wave per tile is cheapest while bins are short; when some bins are very full (tiles
    // that see a floor at grazing angle collect thousands of photons) those waves set the duration of the launch and
    // four waves per tile win.  Deterministic mode always uses one variant: the fold order is part of the result.
    // Measured (tiles kernel, ms): fullest bin 1405 entries (config #3): 0.31 with one wave, 0.18 with four; fullest bin 365
    // (config #4 shape): 0.12 / 0.22.  The fullest bin of the PREVIOUS pass decides (a heuristic either way).
    const bool split_tiles = c->cfg.deterministic ? true : c->last_bin_max >= split_threshold(c, fp->splat_footprint);
    const bool dom = c->profile_kernels;
    launch_splat_tiles(a, split_tiles, c->stream, dom ? c->ev_dom_begin[EVPLP_PASS_SPLAT] : nullptr, dom ? c->ev_dom_end[EVPLP_PASS_SPLAT] : nullptr);
    // The number of (photon, tile) bin entries depends on the photon set and the radius and is known on the device only.  The
    // whole pass is enqueued now; the summary travels to pinned memory and is checked by the next call (settle_splat).
    evplp_context::PendingSplat &slot = c->pend[c->npend];
    HIP_TRY(c, hipMemcpyAsync(slot.h, &c->d_summary[kSummaryFinal], 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));   // entries, fullest bin, overflow
    HIP_TRY(c, hipEventRecord(slot.ev, c->stream));
    slot.args = a; c->npend++;
    c->pass_has_dom[EVPLP_PASS_SPLAT] = dom;
    return pass_end(c, EVPLP_PASS_SPLAT);
}

// [finalize] composite of this context's strip into d_rgb (device); evplp_resolve downloads it, the group all-gathers it
namespace evplp {
// settle = false (the per-iteration composite of a running loop): the composite is enqueued behind the last splat without waiting for
// the verdict on its bins -- the host does not stall; in the rare iteration whose bins overflowed the presented frame lacks that one
// pass (it is run again and lands in the accumulator before the next composite).  Results that leave the device always settle.
int resolve_to_device(evplp_context *c, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, bool settle) {
    CTX_CHECK(c);
    if (settle) { int rc_ = settle_splat(c); if (rc_) return rc_; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc;
    if ((rc = pass_begin(c, EVPLP_PASS_RESOLVE))) return rc;
    launch_resolve(c->st, (const float4 *)c->buf[EVPLP_BUF_VPL_ACCUM], (const float4 *)c->buf[EVPLP_BUF_PHOTON_ACCUM],
                   (const float4 *)c->buf[EVPLP_BUF_LIGHT], vs, ps, ls, mask_emitter, gamma, c->d_rgb, c->stream);
    return pass_end(c, EVPLP_PASS_RESOLVE);
}
} // namespace evplp

extern "C" int evplp_resolve(evplp_context *c, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, float *out_rgb) {
    CTX_CHECK(c);
    if (!out_rgb) { c->set_error("evplp_resolve: null output"); return EVPLP_ERR_INVALID; }
    int rc = evplp::resolve_to_device(c, vs, ps, ls, mask_emitter, gamma, true);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_rgb, c->d_rgb, sizeof(float) * 3 * (size_t)c->st.W * c->st.local_rows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EVPLP_OK;
}

extern "C" int evplp_present(evplp_context *c, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma) {
    CTX_CHECK(c);
    return evplp::resolve_to_device(c, vs, ps, ls, mask_emitter, gamma, !c->aux_stream);      // (overlapped contexts keep the host an iteration ahead)
}

static void count_rows_in_image(evplp_context *c) {
    c->rows_in_image = 0;
    for (int l = 0; l < c->st.local_rows; l++) if (c->st.global_row(l) < c->st.H) c->rows_in_image++;
}
extern "C" int evplp_set_band(evplp_context *c, int32_t first_row, int32_t rows) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (c->st.band_rows <= 0) { c->set_error("evplp_set_band: the context was not created in band mode"); return EVPLP_ERR_INVALID; }
    if (first_row < 0 || (first_row % 16) != 0 || first_row >= c->st.H || rows <= 0 || rows > c->st.local_rows ||
        ((rows % 16) != 0 && first_row + rows < c->st.H) || first_row + rows > ((c->st.H + 15) / 16) * 16) {
        c->set_error("evplp_set_band: rows [%d, %d) are not a band this context can hold (capacity %d rows, multiples of 16)", first_row, first_row + rows, c->st.local_rows); return EVPLP_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); if (c->aux_stream) HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    c->st.band_first = first_row; c->st.band_rows = std::min(rows, c->st.H - first_row);
    c->cfg.band_first_row = first_row; c->cfg.band_rows = rows;
    count_rows_in_image(c);
    c->primary_cuts_valid = false; c->tile_box_valid = false;
    return evplp_clear_accumulators(c);
}

// ---- dealt blocks (include/evplp.h): the owned-block table of a row-strip context, and the per-block cost the deal is made from
extern "C" int evplp_set_blocks(evplp_context *c, const int32_t *image_blocks, int32_t count) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (c->st.strip_count <= 1 || c->st.band_rows > 0) { c->set_error("evplp_set_blocks: the context is not a row-strip context (strip_count > 1)"); return EVPLP_ERR_INVALID; }
    const int cap = c->st.cap_blocks, nb = c->image_blocks;
    if (image_blocks && (count < 0 || count > cap)) { c->set_error("evplp_set_blocks: %d blocks do not fit the context's %d (strip_capacity_rows)", count, cap); return EVPLP_ERR_INVALID; }
    std::vector<int32_t> table;
    if (image_blocks) {
        table.assign((size_t)cap + nb, -1);
        for (int l = 0; l < cap; l++) table[(size_t)l] = nb + l;                   // holds nothing: rows >= H
        for (int l = 0; l < count; l++) {
            const int b = image_blocks[l];
            if (b < 0 || b >= nb || table[(size_t)cap + b] >= 0) { c->set_error("evplp_set_blocks: block %d is outside the image's %d blocks or listed twice", b, nb); return EVPLP_ERR_INVALID; }
            table[(size_t)l] = b; table[(size_t)cap + b] = l;
        }
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); if (c->aux_stream) HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    if (image_blocks) {
        if (!c->d_blocks) HIP_TRY(c, hipMalloc((void **)&c->d_blocks, sizeof(int32_t) * ((size_t)cap + nb)));
        HIP_TRY(c, hipMemcpy(c->d_blocks, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice));
        c->blocks_host.swap(table);
        c->st.blocks = c->d_blocks; c->st.blocks_host = c->blocks_host.data();
    } else { c->st.blocks = nullptr; c->st.blocks_host = nullptr; c->blocks_host.clear(); }       // back to block b -> rank b % strip_count
    count_rows_in_image(c);
    c->primary_cuts_valid = false; c->tile_box_valid = false;
    return evplp_clear_accumulators(c);
}
extern "C" int evplp_get_blocks(evplp_context *c, int32_t *image_blocks, int32_t capacity) {
    CTX_CHECK(c);
    int n = 0;
    const int cap = c->st.band_rows > 0 ? 0 : c->st.local_rows / c->st.strip_rows;
    for (int l = 0; l < cap; l++) {
        const int b = c->st.global_block(l);
        if (b >= c->image_blocks) continue;
        if (image_blocks && n < capacity) image_blocks[n] = b;
        n++;
    }
    return n;
}
extern "C" int evplp_calibrate_blocks(evplp_context *c, int32_t on) {
    CTX_CHECK(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t bytes = sizeof(unsigned long long) * (size_t)std::max(c->st.local_rows / std::max(c->st.strip_rows, 1), 1);
    if (on) {
        if (!c->d_block_cost) HIP_TRY(c, hipMalloc((void **)&c->d_block_cost, bytes));
        HIP_TRY(c, hipMemsetAsync(c->d_block_cost, 0, bytes, c->stream));
    }
    c->calibrate = on != 0;
    return EVPLP_OK;
}
extern "C" int evplp_block_costs(evplp_context *c, uint64_t *cost_per_image_block, int32_t capacity) {
    CTX_CHECK(c);
    if (!cost_per_image_block || capacity < c->image_blocks) { c->set_error("evplp_block_costs: the output needs room for %d blocks", c->image_blocks); return EVPLP_ERR_INVALID; }
    if (!c->d_block_cost) { c->set_error("evplp_block_costs: no calibration has run (evplp_calibrate_blocks)"); return EVPLP_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int cap = c->st.local_rows / c->st.strip_rows;
    std::vector<unsigned long long> local((size_t)cap);
    HIP_TRY(c, hipMemcpyAsync(local.data(), c->d_block_cost, sizeof(unsigned long long) * local.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < c->image_blocks; b++) cost_per_image_block[b] = 0;
    int n = 0;
    for (int l = 0; l < cap; l++) {
        const int b = c->st.band_rows > 0 ? 0 : c->st.global_block(l);
        if (b < c->image_blocks) { cost_per_image_block[b] += local[(size_t)l]; n++; }
    }
    return n;
}

extern "C" int evplp_clear_accumulators(evplp_context *c) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipMemsetAsync(c->buf[EVPLP_BUF_VPL_ACCUM], 0, buffer_bytes(c, EVPLP_BUF_VPL_ACCUM), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->buf[EVPLP_BUF_PHOTON_ACCUM], 0, buffer_bytes(c, EVPLP_BUF_PHOTON_ACCUM), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->buf[EVPLP_BUF_LIGHT], 0, buffer_bytes(c, EVPLP_BUF_LIGHT), c->stream));
    return EVPLP_OK;
}

// ------------------------------------------------------------------------ buffers / stats
extern "C" int evplp_local_rows(const evplp_context *c) { return c ? c->st.local_rows : EVPLP_ERR_INVALID; }

extern "C" int evplp_buffer_info(evplp_context *c, int32_t which, void **ptr, size_t *bytes) {
    CTX_CHECK(c);
    if (which < 0 || which >= EVPLP_BUF_COUNT) { c->set_error("evplp_buffer_info: bad buffer id %d", which); return EVPLP_ERR_INVALID; }
    if (ptr) {
        *ptr = c->buf[which];
        if (which == EVPLP_BUF_GBUF_POSITION) { c->gbuf_pos_exposed = true; c->tile_box_valid = false; }
        if (which >= EVPLP_BUF_GBUF_POSITION && which <= EVPLP_BUF_GBUF_PHONG) c->gbuf_exposed = true;
        if (which == EVPLP_BUF_RECORDS) c->records_exposed = true;       // the pointer must stay the record buffer: no double buffering
    }
    if (bytes) *bytes = buffer_bytes(c, which);
    return EVPLP_OK;
}
extern "C" int evplp_bind_buffer(evplp_context *c, int32_t which, void *ptr, size_t bytes) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (which < 0 || which >= EVPLP_BUF_COUNT || !ptr) { c->set_error("evplp_bind_buffer: bad arguments"); return EVPLP_ERR_INVALID; }
    if (bytes < buffer_bytes(c, which)) { c->set_error("evplp_bind_buffer: %zu bytes given, %zu needed", bytes, buffer_bytes(c, which)); return EVPLP_ERR_INVALID; }
    if (((uintptr_t)ptr & 15u) != 0) { c->set_error("evplp_bind_buffer: pointer must be 16-byte aligned"); return EVPLP_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->buf_owned[which]) hipFree(c->buf[which]);
    c->buf[which] = ptr; c->buf_owned[which] = false;
    if (which == EVPLP_BUF_GBUF_POSITION) { c->gbuf_pos_exposed = true; c->tile_box_valid = false; }
    if (which >= EVPLP_BUF_GBUF_POSITION && which <= EVPLP_BUF_GBUF_PHONG) c->gbuf_exposed = true;
    return EVPLP_OK;
}
extern "C" int evplp_download(evplp_context *c, int32_t which, void *dst, size_t bytes) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (which < 0 || which >= EVPLP_BUF_COUNT || !dst || bytes > buffer_bytes(c, which)) { c->set_error("evplp_download: bad arguments"); return EVPLP_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipMemcpyAsync(dst, c->buf[which], bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EVPLP_OK;
}
extern "C" int evplp_upload(evplp_context *c, int32_t which, const void *src, size_t bytes) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (which < 0 || which >= EVPLP_BUF_COUNT || !src || bytes > buffer_bytes(c, which)) { c->set_error("evplp_upload: bad arguments"); return EVPLP_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipMemcpyAsync(c->buf[which], src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (which == EVPLP_BUF_GBUF_POSITION) c->tile_box_valid = false;
    return EVPLP_OK;
}

// raw device counters of a pass (diagnostic builds fill the histogram part; see kernels.h PassCounters)
extern "C" int evplp_profile_kernels(evplp_context *c, int32_t on) { CTX_CHECK(c); c->profile_kernels = on != 0; return EVPLP_OK; }
extern "C" int evplp_profile_passes(evplp_context *c, int32_t on) { CTX_CHECK(c); c->profile_passes = on != 0; return EVPLP_OK; }

extern "C" int evplp_debug_counters(evplp_context *c, int32_t pass, uint64_t *out, int32_t capacity) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (pass < 0 || pass >= EVPLP_PASS_COUNT || !out || capacity <= 0) { c->set_error("evplp_debug_counters: bad arguments"); return EVPLP_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    PassCounters pc;
    HIP_TRY(c, hipMemcpy(&pc, &c->d_counters[pass], sizeof(pc), hipMemcpyDeviceToHost));
    const int n = std::min<int>(capacity, (int)(sizeof(pc) / sizeof(uint64_t)));
    std::memcpy(out, &pc, sizeof(uint64_t) * (size_t)n);
    return n;
}

extern "C" int evplp_pass_stats_get(evplp_context *c, int32_t pass, evplp_pass_stats *out) {
    CTX_CHECK(c);
    { int rc_ = settle_splat(c); if (rc_) return rc_; }
    if (pass < 0 || pass >= EVPLP_PASS_COUNT || !out) { c->set_error("evplp_pass_stats_get: bad arguments"); return EVPLP_ERR_INVALID; }
    std::memset(out, 0, sizeof(*out));
    if (!c->pass_ran[pass]) return EVPLP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->pass_timed[pass]) {
        HIP_TRY(c, hipEventSynchronize(c->ev_end[pass]));
        HIP_TRY(c, hipEventElapsedTime(&out->ms, c->ev_begin[pass], c->ev_end[pass]));
    } else {                                                              // (run with evplp_profile_passes off: counters only)
        HIP_TRY(c, hipStreamSynchronize(c->stream)); if (c->aux_stream) HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    }
    if (c->pass_has_dom[pass]) { HIP_TRY(c, hipEventElapsedTime(&out->dominant_kernel_ms, c->ev_dom_begin[pass], c->ev_dom_end[pass])); out->launches = 1; }
    else { out->dominant_kernel_ms = out->ms; out->launches = 1; }
    PassCounters pc; uint32_t scal[16];
    HIP_TRY(c, hipMemcpy(&pc, &c->d_counters[pass], sizeof(pc), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(scal, c->d_scalars, sizeof(scal), hipMemcpyDeviceToHost));
    const uint64_t px = (uint64_t)c->st.W * c->rows_in_image;
    if (pass == EVPLP_PASS_GATHER_VPL || pass == EVPLP_PASS_GATHER_VSL) {
        unsigned long long rays = 0, shaded = 0;
        for (int k = 0; k < kCounterShards; k++) { rays += pc.shard_rays[k]; shaded += pc.shard_shaded[k]; }
        out->usable = scal[0]; out->pairs = px * scal[0]; out->rays = rays; out->shaded = shaded; out->reserved[0] = (uint32_t)std::min<unsigned long long>(pc.nodes, 0xffffffffull);
        out->reserved[1] = (uint32_t)(pc.nodes >> 32);
        if (pass == EVPLP_PASS_GATHER_VSL && !EVPLP_TRAVERSAL_STATS) {     // VSL: sample-iterations of the estimators (64 shards), in the same two words
            unsigned long long samples = 0; for (int k = 0; k < 64; k++) samples += pc.hist[k];
            out->reserved[0] = (uint32_t)samples; out->reserved[1] = (uint32_t)(samples >> 32);
        }
    } else if (pass == EVPLP_PASS_SPLAT) {
        // per-tile pair counts (a single counter word would take one device-scope atomic per tile: measured 0.28 ms of
        // a 0.41 ms launch at 1920x1080), summed here
        std::vector<uint32_t> tp((size_t)c->tiles_x * c->tiles_y);
        HIP_TRY(c, hipMemcpy(tp.data(), c->d_tile_pairs, tp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t sum = 0; for (uint32_t v : tp) sum += v;
        uint64_t fragments = 0;
        if (c->last_splat_proxy) {                                        // EVPLP_FOOTPRINT_PROXY: fragments of the proxy mesh
            HIP_TRY(c, hipMemcpy(tp.data(), c->d_tile_frags, tp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t v : tp) fragments += v;
        }
        out->pairs = sum; out->rays = fragments; out->usable = 0; out->reserved[0] = c->last_bin_entries; out->reserved[1] = c->last_bin_max;
        // `shaded`: (photon, pixel) pairs of ALL splat passes of this context so far (device-side running total)
        std::vector<uint32_t> sh((size_t)kSummaryFinal);
        HIP_TRY(c, hipMemcpy(sh.data(), c->d_summary, sh.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (int k = 0; k < kSummaryShards; k++) total += (uint64_t)sh[(size_t)k * kSummaryStride + 4] | ((uint64_t)sh[(size_t)k * kSummaryStride + 5] << 32);
        out->shaded = total;
    } else if (pass == EVPLP_PASS_PATH_TRACE) { out->pairs = pc.pairs; out->rays = pc.rays; }
    else if (pass == EVPLP_PASS_GATHER_LVC) { out->pairs = pc.pairs; out->rays = pc.rays; }
    else if (pass == EVPLP_PASS_PRIMARY) out->rays = c->stats_host[pass].rays;
    return EVPLP_OK;
}
