// evplp_group: the multi-GPU entry of the C ABI (SURVEY 8b "Threading", 8e).  One caller thread drives n contexts -- one per GPU
// of the node -- that own interleaved row strips of the image (include/evplp.h); the scene and the BVH are replicated; large
// light-path sets are traced 1/n per rank and shared by an in-place all-gather of the record buffers; every rank gathers /
// splats its own rows; the composited strips are all-gathered so that every GPU holds the frame.  No other exchange exists on
// the path.  The collectives are RCCL (ncclAllGather over xGMI, one communicator per GPU, issued between ncclGroupStart / End by
// the one host thread); the library is opened at run time so that hosts without it can still use single contexts.  Ranks that
// share one device ("virtual ranks": tests, single-GPU boxes) exchange by device-to-device copies instead.
#include "context.hpp"

#include <rccl/rccl.h>      // types only: the entry points are resolved with dlsym

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

namespace evplp {
int resolve_to_device(evplp_context *c, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, bool settle);   // context.cpp
bool host_would_wait(evplp_context *c);                                                                        // context.cpp
}

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool open(std::string &err) {
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot open RCCL (librccl.so.1): ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather"); GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) { err = "librccl lacks an expected entry point"; return false; }
        return true;
    }
};
} // namespace

struct evplp_group {
    int n = 0;
    std::vector<evplp_context *> ctx;
    std::vector<int> device;
    bool virtual_ranks = false;             // all ranks on one device: exchange by copies
    Rccl rccl; std::vector<ncclComm_t> comms;
    std::vector<float *> d_frame;           // per rank: [n][local_rows * W * 3] the all-gathered composite
    float *d_assembled = nullptr;           // rank 0's device: [H][W][3] the frame in image order (evplp_group_resolve)
    size_t strip_floats = 0;                // local_rows * W * 3
    bool split_paths = false; uint32_t per_rank_paths = 0;
    char error[512] = "";
    void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(error, sizeof(error), fmt, ap); va_end(ap); }
};

static thread_local char g_group_create_error[512] = "";

#define GRP_CHECK(g) do { if (!(g)) return EVPLP_ERR_INVALID; } while (0)
// Every entry point of a context first looks at the verdict of its last photon splat (context.cpp settle_splat) and may wait for
// it.  One host thread feeds all ranks, so the ranks whose verdict has already arrived are fed FIRST and the ones that would make
// the host wait last: no GPU idles behind another rank's wait.  (The ranks are independent contexts; the order of the calls does
// not change any result.)
static void feed_order(const evplp_group *g, int *order) {
    int m = 0; bool late[64];
    for (int r = 0; r < g->n; r++) { late[r] = evplp::host_would_wait(g->ctx[r]); if (!late[r]) order[m++] = r; }
    for (int r = 0; r < g->n; r++) if (late[r]) order[m++] = r;
}
#define GRP_EACH(g, call) do { int order_[64]; feed_order((g), order_); for (int i_ = 0; i_ < (g)->n; i_++) { const int r_ = order_[i_]; evplp_context *c = (g)->ctx[r_]; int rc_ = (call); if (rc_ < 0) { (g)->set_error("rank %d: %s", r_, evplp_last_error(c)); return rc_; } } } while (0)

extern "C" const char *evplp_group_last_error(const evplp_group *g) { return g ? g->error : g_group_create_error; }
extern "C" int evplp_group_size(const evplp_group *g) { return g ? g->n : EVPLP_ERR_INVALID; }
extern "C" evplp_context *evplp_group_context(evplp_group *g, int32_t rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }

extern "C" void evplp_group_destroy(evplp_group *g) {
    if (!g) return;
    for (int r = 0; r < (int)g->d_frame.size(); r++) if (g->d_frame[r]) { hipSetDevice(g->device[r]); hipFree(g->d_frame[r]); }
    if (g->d_assembled) { hipSetDevice(g->device[0]); hipFree(g->d_assembled); }
    for (ncclComm_t c : g->comms) if (c && g->rccl.CommDestroy) g->rccl.CommDestroy(c);
    for (evplp_context *c : g->ctx) evplp_destroy(c);
    delete g;
}

extern "C" int evplp_group_create(const evplp_config *cfg, const evplp_group_config *gc, evplp_group **out) {
    auto fail = [&](int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_group_create_error, sizeof(g_group_create_error), fmt, ap); va_end(ap); return code; };
    if (!cfg || !gc || !out) return fail(EVPLP_ERR_INVALID, "evplp_group_create: null argument");
    *out = nullptr;
    if (gc->n_ranks < 1 || gc->n_ranks > 64) return fail(EVPLP_ERR_INVALID, "evplp_group_create: n_ranks must be 1..64");
    evplp_group *g = new evplp_group();
    g->n = gc->n_ranks;
    for (int r = 0; r < g->n; r++) g->device.push_back(gc->devices ? gc->devices[r] : r);
    bool all_same = true, all_distinct = true;
    for (int r = 0; r < g->n; r++) for (int q = 0; q < r; q++) { if (g->device[r] == g->device[q]) all_distinct = false; else all_same = false; }
    if (g->n > 1 && !all_same && !all_distinct) { delete g; return fail(EVPLP_ERR_INVALID, "evplp_group_create: the ranks' devices must be all distinct (RCCL) or all the same (virtual ranks)"); }
    g->virtual_ranks = g->n > 1 ? all_same : !gc->use_rccl;
    if (gc->use_rccl && g->n > 1 && !all_distinct) { delete g; return fail(EVPLP_ERR_INVALID, "evplp_group_create: RCCL needs one distinct device per rank"); }
    const int strip_rows = gc->strip_rows > 0 ? gc->strip_rows : 8;
    for (int r = 0; r < g->n; r++) {
        evplp_config c = *cfg;
        c.device = g->device[r]; c.strip_rank = r; c.strip_count = g->n; c.strip_rows = strip_rows;
        evplp_context *h = nullptr;
        int rc = evplp_create(&c, &h);
        if (rc < 0) { int code = fail(rc, "rank %d: %s", r, evplp_last_error(nullptr)); evplp_group_destroy(g); return code; }
        g->ctx.push_back(h);
    }
    g->strip_floats = (size_t)g->ctx[0]->st.local_rows * g->ctx[0]->st.W * 3;
    g->d_frame.assign((size_t)g->n, nullptr);
    for (int r = 0; r < g->n; r++) {
        hipSetDevice(g->device[r]);
        hipError_t e = hipMalloc((void **)&g->d_frame[r], sizeof(float) * g->strip_floats * (size_t)g->n);
        if (e != hipSuccess) { int code = fail(EVPLP_ERR_OOM, "rank %d: hipMalloc(frame): %s", r, hipGetErrorString(e)); evplp_group_destroy(g); return code; }
    }
    if (!g->virtual_ranks) {
        std::string err;
        if (!g->rccl.open(err)) { int code = fail(EVPLP_ERR_NO_DEVICE, "%s", err.c_str()); evplp_group_destroy(g); return code; }
        g->comms.assign((size_t)g->n, nullptr);
        ncclResult_t nr = g->rccl.CommInitAll(g->comms.data(), g->n, g->device.data());
        if (nr != ncclSuccess) { int code = fail(EVPLP_ERR_HIP, "ncclCommInitAll: %s", g->rccl.GetErrorString(nr)); g->comms.clear(); evplp_group_destroy(g); return code; }
    }
    // a light-tracing launch is latency-bound (0.26 ms for 1024 paths, 0.25 ms for 128): small path counts are traced redundantly by
    // every rank (identical records, no exchange); large ones are split by path range and shared by one all-gather
    g->split_paths = g->n > 1 && cfg->num_light_paths % (uint32_t)g->n == 0 && cfg->num_light_paths >= 16384;
    g->per_rank_paths = g->split_paths ? cfg->num_light_paths / (uint32_t)g->n : cfg->num_light_paths;
    *out = g;
    return EVPLP_OK;
}

// all-gather of equal chunks: rank r contributes `count` floats at send[r] and receives n * count floats at recv[r]
static int group_all_gather(evplp_group *g, const std::vector<const float *> &send, const std::vector<float *> &recv, size_t count) {
    if (g->n == 1 && g->virtual_ranks) {       // one rank: its chunk goes to its place in stream order, the host does not wait
        if (recv[0] != send[0]) { hipError_t e = hipMemcpyAsync(recv[0], send[0], count * sizeof(float), hipMemcpyDeviceToDevice, g->ctx[0]->stream); if (e != hipSuccess) { g->set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; } }
        return EVPLP_OK;
    }
    if (!g->virtual_ranks) {
        ncclResult_t nr = g->rccl.GroupStart();
        for (int r = 0; r < g->n && nr == ncclSuccess; r++) nr = g->rccl.AllGather(send[r], recv[r], count, ncclFloat, g->comms[r], g->ctx[r]->stream);
        ncclResult_t ne = g->rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) { g->set_error("ncclAllGather: %s", g->rccl.GetErrorString(nr)); return EVPLP_ERR_HIP; }
        return EVPLP_OK;
    }
    // virtual ranks share a device: every producer finishes, then plain device copies on the receivers' streams
    for (int r = 0; r < g->n; r++) { hipError_t e = hipStreamSynchronize(g->ctx[r]->stream); if (e != hipSuccess) { g->set_error("hipStreamSynchronize: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; } }
    for (int r = 0; r < g->n; r++)
        for (int q = 0; q < g->n; q++) {
            float *dst = recv[r] + (size_t)q * count;
            if (dst == send[q]) continue;                          // in-place chunk
            hipError_t e = hipMemcpyAsync(dst, send[q], count * sizeof(float), hipMemcpyDeviceToDevice, g->ctx[r]->stream);
            if (e != hipSuccess) { g->set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; }
        }
    for (int r = 0; r < g->n; r++) hipStreamSynchronize(g->ctx[r]->stream);    // a producer's buffer may be overwritten by its next pass
    return EVPLP_OK;
}

extern "C" int evplp_group_load_scene_json(evplp_group *g, const char *json_path) { GRP_CHECK(g); GRP_EACH(g, evplp_load_scene_json(c, json_path)); return EVPLP_OK; }
extern "C" int evplp_group_clear_accumulators(evplp_group *g) { GRP_CHECK(g); GRP_EACH(g, evplp_clear_accumulators(c)); return EVPLP_OK; }
extern "C" int evplp_group_synchronize(evplp_group *g) { GRP_CHECK(g); GRP_EACH(g, evplp_synchronize(c)); return EVPLP_OK; }
extern "C" int evplp_group_primary(evplp_group *g, const float jitter[2], int32_t light_flags) { GRP_CHECK(g); GRP_EACH(g, evplp_primary(c, jitter, light_flags)); return EVPLP_OK; }

extern "C" int evplp_group_trace_light_paths(evplp_group *g, uint32_t rng_seed) {
    GRP_CHECK(g);
    if (!g->split_paths) { GRP_EACH(g, evplp_trace_light_paths(c, rng_seed, 0, c->cfg.num_light_paths)); return EVPLP_OK; }
    int order[64]; feed_order(g, order);
    for (int i = 0; i < g->n; i++) {
        const int r = order[i];
        // in place: rank r's own slice goes to offset r * chunk of its record buffer.  A partial path range never goes to the second
        // record buffer of overlap_light_tracing (context.cpp only double-buffers whole path sets), so EVPLP_BUF_RECORDS must be the
        // same buffer before and after the call -- checked, because the exchange below would otherwise gather the wrong buffer.
        const void *before = g->ctx[r]->buf[EVPLP_BUF_RECORDS];
        int rc = evplp_trace_light_paths(g->ctx[r], rng_seed, (uint32_t)r * g->per_rank_paths, g->per_rank_paths);
        if (rc < 0) { g->set_error("rank %d: %s", r, evplp_last_error(g->ctx[r])); return rc; }
        if (g->ctx[r]->buf[EVPLP_BUF_RECORDS] != before) { g->set_error("evplp_group_trace_light_paths: rank %d traced a partial path range into a flipped record buffer", r); return EVPLP_ERR_INVALID; }
    }
    const size_t chunk = (size_t)g->per_rank_paths * g->ctx[0]->cfg.photons_per_path * (sizeof(evplp_record) / sizeof(float));
    std::vector<const float *> send((size_t)g->n); std::vector<float *> recv((size_t)g->n);
    for (int r = 0; r < g->n; r++) { recv[r] = (float *)g->ctx[r]->buf[EVPLP_BUF_RECORDS]; send[r] = recv[r] + (size_t)r * chunk; }
    return group_all_gather(g, send, recv, chunk);
}

extern "C" int evplp_group_gather(evplp_group *g, const evplp_frame_params *fp, int32_t kind) {
    GRP_CHECK(g);
    if (kind < 0 || kind > 2) { g->set_error("evplp_group_gather: kind must be 0 (VPL), 1 (VSL) or 2 (light-path windows)"); return EVPLP_ERR_INVALID; }
    GRP_EACH(g, kind == 0 ? evplp_gather_vpl(c, fp) : kind == 1 ? evplp_gather_vsl(c, fp) : evplp_gather_lvc(c, fp));
    return EVPLP_OK;
}
extern "C" int evplp_group_splat_photons(evplp_group *g, const evplp_frame_params *fp, int32_t clear) { GRP_CHECK(g); GRP_EACH(g, evplp_splat_photons(c, fp, clear)); return EVPLP_OK; }
extern "C" int evplp_group_set_splat_proxy(evplp_group *g, const float *vertices, int32_t nverts, const int32_t *indices, int32_t ntris) {
    GRP_CHECK(g); GRP_EACH(g, evplp_set_splat_proxy(c, vertices, nverts, indices, ntris)); return EVPLP_OK;
}
extern "C" int evplp_group_path_trace(evplp_group *g, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces, int32_t do_accumulate) {
    GRP_CHECK(g); GRP_EACH(g, evplp_path_trace(c, camera_pos, rng_seed, max_bounces, do_accumulate)); return EVPLP_OK;
}

// Composite every strip on its GPU and all-gather the strips: every GPU then holds the frame, strip by strip.  This is
// the per-frame exchange of a run that presents every frame; nothing comes to the host.
static int group_present(evplp_group *g, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, bool settle) {
    GRP_EACH(g, evplp::resolve_to_device(c, vs, ps, ls, mask_emitter, gamma, settle || !c->aux_stream));
    std::vector<const float *> send((size_t)g->n); std::vector<float *> recv((size_t)g->n);
    for (int r = 0; r < g->n; r++) { send[r] = g->ctx[r]->d_rgb; recv[r] = g->d_frame[r]; }
    return group_all_gather(g, send, recv, g->strip_floats);
}
extern "C" int evplp_group_present(evplp_group *g, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma) {
    GRP_CHECK(g);
    return group_present(g, vs, ps, ls, mask_emitter, gamma, false);       // (the per-iteration composite: no wait for the splat's verdict)
}

// evplp_group_present, then the frame in image order on rank 0's device and one copy to the caller.
extern "C" int evplp_group_resolve(evplp_group *g, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, float *out_rgb) {
    GRP_CHECK(g);
    if (!out_rgb) { g->set_error("evplp_group_resolve: null output"); return EVPLP_ERR_INVALID; }
    int rc = group_present(g, vs, ps, ls, mask_emitter, gamma, true);
    if (rc < 0) return rc;
    // rank 0 puts the strips into image order on the device; one copy lands the frame in the caller's buffer (no host-side assembly:
    // a run that writes every frame resolves every iteration)
    evplp_context *c0 = g->ctx[0];
    hipSetDevice(g->device[0]);
    const size_t frame_floats = (size_t)c0->st.W * c0->st.H * 3;
    if (!g->d_assembled) {
        hipError_t me = hipMalloc((void **)&g->d_assembled, sizeof(float) * frame_floats);
        if (me != hipSuccess) { g->set_error("evplp_group_resolve: hipMalloc(frame): %s", hipGetErrorString(me)); return EVPLP_ERR_OOM; }
    }
    evplp::launch_assemble_strips(c0->st, g->n, g->d_frame[0], g->d_assembled, c0->stream);
    hipError_t e = hipMemcpyAsync(out_rgb, g->d_assembled, frame_floats * sizeof(float), hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
    if (e != hipSuccess) { g->set_error("frame download: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; }
    return EVPLP_OK;
}
