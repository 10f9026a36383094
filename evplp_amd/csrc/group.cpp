// evplp_group: the multi-GPU entry of the C ABI (SURVEY 8b "Threading", 8e).  n contexts -- one per GPU of the node -- own interleaved
// row strips of the image (include/evplp.h); the scene and the BVH are replicated; large light-path sets are traced 1/n per rank and
// shared by an in-place all-gather of the record buffers; every rank gathers / splats its own rows; the composited strips are
// all-gathered so that every GPU holds the frame.  No other exchange exists on the path.  The collectives are RCCL (ncclAllGather over
// xGMI, one communicator per GPU; the library is opened at run time so that hosts without it can still use single contexts).  Ranks that
// share one device ("virtual ranks": tests, single-GPU boxes) exchange by device-to-device copies instead.
//
// (round 5) ONE WORKER THREAD PER RANK.  Until round 4 the caller's thread issued every rank's launches in turn: at eight ranks that is
// ~50 enqueue calls per iteration against config #4's 0.15-0.6 ms iteration -- the host, not the GPUs, would have set that
// configuration's pace.  Now a group call only POSTS a small command record to each rank's single-producer ring (no lock taken by a
// waiting party, no system call while the workers are awake) and returns; every worker is bound to its device, runs its rank's calls in
// order -- waits for a photon splat's verdict included: nobody else waits with it -- and issues its rank's side of a collective itself
// (RCCL's one-thread-per-GPU model).  Per-pixel results do not depend on any of this: the same calls reach every context in the same order.
//   * errors are sticky: the first failing call of a rank is kept, every later command of that rank is skipped, and the next group call
//     (at the latest evplp_group_synchronize / evplp_group_resolve) returns it;
//   * every worker meets the others at a host-side barrier in front of a collective and looks at the group's failure flag THERE, so that
//     either all ranks enter the collective or none does;
//   * a call made directly on a rank's context (evplp_group_context: statistics, buffers) first waits until that rank's worker has
//     nothing queued (evplp_context::quiesce).
#include "context.hpp"

#include <rccl/rccl.h>      // types only: the entry points are resolved with dlsym

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace evplp {
int resolve_to_device(evplp_context *c, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, bool settle);   // context.cpp
}

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool open(std::string &err) {
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot open RCCL (librccl.so.1): ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllGather || !GetErrorString) { err = "librccl lacks an expected entry point"; return false; }
        return true;
    }
};

enum Op { OP_QUIT = 0, OP_CLEAR, OP_SYNC, OP_PRIMARY, OP_TRACE, OP_GATHER, OP_SPLAT, OP_PATH_TRACE, OP_PRESENT, OP_LOAD_SCENE, OP_SET_PROXY, OP_ASSEMBLE };
// One posted call: plain data, copied into the ring (pointers must stay valid until the caller has drained: load_scene, set_proxy, resolve do)
struct Cmd {
    int op = OP_QUIT;
    evplp_frame_params fp{};
    float f[4] = {}; int32_t i[4] = {}; uint32_t u[4] = {};
    const void *p0 = nullptr, *p1 = nullptr; void *out = nullptr;
};
constexpr int kRing = 64;
constexpr int kSpinBeforeSleep = 200000;     // ~1-2 ms of polling before an idle worker goes to sleep on its condition variable

// sense-reversing barrier of the workers (spins: the waits are microseconds long, in front of a collective).  wait(flag) returns ONE
// verdict for all parties of a generation: the last arriver reads `flag` once, in front of flipping the sense, and everybody leaves with
// what it read -- a rank that fails right behind the barrier cannot make its peers disagree about whether the collective is entered.
struct SpinBarrier {
    std::atomic<int> count{ 0 }; std::atomic<int> sense{ 0 }; std::atomic<int> verdict{ 0 }; int n = 1;
    int wait(const std::atomic<int> *flag = nullptr) {
        const int s = sense.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            verdict.store(flag ? flag->load(std::memory_order_acquire) : 0, std::memory_order_relaxed);
            sense.store(s ^ 1, std::memory_order_release);
        } else { int spins = 0; while (sense.load(std::memory_order_acquire) == s) { if (++spins > 2000) std::this_thread::yield(); } }
        // (the verdict of THIS generation: the next one's last arriver cannot overwrite it before every party of this one has arrived there,
        // i.e. has returned from here)
        return verdict.load(std::memory_order_relaxed);
    }
};
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace

struct evplp_group;
struct Worker {
    evplp_group *g = nullptr; int rank = 0;
    std::thread th;
    Cmd ring[kRing];
    alignas(64) std::atomic<uint64_t> head{ 0 };      // consumed
    alignas(64) std::atomic<uint64_t> tail{ 0 };      // posted
    std::mutex m; std::condition_variable cv; bool sleeping = false;
    std::atomic<int> status{ 0 };                     // first failing call's status (sticky)
    char error[512] = "";
    // host time of this worker (evplp_group_host_stats): inside its rank's enqueue calls / inside exchanges (barriers, copies, collectives)
    double t_calls = 0.0, t_exchange = 0.0; uint64_t n_cmds = 0;
};

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
    bool bands = false; evplp::BandTable band_table{};    // EVPLP_PARTITION_BANDS: first image row of every rank's band (+ H)
    // EVPLP_PARTITION_STRIPS: the blocks as dealt (evplp_group_rebalance); empty = block b belongs to rank b % n
    std::vector<int32_t> owner;             // [image blocks] rank
    uint32_t *d_owner = nullptr;            // rank 0's device: [image blocks] rank << 16 | local block (assemble_strips_kernel)
    int strip_rows = 16, image_blocks = 0, cap_blocks = 0;
    size_t strip_floats_cap = 0;            // d_frame's chunk size (the capacity); strip_floats <= it is what an exchange moves
    std::vector<Worker *> workers;
    SpinBarrier barrier;
    std::atomic<int> failed{ 0 };           // some rank has failed: collectives are skipped by everybody
    char error[512] = "";
    void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(error, sizeof(error), fmt, ap); va_end(ap); }
};

static thread_local char g_group_create_error[512] = "";

// ------------------------------------------------------------------------------------------------ the workers
static void worker_fail(Worker *w, int rc, const char *msg) {
    int expected = 0;
    if (w->status.compare_exchange_strong(expected, rc)) { std::snprintf(w->error, sizeof(w->error), "%s", msg ? msg : ""); w->g->failed.store(1, std::memory_order_release); }
}
// all-gather of equal chunks, this rank's side: it contributes `count` floats at all_send[rank] and receives n * count floats at `recv`
static void worker_all_gather(Worker *w, float *recv, size_t count, const std::vector<const float *> &all_send) {
    evplp_group *g = w->g; evplp_context *c = g->ctx[(size_t)w->rank];
    const float *send = all_send[(size_t)w->rank];
    if (g->n == 1 && g->virtual_ranks) {       // one rank: its chunk goes to its place in stream order, the host does not wait
        if (w->status.load(std::memory_order_relaxed) == 0 && recv != send) {
            hipError_t e = hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) worker_fail(w, EVPLP_ERR_HIP, hipGetErrorString(e));
        }
        return;
    }
    // everybody arrives, then everybody gets the SAME answer to "has somebody failed": all enter the collective or none (the barrier's last
    // arriver reads the flag once for all; a rank that reads it for itself could see a failure its peers, already past the barrier, raised
    // in their NEXT command and skip a collective they have entered)
    if (g->barrier.wait(&g->failed) != 0) return;
    if (!g->virtual_ranks) {
        ncclResult_t nr = g->rccl.AllGather(send, recv, count, ncclFloat, g->comms[(size_t)w->rank], c->stream);
        if (nr != ncclSuccess) worker_fail(w, EVPLP_ERR_HIP, g->rccl.GetErrorString(nr));
        return;
    }
    // virtual ranks share a device: every producer finishes, then plain device copies on the receiver's stream; the producers' buffers
    // may be overwritten by their next pass only after every receiver has its copy
    hipError_t e = hipStreamSynchronize(c->stream);
    g->barrier.wait();
    for (int q = 0; q < g->n && e == hipSuccess; q++) {
        float *dst = recv + (size_t)q * count;
        if (dst != all_send[(size_t)q]) e = hipMemcpyAsync(dst, all_send[(size_t)q], count * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    g->barrier.wait();
    if (e != hipSuccess) worker_fail(w, EVPLP_ERR_HIP, hipGetErrorString(e));
}

static void worker_run(Worker *w, const Cmd &cmd) {
    evplp_group *g = w->g; const int r = w->rank; evplp_context *c = g->ctx[(size_t)r];
    const bool collective = (cmd.op == OP_PRESENT && cmd.i[3] != 0) || (cmd.op == OP_TRACE && g->split_paths);
    int rc = EVPLP_OK;
    const double t0 = now_ms();
    if (w->status.load(std::memory_order_relaxed) == 0) {
        switch (cmd.op) {
        case OP_CLEAR: rc = evplp_clear_accumulators(c); break;
        case OP_SYNC: rc = evplp_synchronize(c); break;
        case OP_PRIMARY: rc = evplp_primary(c, cmd.f, cmd.i[0]); break;
        case OP_TRACE:
            if (!g->split_paths) rc = evplp_trace_light_paths(c, cmd.u[0], 0, c->cfg.num_light_paths);
            else {
                // in place: rank r's own slice goes to offset r * chunk of its record buffer.  A partial path range never goes to the second
                // record buffer of overlap_light_tracing (context.cpp only double-buffers whole path sets), so EVPLP_BUF_RECORDS must be the
                // same buffer before and after the call -- checked, because the exchange below would otherwise gather the wrong buffer.
                const void *before = c->buf[EVPLP_BUF_RECORDS];
                rc = evplp_trace_light_paths(c, cmd.u[0], (uint32_t)r * g->per_rank_paths, g->per_rank_paths);
                if (rc >= 0 && c->buf[EVPLP_BUF_RECORDS] != before) worker_fail(w, EVPLP_ERR_INVALID, "evplp_group_trace_light_paths: a partial path range went into a flipped record buffer");
            }
            break;
        case OP_GATHER: rc = cmd.i[0] == 0 ? evplp_gather_vpl(c, &cmd.fp) : cmd.i[0] == 1 ? evplp_gather_vsl(c, &cmd.fp) : evplp_gather_lvc(c, &cmd.fp); break;
        case OP_SPLAT: rc = evplp_splat_photons(c, &cmd.fp, cmd.i[0]); break;
        case OP_PATH_TRACE: rc = evplp_path_trace(c, cmd.f, cmd.u[0], cmd.u[1], cmd.i[0]); break;
        case OP_PRESENT: rc = evplp::resolve_to_device(c, cmd.f[0], cmd.f[1], cmd.f[2], cmd.i[0], cmd.i[1], cmd.i[2] != 0 || !c->aux_stream); break;
        case OP_LOAD_SCENE: rc = evplp_load_scene_json(c, (const char *)cmd.p0); break;
        case OP_SET_PROXY: rc = evplp_set_splat_proxy(c, (const float *)cmd.p0, cmd.i[0], (const int32_t *)cmd.p1, cmd.i[1]); break;
        case OP_ASSEMBLE: {
            // rank 0 puts the strips into image order on the device; one copy lands the frame in the caller's buffer (no host-side assembly:
            // a run that writes every frame resolves every iteration)
            hipSetDevice(g->device[0]);
            const size_t frame_floats = (size_t)c->st.W * c->st.H * 3;
            hipError_t e = hipSuccess;
            if (!g->d_assembled) e = hipMalloc((void **)&g->d_assembled, sizeof(float) * frame_floats);
            if (e == hipSuccess) {
                evplp::launch_assemble_strips(c->st, g->n, g->bands ? &g->band_table : nullptr, g->owner.empty() ? nullptr : g->d_owner, (int)(g->strip_floats / ((size_t)c->st.W * 3)), g->d_frame[0], g->d_assembled, c->stream);
                e = hipMemcpyAsync(cmd.out, g->d_assembled, frame_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) worker_fail(w, e == hipErrorOutOfMemory ? EVPLP_ERR_OOM : EVPLP_ERR_HIP, hipGetErrorString(e));
            break;
        }
        default: break;
        }
        if (rc < 0) worker_fail(w, rc, evplp_last_error(c));
    }
    const double t1 = now_ms();
    w->t_calls += t1 - t0; w->n_cmds++;
    if (collective) {          // (reached by every rank, failed or not: the barrier inside decides for all)
        std::vector<const float *> all((size_t)g->n);
        if (cmd.op == OP_TRACE) {
            const size_t chunk = (size_t)g->per_rank_paths * c->cfg.photons_per_path * (sizeof(evplp_record) / sizeof(float));
            for (int q = 0; q < g->n; q++) all[(size_t)q] = (const float *)g->ctx[(size_t)q]->buf[EVPLP_BUF_RECORDS] + (size_t)q * chunk;
            worker_all_gather(w, (float *)c->buf[EVPLP_BUF_RECORDS], chunk, all);
        } else {
            for (int q = 0; q < g->n; q++) all[(size_t)q] = g->ctx[(size_t)q]->d_rgb;
            worker_all_gather(w, g->d_frame[(size_t)r], g->strip_floats, all);
        }
        w->t_exchange += now_ms() - t1;
    }
}

static void worker_main(Worker *w) {
    hipSetDevice(w->g->device[(size_t)w->rank]);
    for (;;) {
        const uint64_t h = w->head.load(std::memory_order_relaxed);
        int spins = 0;
        while (w->tail.load(std::memory_order_acquire) == h) {
            if (++spins < kSpinBeforeSleep) { if ((spins & 63) == 0) std::this_thread::yield(); continue; }
            std::unique_lock<std::mutex> lk(w->m);
            w->sleeping = true;
            w->cv.wait(lk, [&] { return w->tail.load(std::memory_order_acquire) != h; });
            w->sleeping = false;
            spins = 0;
        }
        const Cmd cmd = w->ring[h % kRing];
        if (cmd.op == OP_QUIT) { w->head.store(h + 1, std::memory_order_release); return; }
        worker_run(w, cmd);
        w->head.store(h + 1, std::memory_order_release);
    }
}
static void post(Worker *w, const Cmd &cmd) {
    const uint64_t t = w->tail.load(std::memory_order_relaxed);
    while (t - w->head.load(std::memory_order_acquire) >= (uint64_t)kRing) std::this_thread::yield();      // (the ring is full: the caller is 64 calls ahead)
    w->ring[t % kRing] = cmd;
    w->tail.store(t + 1, std::memory_order_seq_cst);
    std::lock_guard<std::mutex> lk(w->m);                  // (uncontended while the worker polls; pairs with the worker's check before it sleeps)
    if (w->sleeping) w->cv.notify_one();
}
static void drain_one(Worker *w) { int spins = 0; while (w->head.load(std::memory_order_acquire) != w->tail.load(std::memory_order_acquire)) { if (++spins > 1000) std::this_thread::yield(); } }
static void drain(evplp_group *g) { for (Worker *w : g->workers) drain_one(w); }
static void quiesce_hook(void *arg) { drain_one((Worker *)arg); }
// the sticky error of the group, if any: the lowest failing rank's
static int group_status(evplp_group *g) {
    for (Worker *w : g->workers) { const int st = w->status.load(std::memory_order_acquire); if (st < 0) { g->set_error("rank %d: %s", w->rank, w->error); return st; } }
    return EVPLP_OK;
}
static int post_all(evplp_group *g, const Cmd &cmd) {
    if (g->failed.load(std::memory_order_acquire)) { drain(g); return group_status(g); }
    for (Worker *w : g->workers) post(w, cmd);
    return EVPLP_OK;
}
// calls whose arguments must outlive them, or whose result the caller needs: post, wait until every worker is idle, report
static int post_and_wait(evplp_group *g, const Cmd &cmd) {
    int rc = post_all(g, cmd);
    if (rc < 0) return rc;
    drain(g);
    return group_status(g);
}

#define GRP_CHECK(g) do { if (!(g)) return EVPLP_ERR_INVALID; } while (0)
// What a pass call can refuse without touching a device is refused HERE, on the caller's thread, at once -- as the plain context does -- and
// leaves the group usable; only failures of the device side are sticky.  (The configuration is the same on every rank and never changes.)
static int check_frame_params(evplp_group *g, const evplp_frame_params *fp, const char *name, bool splat) {
    const evplp_config &cf = g->ctx[0]->cfg;
    if (fp->photons_per_path != cf.photons_per_path || fp->num_light_paths != cf.num_light_paths || fp->num_vpl_light_paths > cf.num_vpl_light_paths) {
        g->set_error("%s: frame params disagree with the configuration (paths / photons per path)", name); return EVPLP_ERR_INVALID;
    }
    if (fp->mis_mode > 5u) { g->set_error("%s: mis_mode %u out of range", name, fp->mis_mode); return EVPLP_ERR_INVALID; }
    if (!splat && fp->num_vpl_light_paths == 0) { g->set_error("%s: num_vpl_light_paths is 0 (the reference disables the pass, rtcomphoton.h:200-203)", name); return EVPLP_ERR_INVALID; }
    if (splat && !(fp->photon_radius > 0.0f)) { g->set_error("%s: photon_radius must be > 0", name); return EVPLP_ERR_INVALID; }
    if (splat && fp->splat_footprint > (uint32_t)EVPLP_FOOTPRINT_PROXY) { g->set_error("%s: splat_footprint %u out of range", name, fp->splat_footprint); return EVPLP_ERR_INVALID; }
    return EVPLP_OK;
}

extern "C" const char *evplp_group_last_error(const evplp_group *g) { return g ? g->error : g_group_create_error; }
extern "C" int evplp_group_size(const evplp_group *g) { return g ? g->n : EVPLP_ERR_INVALID; }
extern "C" evplp_context *evplp_group_context(evplp_group *g, int32_t rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[(size_t)rank] : nullptr; }
extern "C" int evplp_group_host_stats(evplp_group *g, int32_t rank, double out[3]) {
    GRP_CHECK(g);
    if (rank < 0 || rank >= g->n || !out) { g->set_error("evplp_group_host_stats: bad arguments"); return EVPLP_ERR_INVALID; }
    drain_one(g->workers[(size_t)rank]);
    out[0] = g->workers[(size_t)rank]->t_calls; out[1] = g->workers[(size_t)rank]->t_exchange; out[2] = (double)g->workers[(size_t)rank]->n_cmds;
    return EVPLP_OK;
}

extern "C" int evplp_group_profile_passes(evplp_group *g, int32_t on) {
    GRP_CHECK(g);
    drain(g);                                                             // (the flag belongs to the workers' contexts: set between their commands)
    for (evplp_context *c : g->ctx) c->profile_passes = on != 0;
    return EVPLP_OK;
}

extern "C" void evplp_group_destroy(evplp_group *g) {
    if (!g) return;
    for (Worker *w : g->workers) if (w->th.joinable()) { Cmd q; q.op = OP_QUIT; post(w, q); }
    for (Worker *w : g->workers) { if (w->th.joinable()) w->th.join(); delete w; }
    g->workers.clear();
    for (int r = 0; r < (int)g->d_frame.size(); r++) if (g->d_frame[(size_t)r]) { hipSetDevice(g->device[(size_t)r]); hipFree(g->d_frame[(size_t)r]); }
    if (g->d_assembled || g->d_owner) { hipSetDevice(g->device[0]); hipFree(g->d_assembled); hipFree(g->d_owner); }
    for (ncclComm_t c : g->comms) if (c && g->rccl.CommDestroy) g->rccl.CommDestroy(c);
    for (evplp_context *c : g->ctx) { c->quiesce = nullptr; evplp_destroy(c); }
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
    for (int r = 0; r < g->n; r++) for (int q = 0; q < r; q++) { if (g->device[(size_t)r] == g->device[(size_t)q]) all_distinct = false; else all_same = false; }
    if (g->n > 1 && !all_same && !all_distinct) { delete g; return fail(EVPLP_ERR_INVALID, "evplp_group_create: the ranks' devices must be all distinct (RCCL) or all the same (virtual ranks)"); }
    g->virtual_ranks = g->n > 1 ? all_same : !gc->use_rccl;
    if (gc->use_rccl && g->n > 1 && !all_distinct) { delete g; return fail(EVPLP_ERR_INVALID, "evplp_group_create: RCCL needs one distinct device per rank"); }
    // Strips of 16 rows keep a rank's tile rows in neighbouring pairs -- the gathers' entry cuts then cover groups of 2 x 2 tiles as on one
    // GPU (8-row strips: 2 x 1, twice as many cuts per pixel) -- but interleave the image half as finely.  Round 5 took 8 rows from eight
    // ranks on for that; with the blocks dealt by cost and launched most expensive first (evplp_group_rebalance) the finer interleave buys
    // nothing any more and the cheaper cuts win at every rank count (single-GPU projection of config #2, profiles/r06_strip_projection.json:
    // n = 8, 8- / 16-row blocks: slowest rank 8.49 / 8.06 ms; n = 4: 14.97 / 14.60).
    const int strip_rows = gc->strip_rows > 0 ? gc->strip_rows : 16;
    // EVPLP_PARTITION_BANDS: contiguous bands of equal height to begin with (multiples of 16 rows), each with room for twice its share
    g->bands = gc->partition == EVPLP_PARTITION_BANDS && g->n > 1;
    const int rows16 = ((cfg->res_y + 15) / 16) * 16, share = std::max(16, ((rows16 / g->n + 15) / 16) * 16), band_cap = std::min(rows16, 2 * share);
    for (int r = 0; r <= g->n; r++) g->band_table.first[r] = std::min(r * share, cfg->res_y);
    g->band_table.first[g->n] = cfg->res_y;
    if (g->bands && g->band_table.first[g->n - 1] >= cfg->res_y) { delete g; return fail(EVPLP_ERR_INVALID, "evplp_group_create: %d bands of at least 16 rows do not fit %d image rows", gc->n_ranks, cfg->res_y); }
    for (int r = 0; r < g->n; r++) {
        evplp_config c = *cfg;
        c.device = g->device[(size_t)r]; c.strip_rank = r; c.strip_count = g->n; c.strip_rows = strip_rows;
        {   // room for a deal by cost: strip_capacity_pct of the equal share of blocks, rounded up (0 = 150 %)
            const int nb = (cfg->res_y + strip_rows - 1) / strip_rows, share = (nb + g->n - 1) / g->n, pct = gc->strip_capacity_pct > 0 ? std::max(gc->strip_capacity_pct, 100) : 150;
            c.strip_capacity_rows = g->n > 1 ? std::min(nb, (share * pct + 99) / 100) * strip_rows : 0;
        }
        if (g->bands) {
            c.strip_rank = 0; c.strip_count = 1; c.strip_rows = 0;
            c.band_first_row = g->band_table.first[r]; c.band_rows = (r + 1 < g->n ? g->band_table.first[r + 1] : rows16) - g->band_table.first[r]; c.band_capacity_rows = band_cap;
        }
        evplp_context *h = nullptr;
        int rc = evplp_create(&c, &h);
        if (rc < 0) { int code = fail(rc, "rank %d: %s", r, evplp_last_error(nullptr)); evplp_group_destroy(g); return code; }
        g->ctx.push_back(h);
    }
    g->strip_rows = g->ctx[0]->st.strip_rows; g->image_blocks = g->ctx[0]->image_blocks; g->cap_blocks = g->ctx[0]->st.cap_blocks;
    g->strip_floats_cap = (size_t)g->ctx[0]->st.local_rows * g->ctx[0]->st.W * 3;
    // an exchange moves the rows in use: the equal share under the round-robin deal (the capacity beyond it holds nothing), the fullest rank's
    // blocks after a deal by cost
    g->strip_floats = g->bands || g->n == 1 ? g->strip_floats_cap
                                            : (size_t)((g->image_blocks + g->n - 1) / g->n) * (size_t)g->strip_rows * g->ctx[0]->st.W * 3;
    g->d_frame.assign((size_t)g->n, nullptr);
    for (int r = 0; r < g->n; r++) {
        hipSetDevice(g->device[(size_t)r]);
        hipError_t e = hipMalloc((void **)&g->d_frame[(size_t)r], sizeof(float) * g->strip_floats_cap * (size_t)g->n);
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
    // (round 6: until round 5 the rule was "sets of >= 16 384 paths are split".  Config #4 at four ranks: 75 000 paths take 0.24 ms where 300 000
    // take 0.46, and the exchange moves 28.8 MB over every link -- more than the 0.22 ms the split saves at any plausible xGMI rate.)
    g->split_paths = g->n > 1 && cfg->num_light_paths % (uint32_t)g->n == 0 &&
                     (gc->split_light_paths > 0 || (gc->split_light_paths == 0 && evplp_group_split_model(cfg->num_light_paths, cfg->photons_per_path, g->n, nullptr) == 1));
    g->per_rank_paths = g->split_paths ? cfg->num_light_paths / (uint32_t)g->n : cfg->num_light_paths;
    g->barrier.n = g->n;
    for (int r = 0; r < g->n; r++) { Worker *w = new Worker(); w->g = g; w->rank = r; g->workers.push_back(w); }
    for (Worker *w : g->workers) {
        w->th = std::thread(worker_main, w);
        evplp_context *c = g->ctx[(size_t)w->rank];
        c->worker_tid = w->th.get_id(); c->quiesce_arg = w; c->quiesce = quiesce_hook;
    }
    *out = g;
    return EVPLP_OK;
}

// Bands dealt by cost: every rank's device time since the last rebalance, spread evenly over its rows, is a piecewise-constant cost profile of
// the image; the new boundaries cut it into n parts of equal cost (multiples of 16 rows, at least 16, at most the capacity).  A few rounds of
// frame + rebalance converge: a band that was expensive gets shorter, and the next round measures its cost at the new height.
extern "C" int evplp_group_rebalance(evplp_group *g, int32_t *band_first_rows) {
    GRP_CHECK(g);
    drain(g);
    int rc = group_status(g); if (rc < 0) return rc;
    const int H = g->ctx[0]->st.H, n = g->n;
    if (!g->bands && n > 1) {
        // ---- EVPLP_PARTITION_STRIPS: deal the blocks by the cost the gathers clocked (evplp_group_calibrate)
        const int nb = g->image_blocks;
        std::vector<uint64_t> cost((size_t)nb, 0), mine((size_t)nb);
        uint64_t total = 0;
        for (int r = 0; r < n; r++) {
            int rb = evplp_block_costs(g->ctx[(size_t)r], mine.data(), nb);
            if (rb < 0) { g->set_error("rank %d: %s", r, evplp_last_error(g->ctx[(size_t)r])); return rb; }
            for (int b = 0; b < nb; b++) { cost[(size_t)b] += mine[(size_t)b]; total += mine[(size_t)b]; }
        }
        if (total == 0) { g->set_error("evplp_group_rebalance: no block cost was clocked (evplp_group_calibrate, then a frame with a gather)"); return EVPLP_ERR_INVALID; }
        std::vector<int32_t> owner((size_t)nb);
        rc = evplp_deal_blocks(cost.data(), nb, n, g->cap_blocks, owner.data());
        if (rc < 0) { g->set_error("evplp_group_rebalance: %d blocks do not fit %d ranks of %d", nb, n, g->cap_blocks); return rc; }
        // every rank's blocks, the most expensive first (evplp_rank_blocks: the launch order); all tables are built before any is set
        std::vector<std::vector<int32_t>> lists((size_t)n);
        std::vector<uint32_t> packed((size_t)nb);
        size_t most = 0;
        for (int r = 0; r < n; r++) {
            auto &l = lists[(size_t)r];
            l.resize((size_t)nb);
            l.resize((size_t)evplp_rank_blocks(cost.data(), owner.data(), nb, r, l.data(), nb));
            for (size_t i = 0; i < l.size(); i++) packed[(size_t)l[i]] = ((uint32_t)r << 16) | (uint32_t)i;
            most = std::max(most, l.size());
        }
        hipSetDevice(g->device[0]);
        if (!g->d_owner && hipMalloc((void **)&g->d_owner, sizeof(uint32_t) * (size_t)nb) != hipSuccess) { (void)hipGetLastError(); g->set_error("evplp_group_rebalance: hipMalloc(owner table)"); return EVPLP_ERR_OOM; }
        if (hipMemcpy(g->d_owner, packed.data(), sizeof(uint32_t) * packed.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); g->set_error("evplp_group_rebalance: hipMemcpy(owner table)"); return EVPLP_ERR_HIP; }
        for (int r = 0; r < n; r++) {
            int rb = evplp_set_blocks(g->ctx[(size_t)r], lists[(size_t)r].data(), (int32_t)lists[(size_t)r].size());
            if (rb >= 0) rb = evplp_calibrate_blocks(g->ctx[(size_t)r], 0);
            if (rb < 0) {      // (cannot happen with a table evplp_deal_blocks made for this capacity; if it does, everybody returns to the default deal)
                g->set_error("rank %d: %s", r, evplp_last_error(g->ctx[(size_t)r]));
                for (int q = 0; q < n; q++) evplp_set_blocks(g->ctx[(size_t)q], nullptr, 0);
                g->owner.clear(); g->strip_floats = (size_t)((nb + n - 1) / n) * (size_t)g->strip_rows * g->ctx[0]->st.W * 3;
                return rb;
            }
        }
        g->owner.swap(owner);
        g->strip_floats = most * (size_t)g->strip_rows * g->ctx[0]->st.W * 3;
    }
    if (g->bands) {
        static const int kPasses[] = { EVPLP_PASS_PRIMARY, EVPLP_PASS_GATHER_VPL, EVPLP_PASS_GATHER_VSL, EVPLP_PASS_GATHER_LVC, EVPLP_PASS_SPLAT, EVPLP_PASS_PATH_TRACE };
        std::vector<double> cost((size_t)n, 0.0);
        double total = 0.0;
        for (int r = 0; r < n; r++) {
            for (int p : kPasses) {
                evplp_pass_stats ps;
                if (g->ctx[(size_t)r]->pass_ran[p] && evplp_pass_stats_get(g->ctx[(size_t)r], p, &ps) == EVPLP_OK) cost[(size_t)r] += ps.ms;
                g->ctx[(size_t)r]->pass_ran[p] = false;          // (counted once: the next rebalance sees the passes that ran after this one)
            }
            total += cost[(size_t)r];
        }
        if (!(total > 0.0)) { g->set_error("evplp_group_rebalance: no pass was timed since the last rebalance (evplp_group_profile_passes is off, or no frame ran)"); return EVPLP_ERR_INVALID; }
        {
            const int cap = g->ctx[0]->st.local_rows;
            int first[65]; first[0] = 0; first[n] = H;
            // cumulative cost at row y: bands in order, constant density within a band
            auto row_at_cost = [&](double target) {
                double acc = 0.0;
                for (int r = 0; r < n; r++) {
                    const int b0 = g->band_table.first[r], b1 = g->band_table.first[r + 1];
                    if (acc + cost[(size_t)r] >= target || r == n - 1) return b0 + (cost[(size_t)r] > 0.0 ? (target - acc) / cost[(size_t)r] : 0.0) * (double)(b1 - b0);
                    acc += cost[(size_t)r];
                }
                return (double)H;
            };
            for (int r = 1; r < n; r++) {
                int y = (int)(row_at_cost(total * (double)r / (double)n) / 16.0 + 0.5) * 16;
                y = std::max(y, first[r - 1] + 16);                               // at least 16 rows
                y = std::min(y, first[r - 1] + (cap / 16) * 16);                  // at most the capacity
                y = std::min(y, ((H - 1) / 16) * 16 - (n - 1 - r) * 16);          // room for the bands behind it
                first[r] = y;
            }
            // the bands behind a capped one may be pushed beyond THEIR capacity: walk back from the end
            for (int r = n - 1; r >= 1; r--) first[r] = std::max(first[r], ((first[r + 1] + 15) / 16) * 16 - (cap / 16) * 16);
            for (int r = 0; r < n; r++) {
                const int rows = (r + 1 < n ? first[r + 1] : ((H + 15) / 16) * 16) - first[r];
                int rb = evplp_set_band(g->ctx[(size_t)r], first[r], rows);
                if (rb < 0) {      // the group's table follows the contexts: the ranks already moved go back to where the table says they are
                    g->set_error("rank %d: %s", r, evplp_last_error(g->ctx[(size_t)r]));
                    for (int q = 0; q < r; q++) evplp_set_band(g->ctx[(size_t)q], g->band_table.first[q], (q + 1 < n ? g->band_table.first[q + 1] : ((H + 15) / 16) * 16) - g->band_table.first[q]);
                    return rb;
                }
            }
            for (int r = 0; r <= n; r++) g->band_table.first[r] = first[r];
        }
    }
    if (band_first_rows) for (int r = 0; r <= n; r++) band_first_rows[r] = g->bands ? g->band_table.first[r] : 0;
    return EVPLP_OK;
}

extern "C" int evplp_group_split_model(uint32_t num_light_paths, uint32_t photons_per_path, int32_t n_ranks, double out_ms[2]) {
    if (n_ranks < 1) return EVPLP_ERR_INVALID;
    auto trace_ms = [](double paths) { return 0.20 + 1.2e-6 * std::max(0.0, paths - 131072.0); };
    const double all = trace_ms((double)num_light_paths);
    const double chunk_bytes = (double)num_light_paths * photons_per_path * sizeof(evplp_record) / n_ranks;
    const double shared = trace_ms((double)num_light_paths / n_ranks) + (n_ranks > 1 ? 0.02 + chunk_bytes / 48.0e9 * 1.0e3 : 0.0);
    if (out_ms) { out_ms[0] = all; out_ms[1] = shared; }
    return n_ranks > 1 && shared < all ? 1 : 0;
}
extern "C" int evplp_group_calibrate(evplp_group *g, int32_t on) {
    GRP_CHECK(g);
    drain(g);
    int rc = group_status(g); if (rc < 0) return rc;
    for (int r = 0; r < g->n; r++) {
        int rb = evplp_calibrate_blocks(g->ctx[(size_t)r], on);
        if (rb < 0) { g->set_error("rank %d: %s", r, evplp_last_error(g->ctx[(size_t)r])); return rb; }
    }
    return EVPLP_OK;
}
extern "C" int evplp_group_block_owners(evplp_group *g, int32_t *owner_rank, int32_t capacity) {
    GRP_CHECK(g);
    if (g->bands) { g->set_error("evplp_group_block_owners: the group deals bands, not blocks"); return EVPLP_ERR_INVALID; }
    for (int b = 0; b < g->image_blocks && owner_rank && b < capacity; b++) owner_rank[b] = g->owner.empty() ? b % g->n : g->owner[(size_t)b];
    return g->image_blocks;
}

extern "C" int evplp_group_load_scene_json(evplp_group *g, const char *json_path) { GRP_CHECK(g); Cmd c; c.op = OP_LOAD_SCENE; c.p0 = json_path; return post_and_wait(g, c); }
extern "C" int evplp_group_clear_accumulators(evplp_group *g) { GRP_CHECK(g); Cmd c; c.op = OP_CLEAR; return post_all(g, c); }
extern "C" int evplp_group_synchronize(evplp_group *g) { GRP_CHECK(g); Cmd c; c.op = OP_SYNC; return post_and_wait(g, c); }
extern "C" int evplp_group_primary(evplp_group *g, const float jitter[2], int32_t light_flags) {
    GRP_CHECK(g);
    Cmd c; c.op = OP_PRIMARY; c.f[0] = jitter ? jitter[0] : 0.f; c.f[1] = jitter ? jitter[1] : 0.f; c.i[0] = light_flags;
    return post_all(g, c);
}
extern "C" int evplp_group_trace_light_paths(evplp_group *g, uint32_t rng_seed) { GRP_CHECK(g); Cmd c; c.op = OP_TRACE; c.u[0] = rng_seed; return post_all(g, c); }
extern "C" int evplp_group_gather(evplp_group *g, const evplp_frame_params *fp, int32_t kind) {
    GRP_CHECK(g);
    if (kind < 0 || kind > 2) { g->set_error("evplp_group_gather: kind must be 0 (VPL), 1 (VSL) or 2 (light-path windows)"); return EVPLP_ERR_INVALID; }
    if (!fp) { g->set_error("evplp_group_gather: null frame params"); return EVPLP_ERR_INVALID; }
    { int rc = check_frame_params(g, fp, "evplp_group_gather", false); if (rc < 0) return rc; }
    Cmd c; c.op = OP_GATHER; c.fp = *fp; c.i[0] = kind;
    return post_all(g, c);
}
extern "C" int evplp_group_splat_photons(evplp_group *g, const evplp_frame_params *fp, int32_t clear) {
    GRP_CHECK(g);
    if (!fp) { g->set_error("evplp_group_splat_photons: null frame params"); return EVPLP_ERR_INVALID; }
    { int rc = check_frame_params(g, fp, "evplp_group_splat_photons", true); if (rc < 0) return rc; }
    Cmd c; c.op = OP_SPLAT; c.fp = *fp; c.i[0] = clear;
    return post_all(g, c);
}
extern "C" int evplp_group_set_splat_proxy(evplp_group *g, const float *vertices, int32_t nverts, const int32_t *indices, int32_t ntris) {
    GRP_CHECK(g);
    Cmd c; c.op = OP_SET_PROXY; c.p0 = vertices; c.p1 = indices; c.i[0] = nverts; c.i[1] = ntris;
    return post_and_wait(g, c);
}
extern "C" int evplp_group_path_trace(evplp_group *g, const float camera_pos[3], uint32_t rng_seed, uint32_t max_bounces, int32_t do_accumulate) {
    GRP_CHECK(g);
    if (!camera_pos) { g->set_error("evplp_group_path_trace: null camera position"); return EVPLP_ERR_INVALID; }
    Cmd c; c.op = OP_PATH_TRACE; c.f[0] = camera_pos[0]; c.f[1] = camera_pos[1]; c.f[2] = camera_pos[2]; c.u[0] = rng_seed; c.u[1] = max_bounces; c.i[0] = do_accumulate;
    return post_all(g, c);
}

// Composite every strip on its GPU and all-gather the strips: every GPU then holds the frame, strip by strip.  This is
// the per-frame exchange of a run that presents every frame; nothing comes to the host.
static Cmd present_cmd(float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, bool settle, bool exchange) {
    Cmd c; c.op = OP_PRESENT; c.f[0] = vs; c.f[1] = ps; c.f[2] = ls; c.i[0] = mask_emitter; c.i[1] = gamma; c.i[2] = settle ? 1 : 0; c.i[3] = exchange ? 1 : 0;
    return c;
}
extern "C" int evplp_group_present(evplp_group *g, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma) {
    GRP_CHECK(g);
    return post_all(g, present_cmd(vs, ps, ls, mask_emitter, gamma, false, true));       // (the per-iteration composite: no wait for the splat's verdict)
}
// exchange = 0: every rank composites its strip where it is and nobody waits for anybody -- no host barrier, no collective: the iteration of
// a loop whose frame is looked at only now and then (a sub-millisecond iteration pays for the exchange otherwise: DESIGN section 5)
extern "C" int evplp_group_present_ex(evplp_group *g, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, int32_t exchange) {
    GRP_CHECK(g);
    return post_all(g, present_cmd(vs, ps, ls, mask_emitter, gamma, false, exchange != 0));
}

// evplp_group_present (settled), then the frame in image order on rank 0's device and one copy to the caller.
extern "C" int evplp_group_resolve(evplp_group *g, float vs, float ps, float ls, int32_t mask_emitter, int32_t gamma, float *out_rgb) {
    GRP_CHECK(g);
    if (!out_rgb) { g->set_error("evplp_group_resolve: null output"); return EVPLP_ERR_INVALID; }
    int rc = post_all(g, present_cmd(vs, ps, ls, mask_emitter, gamma, true, true));
    if (rc < 0) return rc;
    Cmd c; c.op = OP_ASSEMBLE; c.out = out_rgb;      // (rank 0's stream: behind its side of the exchange)
    post(g->workers[0], c);
    drain(g);
    return group_status(g);
}
