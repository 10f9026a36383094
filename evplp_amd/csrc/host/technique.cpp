// The host side of the reference's technique plugin, restated above the C ABI:
//   main()                      reflectcuts/main.cpp:87-121      -> evplp_render_json
//   RtTechnique::render         rt/rttechnique.h:6-9             -> ComPhotonTechnique::render
//   RtComPhoton::render / run   rt/rtcomphoton/rtcomphoton.h:107-223, 883-1133
//   RtLvcComPhoton              rt/rtcomphoton/rtlvccomphoton.h  -> ComPhotonTechnique(lvc = true)
//   RtPt2::render / run         rt/rtpt/rtpt2.h:84-116, 575-719  -> PathTraceTechnique
// Same JSON keys, defaults, errors and outputs (three images + stat file); no window, no GL:
// the frame loop of common/realtime.h reduces to the iteration cap and the wall-clock limit.
#include "../../../include/evplp.h"
#include "images.hpp"
#include "json.hpp"
#include "scene_io.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace evplp {

namespace {
// build-only key "bvhBuilder"
int parse_bvh_builder(const std::string &v) {
    if (v == "sah") return EVPLP_BVH_SAH;
    if (v == "sbvh") return EVPLP_BVH_SBVH;
    if (v == "lbvh") return EVPLP_BVH_LBVH;
    if (v == "gpu") return EVPLP_BVH_LBVH_GPU;
    throw std::runtime_error("bvhBuilder: expected \"sah\", \"sbvh\", \"lbvh\" or \"gpu\", got \"" + v + "\"");
}
const std::map<std::string, int> kFrameModes = { { "accumulate", 1 }, { "cleareveryframe", 2 } };   // rtcomphoton.h:1194-1197
const std::map<std::string, int> kMisModes = { { "one", 0 }, { "balance", 1 }, { "max", 2 }, { "power2", 3 },
                                               { "geometryClamp", 4 }, { "geometryBrdfClamp", 5 } }; // :1199-1206
constexpr float kInvPi = 0.318309886183790671537767526745028724068919291480912897495f;

// The techniques always run on an evplp_group (one rank = the reference's single device); the `device` block of the technique
// JSON asks for more ranks: {"gpus": N, "virtual": bool, "stripRows": R, "rccl": bool} (build-only key, include/evplp.h).
struct Grp {   // RAII for the C handle
    evplp_group *g = nullptr;
    ~Grp() { if (g) evplp_group_destroy(g); }
};
void check(evplp_group *g, int rc, const char *what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + evplp_group_last_error(g));
}
// What the `device` block says about the RUN (the rest of it configures the group, create_group):
//   "deal": "cost" | "roundRobin" -- row blocks dealt by the cost a calibration frame clocks (evplp_group_calibrate / _rebalance), or block b to
//           rank b % N.  Default: by cost when the run has more than one rank, a VPL / VSL gather, and enough iterations for the calibration
//           frame -- one more frame -- to pay for itself (5 from six ranks on, 25 from three, 100 at two); results do not depend on it.
//   "exchangeEvery": k -- the strips are all-gathered (every GPU holds the frame) in every k-th iteration's composite; 0 = never inside the
//           loop.  Default 0: this loop is headless -- nothing looks at the assembled frame between the frames that are WRITTEN
//           (rtcomphoton.h:1079-1102, 1124-1132), and those always exchange; 1 is what the reference's per-iteration runFinalProgram to
//           the window amounts to (:997-1004): a host barrier and an all-gather per iteration, which a sub-millisecond iteration feels
//           (profiles/r06_host_feed.txt).  Every rank still composites its strip every iteration.  Results do not depend on it.
//   "partition": "strips" | "iterations" -- what the N GPUs share out.  "strips" (default): the image, as above.  "iterations" (round 6; the
//           photonfam techniques in accumulate mode): the ITERATIONS of the progressive run -- GPU g renders iterations g, g + N, g + 2N, ... of
//           the whole image on a context of its own (each has its own seed, jitter and radius: rtcomphoton.h:936-1063 makes an iteration
//           depend on its number only), nothing is exchanged while the loop runs, and the accumulators are summed -- in rank order, on the
//           host -- whenever a frame is written.  N times the iterations per second with no replicated work and nothing to balance; a single
//           iteration is no faster, every GPU holds whole-image buffers, and the sums are associated differently from one GPU's (images agree
//           to fp32 round-off, ~1e-7, not bit for bit).  What config #4 wants: its iteration is two latency-bound walks that row strips cannot
//           shorten.
struct RunOptions { int deal = -1; int exchange_every = 0; bool shard_iterations = false; };       // deal: -1 default, 0 round robin, 1 by cost
RunOptions run_options(const Json &json) {
    RunOptions o;
    if (!json.has("device")) return o;
    const Json &d = json.at("device");
    if (d.has("deal")) {
        const std::string v = d.at("deal").as_string("device.deal");
        if (v == "cost") o.deal = 1; else if (v == "roundRobin") o.deal = 0; else throw JsonError("device.deal: \"cost\" or \"roundRobin\"");
    }
    if (d.has("partition")) {
        const std::string v = d.at("partition").as_string("device.partition");
        if (v == "iterations") o.shard_iterations = true; else if (v != "strips") throw JsonError("device.partition: \"strips\" or \"iterations\"");
    }
    if (d.has("exchangeEvery")) { o.exchange_every = (int)d.at("exchangeEvery").as_int("device.exchangeEvery"); if (o.exchange_every < 0) throw JsonError("device.exchangeEvery: must be >= 0"); }
    return o;
}
// shard >= 0: the group of ONE rank that renders the iterations of shard `shard` ("partition": "iterations"): device + shard, or `device` when virtual
void create_group(Grp &grp, const evplp_config &cfg, const Json &json, int device, int shard = -1) {
    evplp_group_config gc; std::memset(&gc, 0, sizeof(gc));
    gc.n_ranks = 1; gc.strip_rows = 0;      // (0: the group's default, 16-row strips)
    bool virt = false;
    if (json.has("device")) {
        const Json &d = json.at("device");
        if (d.has("gpus")) gc.n_ranks = (int)d.at("gpus").as_int("device.gpus");
        if (d.has("virtual")) virt = d.at("virtual").as_bool("device.virtual");
        if (d.has("stripRows")) gc.strip_rows = (int)d.at("stripRows").as_int("device.stripRows");
        if (d.has("rccl")) gc.use_rccl = d.at("rccl").as_bool("device.rccl") ? 1 : 0;
        if (d.has("stripCapacityPct")) gc.strip_capacity_pct = (int)d.at("stripCapacityPct").as_int("device.stripCapacityPct");
        if (d.has("splitLightPaths")) gc.split_light_paths = d.at("splitLightPaths").as_bool("device.splitLightPaths") ? 1 : -1;   // (absent: the library's cost model)
    }
    if (gc.n_ranks < 1 || gc.n_ranks > 64) throw JsonError("device.gpus: must be 1..64");
    if (shard >= 0) { device = virt ? device : device + shard; gc.n_ranks = 1; }
    std::vector<int32_t> devs((size_t)gc.n_ranks);
    for (int r = 0; r < gc.n_ranks; r++) devs[(size_t)r] = virt ? device : device + r;
    gc.devices = devs.data();
    int rc = evplp_group_create(&cfg, &gc, &grp.g);
    if (rc < 0) throw std::runtime_error(std::string("evplp_group_create: ") + evplp_group_last_error(nullptr));
}
int device_gpus(const Json &json) {
    if (!json.has("device") || !json.at("device").has("gpus")) return 1;
    const int n = (int)json.at("device").at("gpus").as_int("device.gpus");
    if (n < 1 || n > 64) throw JsonError("device.gpus: must be 1..64");
    return n;
}
void upload_scene_group(evplp_group *g, const HostScene &scene) {
    for (int r = 0; r < evplp_group_size(g); r++) {
        evplp_context *h = evplp_group_context(g, r);
        if (upload_scene(h, scene) < 0) throw std::runtime_error(std::string("scene upload: ") + evplp_last_error(h));
    }
}
// setupPhotonSplatIcosohedron("sphere/icosphere.obj") (rtcomphoton.h:632-644, 677): the proxy mesh the photon splat draws around every
// photon.  The reference opens the path relative to its working directory; here it is looked for there and next to the scene JSON, and
// a build-only key "splatProxy" names another file.  The asset is a Git-LFS pointer in the reference's repository: when no mesh file
// is found (or the file is such a pointer) the generated 42-vertex / 80-face icosphere stands in (evplp_default_splat_proxy) and a
// note says so.  A file that IS a mesh but not a closed convex one is an error (evplp_set_splat_proxy refuses it).
// Build-only key "splatFootprint": "proxy" (default: the reference's coverage rule) | "ideal" (the radius test alone).
uint32_t setup_splat_footprint(evplp_group *g, const Json &json, const std::string &json_dir) {
    std::string mode = json.has("splatFootprint") ? json.at("splatFootprint").as_string("splatFootprint") : std::string("proxy");
    if (mode == "ideal") return (uint32_t)EVPLP_FOOTPRINT_IDEAL;
    if (mode != "proxy") throw JsonError("splatFootprint: expected \"proxy\" or \"ideal\", got \"" + mode + "\"");
    std::vector<std::string> candidates;
    const bool named = json.has("splatProxy");
    if (named) candidates.push_back(join_path(json_dir, json.at("splatProxy").as_string("splatProxy")));
    else { candidates.push_back("sphere/icosphere.obj"); candidates.push_back(join_path(json_dir, "sphere/icosphere.obj")); }
    for (const std::string &path : candidates) {
        { std::ifstream probe(path); if (!probe) { if (named) throw std::runtime_error("Impossible to load the scene: " + path); continue; } }
        MeshData m;
        try { m = load_single_mesh_obj(path); }
        catch (const std::exception &e) {
            if (std::string(e.what()).find("Git-LFS pointer") == std::string::npos) throw;
            std::fprintf(stderr, "note: %s is a Git-LFS pointer; the photon splat uses the generated 42-vertex icosphere as its proxy\n", path.c_str());
            break;
        }
        check(g, evplp_group_set_splat_proxy(g, m.verts.data(), (int32_t)(m.verts.size() / 3), m.idx.data(), (int32_t)(m.idx.size() / 3)), ("photon splat proxy " + path).c_str());
        return (uint32_t)EVPLP_FOOTPRINT_PROXY;
    }
    check(g, evplp_group_set_splat_proxy(g, nullptr, 0, nullptr, 0), "photon splat proxy");
    return (uint32_t)EVPLP_FOOTPRINT_PROXY;
}
// Output files named in the technique block.  The shipped scene files carry the authors' Windows paths
// ("C://result/conference/3_pm.pfm"): off Windows a drive-letter path keeps only its file name and lands next to
// the scene JSON, so those files run unchanged; every other path is used as the reference would (relative to the
// working directory there, to the JSON's directory here).
std::string output_path(const std::string &out_dir, const std::string &name) {
    if (name.size() > 1 && name[1] == ':' && ((name[0] >= 'A' && name[0] <= 'Z') || (name[0] >= 'a' && name[0] <= 'z'))) {
        size_t cut = name.find_last_of("/\\");
        std::string base = cut == std::string::npos ? name.substr(2) : name.substr(cut + 1);
        std::fprintf(stderr, "note: output \"%s\" has a drive letter; writing %s/%s\n", name.c_str(), out_dir.c_str(), base.c_str());
        return out_dir + "/" + base;
    }
    return join_path(out_dir, name);
}
// FloatImage::FlipY (floatimage.cpp:114-128) of a bottom-up RGB image + row de-interleave
std::vector<float> flip_y(const std::vector<float> &rgb, int w, int h) {
    std::vector<float> out((size_t)w * h * 3);
    for (int row = 0; row < h; row++) std::memcpy(&out[(size_t)row * w * 3], &rgb[(size_t)(h - 1 - row) * w * 3], sizeof(float) * 3 * w);
    return out;
}
// build-only additions to the stat file: per-pass device times of the last iteration
void add_pass_times(evplp_group *g, Json &st) {
    evplp_context *h = evplp_group_context(g, 0);
    const char *names[EVPLP_PASS_COUNT] = { "primaryMs", "lightTraceMs", "gatherVplMs", "gatherVslMs", "splatMs", "resolveMs", "pathTraceMs", "gatherLvcMs" };
    for (int p = 0; p < EVPLP_PASS_COUNT; p++) { evplp_pass_stats ps; if (evplp_pass_stats_get(h, p, &ps) == EVPLP_OK && ps.ms > 0) st.set(names[p], Json::number(ps.ms)); }
}
// IndependentSampler(mRngOffset).nextVec2() (common/rng.h:9-44, sampler/independent.h:37-40) as the reference's own headers
// behave when compiled here with g++ 11 / libstdc++ (pinned by tests/golden/jitter.npz, generated from oracle/_ref):
//   * std::uniform_real_distribution<float>(0, 1) over std::mt19937 = generate_canonical<float, 24>: float(x) / 2^32 with the
//     32-bit draw x converted to float by round-to-nearest, and a result of 1.0 replaced by the float below it;
//   * Vec2(nextFloat(), nextFloat()): g++ evaluates the two arguments right to left, so .y takes the FIRST draw.
// Both are implementation-defined in C++; the authors built with MSVC, whose library and argument order may differ --
// statistically equivalent, but a different jitter sequence.
struct JitterSampler {
    std::mt19937 rng;
    explicit JitterSampler(uint32_t seed) : rng(seed) {}
    float next() { float u = (float)(uint32_t)rng() / 4294967296.0f; return u >= 1.0f ? 0.99999994f : u; }
    // ndc jitter (2 u - 1) * invResolution (rtcomphoton.h:946-952, rtpt2.h:617-623)
    void next_jitter(int W, int H, float jitter[2]) {
        float uy = next(), ux = next();
        jitter[0] = (2.0f * ux - 1.0f) * (1.0f / (float)W); jitter[1] = (2.0f * uy - 1.0f) * (1.0f / (float)H);
    }
};
} // namespace

// The reference's ground-truth technique: unidirectional path tracing with next-event estimation, one sample per
// pixel per iteration ("numSamplePerPixel" is read and unused, rtpt2.h:109).
class PathTraceTechnique {
public:
    // rtpt2.h:84-116
    void render(const HostScene &scene, int res_x, int res_y, const Json &json, const std::string &out_dir, int device) {
        rng_offset = (uint32_t)json.at("rngOffset").as_int("rngOffset");
        num_max_iteration = (int)json.at("numMaxIteration").as_int("numMaxIteration");
        time_limit_ms = json.at("timeLimitMs").as_float("timeLimitMs");
        {
            const std::string &fm = json.at("frameMode").as_string("frameMode");
            auto it = kFrameModes.find(fm);
            if (it == kFrameModes.end()) throw JsonError("frameMode: unknown value \"" + fm + "\"");
            frame_mode = it->second;
        }
        output_filename = output_path(out_dir, json.at("outputFilename").as_string("outputFilename"));
        stat_filename = output_path(out_dir, json.at("statFilename").as_string("statFilename"));
        use_jitter = json.at("useJitter").as_bool("useJitter");
        use_stat = json.at("useStat").as_bool("useStat");
        (void)json.at("numSamplePerPixel").as_int("numSamplePerPixel");
        num_max_bounce = (int)json.at("numMaxBounces").as_int("numMaxBounces");
        write_every_frame = json.has("writeEveryFrame") ? json.at("writeEveryFrame").as_bool("writeEveryFrame") : false;
        int bvh_builder = EVPLP_BVH_SAH;
        if (json.has("bvhBuilder")) bvh_builder = parse_bvh_builder(json.at("bvhBuilder").as_string("bvhBuilder"));

        evplp_config cfg; std::memset(&cfg, 0, sizeof(cfg));
        cfg.abi_version = EVPLP_ABI_VERSION; cfg.device = device; cfg.res_x = res_x; cfg.res_y = res_y;
        cfg.strip_rank = 0; cfg.strip_count = 1; cfg.strip_rows = 8;
        cfg.num_light_paths = 1; cfg.num_vpl_light_paths = 1; cfg.photons_per_path = 1;   // no light sub-paths in this technique
        cfg.bvh_builder = bvh_builder;
        Grp grp; create_group(grp, cfg, json, device);
        upload_scene_group(grp.g, scene);
        run(grp.g, scene, res_x, res_y);
    }

private:
    // rtpt2.h:575-719
    void run(evplp_group *h, const HostScene &scene, int W, int H) {
        JitterSampler sampler(rng_offset);
        check(h, evplp_group_clear_accumulators(h), "clear");
        int num_iterations = 0;
        auto t0 = std::chrono::steady_clock::now();
        auto elapsed_ms = [&]() { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        std::vector<float> rgb((size_t)W * H * 3);
        const bool clear_every_frame = frame_mode == 2;
        for (;;) {
            if (num_iterations == num_max_iteration) break;                                   // :610-613
            float jitter[2] = { 0.f, 0.f };
            if (use_jitter) sampler.next_jitter(W, H, jitter);                                // :618-624
            check(h, evplp_group_primary(h, jitter, clear_every_frame ? (EVPLP_LIGHT_CLEAR | EVPLP_LIGHT_UNOCCLUDED) : 0), "primary");        // :626-629, 633-643
            check(h, evplp_group_path_trace(h, scene.camera.origin, (uint32_t)num_iterations + rng_offset, (uint32_t)num_max_bounce,
                                            clear_every_frame ? 0 : 1), "path trace");       // :631
            num_iterations++;
            if (write_every_frame) {                                                          // :669-689
                size_t i = output_filename.find_last_of('.');
                save(h, W, H, num_iterations, output_filename.substr(0, i) + "_" + std::to_string(num_iterations) + output_filename.substr(i), rgb);
            }
            if (time_limit_ms < 1e8f) check(h, evplp_group_synchronize(h), "sync");
            if (elapsed_ms() >= time_limit_ms) break;                                         // :667
        }
        check(h, evplp_group_synchronize(h), "sync");
        float time = elapsed_ms();
        if (use_stat) {                                                                       // :694-704
            Json st = Json::object();
            st.set("time", Json::number(time)); st.set("numIterations", Json::number(num_iterations));
            add_pass_times(h, st);
            std::ofstream of(stat_filename);
            if (!of) throw std::runtime_error("cannot write " + stat_filename);
            of << st.dump() << "\n";
        }
        save(h, W, H, num_iterations, output_filename, rgb);                                  // :706-719
    }
    // clear-every-frame: the composite as shown (masked emitter); accumulate: light image + path-traced image / n
    void save(evplp_group *h, int W, int H, int n, const std::string &path, std::vector<float> &rgb) {
        if (frame_mode == 2) check(h, evplp_group_resolve(h, 1.0f, 0.0f, 1.0f, 1, 0, rgb.data()), "resolve");
        else check(h, evplp_group_resolve(h, 1.0f / (float)std::max(n, 1), 0.0f, 1.0f, 0, 0, rgb.data()), "resolve");
        std::vector<float> top = flip_y(rgb, W, H);
        if (save_image(path.c_str(), W, H, top.data()) != EVPLP_OK) throw std::runtime_error("cannot write " + path);
    }

    uint32_t rng_offset = 0; int num_max_iteration = 0, num_max_bounce = 0, frame_mode = 1;
    float time_limit_ms = 0.f;
    bool use_jitter = false, use_stat = false, write_every_frame = false;
    std::string output_filename, stat_filename;
};

class ComPhotonTechnique {
public:
    // lvc: the "lvcphotonfam" twin (rtlvccomphoton.h).  It differs from RtComPhoton in the gather program
    // (per-pixel light-path window, lvclighttracing.cu:348-384), has no forceVsl / writeEveryFrame, and its stat
    // file carries only "time".
    explicit ComPhotonTechnique(bool lvc_variant = false) : lvc(lvc_variant) {}
    // rtcomphoton.h:107-223
    void render(const HostScene &scene, int res_x, int res_y, const Json &json, const std::string &out_dir, int device) {
        num_light_paths = (int)json.at("numLightPaths").as_int("numLightPaths");
        num_vpl_light_paths = (int)json.at("numVplLightPaths").as_int("numVplLightPaths");
        num_max_bounce = (int)json.at("numMaxBounces").as_int("numMaxBounces");
        photons_per_path = num_max_bounce + 1;
        radius_percentage = json.at("radiusPercentage").as_float("radiusPercentage");
        write_every_frame = (!lvc && json.has("writeEveryFrame")) ? json.at("writeEveryFrame").as_bool("writeEveryFrame") : false;
        num_max_iteration = (int)json.at("numMaxIteration").as_int("numMaxIteration");
        time_limit_ms = json.at("timeLimitMs").as_float("timeLimitMs");
        {
            const std::string &fm = json.at("frameMode").as_string("frameMode");
            auto it = kFrameModes.find(fm);
            if (it == kFrameModes.end()) throw JsonError("frameMode: unknown value \"" + fm + "\"");
            frame_mode = it->second;
        }
        if (!json.has("misMode")) mis_mode = 1;   // Balance (:128-131)
        else {
            const std::string &mm = json.at("misMode").as_string("misMode");
            auto it = kMisModes.find(mm);
            if (it == kMisModes.end()) throw JsonError("misMode: unknown value \"" + mm + "\"");
            mis_mode = it->second;
        }
        if (json.has("clampingStart"))            // :137-142
            throw JsonError("clampingStart option is not use anymore; remove it from your JSON file");
        if (json.has("targetRenderingTime")) target_rendering_time = json.at("targetRenderingTime").as_float("targetRenderingTime");
        rng_offset = (uint32_t)json.at("rngOffset").as_int("rngOffset");
        combined_filename = output_path(out_dir, json.at("combinedFilename").as_string("combinedFilename"));
        weighted_photon_filename = output_path(out_dir, json.at("weightedPhotonFilename").as_string("weightedPhotonFilename"));
        weighted_vpl_filename = output_path(out_dir, json.at("weightedVplFilename").as_string("weightedVplFilename"));
        stat_filename = output_path(out_dir, json.at("statFilename").as_string("statFilename"));
        use_jitter = json.at("useJitter").as_bool("useJitter");
        use_stat = json.at("useStat").as_bool("useStat");
        if (json.has("DoProgressive")) do_progressive = json.at("DoProgressive").as_bool("DoProgressive");
        if (json.has("AlphaProgressive")) alpha_progressive = json.at("AlphaProgressive").as_float("AlphaProgressive");
        if (json.has("run")) {                    // :188-197
            const Json &r = json.at("run");
            if (r.has("deferredShading")) do_deferred = r.at("deferredShading").as_bool("run.deferredShading");
            if (r.has("lightTracing")) do_light_tracing = r.at("lightTracing").as_bool("run.lightTracing");
            if (r.has("vplSplat")) do_vpl_splat = r.at("vplSplat").as_bool("run.vplSplat");
            if (r.has("photonSplat")) do_photon_splat = r.at("photonSplat").as_bool("run.photonSplat");
            if (r.has("lightRender")) do_light_render = r.at("lightRender").as_bool("run.lightRender");
            if (r.has("finalize")) do_finalize = r.at("finalize").as_bool("run.finalize");
        }
        if (num_vpl_light_paths == 0) { std::printf("WARN: 0 VPL light paths. Disable mDoVplSplat\n"); do_vpl_splat = false; }   // :200-203
        if (!lvc && json.has("forceVsl")) force_vsl = json.at("forceVsl").as_bool("forceVsl");
        if (json.has("bvhBuilder")) bvh_builder = parse_bvh_builder(json.at("bvhBuilder").as_string("bvhBuilder"));   // build-only key

        // ---- setup(): context + scene upload (replaces GL/OptiX setup :646-708)
        evplp_config cfg; std::memset(&cfg, 0, sizeof(cfg));
        cfg.abi_version = EVPLP_ABI_VERSION; cfg.device = device; cfg.res_x = res_x; cfg.res_y = res_y;
        cfg.strip_rank = 0; cfg.strip_count = 1; cfg.strip_rows = 8;
        cfg.num_light_paths = (uint32_t)num_light_paths; cfg.num_vpl_light_paths = (uint32_t)num_vpl_light_paths;
        cfg.photons_per_path = (uint32_t)photons_per_path; cfg.bvh_builder = bvh_builder;
        if (json.has("deterministic")) cfg.deterministic = json.at("deterministic").as_bool("deterministic") ? 1 : 0;   // build-only key
        cfg.overlap_light_tracing = 1;      // the loop below calls primary, then light tracing: they overlap
        if (json.has("device")) {           // build-only: the bounds of the gathers' scratch buffers (evplp_config; defaults 8 GB / 2 GB)
            const Json &d = json.at("device");
            if (d.has("cutScratchGB")) cfg.cut_scratch_bytes = (uint64_t)(std::max(d.at("cutScratchGB").as_float("device.cutScratchGB"), 0.0f) * 1073741824.0);
            if (d.has("vslMaskGB")) cfg.vsl_mask_bytes = (uint64_t)(std::max(d.at("vslMaskGB").as_float("device.vslMaskGB"), 0.0f) * 1073741824.0);
        }
        run_opts = run_options(json);
        const int gpus = device_gpus(json);
        if (run_opts.shard_iterations && (frame_mode != 1 || gpus < 2)) {
            if (gpus >= 2) std::printf("note: device.partition \"iterations\" needs frameMode \"accumulate\"; running on row strips\n");
            run_opts.shard_iterations = false;
        }
        // one group of N strip ranks -- or N groups of one whole-image rank each, one per shard of the iterations
        std::vector<Grp> grps((size_t)(run_opts.shard_iterations ? gpus : 1));
        std::vector<evplp_group *> hs;
        for (size_t k = 0; k < grps.size(); k++) {
            create_group(grps[k], cfg, json, device, run_opts.shard_iterations ? (int)k : -1);
            upload_scene_group(grps[k].g, scene);
            splat_footprint = setup_splat_footprint(grps[k].g, json, out_dir);                                  // :677
            hs.push_back(grps[k].g);
        }
        Grp &grp = grps[0];
        float bsr = 0.f, total_area = 0.f, light_area = 0.f;
        { evplp_context *h0 = evplp_group_context(grp.g, 0);
          if (evplp_scene_metrics(h0, &bsr, &total_area, &light_area) < 0) throw std::runtime_error(std::string("scene metrics: ") + evplp_last_error(h0)); }

        photon_radius = bsr * radius_percentage;                                                                 // :118-119
        pdf_mc = (float)num_vpl_light_paths / (float)num_light_paths * kInvPi / (photon_radius * photon_radius); // :120
        if (!json.has("clampingCoeff")) {                                                                        // :148-161
            std::printf("Total area computation: %g\n", total_area);
            clamping_value = clamping_start = 1.f / total_area;
        } else clamping_value = clamping_start = json.at("clampingCoeff").as_float("clampingCoeff");
        if (force_vsl) {                                                                                         // :205-218
            float pct = json.at("vslRadiusPercentage").as_float("vslRadiusPercentage");
            vsl_radius = bsr * pct;
            if (vsl_radius <= 0.008f) { vsl_radius = std::max(vsl_radius, 0.008f); std::printf("warning : vslRadius is too small. clamped vslRadius\n"); }
            vsl_inv_pi_radius2 = kInvPi / (vsl_radius * vsl_radius);
        }
        run(hs, scene, res_x, res_y);
    }

private:
    evplp_frame_params params(const HostScene &scene, uint32_t seed, const float jitter[2]) const {
        evplp_frame_params fp; std::memset(&fp, 0, sizeof(fp));
        std::memcpy(fp.camera_pos, scene.camera.origin, 12);
        fp.mis_mode = (uint32_t)mis_mode; fp.pdf_mc = pdf_mc; fp.clamping_value = clamping_value; fp.photon_radius = photon_radius;
        fp.vsl_radius = vsl_radius; fp.vsl_inv_pi_radius2 = vsl_inv_pi_radius2;
        fp.num_light_paths = (uint32_t)num_light_paths; fp.num_vpl_light_paths = (uint32_t)num_vpl_light_paths;
        fp.photons_per_path = (uint32_t)photons_per_path;
        fp.do_accumulate = frame_mode == 2 ? 0u : 1u;   // :923-930
        fp.rng_seed = seed; fp.jitter[0] = jitter[0]; fp.jitter[1] = jitter[1];
        fp.splat_footprint = splat_footprint;
        return fp;
    }

    // rtcomphoton.h:883-1133
    // hs: ONE group (of one or more strip ranks), or one single-rank group per shard of the iterations (RunOptions::shard_iterations)
    void run(const std::vector<evplp_group *> &hs, const HostScene &scene, int W, int H) {
        JitterSampler sampler(rng_offset);
        evplp_group *h = hs[0];
        const int S = (int)hs.size();
        auto all = [&](int (*fn)(evplp_group *), const char *what) { for (evplp_group *q : hs) check(q, fn(q), what); };
        all(evplp_group_clear_accumulators, "clear");
        int num_iterations = 0;
        auto t0 = std::chrono::steady_clock::now();
        // Row blocks dealt by cost (RunOptions above): one frame of the first iteration's light paths and gather with the self-clocking
        // kernels, un-jittered, then the deal.  Nothing of it reaches the images: the rebalance clears the accumulators, the loop below traces
        // the same light paths again, the jitter sequence and the progressive state have not moved.  Its time is part of the run's.
        const bool can_deal = S == 1 && evplp_group_size(h) > 1 && do_vpl_splat && !lvc && do_deferred && do_light_tracing;
        // (default: when the calibration frame pays for itself.  It costs one frame -- clocking a fraction of the VPLs deals worse than round robin,
        // measured -- and a dealt frame is ~1 % / ~4 % / ~20 % shorter than a round-robin one at 2 / 4 / 8 ranks, profiles/r06_strip_projection.json)
        const int ranks = evplp_group_size(h), pays_from = ranks >= 6 ? 5 : ranks >= 3 ? 25 : 100;
        if (can_deal && (run_opts.deal == 1 || (run_opts.deal < 0 && num_max_iteration >= pays_from))) {
            float j0[2] = { 0.f, 0.f };
            evplp_frame_params fp = params(scene, rng_offset, j0);
            check(h, evplp_group_calibrate(h, 1), "calibrate");
            check(h, evplp_group_primary(h, j0, EVPLP_LIGHT_SKIP), "primary (calibration)");
            check(h, evplp_group_trace_light_paths(h, rng_offset), "light tracing (calibration)");
            check(h, evplp_group_gather(h, &fp, force_vsl ? 1 : 0), "gather (calibration)");
            check(h, evplp_group_rebalance(h, nullptr), "rebalance");
        }
        auto elapsed_ms = [&]() { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        float prev_timing = 0.f;
        std::vector<float> rgb((size_t)W * H * 3);
        // The two events per pass behind the stat file's pass times cost a sub-millisecond iteration ~3 % (evplp_profile_passes): a loop
        // that ends by iteration count records them in its last iteration only; one with a time limit (it waits for every frame anyway) always.
        const bool always_profile = time_limit_ms < 1e8f;
        for (evplp_group *q : hs) check(q, evplp_group_profile_passes(q, always_profile ? 1 : 0), "profile");
        for (;;) {
            if (num_iterations == num_max_iteration) break;                                   // :938-941
            h = hs[(size_t)(num_iterations % S)];                                             // (this iteration's shard; S = 1: the one group)
            if (!always_profile && num_iterations + S >= num_max_iteration) check(h, evplp_group_profile_passes(h, 1), "profile");   // (every shard's last iteration)
            float jitter[2] = { 0.f, 0.f };
            if (use_jitter) sampler.next_jitter(W, H, jitter);                                // :946-952
            evplp_frame_params fp = params(scene, (uint32_t)num_iterations + rng_offset, jitter);
            const int light_flags = !do_light_render ? EVPLP_LIGHT_SKIP : frame_mode == 2 ? (EVPLP_LIGHT_CLEAR | EVPLP_LIGHT_UNOCCLUDED) : 0;   // :985-995
            // The reference draws the G-buffer first; the two passes are independent.  Pure photon mapping (no gather): light paths
            // first -- they start on the context's second stream, into the record buffer nobody reads, before the host has waited for
            // the previous iteration's photon bins.  With a gather in the frame they run beside the G-buffer pass instead of beside the
            // gather, whose CUs they would share.
            const bool light_first = !do_vpl_splat;
            if (light_first && do_light_tracing) check(h, evplp_group_trace_light_paths(h, (uint32_t)num_iterations + rng_offset), "light tracing");  // :962-966
            if (do_deferred) check(h, evplp_group_primary(h, jitter, light_flags), "primary");   // :954-960
            if (!light_first && do_light_tracing) check(h, evplp_group_trace_light_paths(h, (uint32_t)num_iterations + rng_offset), "light tracing");
            if (do_vpl_splat) check(h, evplp_group_gather(h, &fp, lvc ? 2 : force_vsl ? 1 : 0), "gather");  // :968-972
            // radius 0 (radiusPercentage 0 of the VPL-only configs): the proxy spheres are degenerate, nothing is drawn
            if (do_photon_splat && photon_radius > 0.0f) check(h, evplp_group_splat_photons(h, &fp, frame_mode == 2 ? 1 : 0), "photon splat");    // :974-983
            // [finalize] runFinalProgram(param, param, 1, true) to the window (:997-1004): headless, the composite still runs -- it is part
            // of the reference's iteration -- and stays on the device
            if (do_finalize) {
                const float param = frame_mode == 2 ? 1.0f : 1.0f / (float)(num_iterations / S + 1);      // (a shard's own frame: over ITS iterations so far)
                const bool exchange = run_opts.exchange_every > 0 && (num_iterations + 1) % run_opts.exchange_every == 0;
                check(h, evplp_group_present_ex(h, param, param, 1.0f, 1, 1, exchange ? 1 : 0), "finalize");       // (doGammaCorrection = true, :1003)
            }
            num_iterations++;
            if (num_iterations % 20 == 0) {                                                   // :1008-1031
                all(evplp_group_synchronize, "sync");
                float cur = elapsed_ms();
                std::printf("numIter: %d | raduis: %g | clamping: %g | timing: %g\n", num_iterations, photon_radius, clamping_value, cur - prev_timing);
                if (target_rendering_time != -1.f) {
                    float frame_time = (cur - prev_timing) / 20.f, factor = target_rendering_time / frame_time;
                    if (factor != 1.f) std::printf("change number of samples: %g | currFrame time: %g\n", factor, frame_time);
                }
                prev_timing = cur;
            }
            if (do_progressive)                                                               // :1033-1063
                evplp_progressive_step(num_iterations, alpha_progressive, clamping_start, (uint32_t)num_vpl_light_paths, (uint32_t)num_light_paths,
                                       &photon_radius, &clamping_value, &pdf_mc, force_vsl ? 1 : 0, &vsl_radius, &vsl_inv_pi_radius2);
            if (write_every_frame) dump_frame(hs, W, H, num_iterations, rgb);                 // :1079-1102
            if (time_limit_ms < 1e8f) all(evplp_group_synchronize, "sync");                   // a wall-clock limit needs finished frames
            if (elapsed_ms() >= time_limit_ms) break;                                         // :1065
        }
        all(evplp_group_synchronize, "sync");
        float time = elapsed_ms();
        for (evplp_group *q : hs) check(q, evplp_group_profile_passes(q, 1), "profile");
        h = hs[0];
        if (use_stat) {                                                                       // :1109-1119
            Json st = Json::object();
            st.set("time", Json::number(time));
            if (!lvc) st.set("numIterations", Json::number(num_iterations));              // rtlvccomphoton.h writes the time only
            add_pass_times(h, st);
            std::ofstream of(stat_filename);
            if (!of) throw std::runtime_error("cannot write " + stat_filename);
            of << st.dump() << "\n";
        }
        float param = frame_mode == 2 ? 1.0f : 1.0f / (float)std::max(num_iterations, 1);     // :1122
        // :1124-1132: three composites, un-masked sums, FlipY, Save
        ShardSum sum(hs);                                    // (S > 1: shard 0's accumulators now hold the sum over the shards)
        auto compose = [&](float vs, float ps, float ls) {
            check(h, evplp_group_resolve(h, vs, ps, ls, 0, 0, rgb.data()), "resolve");
            return flip_y(rgb, W, H);
        };
        std::vector<float> combined = compose(param, param, 1.0f);
        std::vector<float> vpl = compose(param, 0.0f, 1.0f);
        std::vector<float> pm = compose(0.0f, param, 0.0f);
        if (save_image(combined_filename.c_str(), W, H, combined.data()) != EVPLP_OK) throw std::runtime_error("cannot write " + combined_filename);
        if (save_image(weighted_vpl_filename.c_str(), W, H, vpl.data()) != EVPLP_OK) throw std::runtime_error("cannot write " + weighted_vpl_filename);
        if (save_image(weighted_photon_filename.c_str(), W, H, pm.data()) != EVPLP_OK) throw std::runtime_error("cannot write " + weighted_photon_filename);
    }

    // Iterations sharded over several contexts: a written frame needs the SUM of the shards' VPL and photon accumulators.  They are brought to
    // the host, added in shard order (a fixed association: the same run gives the same bits again) and put into shard 0's buffers, whose own
    // contents come back when this object goes (the run may go on accumulating).  The emitter image is the same on every shard: the light
    // mesh is drawn through the un-jittered matrix (rtcomphoton.h:720-727), every iteration writes the same pixels.
    struct ShardSum {
        const std::vector<evplp_group *> &hs; std::vector<std::vector<float>> own;
        explicit ShardSum(const std::vector<evplp_group *> &shards) : hs(shards) {
            if (hs.size() < 2) return;
            static const int kPlanes[2] = { EVPLP_BUF_VPL_ACCUM, EVPLP_BUF_PHOTON_ACCUM };
            for (evplp_group *q : hs) check(q, evplp_group_synchronize(q), "sync");
            for (int k = 0; k < 2; k++) {
                evplp_context *c0 = evplp_group_context(hs[0], 0);
                size_t bytes = 0;
                if (evplp_buffer_info(c0, kPlanes[k], nullptr, &bytes) < 0) throw std::runtime_error(std::string("accumulator size: ") + evplp_last_error(c0));
                std::vector<float> acc(bytes / sizeof(float)), part(bytes / sizeof(float));
                if (evplp_download(c0, kPlanes[k], acc.data(), bytes) < 0) throw std::runtime_error(std::string("accumulator download: ") + evplp_last_error(c0));
                own.push_back(acc);
                for (size_t r = 1; r < hs.size(); r++) {
                    evplp_context *c = evplp_group_context(hs[r], 0);
                    if (evplp_download(c, kPlanes[k], part.data(), bytes) < 0) throw std::runtime_error(std::string("accumulator download: ") + evplp_last_error(c));
                    for (size_t i = 0; i < acc.size(); i++) acc[i] += part[i];
                }
                if (evplp_upload(c0, kPlanes[k], acc.data(), bytes) < 0) throw std::runtime_error(std::string("accumulator upload: ") + evplp_last_error(c0));
            }
        }
        ~ShardSum() {
            static const int kPlanes[2] = { EVPLP_BUF_VPL_ACCUM, EVPLP_BUF_PHOTON_ACCUM };
            for (size_t k = 0; k < own.size(); k++) evplp_upload(evplp_group_context(hs[0], 0), kPlanes[k], own[k].data(), own[k].size() * sizeof(float));
        }
    };
    void dump_frame(const std::vector<evplp_group *> &hs, int W, int H, int iter, std::vector<float> &rgb) {
        float param = frame_mode == 2 ? 1.0f : 1.0f / (float)iter;                            // :1088
        ShardSum sum(hs);
        evplp_group *h = hs[0];
        check(h, evplp_group_resolve(h, param, param, 1.0f, 0, 0, rgb.data()), "resolve");
        std::vector<float> top = flip_y(rgb, W, H);
        size_t i = weighted_photon_filename.find_last_of('.');                                // :1097-1101
        std::string path = weighted_photon_filename.substr(0, i) + "_" + std::to_string(iter) + weighted_photon_filename.substr(i);
        if (save_image(path.c_str(), W, H, top.data()) != EVPLP_OK) throw std::runtime_error("cannot write " + path);
    }

    int num_light_paths = 0, num_vpl_light_paths = 0, num_max_bounce = 0, photons_per_path = 0;
    float radius_percentage = 0.f, photon_radius = 0.f, pdf_mc = 0.f;
    uint32_t rng_offset = 0; int num_max_iteration = 0; int frame_mode = 1; int mis_mode = 1;
    float time_limit_ms = 0.f, clamping_value = 0.f, clamping_start = 0.f;
    bool use_jitter = false, use_stat = false;
    bool do_deferred = true, do_light_tracing = true, do_vpl_splat = true, do_photon_splat = true, do_light_render = true, do_finalize = true;
    bool do_progressive = false, write_every_frame = false; float alpha_progressive = 0.7f;
    float target_rendering_time = -1.f;
    std::string combined_filename, weighted_photon_filename, weighted_vpl_filename, stat_filename;
    bool force_vsl = false; float vsl_radius = 0.f, vsl_inv_pi_radius2 = 0.f;
    bool lvc = false;
    uint32_t splat_footprint = EVPLP_FOOTPRINT_PROXY;
    RunOptions run_opts;
    int bvh_builder = EVPLP_BVH_SAH;   // measured 9% faster frames than the Morton LBVH on the conference stand-in; "bvhBuilder": "lbvh" selects the LBVH
};

} // namespace evplp

extern "C" int evplp_jitter_sequence(uint32_t rng_offset, int32_t count, int32_t res_x, int32_t res_y, float *out_ndc_xy) {
    if (count < 0 || res_x <= 0 || res_y <= 0 || (count > 0 && !out_ndc_xy)) return EVPLP_ERR_INVALID;
    evplp::JitterSampler s(rng_offset);
    for (int32_t i = 0; i < count; i++) s.next_jitter(res_x, res_y, out_ndc_xy + 2 * (size_t)i);
    return EVPLP_OK;
}

// The product's JSON reader as the technique blocks use it (pinned against the reference's nlohmann::json 2.1.1 by
// tests/test_oracle_pins.py): `path` = keys separated by '/', decimal indices into arrays.  want: 0 int, 1 float, 2 bool,
// 3 string, 4 size, 5 kind (0 null, 1 bool, 2 number, 3 string, 4 array, 5 object).
// Returns 0 ok, 1 parse error, 2 key missing / index out of range, 3 conversion error.
extern "C" int evplp_json_query(const char *text, const char *path, int32_t want, double *num, char *str, int32_t cap) {
    using namespace evplp;
    if (!text || !path || !num) return EVPLP_ERR_INVALID;
    Json root;
    try { root = Json::parse(text); } catch (const std::exception &) { return 1; }
    const Json *j = &root;
    const std::string p(path);
    size_t at = 0;
    try {
        while (at < p.size()) {
            size_t e = p.find('/', at); if (e == std::string::npos) e = p.size();
            const std::string key = p.substr(at, e - at);
            at = e + 1;
            if (j->is_array()) { const size_t i = (size_t)std::stoul(key); if (i >= j->size()) return 2; j = &j->at(i); }
            else if (j->is_object()) { if (!j->has(key)) return 2; j = &j->at(key); }
            else return 2;
        }
    } catch (const std::exception &) { return 2; }
    try {
        switch (want) {
        case 0: *num = (double)(int)j->as_int(); break;
        case 1: *num = (double)j->as_float(); break;
        case 2: *num = j->as_bool() ? 1.0 : 0.0; break;
        case 3: { const std::string v = j->as_string(); if (!str || (int32_t)v.size() + 1 > cap) return 3; std::memcpy(str, v.data(), v.size()); str[v.size()] = 0; *num = (double)v.size(); break; }
        case 4: *num = (double)j->size(); break;
        default: *num = (double)j->kind(); break;
        }
    } catch (const std::exception &) { return 3; }
    return 0;
}

extern "C" int evplp_load_scene_json(evplp_context *ctx, const char *json_path) {
    using namespace evplp;
    if (!ctx || !json_path) return EVPLP_ERR_INVALID;
    try {
        Json root = Json::parse(read_text_file(json_path));
        HostScene scene = load_scene(root, json_path);
        return upload_scene(ctx, scene);
    } catch (const JsonError &e) { set_context_error(ctx, e.what()); return EVPLP_ERR_PARSE; }
    catch (const std::exception &e) { set_context_error(ctx, e.what()); return EVPLP_ERR_IO; }   // e.g. a Git-LFS pointer instead of a mesh, a missing file
}

extern "C" int evplp_render_json(const char *json_path, const char *json_overrides, int32_t device, char *err, size_t err_cap) {
    using namespace evplp;
    auto fail = [&](int code, const std::string &msg) { if (err && err_cap) { std::snprintf(err, err_cap, "%s", msg.c_str()); } return code; };
    if (!json_path) return fail(EVPLP_ERR_INVALID, "null json path");
    try {
        std::string text;
        try { text = read_text_file(json_path); } catch (const std::exception &e) { return fail(EVPLP_ERR_IO, e.what()); }
        Json root = Json::parse(text);                                                  // main.cpp:99-102
        HostScene scene;
        try { scene = load_scene(root, json_path); }                                    // main.cpp:104
        catch (const JsonError &e) { return fail(EVPLP_ERR_PARSE, e.what()); }
        catch (const std::exception &e) { return fail(EVPLP_ERR_IO, e.what()); }
        // every technique block that is present runs, in the reference's order (main.cpp:105-121)
        bool ran = false;
        auto block_of = [&](const char *key) {
            Json block = root.at(key);
            if (json_overrides && *json_overrides) block.merge(Json::parse(json_overrides));
            return block;
        };
        if (root.has("pt") && !root.at("pt").is_null()) {                               // main.cpp:105-109
            PathTraceTechnique t;
            t.render(scene, scene.res_x, scene.res_y, block_of("pt"), dirname_of(json_path), device);
            ran = true;
        }
        if (root.has("photonfam") && !root.at("photonfam").is_null()) {                 // main.cpp:111-115
            ComPhotonTechnique t;
            t.render(scene, scene.res_x, scene.res_y, block_of("photonfam"), dirname_of(json_path), device);
            ran = true;
        }
        if (root.has("lvcphotonfam") && !root.at("lvcphotonfam").is_null()) {           // main.cpp:117-121
            ComPhotonTechnique t(/*lvc_variant=*/true);
            t.render(scene, scene.res_x, scene.res_y, block_of("lvcphotonfam"), dirname_of(json_path), device);
            ran = true;
        }
        if (!ran) return fail(EVPLP_ERR_PARSE, "no technique block (\"pt\", \"photonfam\", \"lvcphotonfam\") in the scene JSON");
    } catch (const JsonError &e) { return fail(EVPLP_ERR_PARSE, e.what()); }
    catch (const std::exception &e) { return fail(EVPLP_ERR_HIP, e.what()); }
    return EVPLP_OK;
}
