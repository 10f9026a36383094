// evplp_context: owns all device memory of one GPU's share of the frame.
#pragma once
#include "evplp_types.h"
#include "kernels.h"

#include <string>
#include <thread>
#include <vector>

namespace evplp {
// bvh_gpu.hip: LBVH built on the device (arrays are device allocations owned by the caller)
int build_bvh_gpu(const float *verts_host, int32_t ntri, float pad_scale, hipStream_t stream, BvhDeviceBuild *out);
int build_nodes4(const BvhNode *d_nodes, int32_t nnodes, hipStream_t stream, BvhNode4 **out);
struct HostMesh { std::vector<float> verts, uvs; std::vector<int32_t> idx; int32_t material = 0; };
struct HostTexture { int32_t w = 0, h = 0; std::vector<float> rgba; };
struct HostStats { uint64_t rays = 0; };
// host/proxy_mesh.cpp: the proxy mesh of EVPLP_FOOTPRINT_PROXY as slabs (kernels.h ProxyDev)
struct ProxyHost { std::vector<float4> slabs; std::vector<float> hm; int32_t planes = 0; float rin = 0.f, rout = 0.f; };
void default_splat_proxy(std::vector<float> &verts, std::vector<int32_t> &tris);
bool build_proxy_slabs(const float *verts, int32_t nverts, const int32_t *tris, int32_t ntris, ProxyHost *out, std::string *why);
}

struct evplp_context {
    evplp_config cfg{};
    evplp::StripDev st{};
    int32_t rows_in_image = 0;
    // dealt blocks (evplp_set_blocks): the table behind st.blocks / st.blocks_host, [cap_blocks] local -> image block, then [image blocks] image -> local block
    std::vector<int32_t> blocks_host; int32_t *d_blocks = nullptr; int32_t image_blocks = 0;
    // evplp_calibrate_blocks: while on, the gathers run their self-clocking variants and add every item's resident time to its local block's counter
    bool calibrate = false; unsigned long long *d_block_cost = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev_begin[EVPLP_PASS_COUNT] = {}, ev_end[EVPLP_PASS_COUNT] = {};
    hipEvent_t ev_dom_begin[EVPLP_PASS_COUNT] = {}, ev_dom_end[EVPLP_PASS_COUNT] = {};
    bool pass_ran[EVPLP_PASS_COUNT] = {}, pass_has_dom[EVPLP_PASS_COUNT] = {};
    // evplp_profile_passes: off = a pass records only the events the library itself waits on (the G-buffer pass's start for the
    // overlapped light tracing, the record readers', a splat's verdict); its HIP-event time is then not available (pass_timed)
    bool profile_passes = true, pass_timed[EVPLP_PASS_COUNT] = {};
    // The events around the dominant kernel of the photon splat sit BETWEEN its three dependent launches and hold them apart (18 us of a
    // 227 us pass, round 3): recorded only after evplp_profile_kernels(ctx, 1)
    bool profile_kernels = false;
    evplp::HostStats stats_host[EVPLP_PASS_COUNT];

    void *buf[EVPLP_BUF_COUNT] = {};
    bool buf_owned[EVPLP_BUF_COUNT] = {};

    // host-side scene staging (the RtScene contract, rt/rtcommon.h:816-819)
    std::vector<evplp::HostMesh> meshes;
    std::vector<evplp::Material> materials;
    std::vector<evplp::HostTexture> textures;
    int32_t light_mesh = -1;
    float light_unscaled[4] = {}, light_scaled[4] = {};
    evplp::CamBasis cam{}; evplp_camera cam_in{};
    bool camera_set = false, accel_built = false;
    float bounding_radius = 0.f, total_area = 0.f, light_area = 0.f;
    int32_t accel_nodes = 0, accel_leaves = 0, accel_depth = 0, accel_builder_used = -1; float accel_build_ms = 0.f;

    evplp::SceneDev sc{};
    evplp_record *d_vpls = nullptr; uint32_t *d_vpl_src = nullptr;
    uint32_t *d_scalars = nullptr;           // [0] usable VPL count, [8] splat overflow
    evplp::PassCounters *d_counters = nullptr;
    float *d_rgb = nullptr;
    // gather workspace, allocated on the first gather (path-tracing / photon-only contexts never pay for it)
    float4 *d_partial = nullptr; size_t partial_groups = 0;    // [groups][local_rows * W] per-item partial sums
    int32_t *d_lt_overflow = nullptr; size_t lt_overflow_bytes = 0;   // light tracing: the walk stack beyond its LDS entries (kernels.h)
    char *d_primary_cuts = nullptr; bool primary_cuts_valid = false;   // the eye's entry cuts, one slot per tile group, rebuilt when the camera or the tree changes
    char *d_cuts = nullptr; size_t cut_bytes = 0, cut_cap = 0;  // gathers: entry cuts of every (tile group, VPL) (kernels.h CutArgs), allocated on the first gather; cut_cap: their bound, fixed at the first gather
    size_t mask_cap = 0;                                       // bound of d_vsl_masks, fixed at the first VSL gather
    void *d_vsl_masks = nullptr; size_t vsl_mask_bytes = 0;    // VSL gather: lit masks + per-item ray counts of one launch (kernels.h GatherArgs)

    // splat workspace
    int32_t tiles_x = 0, tiles_y = 0; uint32_t bin_stride = 0, last_bin_entries = 0, last_bin_max = 0;   // bin_stride: slots per tile bin
    // overlap_light_tracing (evplp_config): light tracing on aux_stream, behind ev_records_read (recorded after every pass that reads
    // the record buffer), in front of whatever the main stream is given next (it waits for ev_light_done)
    hipStream_t aux_stream = nullptr; hipEvent_t ev_records_read = nullptr, ev_light_done = nullptr; bool light_in_flight = false;
    // ... and into a SECOND record buffer when the whole path set is traced and the library owns the records (nobody holds a pointer
    // to them): the call flips EVPLP_BUF_RECORDS to the buffer being written, so the light paths of iteration i + 1 do not wait for
    // the passes of iteration i that still read the other one (config #4: light tracing is the long pole and becomes a pipeline)
    void *records_back = nullptr; hipEvent_t ev_back_read = nullptr; bool records_exposed = false;
    // ... and the four G-buffer planes + the tile boxes likewise: evplp_primary writes the set the pending splat does not read, and it
    // is the next evplp_splat_photons call (by then that splat has long finished) that waits for its verdict
    void *gbuf_back[4] = { nullptr, nullptr, nullptr, nullptr }; float4 *d_tile_box_back = nullptr; bool gbuf_exposed = false;
    bool gbuf_pos_exposed = false;             // the caller holds a device pointer to the position plane (buffer_info / bind_buffer): it may write it unseen
    bool tile_box_valid = false;               // d_tile_box describes the current G-buffer (written by evplp_primary; any other way in clears it)
    int32_t num_bin_groups = 0, bucket_w_log2 = 0, bucket_h_log2 = 0, buckets_x = 0, num_buckets = 0;   // two-level binning (kernels.h)
    uint32_t *d_seg = nullptr, *d_big_list = nullptr, *d_big_count = nullptr; uint16_t *d_seg_off = nullptr;
    uint32_t *d_tile_cursor = nullptr, *d_bin_items = nullptr, *d_bin_items_tmp = nullptr;
    float4 *d_compact = nullptr; float4 *d_tile_box = nullptr; uint32_t *d_tile_pairs = nullptr; uint32_t *d_summary = nullptr;
    // EVPLP_FOOTPRINT_PROXY: the proxy mesh as slabs on the device (evplp_set_splat_proxy; the generated icosphere on first use)
    float4 *d_proxy_slabs = nullptr; float *d_proxy_hm = nullptr; int32_t proxy_count = 0; float proxy_rin = 0.f, proxy_rout = 0.f;
    uint32_t *d_tile_frags = nullptr; bool last_splat_proxy = false;
    // The bin sizes of a splat are known only on the device.  The pass is enqueued completely (fill and tiles kernels do
    // nothing when the bins overflowed); the summary arrives in pinned host memory behind ev_summary and is looked at by the
    // NEXT call on the context (settle_splat): no host round trip, no GPU bubble inside the pass.
    // MIXED tile launches of the photon splat (kernels.h SplatArgs): heavy list + per-tile flags; EVPLP_TILE_MIXED=0 keeps the pure variants
    uint32_t *d_heavy_list = nullptr; uint8_t *d_tile_flags = nullptr; uint32_t heavy_cap = 0, heavy_threshold = 256, mixed_trigger = 512; int env_tile_mixed = 1;
    uint32_t *h_summary = nullptr;            // pinned, 4 words per pending pass: [0] bin entries, [1] fullest bin, [2] overflow (slots the fullest bin needed)
    // Up to two passes wait for their verdict (oldest first).  One is the rule: every entry point settles it.  With
    // overlap_light_tracing a second may be in flight: evplp_splat_photons only LOOKS whether the previous one is known yet, and
    // evplp_primary / evplp_trace_light_paths wait for a pending splat only if they are about to overwrite what it read (by then
    // it is two passes old and long finished) -- the host then runs a whole iteration ahead of the GPU.
    struct PendingSplat { evplp::SplatArgs args; hipEvent_t ev = nullptr; uint32_t *h = nullptr; };
    PendingSplat pend[2];
    int npend = 0;

    // Test / developer overrides, read ONCE by evplp_create (never in a pass): EVPLP_BVH_BUILDER (every suite under every builder),
    // EVPLP_BIN_STRIDE (forces the photon-bin overflow path), EVPLP_GATHER_K, EVPLP_TILE_BLOCK_LOG2.  -1 / 0 = not set.
    int32_t env_bvh_builder = -1, env_gather_k = 0, env_tile_block_log2 = -1, env_cuts = -1;      // env_cuts: EVPLP_CUTS=0 walks from the root
    int32_t env_item_deal = -1;                // EVPLP_ITEM_DEAL: 0 tiles dealt to XCDs, 1 a tile's items over all XCDs (default: by launch size)
    int32_t env_split_min = 0;                 // EVPLP_SPLIT_MIN: fullest bin from which the splat's tile kernel runs four waves per tile
    size_t env_cut_bytes = 0;                  // EVPLP_CUT_BYTES: test override of evplp_config.cut_scratch_bytes

    // A context that belongs to an evplp_group is driven by that rank's worker thread (group.cpp).  A call from any other thread -- the
    // caller reading statistics or buffers through evplp_group_context -- first waits until the worker has nothing queued for it.
    void (*quiesce)(void *) = nullptr; void *quiesce_arg = nullptr; std::thread::id worker_tid{};

    char error[512] = "";
    void set_error(const char *fmt, ...);
};
