// Kernel argument blocks and host-side launchers (defined in the .hip files).
#pragma once
#include "evplp_types.h"

namespace evplp {

struct PrimaryArgs {
    SceneDev sc; StripDev st; CamBasis cam;
    float jitter[2]; int32_t clear_light; int32_t pad;
    float4 *g_pos, *g_nrm, *g_dif, *g_phg, *g_light;
    float4 *tile_box;         // [tiles][2] world-space box of every 8x8-px tile's positions (the photon splat's bin cull), or null
    // entry cuts of the primary rays (one slot per tile group, built once per camera by primary_cut_kernel; null: walks from the root)
    const char *cuts; int32_t cut_gw_log2, cut_gh_log2, cut_groups_x, pad2;
};
// The eye's entry cuts: the pyramid through a tile group's pixels -- opened by one pixel on every side, so that it holds for every
// anti-aliasing jitter of the camera -- descends the tree like a (tile group, VPL) pyramid (CutArgs below); lane = tile group.
struct PrimaryCutArgs {
    const BvhNode *nodes; StripDev st; CamBasis cam;
    int32_t tiles_x, tiles_y, gw_log2, gh_log2, groups_x, groups_y;
    char *cuts;
};
void launch_primary_cuts(const PrimaryCutArgs &a, hipStream_t s);

struct LightTraceArgs {
    SceneDev sc;
    uint32_t rng_seed, path_begin, path_count, photons_per_path;
    evplp_record *records;
    int32_t *stack_overflow;      // [lt_overflow_entries(sc)][path_count rounded up to 64]: the walk's stack beyond its LDS entries
    uint32_t overflow_stride, pad;
};
#ifndef EVPLP_LT_STACK
#define EVPLP_LT_STACK 20
#endif
constexpr int kLtLdsStack = EVPLP_LT_STACK;   // LDS entries of light tracing's walk stack (20: 5 KB per wave); the rest of the worst case lives in global memory
inline int lt_overflow_entries(const SceneDev &sc) {
    const int generic = 3 * ((sc.bvh_depth + 1) / 2) + 4;
    const int worst = sc.stack4_entries > 0 && sc.stack4_entries < generic ? sc.stack4_entries : generic;
    return worst > kLtLdsStack ? worst - kLtLdsStack : 0;
}

// Device-side counters of one pass (zeroed by the host before the launch)
struct PassCounters {
    unsigned long long rays;          // shadow / closest-hit rays actually traced
    unsigned long long nodes;         // BVH nodes visited (wave-level visits for the gather)
    unsigned long long pairs;         // splat: (photon, pixel) pairs inside the kernel radius
    unsigned long long aux;
    // EVPLP_TRAVERSAL_STATS builds only (tools/traversal_stats.py): histogram of leaf blocks tested per (wave, VPL) walk
    // ([31] = 31 or more), [32] = walks, [33] = triangle pairs tested, [34] = walks that ended with every lane occluded,
    unsigned long long hist[64];
    // gather: shadow rays / unoccluded pairs, summed by gather_reduce_kernel into 64 shards (one device-scope atomic per
    // workgroup; a single word saturates near 90 atomics per microsecond)
    unsigned long long shard_rays[64], shard_shaded[64];
    // -DEVPLP_DEBUG_NAN builds only (`make nan`): pixels whose partial sum / splat sum came out non-finite -- the reference's ASSERT under
    // DEBUG (realtimetechniques/all.cuh:10-17) as a counter (word 196 of evplp_debug_counters)
    unsigned long long nonfinite;
};
#ifndef EVPLP_DEBUG_NAN
#define EVPLP_DEBUG_NAN 0
#endif
constexpr int kCounterShards = 64;

struct GatherArgs {
    SceneDev sc; StripDev st;
    const float4 *g_pos, *g_nrm, *g_dif, *g_phg;
    const evplp_record *vpls;         // compacted usable VPL records
    const uint32_t *vpl_src_index;    // VSL only: original record index of each compacted VPL (RNG substream)
    const uint32_t *nvpl;             // device scalar written by compact_vpl_kernel
    evplp_frame_params fp;
    float pdf_mc2; int32_t pad0;      // fp.pdf_mc squared (power2 heuristic)
    float4 *out;
    float4 *partial;                  // [kVplSplit / splits_per_wave][partial_stride] per-item partial sums
    size_t partial_stride;            // W * local_rows
    PassCounters *counters;
    int32_t splits_per_wave;          // k: a wave sums k consecutive splits (a power of two <= 32) and folds them in tree order
    int32_t block_h_log2;             // tiles are enumerated in blocks of 8 x (1 << block_h_log2) tiles
    // a launch covers the groups [group_first, group_first + group_count) of every tile (group_count = 0: all 128 / k of them)
    int32_t group_first, group_count;
    // VSL gather (two kernels): lit-lane masks of every (item, VSL) of the launch, [item][k * masks_per_split], written by
    // gather_vsl_walk_kernel and read by gather_vsl_shade_kernel
    unsigned long long *vsl_masks; int32_t masks_per_split; int32_t pad1;
    // entry cuts (round 4, CutArgs below): the walks of tile (tx, ty) for VPL i start from the cut of its tile group, at
    // cuts + ((group index) * cut_vpl_stride + i) * kCutSlotBytes; null = walks start at the root
    const char *cuts; uint32_t cut_vpl_stride; int32_t cut_gw_log2, cut_gh_log2, cut_groups_x;
    // a launch covers the rows [band_first, band_first + band_rows) of tile BLOCKS only (band_rows = 0: all of them): the entry cuts of a
    // large configuration are built and consumed band by band so that their scratch stays bounded (config #5: 137 GB for all at once).
    // cut_group_row_first: the first row of tile groups that has slots in `cuts`
    int32_t band_first, band_rows, cut_group_row_first;
    int32_t item_deal;                // 1: a tile's items go to all XCDs (small launches: strips); 0: tiles are dealt to XCDs (kernels_gather.hip item_index)
    // calibration launches only (evplp_calibrate_blocks; the kernels' COST variants): the 100 MHz clock ticks every item was resident, added to
    // its local block's counter -- what evplp_group_rebalance deals the blocks by
    unsigned long long *block_cost;
};
// Entry cuts.  A frustum around ALL segments between one VPL and the pixels of a group of 2 x 2 tiles (the box of their end points
// + four planes through the VPL) descends the tree breadth-first, dropping every subtree it cannot reach, until the surviving cut
// would exceed kCutEntries; the group's (tile, VPL) packet walks then start from the cut -- two cut entries per synthetic node, in
// the BvhNode format, so the walk's node visit tests them -- instead of from the root.  Measured on the walk population of the
// bench configuration with the CPU replay (tools/bvh_eval/walk_proxy.cpp): 34.8 -> 16.0 node visits per walk, half of the walks
// start from an empty cut; 8.7 frustum steps per tile walk at ~1.5 vector instructions per step and lane.
constexpr int kCutEntries = 8;                     // cut entries per (group, VPL): 17.6 node visits per walk in the CPU replay against 16.9 with 16 entries, for a quarter fewer descent steps, half the slot and half as many synthetic visits
constexpr int kCutNodes = kCutEntries / 2;         // synthetic nodes per slot
constexpr int kCutSlotBytes = kCutNodes * 64;
struct CutArgs {
    const BvhNode *nodes;
    const float4 *tile_box;           // [tiles][2] world-space box of every tile's G-buffer positions (primary_kernel / tile_box_kernel)
    int32_t tiles_x, tiles_y;         // tiles of this context's strip
    int32_t gw_log2, gh_log2;         // tiles per group: 2 x 2 (2 x 1 when the strip's tile rows are not neighbours in the image)
    int32_t groups_x, groups_y;       // groups per row; rows of groups of THIS launch
    int32_t group_row_first, pad;     // first row of groups of this launch (slot 0 of `cuts` belongs to its first group)
    const evplp_record *vpls; const uint32_t *nvpl; uint32_t vpl_stride;     // compacted usable VPLs; slots per group in `cuts`
    char *cuts;
};
void launch_gather_cuts(const CutArgs &a, hipStream_t s);
void launch_tile_boxes(const StripDev &st, const float4 *g_pos, float4 *tile_box, int tiles_x, int tiles_y, hipStream_t s);
#ifndef EVPLP_VPL_SPLIT
#define EVPLP_VPL_SPLIT 128
#endif
// VPL i belongs to split i % kVplSplit; a pixel's sum is the balanced binary tree over the kVplSplit per-split sums, each
// split summed in increasing i.  A constant, and a fixed tree: results must not depend on the GPU count or on splits_per_wave.
constexpr int kVplSplit = EVPLP_VPL_SPLIT;
#ifndef EVPLP_GATHER_K
#define EVPLP_GATHER_K 2
#endif
// (round 4) k = 2: with the entry cuts an item of four splits is long against its set-up again (k = 2 / 4 / 8: 57.6 / 58.6 / 71.1 ms);
// 64 partial sums per pixel, 1 GB at 1024^2.  Rounds 1-3, k = 4: 32 partial sums per pixel (512 MB at 1024^2 instead of 2 GB for k = 1) at the same speed (hard scene, cfg2, one GPU:
// k = 1 113.9 ms, 2 114.3, 4 114.0 on one box); k = 16 was 38 % slower (long items: launch tail)
constexpr int kDefaultSplitsPerWave = EVPLP_GATHER_K;

struct PathTraceArgs {
    SceneDev sc; StripDev st;
    const float4 *g_pos, *g_nrm, *g_dif, *g_phg;
    float camera_pos[3]; uint32_t rng_seed;
    uint32_t max_bounces, do_accumulate;
    float4 *out;
    PassCounters *counters;
};

constexpr int kSummaryShards = 1024, kSummaryStride = 32, kSummaryFinal = kSummaryShards * kSummaryStride;
constexpr int kSummaryHeavy = kSummaryFinal + 8;     // tiles on the heavy list of this pass (splat_heavy_kernel)
// Two-level binning of the photon splat, without contended atomics.  Measured (tools/ub/atomics.hip): returning atomics on
// scattered addresses retire at 27 G/s chip-wide (64-byte requests at the memory side), atomics on ONE 128-byte line at 88 M/s
// (11 ns each, whichever words of the line), lines in parallel.  One atomic per (photon, tile) entry on per-tile cursors cost
// 100 us for the 2 M entries of config #3; one per (workgroup, coarse bucket) cost 1.2 ms on 128 bucket cursors packed into four
// lines and 115 us on cursors with a line each (8 k adds per line: a 90 us chain).  So:
//   splat_bin     : a workgroup (kBinChunks x 256 consecutive records) ranks its entries per BUCKET (a rectangle of
//                   2^bucket_w_log2 x 2^bucket_h_log2 = 128 tiles) in LDS and writes them, sorted by bucket, into a segment of its
//                   own + a table of bucket offsets -- no global atomic.
//   splat_scatter : workgroup (slice, bucket): thread t takes the run of bucket entries of bin-workgroup slice * 256 + t, the
//                   workgroup ranks them per tile in LDS and reserves the tiles' bin slots with ONE atomic per (workgroup,
//                   tile): a tile cursor sees one add per slice.
//   splat_big     : photons whose rectangle is larger than 2x2 tiles (huge radii) place their entries one atomic each.
#ifndef EVPLP_BIN_CHUNKS
#define EVPLP_BIN_CHUNKS 1
#endif
// 256-record chunks per workgroup of splat_bin_kernel.  Measured (bin + scatter + big kernels, us, configs #3 / #4):
// 1 chunk 155 / 105, 2 chunks 186 / 134, 4 chunks 182 / 128: short workgroups, four per CU, hide latency best.
constexpr int kBinChunks = EVPLP_BIN_CHUNKS;
constexpr int kBinGroup = 256 * kBinChunks;   // records per workgroup
constexpr int kSegCap = 4 * kBinGroup;        // entries of a workgroup's segment (a photon of the LDS path has at most 2x2)
#ifndef EVPLP_SCATTER_G
#define EVPLP_SCATTER_G 2
#endif
constexpr int kScatterG = EVPLP_SCATTER_G;    // bin-groups per thread of splat_scatter_kernel (a slice = 256 * kScatterG groups)
constexpr int kMaxBuckets = 1024;
constexpr int kBucketTilesLog2 = 7;   // tiles per bucket: 16 x 8 ...
constexpr int kMaxBucketTilesLog2 = 9; // ... up to 32 x 16 for images of more than 131 072 tiles (kMaxBuckets buckets at most)
struct SplatArgs {
    StripDev st; CamBasis cam;
    evplp_frame_params fp;
    const float4 *g_pos, *g_nrm, *g_dif, *g_phg;
    const evplp_record *records; uint32_t num_records;
    float4 *out;
    // binning workspace: every 8x8-px tile owns a fixed slab of bin_stride item slots
    uint32_t *tile_pairs;     // [ntiles] (photon, pixel) pairs accepted in the tile (statistics; summed by the host on demand)
    float4 *tile_box;         // [ntiles][2] world-space box (lo, hi) of the tile's G-buffer positions
    uint32_t *tile_cursor;    // [ntiles] entries the photons wanted to put into the tile's bin (may exceed bin_stride: overflow)
    uint32_t *bin_items;      // [ntiles][bin_stride] compact photon ids
    uint32_t bin_stride;
    uint32_t *seg;            // [num_bin_groups][kSegCap] entries: record - group base (10 bits) | tile within the bucket (<= 9 bits) << 10, sorted by bucket
    uint16_t *seg_off;        // [num_bin_groups][num_buckets + 1] first entry of every bucket in the group's segment
    uint32_t *big_list;       // [num_bin_groups][kBinGroup] records whose rectangle is larger than 2x2 tiles
    uint32_t *big_count;      // [num_bin_groups]
    int32_t num_bin_groups, bucket_w_log2, bucket_h_log2, buckets_x, num_buckets;
    uint32_t *bin_items_tmp;  // [ntiles][bin_stride] (deterministic mode: unsorted fill target)
    float4 *compact;          // [num_records * kCompactF4] per-photon pre-shaded data
    uint32_t *overflow;       // device flag: the slots the fullest bin wanted, when that is more than bin_stride (tiles kernel then does nothing; the host re-runs) (fill / tiles then do nothing; the host re-runs)
    uint32_t *summary;        // device: kSummaryShards x {entries, fullest bin} on separate 128-byte lines (scatter / big kernels, per workgroup),
                              // folded into [kSummaryFinal + 0] total bin entries, [+1] entries of the fullest bin by the tile kernel
    int32_t tiles_x, tiles_y; int32_t deterministic; int32_t boxes_valid;   // boxes_valid: tile_box was written by the primary pass
    PassCounters *counters;
    // EVPLP_FOOTPRINT_PROXY (ProxyDev below): the slabs of the proxy mesh in units of the radius, and the per-tile fragment counts
    const float4 *proxy_slabs; const float *proxy_hm; int32_t proxy_count; float proxy_rin, proxy_rout;
    uint32_t *tile_frags;     // [ntiles] proxy fragments accepted in the tile (statistics, like tile_pairs)
    // MIXED tile launches (kernels_splat.hip splat_heavy_kernel): tiles with >= heavy_threshold bin entries, at most heavy_cap of them; null = the pure variants
    uint32_t *heavy_list; uint8_t *tile_flags; uint32_t heavy_cap, heavy_threshold;
};
// The proxy mesh of the reference's photon splat as the tile kernel wants it.  A convex mesh is the intersection of its face planes
// n . x <= h; two faces with opposite normals form a SLAB -h- <= n . x <= h+, and a ray's parameters at the two planes of a slab
// come from one pair of dot products and one reciprocal.  slab i = (n, h+) and hm[i] = h- (kProxyOpen for a face without an opposite
// one).  rin = the smallest h (radius of the largest sphere around the origin inside the mesh), rout = the largest |vertex|.
constexpr int kMaxProxySlabs = EVPLP_MAX_PROXY_PLANES;
constexpr float kProxyOpen = 1.0e15f;
constexpr int kSplatTile = 8;        // pixels per tile edge (one wave per tile)
constexpr int kCompactF4 = 4;        // float4 per compact photon

// dynamic LDS of the kernels that walk the BVH one ray per lane: an [entry][lane] stack as deep as the tree
inline size_t lane_stack_bytes(const SceneDev &sc) { return (size_t)(sc.bvh_depth + 2) * 64 * sizeof(int32_t); }
// (four-wide walk: every second level of the binary tree, up to three pushes per level)
// (four-wide walk: up to three pushes per visit.  3 x levels / 2 is the bound for ANY tree of that depth -- 13 KB per wave for a 31-level
// tree, three waves per SIMD; the bound for THE tree that was built, computed once by evplp_build_accel, is about half of it)
inline size_t lane_stack_bytes4(const SceneDev &sc) {
    const int generic = 3 * ((sc.bvh_depth + 1) / 2) + 4;
    const int entries = sc.stack4_entries > 0 ? (sc.stack4_entries < generic ? sc.stack4_entries : generic) : generic;
    return (size_t)entries * 64 * sizeof(int32_t);
}

void launch_primary(const PrimaryArgs &a, hipStream_t s);
void launch_light_trace(const LightTraceArgs &a, hipStream_t s);
void launch_compact_vpl(const evplp_record *records, uint32_t nrec, evplp_record *out, uint32_t *src_index,
                        uint32_t *count_out, hipStream_t s);
// VPL / VSL gather = the items, then one reduce
void launch_gather_vpl_items(const GatherArgs &a, hipStream_t s);
void launch_gather_vsl(const GatherArgs &a, hipStream_t s);        // walks, then estimators, of the launch's group range
int gather_launch_tiles(const GatherArgs &a);                      // tiles a gather launch enumerates (whole blocks)
void launch_gather_reduce(const GatherArgs &a, int stencil_test, hipStream_t s);
void launch_gather_lvc(const GatherArgs &a, const evplp_record *records, hipStream_t s);
void launch_path_trace(const PathTraceArgs &a, hipStream_t s);
// binning (tile depth ranges, compact photons + bin fill, summary); then the per-tile accumulation
void launch_splat_bin(const SplatArgs &a, hipStream_t s);
void launch_splat_tiles(const SplatArgs &a, bool split_tiles, hipStream_t s, hipEvent_t dominant_begin, hipEvent_t dominant_end);
void launch_resolve(const StripDev &st, const float4 *vpl, const float4 *pm, const float4 *light,
                    float vs, float ps, float ls, int mask_emitter, int gamma, float *out_rgb, hipStream_t s);
struct BandTable { int32_t first[65]; };      // first[r] = first image row of rank r's band, first[n] = H
void launch_assemble_strips(const StripDev &st, int nranks, const BandTable *bands, const uint32_t *owner, int chunk_rows, const float *gathered, float *frame, hipStream_t s);
void launch_fill_zero(void *p, size_t bytes, hipStream_t s);

} // namespace evplp
