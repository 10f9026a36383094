// Kernel argument blocks and host-side launchers (defined in the .hip files).
#pragma once
#include "evplp_types.h"

namespace evplp {

struct PrimaryArgs {
    SceneDev sc; StripDev st; CamBasis cam;
    float jitter[2]; int32_t clear_light; int32_t pad;
    float4 *g_pos, *g_nrm, *g_dif, *g_phg, *g_light;
};

struct LightTraceArgs {
    SceneDev sc;
    uint32_t rng_seed, path_begin, path_count, photons_per_path;
    evplp_record *records;
};

// Device-side counters of one pass (zeroed by the host before the launch)
struct PassCounters {
    unsigned long long rays;          // shadow / closest-hit rays actually traced
    unsigned long long nodes;         // BVH nodes visited (wave-level visits for the gather)
    unsigned long long pairs;         // splat: (photon, pixel) pairs inside the kernel radius
    unsigned long long aux;
    // EVPLP_TRAVERSAL_STATS builds only (tools/traversal_stats.py): histogram of leaf blocks tested per (wave, VPL) walk
    // ([31] = 31 or more), [32] = walks, [33] = triangle pairs tested, [34] = walks that ended with every lane occluded,
    // beam_visibility_kernel: [35] = (super-tile, VPL) beams walked, [36] = node visits, [37] = leaf blocks met by some shaft,
    // [38] = (tile, leaf) exact tests run with pixel lanes, [39] = triangle pairs passed to the exact predicate, [40] = pairs
    // rejected by the plane-distance pre-test, [41] = tiles that ended fully occluded, [42] = (tile, VPL) pairs culled by the cosine bounds
    unsigned long long hist[64];
    // gather: shadow rays / unoccluded pairs, summed by gather_reduce_kernel into 64 shards (one device-scope atomic per
    // workgroup; a single word saturates near 90 atomics per microsecond)
    unsigned long long shard_rays[64], shard_shaded[64];
};
constexpr int kCounterShards = 64;

// The beam pass bounds the pixels of a tile by up to kSubs position boxes ("sub-tiles").  A tile's lit pixels (non-zero normal:
// a zero normal makes the receiver cosine of lighttracing.cu:284 exactly 0, such a pixel never traces a shadow ray) are sorted by
// their distance to the camera and cut into kSubs groups -- at depth discontinuities where there are any, at the middle of the
// depth range otherwise -- so that a tile seen at a grazing angle (metres deep, centimetres wide) or holding a silhouette edge is
// still covered by compact boxes.  Written once per frame by tile_clusters_kernel.
constexpr int kSubs = 4;
struct SubBound {
    float lo[3]; uint32_t flags;      // AABB of the member pixels' positions
    float hi[3]; uint32_t mem_lo;     // mem_lo | mem_hi << 32: bit l = pixel (l & 7, l >> 3) of the tile belongs to this sub-tile
    float n[3]; uint32_t mem_hi;      // n: the common normal when every member has the same one (kTileFlat)
};
static_assert(sizeof(SubBound) == 48, "SubBound must be 48 bytes");
constexpr uint32_t kTileLit = 1u, kTileFlat = 2u;
// kTileFat (set on every sub-tile of the tile): some sub-tile still spans far more space than neighbouring surface points would;
// its shaft would meet hundreds of leaves, so the tile is left out of the beam pass and its items walk the tree with their 64
// exact segments instead.
constexpr uint32_t kTileFat = 4u;

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
    // tile enumeration: super-tiles of (1 << super_w_log2) x (64 >> super_w_log2) tiles, tile id = super-tile * 64 + lane
    SubBound *tile_bounds;            // [nsx * nsy * 64][kSubs]
    unsigned long long *vis;          // [max usable VPLs][band_supers * 64] occlusion mask of every (tile, VPL): bit l = the shadow ray of
                                      // pixel l is blocked (or the pixel cannot be lit); null: every gather item walks the tree itself
    int32_t super_w_log2, nsx, nsy;
    int32_t band_first_super, band_supers;   // the super-tiles this launch covers
    int32_t splits_per_wave;          // k: a wave sums k consecutive splits (a power of two <= 32) and folds them in tree order
    uint32_t max_vpls;                // allocated VPL slots (grid bound of beam_visibility_kernel)
    uint32_t *dbg;                    // EVPLP_TRAVERSAL_STATS builds: [tile ids] + [max_vpls] exact-test counts of the beam pass; else null
    float fat_ratio;                  // a sub-tile is fat when its box extent exceeds fat_ratio x 8 x the smallest spacing of adjacent lit pixels
};
#ifndef EVPLP_VPL_SPLIT
#define EVPLP_VPL_SPLIT 128
#endif
// VPL i belongs to split i % kVplSplit; a pixel's sum is the balanced binary tree over the kVplSplit per-split sums, each
// split summed in increasing i.  A constant, and a fixed tree: results must not depend on the GPU count or on splits_per_wave.
constexpr int kVplSplit = EVPLP_VPL_SPLIT;

struct PathTraceArgs {
    SceneDev sc; StripDev st;
    const float4 *g_pos, *g_nrm, *g_dif, *g_phg;
    float camera_pos[3]; uint32_t rng_seed;
    uint32_t max_bounces, do_accumulate;
    float4 *out;
    PassCounters *counters;
};

struct SplatArgs {
    StripDev st; CamBasis cam;
    evplp_frame_params fp;
    const float4 *g_pos, *g_nrm, *g_dif, *g_phg;
    const evplp_record *records; uint32_t num_records;
    float4 *out;
    // binning workspace
    uint32_t *tile_count;     // [ntiles + 1]
    uint32_t *tile_pairs;     // [ntiles] (photon, pixel) pairs accepted in the tile (statistics; summed by the host on demand)
    float2 *tile_z;           // [ntiles] view-depth range (min, max) of the tile's G-buffer positions
    uint32_t *tile_offset;    // [ntiles + 1] exclusive scan
    uint32_t *tile_cursor;    // [ntiles]
    uint32_t *bin_items;      // [bin_capacity] compact photon ids
    uint32_t bin_capacity;
    uint32_t *bin_items_tmp;  // [bin_capacity] (deterministic mode: unsorted fill target)
    float4 *compact;          // [num_records * kCompactF4] per-photon pre-shaded data
    uint4 *rect;              // [num_records] tile rectangle (x0 | x1<<16, y0 | y1<<16; x0 > x1 = none) + 64-bit mask of its tiles that survive the depth cull
    uint32_t *overflow;       // device flag: bins did not fit
    uint32_t *summary;        // device: [0] total bin entries, [1] entries of the fullest bin
    int32_t tiles_x, tiles_y; int32_t deterministic; int32_t pad;
    PassCounters *counters;
};
constexpr int kSplatTile = 8;        // pixels per tile edge (one wave per tile)
constexpr int kCompactF4 = 4;        // float4 per compact photon

// dynamic LDS of the kernels that walk the BVH one ray per lane: an [entry][lane] stack as deep as the tree
inline size_t lane_stack_bytes(const SceneDev &sc) { return (size_t)(sc.bvh_depth + 2) * 64 * sizeof(int32_t); }

void launch_primary(const PrimaryArgs &a, hipStream_t s);
void launch_light_trace(const LightTraceArgs &a, hipStream_t s);
void launch_compact_vpl(const evplp_record *records, uint32_t nrec, evplp_record *out, uint32_t *src_index,
                        uint32_t *count_out, hipStream_t s);
// VPL gather = tile bounds, then per band of super-tiles: beam visibility (when a.vis) + the gather items, then one reduce
void launch_tile_bounds(const GatherArgs &a, hipStream_t s);
void launch_beam_visibility(const GatherArgs &a, hipStream_t s);
void launch_gather_vpl_items(const GatherArgs &a, hipStream_t s);
void launch_gather_vsl(const GatherArgs &a, hipStream_t s);
void launch_gather_reduce(const GatherArgs &a, int stencil_test, hipStream_t s);
void launch_gather_lvc(const GatherArgs &a, const evplp_record *records, hipStream_t s);
void launch_path_trace(const PathTraceArgs &a, hipStream_t s);
void launch_splat_count(const SplatArgs &a, hipStream_t s);
void launch_splat_tiles(const SplatArgs &a, bool split_tiles, hipStream_t s, hipEvent_t dominant_begin, hipEvent_t dominant_end);
void launch_resolve(const StripDev &st, const float4 *vpl, const float4 *pm, const float4 *light,
                    float vs, float ps, float ls, int mask_emitter, int gamma, float *out_rgb, hipStream_t s);
void launch_fill_zero(void *p, size_t bytes, hipStream_t s);

} // namespace evplp
