// Internal data layout of libevplp_hip.so (host + device).  See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/evplp.h"

namespace evplp {

// Flattened binary BVH node, 64 B = one s_load_dwordx16 / one cache line.  A node stores the boxes
// of BOTH children, interleaved component by component so that the two slab tests run as one
// stream of packed-fp32 instructions (half 0 = child 0, half 1 = child 1).  Boxes are stored as
// centre + half-size: with A = ctr * (1/d) - o/d and B = hal * |1/d| the slab entry / exit parameters
// are A - B and A + B, i.e. three v_pk_fma_f32 per axis for BOTH children and no per-lane min/max to
// sort the near and far plane.  child >= 0: inner node index; child < 0: leaf, id = ~child,
// leaf block = id >> 2, triangle count = (id & 3) + 1.  An absent child has a negative half-size and
// child = kNoChild.
struct BvhNode {
    float ctr[3][2];  // box centre   [axis][child]
    float hal[3][2];  // box half-size [axis][child] (>= 0; an absent child has hal = -big: never hit)
    int32_t c0, c1;
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");
// Four-wide node for the per-lane walks (incoherent rays: light tracing, path tracer, light-path windows): the two children of
// a binary node replaced by its (up to four) grandchildren -- half the dependent fetches per ray.  node4[i] is derived from
// binary node i on the device (bvh_gpu.hip build_nodes4); the walks only ever visit every second level of it.
struct BvhNode4 {
    float lo[3][4], hi[3][4];   // [axis][child]; an absent child has lo > hi
    int32_t child[4];           // binary node index (>= 0), ~leaf reference, or kNoChild
    int32_t pad[4];
};
static_assert(sizeof(BvhNode4) == 128, "BvhNode4 must be 128 bytes");
constexpr int32_t kNoChild = INT32_MIN;
constexpr int kMaxLeafTris = 4;
constexpr int kMaxDepth = 64;

// Leaf block, 192 B = three s_load_dwordx16: up to four triangles as TWO PAIRS; every operand of
// optix::intersect_triangle_branchless (p0, e0 = p1-p0, e1 = p0-p2, n = cross(e1, e0)) is stored as
// {triangle A, triangle B} so that a pair is tested with packed-fp32 instructions.  Unused slots
// are zero (den = 0 -> the test is false).
struct TriPair {
    float p0[3][2], e0[3][2], e1[3][2], n[3][2];   // [component][A|B]
};
static_assert(sizeof(TriPair) == 96, "TriPair must be 96 bytes");
struct LeafBlock { TriPair pair[2]; };
static_assert(sizeof(LeafBlock) == 192, "LeafBlock must be 192 bytes");
// The same operands once more, one triangle per 48 contiguous bytes (slot = 4 * leaf block + k): the per-lane walks
// fetch a triangle with three 16-byte loads instead of twelve strided words of the pair layout.
struct TriFlat { float p0[3], e0[3], e1[3], n[3]; };
static_assert(sizeof(TriFlat) == 48, "TriFlat must be 48 bytes");

// Shading attributes per ORIGINAL triangle.
struct TriAttr {
    float v[9];      // p0,p1,p2 as uploaded (G-buffer position / normal use the originals)
    float uv[6];
    int32_t material;
};
static_assert(sizeof(TriAttr) == 64, "TriAttr must be 64 bytes");

struct Material {   // 64 B
    float kd[3]; float ns;
    float ks[3]; int32_t tex_kd;
    float light[4];           // mLightIntensity (I*pi, w); zeros for non-emitters
    int32_t tex_ks, tex_ns, pad0, pad1;
};
static_assert(sizeof(Material) == 64, "Material must be 64 bytes");

struct TexDesc { int32_t w, h; uint32_t offset; uint32_t pad; };  // offset in float4 units into the pool

// Camera basis (glm::lookAt RH + glm::perspective, rt/rtcommon.h:586-591)
struct CamBasis {
    float eye[3]; float tan_half;
    float s[3];   float aspect;
    float u[3];   float pad0;
    float f[3];   float pad1;
};

// Everything a traversal kernel needs, passed by value as a kernel argument (all pointers
// wave-uniform => scalar loads).
struct SceneDev {
    const BvhNode *nodes;
    const BvhNode4 *nodes4;      // per-lane walks
    const LeafBlock *leaves;     // one block per leaf
    const TriFlat *tri_flat;     // 4 slots per leaf (per-lane walks)
    const int32_t *tri_index;    // 4 slots per leaf: original triangle index or -1
    const TriAttr *attrs;        // original order
    const Material *materials;
    const TexDesc *textures;
    const float4 *tex_pool;
    const float *light_cdf;      // normalised area CDF of the light mesh
    int32_t ntris;               // triangles in the BVH (degenerate ones dropped)
    int32_t bvh_depth;           // deepest leaf: sizes the per-lane traversal stacks (dynamic LDS)
    int32_t stack4_entries;      // worst-case stack of the four-wide per-lane walk over THIS tree (computed at build time; see evplp_build_accel)
    int32_t pad_sc;
    int32_t light_first, light_count; // ORIGINAL triangle index range of the light mesh
    float light_area;
    float light_intensity[4];    // (I*pi, w)
    float light_unscaled[4];     // (I, w)
    float light_lo[3], light_hi[3];   // bounds of the light mesh, padded (primary_kernel skips its walk for tiles that cannot see it)
};

// Strip geometry shared by all per-pixel kernels.
#ifndef EVPLP_BLOCK_TABLE
#define EVPLP_BLOCK_TABLE 1      // developer knob (A/B builds): 0 compiles the owned-block table out of the kernels (round-robin deal only)
#endif
struct StripDev {
    int32_t W, H;
    int32_t strip_rank, strip_count, strip_rows;
    int32_t local_rows;
    // band mode (evplp_config.band_rows > 0; strip_count = 1): the context owns the image rows [band_first, band_first + band_rows), stored
    // from local row 0; local rows beyond the band (padding up to the capacity) lie outside the image
    int32_t band_first, band_rows;
    // DEALT blocks (evplp_set_blocks; strip_count > 1): the owned-block table replaces "block b belongs to rank b % strip_count".
    //   blocks[l], l < cap_blocks = local_rows / strip_rows: the image block stored in local block l; a local block that holds nothing carries an
    //     index >= the image's block count, i.e. rows >= H: outside the image, as the padding rows of the round-robin deal are;
    //   blocks[cap_blocks + b], b < image blocks: the local block that holds image block b, or -1 (another rank's).
    // nullptr = the round-robin deal.  Two copies of the same table: `blocks` in device memory, `blocks_host` for the host side of global_row.
    const int32_t *blocks, *blocks_host;
    int32_t cap_blocks, pad_st;
    __host__ __device__ inline int32_t global_block(int32_t local_blk) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int32_t *t = EVPLP_BLOCK_TABLE ? blocks : nullptr;
#else
        const int32_t *t = blocks_host;
#endif
        return t ? t[local_blk] : local_blk * strip_count + strip_rank;
    }
    __host__ __device__ inline int32_t global_row(int32_t local) const {
        if (band_rows > 0) return local < band_rows ? band_first + local : H + local;
        int32_t blk = local / strip_rows;
        return global_block(blk) * strip_rows + (local - blk * strip_rows);
    }
    // the local block that holds image block b, or -1 (strip_count > 1 only)
    __host__ __device__ inline int32_t local_block(int32_t b) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int32_t *t = EVPLP_BLOCK_TABLE ? blocks : nullptr;
#else
        const int32_t *t = blocks_host;
#endif
        if (t) return t[cap_blocks + b];
        return b % strip_count == strip_rank ? b / strip_count : -1;
    }
};

// Host-side acceleration structure build result
struct BvhBuild {
    BvhNode *nodes = nullptr; int32_t nnodes = 0;
    LeafBlock *leaves = nullptr;  int32_t *tri_index = nullptr; int32_t ntris = 0;
    TriFlat *tri_flat = nullptr;
    int32_t nleaves = 0, depth = 0;
    float build_ms = 0.f;
};
// Device-side build (bvh_gpu.hip): the four arrays are device allocations handed to the caller.
struct BvhDeviceBuild {
    BvhNode *nodes = nullptr; LeafBlock *leaves = nullptr; TriFlat *tri_flat = nullptr; int32_t *tri_index = nullptr;
    int32_t nnodes = 0, nleaves = 0, ntris = 0, depth = 0;
    float build_ms = 0.f;
};
// verts: 9 floats per original triangle.  Degenerate triangles (rt/triangleintersect.cu:62-81
// meshBound invalidates them) are dropped.  Returns 0 on success.
int build_bvh(const float *verts, int32_t ntri, int builder, BvhBuild *out);
// box padding of both builders, as a fraction of the scene diagonal (bvh_build.cpp explains the value)
float bvh_pad_scale();
void free_bvh(BvhBuild *b);

} // namespace evplp
