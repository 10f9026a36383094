// LBVH built on the device: the same flattened node / leaf-block format as the host builders (bvh_build.cpp), for scenes
// that change between frames or are too large to wait for a host build.
// Replaces OptiX's closed-source "Trbvh" acceleration build (rt/rtcomphoton/rtcomphoton.h:705-707) and the meshBound
// program (rt/triangleintersect.cu:62-81).
//
//   1. tri_setup    : per triangle its box, box centre and validity (meshBound: area > 0 and finite); scene bounds.
//   2. morton       : 63-bit Morton code of the box centre (21 bits per axis, same quantisation as the host LBVH);
//                     invalid triangles get the largest key.  hipCUB radix sort of (code, triangle) pairs (stable: equal
//                     codes keep triangle order, as std::sort of the pairs does on the host).
//   3. hierarchy    : Karras 2012 -- every internal node of the binary radix tree finds its range and split on its own
//                     (keys made unique by their position).
//   4. refit        : leaf boxes, then internal boxes bottom-up (the second thread to arrive at a node owns it).
//   5. collapse     : subtrees of <= 4 triangles become leaf blocks (the walks test triangles two at a time, 4 per block);
//                     nodes with more than 4 triangles are kept and renumbered by a prefix sum.
//   6. emit         : kept nodes with both child boxes (padded, centre / half-size form), leaf blocks with the
//                     precomputed operands of the exact triangle test.
#include "evplp_types.h"
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace evplp {
namespace {

struct Bx { float lo[3], hi[3]; };

// order-preserving float <-> uint (for atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t f2o(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float o2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; std::memcpy(&f, &u, 4); return f;
#endif
}

// bounds[0..2] centroid lo, [3..5] centroid hi, [6..8] scene lo, [9..11] scene hi (ordered uints); bounds[12] = valid count
__global__ __launch_bounds__(256) void tri_setup_kernel(const float *verts, int ntri, Bx *tbox, uint32_t *bounds, uint8_t *valid) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    float clo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, chi[3] = { -3.0e38f, -3.0e38f, -3.0e38f }, slo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, shi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    uint32_t ok = 0u;
    if (i < ntri) {
        const float *v = verts + 9 * (size_t)i;
        // rt/triangleintersect.cu:62-81 meshBound: area = |cross(v1-v0, v2-v0)| must be > 0 and finite
        const float a[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] }, b[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
        const float c[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
        const float area = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        Bx t;
        for (int k = 0; k < 3; k++) { t.lo[k] = fminf(fminf(v[k], v[3 + k]), v[6 + k]); t.hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), v[6 + k]); }
        tbox[i] = t;
        ok = (area > 0.0f && !isinf(area)) ? 1u : 0u;
        valid[i] = (uint8_t)ok;
        if (ok) for (int k = 0; k < 3; k++) { const float ce = 0.5f * (t.lo[k] + t.hi[k]); clo[k] = chi[k] = ce; slo[k] = t.lo[k]; shi[k] = t.hi[k]; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; k++) {
            clo[k] = fminf(clo[k], __shfl_xor(clo[k], off)); chi[k] = fmaxf(chi[k], __shfl_xor(chi[k], off));
            slo[k] = fminf(slo[k], __shfl_xor(slo[k], off)); shi[k] = fmaxf(shi[k], __shfl_xor(shi[k], off));
        }
        ok += __shfl_xor(ok, off);
    }
    if ((threadIdx.x & 63) == 0 && ok) {
        for (int k = 0; k < 3; k++) {
            atomicMin(&bounds[k], f2o(clo[k])); atomicMax(&bounds[3 + k], f2o(chi[k]));
            atomicMin(&bounds[6 + k], f2o(slo[k])); atomicMax(&bounds[9 + k], f2o(shi[k]));
        }
        atomicAdd(&bounds[12], ok);
    }
}

__device__ __forceinline__ uint64_t expand21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
__global__ __launch_bounds__(256) void morton_kernel(const Bx *tbox, const uint8_t *valid, int ntri, const uint32_t *bounds, uint64_t *keys, int32_t *ids) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntri) return;
    ids[i] = i;
    if (!valid[i]) { keys[i] = ~0ull; return; }
    uint64_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = o2f(bounds[k]), ext = fmaxf(o2f(bounds[3 + k]) - lo, 1e-30f);
        const float c = 0.5f * (tbox[i].lo[k] + tbox[i].hi[k]);
        double t = ((double)c - (double)lo) / (double)ext;
        t = fmin(fmax(t, 0.0), 1.0);
        q[k] = (uint64_t)fmin(t * 2097152.0, 2097151.0);
    }
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
}

// Karras 2012.  Internal node i < n - 1; a child reference is an internal index, or ~leaf for sorted position `leaf`.
__device__ __forceinline__ int delta(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}
struct Topo { int32_t left, right, first, last; };
__global__ __launch_bounds__(256) void hierarchy_kernel(const uint64_t *keys, int n, Topo *topo, int32_t *parent_int, int32_t *parent_leaf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do { t = (t + 1) / 2; if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t; } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    Topo tp; tp.first = first; tp.last = last;
    if (first == gamma) { tp.left = ~gamma; parent_leaf[gamma] = i; } else { tp.left = gamma; parent_int[gamma] = i; }
    if (last == gamma + 1) { tp.right = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { tp.right = gamma + 1; parent_int[gamma + 1] = i; }
    topo[i] = tp;
    if (i == 0) parent_int[0] = -1;
}

// a box another workgroup (possibly on another XCD) wrote before it bumped the visit counter: loads that bypass this CU's L1
__device__ __forceinline__ Bx load_box_agent(const Bx *p) {
    Bx b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = __hip_atomic_load(&p->lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.hi[k] = __hip_atomic_load(&p->hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}
// bottom-up boxes; also the height of the tree in KEPT nodes (more than 4 triangles), carried up with the boxes: the second
// arrival at a node owns it and takes max(children) + 1.  (Round 2 let every leaf thread walk its whole ancestor chain to count:
// O(n depth), quadratic on the long chains that clustered or duplicate Morton codes produce.)
__global__ __launch_bounds__(256) void refit_kernel(const Bx *tbox, const int32_t *ids, int n, const Topo *topo, const int32_t *parent_int, const int32_t *parent_leaf,
                                                    Bx *lbox, Bx *ibox, uint32_t *visits, uint32_t *height, uint32_t *max_depth) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    Bx b = tbox[ids[j]];
    lbox[j] = b;
    if (n == 1) return;
    __threadfence();
    int cur = parent_leaf[j];
    while (cur >= 0) {
        if (atomicAdd(&visits[cur], 1u) == 0u) break;                    // the first arrival leaves; its subtree is complete and visible
        __threadfence();
        const Topo tp = topo[cur];
        const Bx l = load_box_agent(tp.left < 0 ? &lbox[~tp.left] : &ibox[tp.left]), r = load_box_agent(tp.right < 0 ? &lbox[~tp.right] : &ibox[tp.right]);
        const uint32_t hl = tp.left < 0 ? 0u : __hip_atomic_load(&height[tp.left], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t hr = tp.right < 0 ? 0u : __hip_atomic_load(&height[tp.right], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t h = max(hl, hr) + (tp.last - tp.first + 1 > kMaxLeafTris ? 1u : 0u);
        Bx u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(l.lo[k], r.lo[k]); u.hi[k] = fmaxf(l.hi[k], r.hi[k]); }
        ibox[cur] = u;
        height[cur] = h;
        __threadfence();
        const int up = parent_int[cur];
        if (up < 0) *max_depth = h;                                       // the root: one writer
        cur = up;
    }
}

// kept[i] = node i has more than 4 triangles; head[p] = triangles of the leaf block that starts at sorted position p (0: none)
__global__ __launch_bounds__(256) void collapse_kernel(const Topo *topo, int n, uint32_t *kept, uint32_t *head) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const Topo tp = topo[i];
    const bool k = tp.last - tp.first + 1 > kMaxLeafTris;
    kept[i] = k ? 1u : 0u;
    if (!k) { if (i == 0) head[0] = (uint32_t)n; return; }            // (the whole tree is one block)
    const int32_t ch[2] = { tp.left, tp.right };
    for (int s = 0; s < 2; s++) {
        if (ch[s] < 0) head[~ch[s]] = 1u;
        else { const Topo c = topo[ch[s]]; const int cnt = c.last - c.first + 1; if (cnt <= kMaxLeafTris) head[c.first] = (uint32_t)cnt; }
    }
}
__global__ __launch_bounds__(256) void flag_kernel(const uint32_t *head, int n, uint32_t *flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = head[i] ? 1u : 0u;
}

__device__ __forceinline__ void set_box(BvhNode &f, int child, const Bx &b, float pad) {
#pragma clang fp contract(off)
    for (int k = 0; k < 3; k++) {
        const float lo = b.lo[k] - pad, hi = b.hi[k] + pad;
        const float c = 0.5f * (lo + hi);
        float h = fmaxf(hi - c, c - lo);
        h = h + fabsf(h) * 1e-6f + 1e-30f;
        f.ctr[k][child] = c; f.hal[k][child] = h;
    }
}
__global__ __launch_bounds__(256) void emit_nodes_kernel(const Topo *topo, int n, const uint32_t *kept, const uint32_t *new_id, const uint32_t *head, const uint32_t *block_id,
                                                         const Bx *lbox, const Bx *ibox, const uint32_t *bounds, float pad_scale, BvhNode *nodes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float dx = o2f(bounds[9]) - o2f(bounds[6]), dy = o2f(bounds[10]) - o2f(bounds[7]), dz = o2f(bounds[11]) - o2f(bounds[8]);
    float coord = 0.f;
    for (int k = 0; k < 3; k++) coord = fmaxf(coord, fmaxf(fabsf(o2f(bounds[6 + k])), fabsf(o2f(bounds[9 + k]))));
    const float pad = pad_scale * fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), coord) + 1e-30f;
    if (n - 1 < 1 || !kept[0]) {
        // at most 4 triangles: a root with the block as its only child
        if (i == 0) {
            BvhNode f; for (int k = 0; k < 3; k++) { f.ctr[k][1] = 0.f; f.hal[k][1] = -3.0e38f; }
            set_box(f, 0, n == 1 ? lbox[0] : ibox[0], pad);
            f.c0 = ~((0 << 2) | (n - 1)); f.c1 = kNoChild; f.pad[0] = f.pad[1] = 0;
            nodes[0] = f;
        }
        return;
    }
    if (i >= n - 1 || !kept[i]) return;
    const Topo tp = topo[i];
    BvhNode f; f.pad[0] = f.pad[1] = 0;
    const int32_t ch[2] = { tp.left, tp.right };
    int32_t ref[2];
    for (int s = 0; s < 2; s++) {
        if (ch[s] < 0) { set_box(f, s, lbox[~ch[s]], pad); ref[s] = ~((int32_t)(block_id[~ch[s]] << 2) | 0); }
        else {
            set_box(f, s, ibox[ch[s]], pad);
            if (kept[ch[s]]) ref[s] = (int32_t)new_id[ch[s]];
            else { const int first = topo[ch[s]].first; ref[s] = ~((int32_t)(block_id[first] << 2) | (int32_t)(head[first] - 1u)); }
        }
    }
    f.c0 = ref[0]; f.c1 = ref[1];
    nodes[new_id[i]] = f;
}
__global__ __launch_bounds__(256) void emit_leaves_kernel(const float *verts, const int32_t *ids, int n, const uint32_t *head, const uint32_t *block_id,
                                                          LeafBlock *leaves, TriFlat *tri_flat, int32_t *tri_index) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n || !head[p]) return;
    const uint32_t blk = block_id[p], cnt = head[p];
    for (uint32_t k = 0; k < (uint32_t)kMaxLeafTris; k++) {
        const size_t slot = 4 * (size_t)blk + k;
        if (k >= cnt) { tri_index[slot] = -1; continue; }
        const int32_t tri = ids[p + k];
        tri_index[slot] = tri;
        const float *v = verts + 9 * (size_t)tri;
        // same operation order as the host builder and the oracle's tri_test: e0 = p1-p0, e1 = p0-p2, n = cross(e1, e0)
        float e0[3], e1[3];
        for (int c = 0; c < 3; c++) { e0[c] = v[3 + c] - v[c]; e1[c] = v[c] - v[6 + c]; }
        const float nn[3] = { e1[1] * e0[2] - e1[2] * e0[1], e1[2] * e0[0] - e1[0] * e0[2], e1[0] * e0[1] - e1[1] * e0[0] };
        TriPair &tp = leaves[blk].pair[(k >> 1) & 1]; const int h = (int)(k & 1);
        TriFlat &tf = tri_flat[slot];
        for (int c = 0; c < 3; c++) {
            tp.p0[c][h] = v[c]; tp.e0[c][h] = e0[c]; tp.e1[c][h] = e1[c]; tp.n[c][h] = nn[c];
            tf.p0[c] = v[c]; tf.e0[c] = e0[c]; tf.e1[c] = e1[c]; tf.n[c] = nn[c];
        }
    }
}

// node4[i] from binary node i: child s of i, if it is an inner node, is replaced by ITS two children (their boxes are stored in
// that child's own record); a leaf or absent child stays.  lo / hi are computed exactly as the binary per-lane walk computes them.
__global__ __launch_bounds__(256) void node4_kernel(const BvhNode *nodes, int n, BvhNode4 *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const BvhNode b = nodes[i];
    BvhNode4 r;
    for (int q = 0; q < 4; q++) { r.child[q] = kNoChild; r.pad[q] = 0; for (int k = 0; k < 3; k++) { r.lo[k][q] = 3.0e38f; r.hi[k][q] = -3.0e38f; } }
    const int32_t ch[2] = { b.c0, b.c1 };
    for (int s = 0; s < 2; s++) {
        if (ch[s] >= 0) {
            const BvhNode c = nodes[ch[s]];
            const int32_t gc[2] = { c.c0, c.c1 };
            for (int q = 0; q < 2; q++) {
                r.child[2 * s + q] = gc[q];
                for (int k = 0; k < 3; k++) { r.lo[k][2 * s + q] = c.ctr[k][q] - c.hal[k][q]; r.hi[k][2 * s + q] = c.ctr[k][q] + c.hal[k][q]; }
            }
        } else if (ch[s] != kNoChild) {
            r.child[2 * s] = ch[s];
            for (int k = 0; k < 3; k++) { r.lo[k][2 * s] = b.ctr[k][s] - b.hal[k][s]; r.hi[k][2 * s] = b.ctr[k][s] + b.hal[k][s]; }
        }
    }
    out[i] = r;
}

#define GB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = e_; goto done; } } while (0)

} // namespace

// The four-wide nodes of a flattened binary tree that is already on the device (all builders); *out is a device allocation.
int build_nodes4(const BvhNode *d_nodes, int32_t nnodes, hipStream_t stream, BvhNode4 **out) {
    BvhNode4 *p = nullptr;
    hipError_t e = hipMalloc((void **)&p, sizeof(BvhNode4) * (size_t)std::max(nnodes, 1));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(node4_kernel, dim3((unsigned)((nnodes + 255) / 256)), dim3(256), 0, stream, d_nodes, nnodes, p);
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { hipFree(p); return (int)e; }
    *out = p;
    return 0;
}

// Builds on `stream` from the host triangle list; the four output arrays are device allocations owned by the caller
// (hipFree).  Returns hipSuccess or the failing HIP status.
int build_bvh_gpu(const float *verts_host, int32_t ntri, float pad_scale, hipStream_t stream, BvhDeviceBuild *out) {
    auto t0 = std::chrono::steady_clock::now();
    hipError_t err = hipSuccess;
    const int nt = std::max(ntri, 1);
    float *d_verts = nullptr; Bx *tbox = nullptr, *lbox = nullptr, *ibox = nullptr; uint8_t *valid = nullptr; uint32_t *bounds = nullptr;
    uint64_t *keys = nullptr, *keys2 = nullptr; int32_t *ids = nullptr, *ids2 = nullptr, *parent_int = nullptr, *parent_leaf = nullptr;
    Topo *topo = nullptr; uint32_t *visits = nullptr, *kept = nullptr, *new_id = nullptr, *head = nullptr, *flag = nullptr, *block_id = nullptr, *scal = nullptr;
    void *tmp = nullptr; size_t tmp_bytes = 0, need = 0;
    uint32_t h_bounds[13], h_counts[4];
    int n = 0, nnodes = 0, nblocks = 0;
    const unsigned gt = (unsigned)((nt + 255) / 256);
    BvhNode *nodes = nullptr; LeafBlock *leaves = nullptr; TriFlat *tri_flat = nullptr; int32_t *tri_index = nullptr;

    GB_TRY(hipMalloc((void **)&d_verts, sizeof(float) * 9 * (size_t)nt));
    GB_TRY(hipMalloc((void **)&tbox, sizeof(Bx) * (size_t)nt)); GB_TRY(hipMalloc((void **)&lbox, sizeof(Bx) * (size_t)nt)); GB_TRY(hipMalloc((void **)&ibox, sizeof(Bx) * (size_t)nt));
    GB_TRY(hipMalloc((void **)&valid, (size_t)nt)); GB_TRY(hipMalloc((void **)&bounds, sizeof(uint32_t) * 16));
    GB_TRY(hipMalloc((void **)&keys, 8 * (size_t)nt)); GB_TRY(hipMalloc((void **)&keys2, 8 * (size_t)nt));
    GB_TRY(hipMalloc((void **)&ids, 4 * (size_t)nt)); GB_TRY(hipMalloc((void **)&ids2, 4 * (size_t)nt));
    GB_TRY(hipMalloc((void **)&parent_int, 4 * (size_t)nt)); GB_TRY(hipMalloc((void **)&parent_leaf, 4 * (size_t)nt));
    GB_TRY(hipMalloc((void **)&topo, sizeof(Topo) * (size_t)nt));
    GB_TRY(hipMalloc((void **)&visits, 4 * (size_t)nt)); GB_TRY(hipMalloc((void **)&kept, 4 * (size_t)nt)); GB_TRY(hipMalloc((void **)&new_id, 4 * (size_t)nt));
    GB_TRY(hipMalloc((void **)&head, 4 * (size_t)nt)); GB_TRY(hipMalloc((void **)&flag, 4 * (size_t)nt)); GB_TRY(hipMalloc((void **)&block_id, 4 * (size_t)nt));
    GB_TRY(hipMalloc((void **)&scal, 4 * 4));
    if (ntri > 0) GB_TRY(hipMemcpyAsync(d_verts, verts_host, sizeof(float) * 9 * (size_t)ntri, hipMemcpyHostToDevice, stream));
    {
        const uint32_t init[13] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u };
        GB_TRY(hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(tri_setup_kernel, dim3(gt), dim3(256), 0, stream, d_verts, ntri, tbox, bounds, valid);
    hipLaunchKernelGGL(morton_kernel, dim3(gt), dim3(256), 0, stream, tbox, valid, ntri, bounds, keys, ids);
    if (ntri > 0) {
        GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, keys2, ids, ids2, ntri, 0, 64, stream));
        tmp_bytes = need;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, kept, new_id, nt, stream));
        tmp_bytes = std::max(tmp_bytes, need);
        GB_TRY(hipMalloc(&tmp, tmp_bytes));
        need = tmp_bytes;
        GB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, need, keys, keys2, ids, ids2, ntri, 0, 64, stream));
    }
    GB_TRY(hipMemcpyAsync(h_bounds, bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    n = (int)h_bounds[12];                                                // valid triangles: the first n sorted positions
    GB_TRY(hipMemsetAsync(visits, 0, 4 * (size_t)nt, stream)); GB_TRY(hipMemsetAsync(head, 0, 4 * (size_t)nt, stream));
    GB_TRY(hipMemsetAsync(kept, 0, 4 * (size_t)nt, stream)); GB_TRY(hipMemsetAsync(scal, 0, 16, stream));
    if (n > 0) {
        const unsigned gn = (unsigned)((n + 255) / 256);
        if (n > 1) hipLaunchKernelGGL(hierarchy_kernel, dim3(gn), dim3(256), 0, stream, keys2, n, topo, parent_int, parent_leaf);
        hipLaunchKernelGGL(refit_kernel, dim3(gn), dim3(256), 0, stream, tbox, ids2, n, topo, parent_int, parent_leaf, lbox, ibox, visits, new_id /* free until the scan below: node heights */, &scal[0]);
        if (n > 1) hipLaunchKernelGGL(collapse_kernel, dim3(gn), dim3(256), 0, stream, topo, n, kept, head);
        else GB_TRY(hipMemsetAsync(head, 0, 4, stream));
        if (n == 1) { const uint32_t one = 1u; GB_TRY(hipMemcpyAsync(head, &one, 4, hipMemcpyHostToDevice, stream)); }
        hipLaunchKernelGGL(flag_kernel, dim3(gn), dim3(256), 0, stream, head, n, flag);
        need = tmp_bytes; GB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, need, kept, new_id, n, stream));
        need = tmp_bytes; GB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, need, flag, block_id, n, stream));
        // counts: kept nodes = new_id[n-2] + kept[n-2] (n >= 2), blocks = block_id[n-1] + flag[n-1]
        GB_TRY(hipMemcpyAsync(&h_counts[0], &block_id[n - 1], 4, hipMemcpyDeviceToHost, stream));
        GB_TRY(hipMemcpyAsync(&h_counts[1], &flag[n - 1], 4, hipMemcpyDeviceToHost, stream));
        if (n > 1) { GB_TRY(hipMemcpyAsync(&h_counts[2], &new_id[n - 2], 4, hipMemcpyDeviceToHost, stream)); GB_TRY(hipMemcpyAsync(&h_counts[3], &kept[n - 2], 4, hipMemcpyDeviceToHost, stream)); }
        else h_counts[2] = h_counts[3] = 0;
        GB_TRY(hipStreamSynchronize(stream));
        nblocks = (int)(h_counts[0] + h_counts[1]);
        nnodes = std::max((int)(h_counts[2] + h_counts[3]), 1);           // (<= 4 triangles: the wrapper root)
    } else nnodes = 1;

    GB_TRY(hipMalloc((void **)&nodes, sizeof(BvhNode) * (size_t)nnodes));
    GB_TRY(hipMalloc((void **)&leaves, sizeof(LeafBlock) * (size_t)std::max(nblocks, 1)));
    GB_TRY(hipMalloc((void **)&tri_flat, sizeof(TriFlat) * 4 * (size_t)std::max(nblocks, 1)));
    GB_TRY(hipMalloc((void **)&tri_index, sizeof(int32_t) * 4 * (size_t)std::max(nblocks, 1)));
    GB_TRY(hipMemsetAsync(leaves, 0, sizeof(LeafBlock) * (size_t)std::max(nblocks, 1), stream));
    GB_TRY(hipMemsetAsync(tri_flat, 0, sizeof(TriFlat) * 4 * (size_t)std::max(nblocks, 1), stream));
    GB_TRY(hipMemsetAsync(tri_index, 0xff, sizeof(int32_t) * 4 * (size_t)std::max(nblocks, 1), stream));
    if (n > 0) {
        const unsigned gn = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(emit_nodes_kernel, dim3(gn), dim3(256), 0, stream, topo, n, kept, new_id, head, block_id, lbox, ibox, bounds, pad_scale, nodes);
        hipLaunchKernelGGL(emit_leaves_kernel, dim3(gn), dim3(256), 0, stream, d_verts, ids2, n, head, block_id, leaves, tri_flat, tri_index);
    } else {
        BvhNode r; std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) { r.hal[k][0] = r.hal[k][1] = -3.0e38f; }
        r.c0 = r.c1 = kNoChild;
        GB_TRY(hipMemcpyAsync(nodes, &r, sizeof(r), hipMemcpyHostToDevice, stream));
    }
    GB_TRY(hipMemcpyAsync(&h_counts[0], &scal[0], 4, hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    GB_TRY(hipGetLastError());
    out->nodes = nodes; out->leaves = leaves; out->tri_flat = tri_flat; out->tri_index = tri_index;
    out->nnodes = nnodes; out->nleaves = nblocks; out->ntris = n; out->depth = (int32_t)h_counts[0] + 2;
    nodes = nullptr; leaves = nullptr; tri_flat = nullptr; tri_index = nullptr;
done:
    hipFree(d_verts); hipFree(tbox); hipFree(lbox); hipFree(ibox); hipFree(valid); hipFree(bounds); hipFree(keys); hipFree(keys2); hipFree(ids); hipFree(ids2);
    hipFree(parent_int); hipFree(parent_leaf); hipFree(topo); hipFree(visits); hipFree(kept); hipFree(new_id); hipFree(head); hipFree(flag); hipFree(block_id); hipFree(scal);
    hipFree(tmp); hipFree(nodes); hipFree(leaves); hipFree(tri_flat); hipFree(tri_index);
    out->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return (int)err;
}

} // namespace evplp
