// Micro-benchmark: device-scope integer atomics on gfx950 as the photon binning uses them (developer tool).
//   1. returning atomicAdd, every lane its own address scattered over a table (the per-tile bin cursors of a one-level design)
//   2. returning atomicAdd, one lane per wave, ALL waves on one address (a statistics word, a hot bin cursor)
//   3. the same on 1024 addresses 128 bytes apart (sharded counters)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ub/atomics tools/ub/atomics.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void scattered(uint32_t *table, uint32_t mask, uint32_t *sink, int per_lane) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u, acc = 0;
    for (int i = 0; i < per_lane; i++) { x = x * 1664525u + 1013904223u; acc += atomicAdd(&table[(x >> 8) & mask], 1u); }
    if (acc == 0xffffffffu) sink[0] = acc;
}
__global__ __launch_bounds__(256) void one_lane_per_wave(uint32_t *table, uint32_t shards, uint32_t *sink) {
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
        const uint32_t r = atomicAdd(&table[(w % shards) * 32u], 1u);
        if (r == 0xffffffffu) sink[0] = r;
    }
}
static float timed(void (*launch)(void *), void *arg) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(arg); hipDeviceSynchronize();
    hipEventRecord(e0); launch(arg); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
struct A { uint32_t *table, *sink; uint32_t mask, shards; int grid, per_lane; };
static void l1(void *p) { A *a = (A *)p; hipLaunchKernelGGL(scattered, dim3(a->grid), dim3(256), 0, 0, a->table, a->mask, a->sink, a->per_lane); }
static void l2(void *p) { A *a = (A *)p; hipLaunchKernelGGL(one_lane_per_wave, dim3(a->grid), dim3(256), 0, 0, a->table, a->shards, a->sink); }
int main() {
    A a; hipMalloc(&a.table, 64u << 20); hipMalloc(&a.sink, 64); hipMemset(a.table, 0, 64u << 20);
    for (uint32_t entries : { 16384u, 1u << 20 }) {
        a.mask = entries - 1; a.grid = 8192; a.per_lane = 4;
        const float ms = timed(l1, &a);
        printf("scattered returning atomicAdd, %u cursors: %.1f G atomics/s (%d atomics in %.1f us)\n", entries, a.grid * 256.0 * a.per_lane / ms / 1e6, a.grid * 256 * a.per_lane, ms * 1e3);
    }
    for (uint32_t shards : { 1u, 8u, 64u, 1024u }) {
        a.shards = shards; a.grid = 8192;                      // 32768 waves, one atomic each
        const float ms = timed(l2, &a);
        printf("one returning atomicAdd per wave, 32768 waves on %4u address(es) 128 B apart: %7.1f us = %.0f ns per atomic and address\n", shards, ms * 1e3, ms * 1e6 / (32768.0 / shards));
    }
    return 0;
}
