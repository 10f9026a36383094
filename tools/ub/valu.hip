// Micro-benchmark: VALU issue rate of plain vs packed fp32 FMA / min-max on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2f pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                                          x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; j++) { p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb); p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb); }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) { x0 = fmaxf(x0, a + j); x1 = fminf(x1, b + j); x2 = fmaxf(x2, a - j); x3 = fminf(x3, b - j); x4 = fmaxf(x4, a * j); x5 = fminf(x5, b * j); x6 = fmaxf(x6, a + 2 * j); x7 = fminf(x7, b + 3 * j); }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) { p0 = __builtin_elementwise_max(p0, pa + (float)j); p1 = __builtin_elementwise_min(p1, pb + (float)j); p2 = __builtin_elementwise_max(p2, pa - (float)j); p3 = __builtin_elementwise_min(p3, pb - (float)j); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, int blocks_per_cu) {
    float *out; hipMalloc(&out, 256 * 256 * 8 * 4 * 4);
    int iters = 20000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<MODE><<<grid, 256>>>(out, 1.0001f, 0.5f, 10);
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, 1.0001f, 0.5f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)grid * 4 * iters * 64;   // wave-instructions of the unrolled body
    printf("%-22s %d waves/SIMD: %.2f ms, %.3f wave-instr/cycle/SIMD at 2.4 GHz\n", name, blocks_per_cu, ms, insts / (ms * 1e-3 * 2.4e9) / 1024);
    hipFree(out);
}
int main() {
    for (int b : {1, 2, 4, 8}) { run<0>("v_fma_f32", b); run<1>("v_pk_fma_f32", b); run<2>("v_max/min_f32", b); run<3>("v_pk_max/min_f32", b); }
    return 0;
}
