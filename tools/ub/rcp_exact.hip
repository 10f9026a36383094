// Exhaustive check: is  rcp_exact(x) == 1.0f / x  (IEEE-correct division as the compiler expands it) for EVERY float bit pattern?
// rcp_exact is the division's own Newton-Raphson + two residual corrections on v_rcp_f32 without the range scaling
// (v_div_scale / v_div_fixup) that only matters for results near the ends of the exponent range.  Prints the number of
// patterns where the two differ, split by class, and the range of |x| over which they agree everywhere.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ub/rcp_exact tools/ub/rcp_exact.hip && tools/ub/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__device__ __forceinline__ float rcp_exact(float x) {
#pragma clang fp contract(off)
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    float err = __builtin_fmaf(-x, q, 1.0f);
    q = __builtin_fmaf(err, r, q);
    err = __builtin_fmaf(-x, q, 1.0f);
    return __builtin_fmaf(err, r, q);
}
__global__ void check(unsigned long long *bad, unsigned *lo_bad, unsigned *hi_bad) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)i;
        const float x = __uint_as_float(b);
        const float ref = 1.0f / x, got = rcp_exact(x);
        const uint32_t rb = __float_as_uint(ref), gb = __float_as_uint(got);
        const bool same = rb == gb || (ref != ref && got != got);
        if (!same) {
            const uint32_t ex = (b >> 23) & 0xff;
            atomicAdd(&bad[0], 1ull);
            if (ex == 0) atomicAdd(&bad[1], 1ull);               // zero / denormal input
            else if (ex == 255) atomicAdd(&bad[2], 1ull);        // inf / nan input
            else { atomicAdd(&bad[3], 1ull); atomicMin(lo_bad, ex); atomicMax(hi_bad, ex); }
        }
    }
}
int main() {
    unsigned long long *bad; unsigned *lo, *hi;
    hipMalloc(&bad, 32); hipMemset(bad, 0, 32); hipMalloc(&lo, 4); hipMalloc(&hi, 4);
    unsigned l0 = 255, h0 = 0; hipMemcpy(lo, &l0, 4, hipMemcpyHostToDevice); hipMemcpy(hi, &h0, 4, hipMemcpyHostToDevice);
    check<<<4096, 256>>>(bad, lo, hi);
    unsigned long long h[4]; hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(&l0, lo, 4, hipMemcpyDeviceToHost); hipMemcpy(&h0, hi, 4, hipMemcpyDeviceToHost);
    printf("patterns that differ: %llu (zero/denormal input %llu, inf/nan input %llu, normal input %llu; biased exponents of the normal ones %u..%u)\n", h[0], h[1], h[2], h[3], l0, h0);
    return 0;
}
