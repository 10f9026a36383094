// Device-side self checks reachable through the C ABI (evplp_selftest): facts the kernels rely on, verified on the part they
// run on rather than assumed.
#include "device_common.hpp"
#include "context.hpp"
#include "ev_math.h"

namespace evplp {

// which = 0: rcp_exact(x) against the IEEE division 1.0f / x on ALL 2^32 bit patterns.
//   out[0] patterns where the bits differ (NaN == NaN), out[1] of them zero / denormal x, out[2] infinite / NaN x, out[3] normal x,
//   out[4] / out[5] smallest / largest biased exponent among the differing normal x (255 / 0 if none).
__global__ __launch_bounds__(256) void selftest_rcp_kernel(unsigned long long *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)i;
        const float x = __uint_as_float(b);
        const float ref = 1.0f / x, got = rcp_exact(x);
        v2f xx; xx.x = x; xx.y = -x;
        const v2f g2 = rcp_exact2(xx);                                   // the packed flavour must agree with the scalar one
        const bool same = (__float_as_uint(ref) == __float_as_uint(got) || (ref != ref && got != got)) &&
                          (__float_as_uint(g2.x) == __float_as_uint(got) || (g2.x != g2.x && got != got)) &&
                          (__float_as_uint(g2.y) == (__float_as_uint(got) ^ 0x80000000u) || (g2.y != g2.y && got != got));
        if (!same) {
            const uint32_t ex = (b >> 23) & 0xffu;
            atomicAdd(&out[0], 1ull);
            if (ex == 0u) atomicAdd(&out[1], 1ull);
            else if (ex == 255u) atomicAdd(&out[2], 1ull);
            else { atomicAdd(&out[3], 1ull); atomicMin(&out[4], (unsigned long long)ex); atomicMax(&out[5], (unsigned long long)ex); }
        }
    }
}

// which = 1: d^e = exp2(e log2 d) on the hardware transcendentals (the Phong lobes of the VPL gather and of the splat) against the
// double-precision pow, for e = 1, 5, 20, 100, 1000, 10000 and 2^22 values of d in (1e-6, 1]: out[k] = the largest relative error over
// the d whose lobe is at least 1e-4 of its peak, in units of 1e-12.
__global__ __launch_bounds__(256) void selftest_pow_kernel(unsigned long long *out) {
    const float es[6] = { 1.0f, 5.0f, 20.0f, 100.0f, 1000.0f, 10000.0f };
    const uint32_t n = 1u << 22;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float d = 1.0f - (float)i / (float)n * 0.999999f;
        for (int k = 0; k < 6; k++) {
            const double want = pow((double)d, (double)es[k]);
            if (want < 1e-4) continue;
            const double got = (double)__builtin_amdgcn_exp2f(es[k] * __builtin_amdgcn_logf(d));
            const double rel = fabs(got - want) / want;
            atomicMax(&out[k], (unsigned long long)(rel * 1e12));
        }
    }
}

// evplp_debug_ev_math: ev_math.h's functions AS THE DEVICE COMPUTES THEM, on the caller's inputs.  The light-tracing records are compared with
// the oracle's byte for byte, and the oracle #includes the same header: that comparison vouches for the walk and the draw order, not for
// these functions -- unless the header really gives the same bits on both machines, which is what tests/test_gpu_parity.py checks with this
// entry point (device results against the oracle's gcc build of the header, dense grids over the callers' input ranges).
__global__ __launch_bounds__(256) void ev_math_kernel(int which, const float *x, const float *y, uint32_t n, float *o0, float *o1) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (which == 0) { float s, c; evm_sincosf(x[i], &s, &c); o0[i] = s; o1[i] = c; }
    else o0[i] = evm_powf(x[i], y[i]);
}

} // namespace evplp

extern "C" int evplp_debug_ev_math(evplp_context *c, int32_t which, const float *x, const float *y, int32_t n, float *out0, float *out1) {
    if (!c || !x || !out0 || n <= 0 || which < 0 || which > 1 || (which == 0 && !out1) || (which == 1 && !y)) { if (c) c->set_error("evplp_debug_ev_math: bad arguments"); return EVPLP_ERR_INVALID; }
    if (hipSetDevice(c->cfg.device) != hipSuccess) return EVPLP_ERR_HIP;
    float *d = nullptr;
    const size_t bytes = sizeof(float) * (size_t)n;
    if (hipMalloc((void **)&d, 4 * bytes) != hipSuccess) { (void)hipGetLastError(); c->set_error("evplp_debug_ev_math: out of memory"); return EVPLP_ERR_OOM; }
    float *dx = d, *dy = d + n, *d0 = d + 2 * (size_t)n, *d1 = d + 3 * (size_t)n;
    hipError_t e = hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && which == 1) e = hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(evplp::ev_math_kernel, dim3(((uint32_t)n + 255u) / 256u), dim3(256), 0, c->stream, which, dx, dy, (uint32_t)n, d0, d1);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out0, d0, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess && which == 0) e = hipMemcpy(out1, d1, bytes, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { c->set_error("evplp_debug_ev_math: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; }
    return EVPLP_OK;
}

extern "C" int evplp_selftest(evplp_context *c, int32_t which, uint64_t *out, int32_t capacity) {
    if (!c || !out || capacity < 6 || which < 0 || which > 1) { if (c) c->set_error("evplp_selftest: bad arguments"); return EVPLP_ERR_INVALID; }
    if (hipSetDevice(c->cfg.device) != hipSuccess) return EVPLP_ERR_HIP;
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 8 * sizeof(unsigned long long)) != hipSuccess) return EVPLP_ERR_OOM;
    const unsigned long long init[8] = { 0, 0, 0, 0, which == 0 ? 255ull : 0ull, 0, 0, 0 };
    hipError_t e = hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (which == 0) hipLaunchKernelGGL(evplp::selftest_rcp_kernel, dim3(4096), dim3(256), 0, c->stream, d);
        else hipLaunchKernelGGL(evplp::selftest_pow_kernel, dim3(4096), dim3(256), 0, c->stream, d);
        e = hipStreamSynchronize(c->stream);
    }
    unsigned long long h[8] = {};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { c->set_error("evplp_selftest: %s", hipGetErrorString(e)); return EVPLP_ERR_HIP; }
    for (int k = 0; k < 6; k++) out[k] = h[k];
    return 6;
}
