// Accuracy of d^e = exp2(e * log2 d) on the hardware transcendentals (v_log_f32 / v_exp_f32) against double-precision pow, for the
// Phong lobes of the gather and the splat: maximum relative error over d in (1e-6, 1], per exponent, where the lobe is >= 1e-4 of its peak.
// hipcc --offload-arch=gfx950 -O2 -o tools/ub/pow_hw tools/ub/pow_hw.hip && tools/ub/pow_hw
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *d, float e, float *out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(d[i]));
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n), r(n);
    for (int i = 0; i < n; i++) h[i] = 1.0f - (float)i / (float)n * 0.999999f;       // (1e-6, 1]
    float *dd, *dout; hipMalloc(&dd, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dd, h.data(), n * 4, hipMemcpyHostToDevice);
    const float es[] = { 1.0f, 5.0f, 20.0f, 100.0f, 1000.0f, 10000.0f };
    for (float e : es) {
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dd, e, dout, n);
        hipMemcpy(r.data(), dout, n * 4, hipMemcpyDeviceToHost);
        double worst = 0, worst_d = 0; double worst_lib = 0;
        for (int i = 0; i < n; i++) {
            const double want = std::pow((double)h[i], (double)e);
            if (want < 1e-4) continue;
            const double rel = std::fabs((double)r[i] - want) / want;
            if (rel > worst) { worst = rel; worst_d = h[i]; }
            const double rl = std::fabs((double)powf(h[i], e) - want) / want; if (rl > worst_lib) worst_lib = rl;
        }
        std::printf("e = %7.0f: max relative error %.3e at d = %.7f   (host powf: %.3e)\n", e, worst, worst_d, worst_lib);
    }
    return 0;
}
