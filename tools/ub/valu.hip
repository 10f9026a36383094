// Micro-benchmark (developer tool): issue rate of the vector instructions the packet walk is made of, on gfx950, at 1 / 2 / 4 / 8
// waves per SIMD.  Every body is inline asm on eight independent registers (no dependence between consecutive instructions), so
// the compiler can neither fuse, pack nor drop anything.  Reported per line: nanoseconds per wave-instruction per SIMD from the event
// time of the launch (what a kernel pays), the clock the chip held inside the loop (s_memtime ticks / s_memrealtime ticks x 100 MHz,
// median over workgroups: the chip lowers its clock under load, by instruction mix) and their product, cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ub/valu tools/ub/valu.hip && tools/ub/valu > profiles/ub_valu_r04.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

enum { FMA, PK_FMA, PK_FMA_OPSEL, PK_MUL, MAX3, MAX3_CLAMP, CMP_VCC, CMP_SGPR, EXP, LOG, RCP, MOV, WRITELANE, MIX_VISIT, MUL_LO_U32, MAD_U64_U32, MUL_F32, SQRT, SIN, ALIGNBIT, NMODES };
static const char *kNames[NMODES] = { "v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel", "v_pk_mul_f32", "v_max3_f32", "v_max3_f32 clamp", "v_cmp_lt_f32 vcc",
                                      "v_cmp_lt_f32 s[..]", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_mov_b32", "v_writelane_b32", "node visit (9 pk_fma + 4 max3/min3 + 2 cmp)",
                                      "v_mul_lo_u32", "v_mad_u64_u32", "v_mul_f32", "v_sqrt_f32", "v_sin_f32", "v_alignbit_b32" };
static const int kPerIter[NMODES] = { 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 60, 64, 64, 64, 64, 64, 64 };

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float a, float b, int iters) {
    float x[8]; v2f p[8]; unsigned long long q[8];
    for (int j = 0; j < 8; j++) { x[j] = threadIdx.x + j; p[j].x = x[j]; p[j].y = x[j] + 0.5f; q[j] = threadIdx.x * 77u + j; }
    v2f pa = { a, a * 1.5f }, pb = { b, b * 0.5f };
    unsigned long long sg = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == FMA) {
#define OP(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
                R8(OP)
#undef OP
            } else if (MODE == PK_FMA) {
#define OP(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pa), "v"(pb));
                R8(OP)
#undef OP
            } else if (MODE == PK_FMA_OPSEL) {
#define OP(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(p[j]) : "v"(pa), "v"(pb));
                R8(OP)
#undef OP
            } else if (MODE == PK_MUL) {
#define OP(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pa));
                R8(OP)
#undef OP
            } else if (MODE == MAX3) {
#define OP(j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
                R8(OP)
#undef OP
            } else if (MODE == MAX3_CLAMP) {
#define OP(j) asm volatile("v_max3_f32 %0, %0, %1, %2 clamp" : "+v"(x[j]) : "v"(a), "v"(b));
                R8(OP)
#undef OP
            } else if (MODE == CMP_VCC) {
#define OP(j) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[j]), "v"(a) : "vcc");
                R8(OP)
#undef OP
            } else if (MODE == CMP_SGPR) {
#define OP(j) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(sg) : "v"(x[j]), "v"(a));
                R8(OP)
#undef OP
            } else if (MODE == EXP) {
#define OP(j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
                R8(OP)
#undef OP
            } else if (MODE == LOG) {
#define OP(j) asm volatile("v_log_f32 %0, %0" : "+v"(x[j]));
                R8(OP)
#undef OP
            } else if (MODE == RCP) {
#define OP(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
                R8(OP)
#undef OP
            } else if (MODE == MOV) {
#define OP(j) asm volatile("v_mov_b32 %0, %1" : "+v"(x[j]) : "v"(a));
                R8(OP)
#undef OP
            } else if (MODE == MUL_LO_U32) {
#define OP(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                R8(OP)
#undef OP
            } else if (MODE == MAD_U64_U32) {      // the 64-bit multiply-add an LCG step is made of (plus two v_mul_lo_u32 for the high word)
#define OP(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[j]) : "v"(a), "v"(b) : "vcc");
                R8(OP)
#undef OP
            } else if (MODE == MUL_F32) {
#define OP(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                R8(OP)
#undef OP
            } else if (MODE == SQRT) {
#define OP(j) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[j]));
                R8(OP)
#undef OP
            } else if (MODE == SIN) {
#define OP(j) asm volatile("v_sin_f32 %0, %0" : "+v"(x[j]));
                R8(OP)
#undef OP
            } else if (MODE == ALIGNBIT) {
#define OP(j) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(x[j]) : "v"(a));
                R8(OP)
#undef OP
            } else if (MODE == WRITELANE) {
                int sp = (i + u) & 63, val = i;
#define OP(j) asm volatile("s_mov_b32 m0, %1\n\tv_writelane_b32 %0, %2, m0" : "+v"(x[j]) : "s"(sp), "s"(val) : "m0");
                R8(OP)
#undef OP
            }
        }
        if (MODE == MIX_VISIT) {
            // the vector instructions of one node visit of the packet walk, in its order, on registers of their own (4 copies per iteration)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile(
                    "v_pk_fma_f32 %0, %10, %11, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                    "v_pk_fma_f32 %1, %10, %11, %12 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"
                    "v_pk_fma_f32 %2, %10, %12, %11 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                    "v_pk_fma_f32 %3, %10, %12, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
                    "v_pk_fma_f32 %4, %10, %11, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
                    "v_pk_fma_f32 %5, %10, %11, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"
                    "v_pk_fma_f32 %0, %10, %12, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                    "v_pk_fma_f32 %1, %10, %11, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                    "v_pk_fma_f32 %2, %10, %11, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                    "v_max3_f32 %7, %7, %8, %9 clamp\n\t"
                    "v_min3_f32 %8, %8, %9, %7 clamp\n\t"
                    "v_max3_f32 %9, %9, %7, %8 clamp\n\t"
                    "v_min3_f32 %7, %7, %8, %9 clamp\n\t"
                    "v_cmp_lt_f32 vcc, %7, %8\n\t"
                    "v_cmp_lt_f32 %6, %9, %7\n\t"
                    : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+s"(sg), "+v"(x[0]), "+v"(x[1]), "+v"(x[2])
                    : "s"(pa), "v"(pb), "v"(p[6])
                    : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)(sg & 1);
    for (int j = 0; j < 8; j++) s += x[j] + p[j].x + p[j].y + (float)(q[j] & 0xffull);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE> void run(int waves_per_simd, FILE *f) {
    const int grid = 256 * waves_per_simd, iters = 4000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&cyc, (size_t)grid * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, cyc, 1.0001f, 0.5f, 10);
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, cyc, 1.0001f, 0.5f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 2); hipMemcpy(h.data(), cyc, (size_t)grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz((size_t)grid);
    for (int b = 0; b < grid; b++) ghz[b] = (double)h[2 * b] / (double)std::max<unsigned long long>(h[2 * b + 1], 1ull) * 0.1;      // s_memrealtime ticks at 100 MHz
    std::sort(ghz.begin(), ghz.end());
    const double clock = ghz[grid / 2];
    const double per_simd = (double)iters * kPerIter[MODE] * waves_per_simd;   // wave-instructions one SIMD issued (every SIMD holds waves_per_simd of the launch's waves)
    const double ns = ms * 1e6 / per_simd;
    std::fprintf(f, "%-46s %d waves/SIMD: %7.3f ms  %6.3f ns per wave-instruction per SIMD  clock %.2f GHz  -> %5.2f cycles\n", kNames[MODE], waves_per_simd, ms, ns, clock, ns * clock);
    hipFree(out); hipFree(cyc);
}
template <int M> void all(FILE *f) { for (int w : { 1, 2, 4, 8 }) run<M>(w, f); std::fprintf(f, "\n"); }
int main() {
    FILE *f = stdout;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    std::fprintf(f, "# %s, %d CUs, clockRate %d kHz; 256-thread workgroups, one per CU per wave-per-SIMD step; eight independent registers per lane\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    all<FMA>(f); all<PK_FMA>(f); all<PK_FMA_OPSEL>(f); all<PK_MUL>(f); all<MAX3>(f); all<MAX3_CLAMP>(f); all<CMP_VCC>(f); all<CMP_SGPR>(f);
    all<EXP>(f); all<LOG>(f); all<RCP>(f); all<MOV>(f); all<WRITELANE>(f); all<MIX_VISIT>(f);
    all<MUL_LO_U32>(f); all<MAD_U64_U32>(f); all<MUL_F32>(f); all<SQRT>(f); all<SIN>(f); all<ALIGNBIT>(f);
    return 0;
}
