/*
 * ev_math.h -- sin / cos / pow for the DIRECTION SAMPLING of light paths, written so that the CPU oracle (gcc, plain C) and the
 * HIP kernels (clang, device code) produce the same bits.
 *
 * Why: light tracing makes discrete decisions downstream of sinf / cosf / powf (which triangle a sub-path hits next).  glibc and
 * ROCm's ocml agree only to an ulp or two, and with library functions ~0.08 % of the light paths of a 300 000-path iteration end
 * up on another triangle somewhere along their way: the record sets of CPU and GPU are then different Monte-Carlo samples of the
 * same integral and every end-to-end comparison is limited to ~1e-3.  The reference itself (CUDA 8 sinf / cosf / powf under nvcc)
 * defines these values no more precisely than "a faithful float result"; here they are pinned to one implementation:
 *   evm_sincosf : Cody-Waite reduction by pi/4 octants + the Cephes single-precision minimax polynomials (|error| < 2 ulp on |x| < 8192)
 *   evm_powf    : x^y = 2^(y log2 x) evaluated in double (log via atanh series of (m-1)/(m+1), exp via its Taylor series, both to
 *                 < 1e-13), rounded once to float: far below half an ulp of the float result in all but ~1e-6 of the cases
 * Only + - * / fma, rint and integer operations are used, every one of them correctly rounded on both machines; fused
 * multiply-adds are written out (__builtin_fma*), everything else must NOT be contracted: clang gets the pragma below, the
 * oracle is compiled with -ffp-contract=off (oracle/Makefile).  tests/test_oracle_selfcheck.py checks both functions against libm.
 * BRDF evaluation (continuous, compared under a tolerance) keeps the library powf.
 */
#ifndef EVPLP_EV_MATH_H
#define EVPLP_EV_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define EVM_FN __host__ __device__ static inline
#else
#define EVM_FN static inline
#endif
#if defined(__clang__)
#define EVM_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define EVM_NO_CONTRACT
#endif

EVM_FN uint64_t evm_bits64(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
EVM_FN double evm_from_bits64(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }

/* sin(x), cos(x) for finite |x| < 8192 (the callers pass phi = 2 pi u, u in (0, 1]) */
EVM_FN void evm_sincosf(float x, float *s_out, float *c_out) {
    EVM_NO_CONTRACT
    const float ax = x < 0.0f ? -x : x;
    /* octant: j = floor(ax * 4/pi), made even by rounding up */
    int j = (int)(ax * 1.27323954473516f);
    j = (j + 1) & ~1;
    const float y = (float)j;
    /* ax - y * pi/4 in three exactly representable pieces (Cephes DP1..DP3) */
    float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const int q = (j >> 1) & 3;         /* quadrant of the reduced angle */
    float s = (q & 1) ? pc : ps, c = (q & 1) ? ps : pc;
    if (q == 1 || q == 2) c = -c;
    if (q == 2 || q == 3) s = -s;
    if (x < 0.0f) s = -s;
    *s_out = s; *c_out = c;
}

/* x^y for x >= 0, finite y (callers: x in (0, 1], y >= 0).  0^y = 0 for y > 0, x^0 = 1. */
EVM_FN float evm_powf(float xf, float yf) {
    EVM_NO_CONTRACT
    if (yf == 0.0f) return 1.0f;
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    /* x = m 2^e, m in [sqrt(1/2), sqrt(2)) */
    uint64_t b = evm_bits64((double)xf);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = evm_from_bits64((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    /* ln m = 2 atanh(f), f = (m - 1) / (m + 1), |f| <= 0.1716 */
    const double f = (m - 1.0) / (m + 1.0), f2 = f * f;
    double p = 1.0 / 17.0;
    p = __builtin_fma(p, f2, 1.0 / 15.0); p = __builtin_fma(p, f2, 1.0 / 13.0); p = __builtin_fma(p, f2, 1.0 / 11.0);
    p = __builtin_fma(p, f2, 1.0 / 9.0); p = __builtin_fma(p, f2, 1.0 / 7.0); p = __builtin_fma(p, f2, 1.0 / 5.0);
    p = __builtin_fma(p, f2, 1.0 / 3.0);
    const double ln_m = 2.0 * __builtin_fma(f * f2, p, f);
    const double log2x = __builtin_fma(ln_m, 1.4426950408889634, (double)e);
    double z = (double)yf * log2x;
    if (z < -1080.0) return 0.0f;
    if (z > 1030.0) z = 1030.0;
    const double n = __builtin_rint(z);
    const double r = (z - n) * 0.6931471805599453;      /* |r| <= 0.3466 */
    double q = 1.0 / 479001600.0;                        /* Taylor series of exp to r^12 / 12! (< 7e-15 here) */
    q = __builtin_fma(q, r, 1.0 / 39916800.0); q = __builtin_fma(q, r, 1.0 / 3628800.0); q = __builtin_fma(q, r, 1.0 / 362880.0);
    q = __builtin_fma(q, r, 1.0 / 40320.0); q = __builtin_fma(q, r, 1.0 / 5040.0); q = __builtin_fma(q, r, 1.0 / 720.0);
    q = __builtin_fma(q, r, 1.0 / 120.0); q = __builtin_fma(q, r, 1.0 / 24.0); q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5); q = __builtin_fma(q, r, 1.0); q = __builtin_fma(q, r, 1.0);
    /* scale by 2^n: n in [-1080, 1030]; two steps keep every factor a normal double */
    int ni = (int)n;
    double scale = 1.0;
    if (ni < -1000) { scale = evm_from_bits64((uint64_t)(1023 - 1000) << 52); ni += 1000; }
    const double pw = evm_from_bits64((uint64_t)(1023 + ni) << 52);
    return (float)(q * pw * scale);
}

#endif /* EVPLP_EV_MATH_H */
