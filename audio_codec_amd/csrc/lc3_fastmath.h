/* lc3_fastmath.h -- log2, log10 and exp2 of a FLOAT, evaluated in double and rounded to float: the device's form of the reference's run-time
 * libm calls (the reference calls log2f / log10f / powf(2, x) of the host's libm; the device - and oracle/liblc3_oracle_pm.so -
 * evaluate (float)f((double)x)).  Until round 3 that f was the device library's fp64 log2 / log10 / exp2: 40 ... 80 double-precision instructions per
 * call at half rate - 42 % of lc3_enc_scf_lane_kernel's and 19 % of lc3_enc_shape_lane_kernel's vector instructions - and equal to glibc's double
 * functions "almost always".  Because the ARGUMENT is a float there are only 2^31 positive inputs, so a short table-driven evaluation can be checked
 * against glibc for EVERY input (tools/fastmath_check.c, recorded in profiles/r04_fastmath_check.txt; a sample of it runs in
 * tests/test_fastmath.py), and the same few IEEE operations (fma, +, *, integer shifts) give the same bits on the host and on the device.
 *
 *   log:  x = 2^k z, z in [0.6875, 1.375); interval i = top 7 bits of bits(z) - bits(0.6875); r = z * invc[i] - 1 EXACTLY (invc has 28 bits, z 24);
 *         log2 x = (k + hi[i]) + (lo[i] + r P(r)) with k + hi[i] exact (hi on a 2^-40 grid), P of degree 8; the two intervals around 1 have invc = 1, hi = lo = 0.
 *         log10 likewise with k LOG10_2_HI + hi[i] exact.
 *   exp2: y = (k + j / 64) + r, |r| <= 1 / 128 exactly; 2^y = 2^k T[j] (1 + r Q(r)), Q of degree 5.
 * This header is compiled by hipcc into the kernels and by gcc into the checker; it has no other users (the oracle keeps calling glibc: that IS the definition). */
#ifndef LC3_FASTMATH_H
#define LC3_FASTMATH_H
#include <stdint.h>
#if defined(__HIPCC__)
#define LC3M_FN __device__ __forceinline__
#define LC3M_TABLE static __device__ const
#else
#include <math.h>
#define LC3M_FN static inline
#define LC3M_TABLE static const
#endif
#include "lc3_fastmath_tables.h"

LC3M_FN uint64_t lc3m_bits(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
LC3M_FN double lc3m_dbl(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }

/* the common part of both logarithms: k, r and the table entry; 0 for arguments the caller hands to the library function (<= 0, inf, NaN) */
LC3M_FN int lc3m_log_reduce(float x, const double* __restrict__ tab, double* kd, double* r, double* hi, double* lo)
{
    const uint64_t ix = lc3m_bits((double)x);
    if (!(ix - 0x0010000000000000ULL < 0x7FE0000000000000ULL)) return 0;       /* not a positive finite number (a float subnormal is a normal double) */
    const uint64_t tmp = ix - 0x3FE6000000000000ULL;
    const int i = (int)(tmp >> 45) & 127;
    const int64_t k = (int64_t)tmp >> 52;
    const double z = lc3m_dbl(ix - (tmp & 0xFFF0000000000000ULL));
    const double* T = tab + 3 * i;
    *r = __builtin_fma(z, T[0], -1.0);
    *hi = T[1]; *lo = T[2]; *kd = (double)(int)k;
    return 1;
}
LC3M_FN double lc3m_poly8(const double* __restrict__ c, double r)
{
    double p = c[8];
    p = __builtin_fma(p, r, c[7]); p = __builtin_fma(p, r, c[6]); p = __builtin_fma(p, r, c[5]); p = __builtin_fma(p, r, c[4]);
    p = __builtin_fma(p, r, c[3]); p = __builtin_fma(p, r, c[2]); p = __builtin_fma(p, r, c[1]); p = __builtin_fma(p, r, c[0]);
    return p;
}
/* Branch-free forms for code that evaluates several logarithms in a row (a branch per call keeps the compiler from overlapping their table reads): the
 * result for a positive finite x, anything for the rest; *special is set for the rest, and the caller takes lc3m_log*f() for those afterwards (rare: one
 * wave-level test behind the group). */
LC3M_FN double lc3m_log_reduce_nb(float x, const double* __restrict__ tab, double* kd, double* hi, double* lo, int* special)
{
    const uint64_t ix = lc3m_bits((double)x);
    *special = !(ix - 0x0010000000000000ULL < 0x7FE0000000000000ULL);
    const uint64_t tmp = ix - 0x3FE6000000000000ULL;
    const int i = (int)(tmp >> 45) & 127;
    const double z = lc3m_dbl(ix - (tmp & 0xFFF0000000000000ULL));
    const double* T = tab + 3 * i;
    *hi = T[1]; *lo = T[2]; *kd = (double)(int)((int64_t)tmp >> 52);
    return __builtin_fma(z, T[0], -1.0);
}
LC3M_FN float lc3m_log2f_nb(float x, const double* __restrict__ tab, int* special)
{
    double kd, hi, lo;
    const double r = lc3m_log_reduce_nb(x, tab, &kd, &hi, &lo, special);
    return (float)((kd + hi) + __builtin_fma(r, lc3m_poly8(lc3m_log2_poly, r), lo));
}
LC3M_FN float lc3m_log10f_nb(float x, const double* __restrict__ tab, int* special)
{
    double kd, hi, lo;
    const double r = lc3m_log_reduce_nb(x, tab, &kd, &hi, &lo, special);
    return (float)(__builtin_fma(kd, LC3M_LOG10_2_HI, hi) + __builtin_fma(r, lc3m_poly8(lc3m_log10_poly, r), __builtin_fma(kd, LC3M_LOG10_2_LO, lo)));
}
/* (float)log2((double)x); tab = lc3m_log2_tab or a copy of it */
LC3M_FN float lc3m_log2f(float x, const double* __restrict__ tab)
{
    double kd, r, hi, lo;
    if (!lc3m_log_reduce(x, tab, &kd, &r, &hi, &lo)) return (float)log2((double)x);
    return (float)((kd + hi) + __builtin_fma(r, lc3m_poly8(lc3m_log2_poly, r), lo));
}
/* (float)log10((double)x); tab = lc3m_log10_tab or a copy of it */
LC3M_FN float lc3m_log10f(float x, const double* __restrict__ tab)
{
    double kd, r, hi, lo;
    if (!lc3m_log_reduce(x, tab, &kd, &r, &hi, &lo)) return (float)log10((double)x);
    return (float)(__builtin_fma(kd, LC3M_LOG10_2_HI, hi) + __builtin_fma(r, lc3m_poly8(lc3m_log10_poly, r), __builtin_fma(kd, LC3M_LOG10_2_LO, lo)));
}
/* (float)exp2((double)x); tab = lc3m_exp2_tab or a copy of it */
LC3M_FN float lc3m_exp2f(float x, const double* __restrict__ tab)
{
    const double y = (double)x;
    if (!(__builtin_fabs(y) < 1000.0)) return (float)exp2(y);                  /* far outside the float range either way; NaN */
    double kd = y * 64.0 + 0x1.8p52;                                          /* the product is exact, the sum rounds it to an integer */
    const uint64_t ki = lc3m_bits(kd);
    kd -= 0x1.8p52;
    const double r = __builtin_fma(kd, -0x1p-6, y);                           /* exact: |r| <= 1 / 128 */
    const double scale = lc3m_dbl(lc3m_bits(tab[ki & 63]) + ((ki >> 6) << 52));
    const double* c = lc3m_exp2_poly;
    double q = c[5];
    q = __builtin_fma(q, r, c[4]); q = __builtin_fma(q, r, c[3]); q = __builtin_fma(q, r, c[2]); q = __builtin_fma(q, r, c[1]); q = __builtin_fma(q, r, c[0]);
    return (float)__builtin_fma(scale, r * q, scale);
}
#endif
