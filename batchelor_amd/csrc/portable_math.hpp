// Bit-reproducible exp / log1p / logspace_add for the adjust_shift_variance quantile walk (the product's copy; the
// tests' CPU checker carries its own copy of the same arithmetic).
#pragma once
#include <hip/hip_runtime.h>
#define BMX_PM_FN __host__ __device__ __forceinline__
namespace bmx {
/* exp(x) for x <= 0 and log1p(y) for 0 <= y <= 1 from +, -, *, / and integer operations only, so that a C compiler
 * (-ffp-contract=off) and hipcc (-ffp-contract=off) produce the same bits for the same input: R::logspace_add
 * (Rmath) calls the platform's exp and log1p, whose last bit differs between math libraries -- and the quantile walk
 * of adjust_shift_variance (src/adjust_shift_variance.cpp:137-157) turns a last-bit difference into a different
 * cell.  Accuracy: a few units in the last place (argument reduction by ln 2 in two pieces + degree-13 Taylor
 * polynomial; 2 atanh(y / (2 + y)) as an 18-term odd series). */
BMX_PM_FN double bmx_pm_exp_neg(double x) {
    if (!(x > -700.0)) return 0.0; /* below ~1e-304 (and NaN): nothing an addend of >= 1 ulp could notice */
    if (x > 0.0) x = 0.0;
    const double inv_ln2 = 1.4426950408889634074;
    const double ln2_hi = 6.93147180369123816490e-01; /* 32 significant bits: n * ln2_hi is exact */
    const double ln2_lo = 1.90821492927058770002e-10;
    const long long n = (long long)(x * inv_ln2 - 0.5); /* x <= 0: truncation rounds to nearest */
    const double r = (x - (double)n * ln2_hi) - (double)n * ln2_lo; /* |r| <= 0.35 */
    double p = 1.0 / 6227020800.0; /* 1 / 13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    /* 2^n, n in [-1011, 0]: a normal number built from its exponent field */
    union {
        unsigned long long u;
        double d;
    } two_n;
    two_n.u = (unsigned long long)(n + 1023) << 52;
    return p * two_n.d;
}

BMX_PM_FN double bmx_pm_log1p_unit(double y) {
    const double s = y / (2.0 + y); /* <= 1/3 */
    const double z = s * s;
    double q = 1.0 / 35.0;
    q = q * z + 1.0 / 33.0;
    q = q * z + 1.0 / 31.0;
    q = q * z + 1.0 / 29.0;
    q = q * z + 1.0 / 27.0;
    q = q * z + 1.0 / 25.0;
    q = q * z + 1.0 / 23.0;
    q = q * z + 1.0 / 21.0;
    q = q * z + 1.0 / 19.0;
    q = q * z + 1.0 / 17.0;
    q = q * z + 1.0 / 15.0;
    q = q * z + 1.0 / 13.0;
    q = q * z + 1.0 / 11.0;
    q = q * z + 1.0 / 9.0;
    q = q * z + 1.0 / 7.0;
    q = q * z + 1.0 / 5.0;
    q = q * z + 1.0 / 3.0;
    q = q * z + 1.0;
    return 2.0 * s * q;
}

/* R::logspace_add: log(exp(lx) + exp(ly)) = max + log1p(exp(-|lx - ly|)) */
BMX_PM_FN double bmx_pm_logspace_add(double lx, double ly) {
    const double m = lx > ly ? lx : ly;
    const double dlt = lx > ly ? ly - lx : lx - ly; /* -|lx - ly| */
    return m + bmx_pm_log1p_unit(bmx_pm_exp_neg(dlt));
}
}  // namespace bmx
