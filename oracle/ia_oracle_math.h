/*
 * TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/README.md).
 *
 * fp32 elementary functions used by the oracle.  The reference gets exp / log /
 * sigmoid from PyTorch (third-party, pinned torch 1.1.0 by /root/reference
 * INSTALL.md:9; call sites iou_aware_retina_head.py:505,513,531,
 * transforms.py:63-64, losses.py:232-238,478).  PyTorch's vectorised CPU
 * kernels use Sleef (<= 1-2 ulp); we restate them with a classic
 * Cody-Waite + minimax polynomial (Cephes expf/logf coefficients, public
 * domain), accurate to ~1 ulp, built only from IEEE-exact primitives
 * (+ - * / fmaf sqrtf rintf) so that the HIP kernels, which restate the
 * SAME sequence of operations independently in
 * iou-aware-single-stage-object-detector_amd/csrc/ia_math.hpp, are
 * bit-identical to this file on every input.  Oracle-vs-reference agreement
 * (<= 1e-4) is pinned by tests/golden fixtures generated from the imported
 * reference.
 *
 * Compile with -ffp-contract=off (no implicit fusing) and -mfma (fmaf inline).
 */
#ifndef IA_ORACLE_MATH_H
#define IA_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float ia_o_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t ia_o_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x), fp32.  ~1 ulp.  Denormal results are produced (not flushed). */
static inline float ia_o_expf(float x)
{
    if (x != x) return x;
    if (x > 88.7228394f) return INFINITY;
    if (x < -103.972084f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2;
    int n2 = ni - n1;
    float s1 = ia_o_from_bits((uint32_t)(n1 + 127) << 23);
    float s2 = ia_o_from_bits((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

/* log(x), fp32, x > 0 (0 -> -inf, negative -> nan, denormals handled). ~1 ulp. */
static inline float ia_o_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    uint32_t ix = ia_o_bits(x);
    int e = (int)((ix >> 23) & 0xffu) - 126 + eadj;
    float m = ia_o_from_bits((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e = e - 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* sigmoid as the reference's torch CPU kernel states it: 1 / (1 + exp(-x)). */
static inline float ia_o_sigmoidf(float x)
{
    return 1.0f / (1.0f + ia_o_expf(-x));
}

/* log(1 + exp(-|x|)) : the softplus tail of BCE-with-logits. */
static inline float ia_o_softplus_negabs(float x)
{
    float a = (x < 0.0f) ? x : -x;      /* -|x| */
    float u = ia_o_expf(a);              /* in (0, 1] */
    /* log1p(u): for tiny u the series is exact to fp32; else log(1+u). */
    if (u < 2.44140625e-4f) return fmaf(-0.5f * u, u, u);
    return ia_o_logf(1.0f + u);
}

/* x ** g for x >= 0 the way torch special-cases the exponent
 * (pow(x,2)=x*x, pow(x,1)=x, pow(x,0.5)=sqrt, else exp(g*log x)). */
static inline float ia_o_powf_pos(float x, float g)
{
    if (g == 2.0f) return x * x;
    if (g == 1.0f) return x;
    if (g == 0.5f) return sqrtf(x);
    if (g == 0.0f) return 1.0f;
    if (x == 0.0f) return 0.0f;
    return ia_o_expf(g * ia_o_logf(x));
}

/* exp(x), fp64, ~1 ulp: k = rint(x/ln2), r = x - k ln2 (two-part), Taylor to r^13, 2^k by
 * exponent bits.  Used where the reference calls numpy's float64 exp
 * (soft_nms_cpu.pyx:104, gaussian weights); after the cast to fp32 it agrees with any
 * faithfully rounded fp64 exp except on double-rounding coincidences (~1e-9 per value).
 * Built from IEEE-exact fp64 operations only, restated independently in ia_math.hpp.   */
static inline double ia_o_from_bits64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline double ia_o_exp_f64(double x)
{
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -745.0) return 0.0;
    double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;                  /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;
    double s1 = ia_o_from_bits64((uint64_t)(k1 + 1023) << 52);
    double s2 = ia_o_from_bits64((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

#endif
