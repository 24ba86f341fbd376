// fp32 elementary functions of the gfx950 kernels.
//
// Built only from IEEE-exact primitives (+ - * / fma sqrt rint, bit casts) in
// a FIXED operation order, so results are reproducible bit-for-bit on any
// IEEE-754 implementation: v_exp_f32 / v_log_f32 / v_rcp_f32 (approximate,
// hardware specific) are deliberately not used.  Compile with
// -ffp-contract=off; hipcc's default correctly-rounded fp32 divide / sqrt is
// required (-fhip-fp32-correctly-rounded-divide-sqrt, on by default).
//
// Accuracy ~1 ulp (Cody-Waite reduction + Cephes minimax polynomials), the
// same class as the Sleef kernels behind the reference's torch CPU ops
// (call sites: reference iou_aware_retina_head.py:505,513,531;
// mmdet/core/bbox/transforms.py:63-64; mmdet/core/loss/losses.py:232-238,478).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ia {

__device__ __forceinline__ float from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float expf_(float x)
{
    if (x != x) return x;
    if (x > 88.7228394f) return __builtin_inff();
    if (x < -103.972084f) return 0.0f;
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, z, r);
    y = y + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2;
    int n2 = ni - n1;
    float s1 = from_bits((uint32_t)(n1 + 127) << 23);
    float s2 = from_bits((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

__device__ __forceinline__ float logf_(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    uint32_t ix = to_bits(x);
    int e = (int)((ix >> 23) & 0xffu) - 126 + eadj;
    float m = from_bits((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e = e - 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, m, -1.1514610310e-1f);
    y = __builtin_fmaf(y, m, 1.1676998740e-1f);
    y = __builtin_fmaf(y, m, -1.2420140846e-1f);
    y = __builtin_fmaf(y, m, 1.4249322787e-1f);
    y = __builtin_fmaf(y, m, -1.6668057665e-1f);
    y = __builtin_fmaf(y, m, 2.0000714765e-1f);
    y = __builtin_fmaf(y, m, -2.4999993993e-1f);
    y = __builtin_fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf_(-x)); }

// sqrt(sigmoid(x)): the per-factor term of the fused detection score
// (reference iou_aware_retina_head.py:531, alpha = 0.5 -> pow(.,0.5) = sqrt).
__device__ __forceinline__ float sqrt_sigmoidf_(float x) { return __builtin_sqrtf(sigmoidf_(x)); }

// log(1 + exp(-|x|))
__device__ __forceinline__ float softplus_negabs_(float x)
{
    float a = (x < 0.0f) ? x : -x;
    float u = expf_(a);
    if (u < 2.44140625e-4f) return __builtin_fmaf(-0.5f * u, u, u);
    return logf_(1.0f + u);
}

// x ** g for x >= 0 with torch's exponent special cases
__device__ __forceinline__ float powf_pos_(float x, float g)
{
    if (g == 2.0f) return x * x;
    if (g == 1.0f) return x;
    if (g == 0.5f) return __builtin_sqrtf(x);
    if (g == 0.0f) return 1.0f;
    if (x == 0.0f) return 0.0f;
    return expf_(g * logf_(x));
}

// BCE-with-logits, target t in [0,1]
__device__ __forceinline__ float bce_logits_(float x, float t)
{
    float mx = (x > 0.0f) ? x : 0.0f;
    return (mx - x * t) + softplus_negabs_(x);
}

// exp in fp64 (soft-NMS gaussian weights: the reference evaluates numpy's float64 exp and
// casts to fp32, soft_nms_cpu.pyx:99).  Range reduction by ln2 in two parts, 13th-order
// series in Horner form with fused multiply-adds, scaling by exponent bits in two steps:
// the same operation sequence as the oracle's, hence the same bits.
__device__ __forceinline__ double exp_f64_(double x)
{
    if (x != x) return x;
    if (x > 709.0) return __builtin_huge_val();
    if (x < -745.0) return 0.0;
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    const double inv_fact[12] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0,
                                 1.0 / 362880.0,    1.0 / 40320.0,    1.0 / 5040.0,
                                 1.0 / 720.0,       1.0 / 120.0,      1.0 / 24.0,
                                 1.0 / 6.0,         0.5,              1.0};
    double p = 1.0 / 6227020800.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) p = __builtin_fma(p, r, inv_fact[i]);
    p = __builtin_fma(p, r, 1.0);
    const int ki = (int)k;
    const int k1 = ki / 2, k2 = ki - k1;
    const double s1 = __builtin_bit_cast(double, (uint64_t)(k1 + 1023) << 52);
    const double s2 = __builtin_bit_cast(double, (uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

// order-preserving map float -> uint32 (larger float <-> larger key)
__device__ __forceinline__ uint32_t ordered_key(float f)
{
    uint32_t u = to_bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_key_inv(uint32_t k)
{
    return from_bits((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// bf16 (raw bits) -> fp32
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return from_bits((uint32_t)h << 16); }

template <typename T> __device__ __forceinline__ float load_f32(const T *p);
template <> __device__ __forceinline__ float load_f32<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float load_f32<uint16_t>(const uint16_t *p) { return bf16_to_f32(*p); }

}  // namespace ia
