// Pieces of the NMS stage shared by nms.hip and lazynms.hip: the exact suppression test.
#pragma once
#include <math.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ia {

struct IouThr {
    double mid;      // midpoint between thr and its fp32 predecessor
    float thr;
    int32_t inclusive;
    int32_t is_half; // thr == 0.5f: fl(inter/uni) >= 0.5  <=>  2*inter >= uni, exact in fp32
};

inline IouThr make_thr(float thr)
{
    IouThr t;
    t.thr = thr;
    float pred = nextafterf(thr, -INFINITY);
    t.mid = ((double)pred + (double)thr) * 0.5;
    uint32_t bits = __builtin_bit_cast(uint32_t, thr);
    t.inclusive = (bits & 1u) == 0u;
    t.is_half = (thr == 0.5f);
    return t;
}

// suppressor box s (earlier in the order, "i" of nms_cpu.cpp:37-55) vs candidate c ("j")
__device__ __forceinline__ bool suppresses(float sx1, float sy1, float sx2, float sy2, float sarea,
                                           float cx1, float cy1, float cx2, float cy2, float carea,
                                           const IouThr &t)
{
    float xx1 = (sx1 < cx1) ? cx1 : sx1;
    float yy1 = (sy1 < cy1) ? cy1 : sy1;
    float xx2 = (cx2 < sx2) ? cx2 : sx2;
    float yy2 = (cy2 < sy2) ? cy2 : sy2;
    float w = (xx2 - xx1) + 1.0f;  w = (0.0f < w) ? w : 0.0f;
    float h = (yy2 - yy1) + 1.0f;  h = (0.0f < h) ? h : 0.0f;
    float inter = w * h;
    float uni = (sarea + carea) - inter;
    if (uni > 0.0f) {
        // thr = 0.5 (every reference config): q = inter/uni rounds to >= 0.5 iff q >= 0.5 - 2^-26,
        // and no two fp32 numbers 2*inter < uni are closer than 2^-24 * uni, so the test is the
        // exact fp32 comparison 2*inter >= uni (2*inter is exact).
        if (t.is_half) return (inter + inter) >= uni;
        double lhs = (double)inter, rhs = t.mid * (double)uni;
        return t.inclusive ? (lhs >= rhs) : (lhs > rhs);
    }
    return (inter / uni) >= t.thr;
}

}  // namespace ia
