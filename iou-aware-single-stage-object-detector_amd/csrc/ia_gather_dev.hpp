// Device pieces of the gather / decode step shared by decode.hip (k_gather, k_gather_nhwc) and
// select.hip (the one-launch decode stage): argument block, delta2bbox on the regenerated anchor,
// per-lane level lookup.
#pragma once
#include "ia_internal.hpp"
#include "ia_math.hpp"

namespace ia {

struct GatherArgs {
    LevelTable t;
    BaseAnchors ba;
    float means[4], stds[4];
    ia_level_ptrs p;
    const int32_t *cand_idx;
    const float *img_hw;
    const float *scale_factor;
    float *boxes;
    float *scores_t;
    float *best_score;       // (B, R) max over classes of the fused score (NMS activity filter)
    int32_t R, Rs, rescale;
};

// delta2bbox of one candidate (reference mmdet/core/bbox/transforms.py:50-76) on the anchor
// regenerated from (level, position, anchor), clamp to img_shape, true division by scale_factor
template <typename AT>
__device__ __forceinline__ float4 decode_box(const AT &a, int b, float ba0, float ba1,
                                             float ba2, float ba3, int W, int stride, int pos,
                                             float r0, float r1, float r2, float r3)
{
    const int y = pos / W, x = pos - y * W;
    const float sx = (float)(x * stride), sy = (float)(y * stride);
    const float ax1 = ba0 + sx, ay1 = ba1 + sy, ax2 = ba2 + sx, ay2 = ba3 + sy;
    const float max_ratio = 4.135166556742356f;
    float dx = r0 * a.stds[0] + a.means[0];
    float dy = r1 * a.stds[1] + a.means[1];
    float dw = r2 * a.stds[2] + a.means[2];
    float dh = r3 * a.stds[3] + a.means[3];
    dw = (dw < -max_ratio) ? -max_ratio : dw;  dw = (dw > max_ratio) ? max_ratio : dw;
    dh = (dh < -max_ratio) ? -max_ratio : dh;  dh = (dh > max_ratio) ? max_ratio : dh;
    float px = (ax1 + ax2) * 0.5f;
    float py = (ay1 + ay2) * 0.5f;
    float pw = (ax2 - ax1) + 1.0f;
    float ph = (ay2 - ay1) + 1.0f;
    float gw = pw * expf_(dw);
    float gh = ph * expf_(dh);
    float gx = px + pw * dx;
    float gy = py + ph * dy;
    float x1 = (gx - gw * 0.5f) + 0.5f;
    float y1 = (gy - gh * 0.5f) + 0.5f;
    float x2 = (gx + gw * 0.5f) - 0.5f;
    float y2 = (gy + gh * 0.5f) - 0.5f;
    const float mx = a.img_hw[2 * b + 1] - 1.0f, my = a.img_hw[2 * b] - 1.0f;
    x1 = (x1 < 0.0f) ? 0.0f : x1;  x1 = (x1 > mx) ? mx : x1;
    y1 = (y1 < 0.0f) ? 0.0f : y1;  y1 = (y1 > my) ? my : y1;
    x2 = (x2 < 0.0f) ? 0.0f : x2;  x2 = (x2 > mx) ? mx : x2;
    y2 = (y2 < 0.0f) ? 0.0f : y2;  y2 = (y2 > my) ? my : y2;
    if (a.rescale) {
        const float *sf = a.scale_factor + 4 * b;
        x1 = x1 / sf[0]; y1 = y1 / sf[1]; x2 = x2 / sf[2]; y2 = y2 / sf[3];
    }
    return make_float4(x1, y1, x2, y2);
}

// Per-lane level lookup without per-lane memory traffic.  Indexing the kernel-argument tables
// with a per-lane level (`a.t.cand_off[l]`, `a.p.cls[l]`, ...) compiles to VECTOR loads from the
// kernarg segment: the old `while (r >= cand_off[l + 1]) ++l` was a chain of up to L dependent
// memory round trips, followed by more for H, W and the three pointers (most of the gather
// kernels' 17-20 us).  With constant indices the tables are scalar loads and the per-lane part is
// compares and selects.
struct LevelSel { int l, H, W, stride; const void *cls, *reg, *iou; };

// MAXL: levels the caller guarantees at most (the per-level scalars of all MAXL levels stay live
// in SGPRs: 8 levels cost 90 SGPRs, which caps the residency at 7 wavefronts per SIMD)
template <int MAXL = IA_MAX_LEVELS, typename AT = GatherArgs>
__device__ __forceinline__ LevelSel level_of_candidate(const AT &a, int r)
{
    LevelSel s;
    s.l = 0;
#pragma unroll
    for (int i = 1; i < MAXL; ++i)
        s.l += (i < a.t.num_levels && r >= a.t.cand_off[i]) ? 1 : 0;
    s.H = a.t.H[0]; s.W = a.t.W[0]; s.stride = a.t.stride[0];
    s.cls = a.p.cls[0]; s.reg = a.p.reg[0]; s.iou = a.p.iou[0];
#pragma unroll
    for (int i = 1; i < MAXL; ++i) {
        const bool m = s.l == i;
        s.H = m ? a.t.H[i] : s.H; s.W = m ? a.t.W[i] : s.W; s.stride = m ? a.t.stride[i] : s.stride;
        s.cls = m ? a.p.cls[i] : s.cls; s.reg = m ? a.p.reg[i] : s.reg; s.iou = m ? a.p.iou[i] : s.iou;
    }
    return s;
}

}  // namespace ia
