// Streaming stages of the inference head path (HBM-bound):
//
//   k_rowmax  -- reads the (B, A*C, H, W) class logits and (B, A, H, W) IoU
//                logits of every level ONCE, coalesced along W, and writes one
//                fused row-max score per anchor
//                (reference iou_aware_retina_head.py:502-513,528-531,539).
//   k_gather  -- for the selected candidates only: regenerate the anchor from
//                its index, delta2bbox, clamp, rescale; emit the 80 fused
//                scores class-major (reference :545-558, transforms.py:44-78,
//                anchor_generator.py:53-70).
//
// Layout notes (gfx950): a wavefront owns one anchor `a` and a run of 256
// consecutive positions of one channel plane, so every global load
// instruction is a contiguous 1 KiB (fp32, 16 B per lane) segment; the A
// waves of a workgroup share the position tile and transpose their results
// through LDS so the (position-major, anchor-minor) row-max array is written
// as one contiguous block.  sqrt(sigmoid(.)) is monotone non-decreasing in
// fp32 (exhaustively checked by tests/test_oracle_math.py), so the max over
// classes is taken on the raw logits and the transcendental part runs once
// per anchor instead of once per class.
#include "ia_internal.hpp"
#include "ia_math.hpp"

namespace ia {

struct RowmaxArgs {
    LevelTable t;
    ia_level_ptrs p;
    float *rowmax;
    int32_t tiles_per_img;
    int32_t anchors_per_img;
};

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    using type = float4;
    static __device__ __forceinline__ void load(const float *p, float (&v)[4])
    {
        float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
};
template <> struct Vec4<uint16_t> {
    using type = ushort4;
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[4])
    {
        ushort4 q = *reinterpret_cast<const ushort4 *>(p);
        v[0] = bf16_to_f32(q.x); v[1] = bf16_to_f32(q.y);
        v[2] = bf16_to_f32(q.z); v[3] = bf16_to_f32(q.w);
    }
};

constexpr int kTile = 256;   // positions per workgroup

template <typename T>
__global__ void __launch_bounds__(1024) k_rowmax(RowmaxArgs a)
{
    extern __shared__ float tile[];             // kTile * A
    const int lane = threadIdx.x;               // 0..63
    const int an = threadIdx.y;                 // anchor owned by this wave
    const int A = a.t.A, C = a.t.C;
    const int b = blockIdx.x / a.tiles_per_img;
    const int rem = blockIdx.x - b * a.tiles_per_img;
    int l = 0;
    while (rem >= a.t.tile_off[l + 1]) ++l;
    const int HW = a.t.H[l] * a.t.W[l];
    const int p0 = (rem - a.t.tile_off[l]) * kTile;
    const T *cls = static_cast<const T *>(a.p.cls[l]) + ((size_t)b * A + an) * C * HW;
    const T *iou = static_cast<const T *>(a.p.iou[l]) + ((size_t)b * A + an) * HW;

    const float ninf = -__builtin_inff();
    float m[4] = {ninf, ninf, ninf, ninf};
    float il[4] = {0.f, 0.f, 0.f, 0.f};
    int slot[4];
    bool ok[4];
    if ((HW & 3) == 0) {
        // 4 consecutive positions per lane: one 16 B (fp32) / 8 B (bf16) load per class
        const int pp = p0 + lane * 4;
        const bool in = pp < HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) { slot[j] = lane * 4 + j; ok[j] = in; }
        if (in) {
            const T *src = cls + pp;
#pragma unroll 8
            for (int c = 0; c < C; ++c) {
                float v[4];
                Vec4<T>::load(src + (size_t)c * HW, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = (m[j] < v[j]) ? v[j] : m[j];
            }
            Vec4<T>::load(iou + pp, il);
        }
    } else {
        // plane base only 4 B aligned: 4 strided positions per lane, each load
        // instruction still covers 64 consecutive positions
#pragma unroll
        for (int j = 0; j < 4; ++j) { slot[j] = lane + 64 * j; ok[j] = (p0 + slot[j]) < HW; }
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const T *src = cls + (size_t)c * HW + p0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ok[j]) {
                    float v = load_f32<T>(src + slot[j]);
                    m[j] = (m[j] < v) ? v : m[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ok[j]) il[j] = load_f32<T>(iou + p0 + slot[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ok[j]) tile[slot[j] * A + an] = sqrt_sigmoidf_(m[j]) * sqrt_sigmoidf_(il[j]);
    __syncthreads();
    const int npos = (HW - p0 < kTile) ? (HW - p0) : kTile;
    const int cnt = npos * A;
    float *out = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l] + (size_t)p0 * A;
    for (int i = an * 64 + lane; i < cnt; i += 64 * A) out[i] = tile[i];
}

int launch_rowmax(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype, float *rowmax,
                  hipStream_t s)
{
    if (batch < 1 || !rowmax) return IA_E_ARG;
    RowmaxArgs a;
    a.t = t; a.p = p; a.rowmax = rowmax;
    a.tiles_per_img = t.tile_off[t.num_levels];
    a.anchors_per_img = t.anchor_off[t.num_levels];
    dim3 block(64, t.A);
    dim3 grid((unsigned)(a.tiles_per_img * batch));
    size_t lds = sizeof(float) * kTile * t.A;
    if (dtype == IA_F32) hipLaunchKernelGGL(k_rowmax<float>, grid, block, lds, s, a);
    else if (dtype == IA_BF16) hipLaunchKernelGGL(k_rowmax<uint16_t>, grid, block, lds, s, a);
    else return IA_E_ARG;
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------
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
    int32_t R, Rs, rescale;
};

constexpr int kGroups = 4;   // class groups per candidate (threadIdx.y)

template <typename T>
__global__ void __launch_bounds__(64 * kGroups) k_gather(GatherArgs a)
{
    const int lane = threadIdx.x, grp = threadIdx.y;
    const int b = blockIdx.y;
    const int r = blockIdx.x * 64 + lane;
    if (r >= a.R) return;
    const int A = a.t.A, C = a.t.C;
    int l = 0;
    while (r >= a.t.cand_off[l + 1]) ++l;
    const int W = a.t.W[l], HW = a.t.H[l] * W;
    const int idx = a.cand_idx[(size_t)b * a.R + r];
    const int pos = idx / A, an = idx - pos * A;
    const T *cls = static_cast<const T *>(a.p.cls[l]) + ((size_t)b * A + an) * C * HW + pos;
    const T *iou = static_cast<const T *>(a.p.iou[l]) + ((size_t)b * A + an) * HW + pos;
    const float sq_iou = sqrt_sigmoidf_(load_f32<T>(iou));
    const int cpg = (C + kGroups - 1) / kGroups;
    const int c0 = grp * cpg;
    const int c1 = (c0 + cpg < C) ? (c0 + cpg) : C;
    float *so = a.scores_t + (size_t)b * C * a.Rs + r;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        float x = load_f32<T>(cls + (size_t)c * HW);
        so[(size_t)c * a.Rs] = sqrt_sigmoidf_(x) * sq_iou;
    }
    if (grp == 0) {
        const T *reg = static_cast<const T *>(a.p.reg[l]) + ((size_t)b * A + an) * 4 * HW + pos;
        const int y = pos / W, x = pos - y * W;
        const float sx = (float)(x * a.t.stride[l]), sy = (float)(y * a.t.stride[l]);
        const float *ba = a.ba.v[l][an];
        const float ax1 = ba[0] + sx, ay1 = ba[1] + sy, ax2 = ba[2] + sx, ay2 = ba[3] + sy;
        // delta2bbox, reference mmdet/core/bbox/transforms.py:50-76
        const float max_ratio = 4.135166556742356f;
        float dx = load_f32<T>(reg) * a.stds[0] + a.means[0];
        float dy = load_f32<T>(reg + (size_t)HW) * a.stds[1] + a.means[1];
        float dw = load_f32<T>(reg + (size_t)2 * HW) * a.stds[2] + a.means[2];
        float dh = load_f32<T>(reg + (size_t)3 * HW) * a.stds[3] + a.means[3];
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
        reinterpret_cast<float4 *>(a.boxes)[(size_t)b * a.R + r] = make_float4(x1, y1, x2, y2);
    }
}

int launch_gather(const LevelTable &t, const BaseAnchors &ba, const float *means, const float *stds,
                  const ia_level_ptrs &p, int batch, int dtype, const int32_t *cand_idx,
                  const float *img_hw, const float *scale_factor, int rescale, float *boxes,
                  float *scores_t, int Rs, hipStream_t s)
{
    if (batch < 1 || !cand_idx || !img_hw || !boxes || !scores_t) return IA_E_ARG;
    if (rescale && !scale_factor) return IA_E_ARG;
    GatherArgs a;
    a.t = t; a.ba = ba; a.p = p;
    for (int i = 0; i < 4; ++i) { a.means[i] = means[i]; a.stds[i] = stds[i]; }
    a.cand_idx = cand_idx; a.img_hw = img_hw; a.scale_factor = scale_factor;
    a.boxes = boxes; a.scores_t = scores_t;
    a.R = t.cand_off[t.num_levels]; a.Rs = Rs; a.rescale = rescale;
    dim3 block(64, kGroups);
    dim3 grid((unsigned)((a.R + 63) / 64), (unsigned)batch);
    if (dtype == IA_F32) hipLaunchKernelGGL(k_gather<float>, grid, block, 0, s, a);
    else if (dtype == IA_BF16) hipLaunchKernelGGL(k_gather<uint16_t>, grid, block, 0, s, a);
    else return IA_E_ARG;
    return hip_status(hipGetLastError());
}

}  // namespace ia
