// Training-target assignment on the device: MaxIoUAssigner + PseudoSampler +
// bbox2delta + unmap of the reference, for a whole batch in two launches, with the anchors
// regenerated from their index (reference mmdet/core/anchor/anchor_target.py:129-242,
// mmdet/core/bbox/assigners/max_iou_assigner.py:98-201, mmdet/core/bbox/geometry.py:48-64,
// mmdet/core/bbox/transforms.py:6-41).  The reference evaluates a (G x 201 600) IoU matrix per
// image with torch ops and then loops over the gts in Python (one host sync per gt); here the
// IoUs are recomputed where needed instead of stored:
//   k_assign_gtmax   per anchor: IoU with every gt (gt boxes in LDS); per-gt maximum over all
//                    valid anchors via atomicMax on the float bits (IoU >= 0 orders as uint);
//   k_assign_write   per anchor: the assignment rules in the reference's order (negative <
//                    neg_iou_thr, positive >= pos_iou_thr on the arg-max gt, then every gt claims
//                    the anchors that attain its maximum -- later gts overwrite), labels /
//                    weights / encoded deltas, positive and negative counts.
// Outputs are written level-major ((B, N_l) blocks one after the other), i.e. exactly the
// per-level tensors `images_to_levels` builds, so the loss kernels read them without copies.
#include <string.h>
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"

namespace ia {

constexpr int kMaxGt = 512;           // gt boxes per image held in LDS

struct AssignArgs {
    LevelTable t;
    BaseAnchors ba;
    const float *gt_boxes;            // (B, Gmax, 4)
    const int64_t *gt_labels;         // (B, Gmax) or NULL (RPN-style: label 1)
    const int32_t *num_gt;            // (B)
    const int32_t *valid_hw;          // (B, L, 2): valid feature rows / cols per level (pad_shape)
    // by-value variant (ia_anchor_targets_ptrs): per-image pointers and sizes in the arguments
    const float *gt_ptr[IA_MAX_TARGET_BATCH];
    const int64_t *gl_ptr[IA_MAX_TARGET_BATCH];
    int16_t num_gt_v[IA_MAX_TARGET_BATCH];
    int16_t vhw_v[IA_MAX_TARGET_BATCH][IA_MAX_LEVELS][2];
    int32_t by_value;
    uint32_t *gt_max;                 // (B, Gmax) float bits, zero-initialised
    int64_t *labels;                  // level-major (B, N_l) blocks
    float *label_weights;
    float *bbox_targets;              // level-major (B, N_l, 4) blocks
    float *bbox_weights;
    int32_t *counts;                  // (B, 2): positives, negatives (zero-initialised)
    float means[4], stds[4];
    float pos_iou_thr, neg_iou_thr, min_pos_iou, pos_weight;
    int32_t B, Gmax, N;
};

// bbox_overlaps(gt, anchor) of geometry.py:48-64 for one pair (bboxes1 = gt, bboxes2 = anchor)
__device__ __forceinline__ float iou_pair(const float4 &g, float garea, float ax1, float ay1,
                                          float ax2, float ay2, float aarea)
{
    float ltx = (g.x < ax1) ? ax1 : g.x;
    float lty = (g.y < ay1) ? ay1 : g.y;
    float rbx = (ax2 < g.z) ? ax2 : g.z;
    float rby = (ay2 < g.w) ? ay2 : g.w;
    float w = (rbx - ltx) + 1.0f;  w = (w < 0.0f) ? 0.0f : w;
    float h = (rby - lty) + 1.0f;  h = (h < 0.0f) ? 0.0f : h;
    float ov = w * h;
    return ov / ((garea + aarea) - ov);
}

struct AnchorRef { int l, pos, an; bool valid; float x1, y1, x2, y2; size_t out; };

__device__ __forceinline__ AnchorRef locate_anchor(const AssignArgs &a, int b, int n)
{
    AnchorRef r;
    int l = 0;
    while (n >= a.t.anchor_off[l + 1]) ++l;
    const int i = n - a.t.anchor_off[l];
    const int A = a.t.A, W = a.t.W[l];
    r.l = l; r.pos = i / A; r.an = i - r.pos * A;
    const int y = r.pos / W, x = r.pos - y * W;
    const int vh = a.by_value ? (int)a.vhw_v[b][l][0] : a.valid_hw[((size_t)b * a.t.num_levels + l) * 2];
    const int vw = a.by_value ? (int)a.vhw_v[b][l][1] : a.valid_hw[((size_t)b * a.t.num_levels + l) * 2 + 1];
    r.valid = (y < vh) && (x < vw);                   // AnchorGenerator.valid_flags
    const float sx = (float)(x * a.t.stride[l]), sy = (float)(y * a.t.stride[l]);
    const float *ba = a.ba.v[l][r.an];
    r.x1 = ba[0] + sx; r.y1 = ba[1] + sy; r.x2 = ba[2] + sx; r.y2 = ba[3] + sy;
    const size_t nl = (size_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    r.out = (size_t)a.B * a.t.anchor_off[l] + (size_t)b * nl + i;   // level-major block layout
    return r;
}

template <bool WRITE>
__global__ void __launch_bounds__(256) k_assign(AssignArgs a)
{
    __shared__ float4 s_gt[kMaxGt];
    __shared__ float s_area[kMaxGt];
    __shared__ float s_gmax[kMaxGt];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_best[WRITE ? 1 : kMaxGt];     // first pass: the workgroup's per-gt maxima
    const int b = blockIdx.y;
    const int G = a.by_value ? (int)a.num_gt_v[b] : a.num_gt[b];
    const float4 *gsrc = a.by_value ? reinterpret_cast<const float4 *>(a.gt_ptr[b])
                                    : reinterpret_cast<const float4 *>(a.gt_boxes) + (size_t)b * a.Gmax;
    const int64_t *lsrc = a.by_value ? a.gl_ptr[b]
                                     : (a.gt_labels ? a.gt_labels + (size_t)b * a.Gmax : nullptr);
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float4 q = gsrc[g];
        s_gt[g] = q;
        s_area[g] = ((q.z - q.x) + 1.0f) * ((q.w - q.y) + 1.0f);
        if (WRITE) s_gmax[g] = from_bits(a.gt_max[(size_t)b * a.Gmax + g]);
        else s_best[g] = 0u;
    }
    if (WRITE && threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_pos = false, is_neg = false;
    if (n < a.N) {
        const AnchorRef r = locate_anchor(a, b, n);
        const float aarea = ((r.x2 - r.x1) + 1.0f) * ((r.y2 - r.y1) + 1.0f);
        if (!WRITE) {
            if (r.valid)
                for (int g = 0; g < G; ++g) {
                    const float v = iou_pair(s_gt[g], s_area[g], r.x1, r.y1, r.x2, r.y2, aarea);
                    // overlaps.max(dim=1): per-gt maximum over the valid anchors -- reduced in
                    // LDS first (IoU >= 0 orders like its bit pattern), one global atomic per
                    // (workgroup, gt) afterwards instead of one per (anchor, gt)
                    const uint32_t vb = to_bits(v);
                    if (v > 0.0f && vb > s_best[g]) atomicMax(&s_best[g], vb);
                }
        } else {
            int64_t label = 0;
            float lw = 0.0f, bw = 0.0f;
            float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r.valid) {
                float best = -1.0f;
                int arg = 0, claimed = -1;
                for (int g = 0; g < G; ++g) {
                    const float v = iou_pair(s_gt[g], s_area[g], r.x1, r.y1, r.x2, r.y2, aarea);
                    if (v > best) { best = v; arg = g; }                   // first maximum
                    // step 4 (max_iou_assigner.py:183-189): gt g claims its best anchors
                    if (s_gmax[g] >= a.min_pos_iou && v == s_gmax[g]) claimed = g;
                }
                int assigned = -1;                                          // step 1
                if (best >= 0.0f && best < a.neg_iou_thr) assigned = 0;     // step 2
                if (best >= a.pos_iou_thr) assigned = arg + 1;              // step 3
                if (claimed >= 0) assigned = claimed + 1;                   // step 4
                if (assigned > 0) {
                    is_pos = true;
                    const float4 g4 = s_gt[assigned - 1];
                    label = lsrc ? lsrc[assigned - 1] : 1;
                    lw = (a.pos_weight <= 0.0f) ? 1.0f : a.pos_weight;
                    bw = 1.0f;
                    // bbox2delta (transforms.py:21-39)
                    const float px = (r.x1 + r.x2) * 0.5f, py = (r.y1 + r.y2) * 0.5f;
                    const float pw = (r.x2 - r.x1) + 1.0f, ph = (r.y2 - r.y1) + 1.0f;
                    const float gx = (g4.x + g4.z) * 0.5f, gy = (g4.y + g4.w) * 0.5f;
                    const float gw = (g4.z - g4.x) + 1.0f, gh = (g4.w - g4.y) + 1.0f;
                    bt.x = ((gx - px) / pw - a.means[0]) / a.stds[0];
                    bt.y = ((gy - py) / ph - a.means[1]) / a.stds[1];
                    bt.z = (logf_(gw / pw) - a.means[2]) / a.stds[2];
                    bt.w = (logf_(gh / ph) - a.means[3]) / a.stds[3];
                } else if (assigned == 0) {
                    is_neg = true;
                    lw = 1.0f;
                }
            }
            a.labels[r.out] = label;
            a.label_weights[r.out] = lw;
            reinterpret_cast<float4 *>(a.bbox_targets)[r.out] = bt;
            reinterpret_cast<float4 *>(a.bbox_weights)[r.out] = make_float4(bw, bw, bw, bw);
        }
    }
    if (!WRITE) {
        __syncthreads();
        for (int g = threadIdx.x; g < G; g += blockDim.x)
            if (s_best[g]) atomicMax(&a.gt_max[(size_t)b * a.Gmax + g], s_best[g]);
    }
    if (WRITE) {
        const uint64_t mp = __ballot(is_pos), mn = __ballot(is_neg);
        if ((threadIdx.x & 63) == 0) {
            if (mp) atomicAdd(&s_cnt[0], (uint32_t)__builtin_popcountll(mp));
            if (mn) atomicAdd(&s_cnt[1], (uint32_t)__builtin_popcountll(mn));
        }
        __syncthreads();
        if (threadIdx.x < 2 && s_cnt[threadIdx.x])
            atomicAdd(&a.counts[2 * b + threadIdx.x], (int32_t)s_cnt[threadIdx.x]);
    }
}

}  // namespace ia

namespace ia {
static int run_assign(AssignArgs &a, const ia_head_geom *g, int batch, int gmax, float pos_iou_thr,
                      float neg_iou_thr, float min_pos_iou, float pos_weight,
                      uint32_t *gt_max_scratch, int64_t *labels, float *label_weights,
                      float *bbox_targets, float *bbox_weights, int32_t *counts, void *stream)
{
    int rc = make_level_table(g, a.t);
    if (rc) return rc;
    if (batch < 1 || gmax < 1 || gmax > kMaxGt) return IA_E_ARG;
    if (!gt_max_scratch || !labels || !label_weights || !bbox_targets || !bbox_weights || !counts)
        return IA_E_ARG;
    memcpy(a.ba.v, g->base_anchors, sizeof(a.ba.v));
    a.gt_max = gt_max_scratch; a.labels = labels; a.label_weights = label_weights;
    a.bbox_targets = bbox_targets; a.bbox_weights = bbox_weights; a.counts = counts;
    for (int k = 0; k < 4; ++k) { a.means[k] = g->means[k]; a.stds[k] = g->stds[k]; }
    a.pos_iou_thr = pos_iou_thr; a.neg_iou_thr = neg_iou_thr; a.min_pos_iou = min_pos_iou;
    a.pos_weight = pos_weight; a.B = batch; a.Gmax = gmax; a.N = a.t.anchor_off[a.t.num_levels];
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gt_max_scratch, 0, sizeof(uint32_t) * (size_t)batch * gmax, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * (size_t)batch, s);
    if (e != hipSuccess) return (int)e;
    dim3 grid((unsigned)((a.N + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL(k_assign<false>, grid, dim3(256), 0, s, a);
    if ((rc = hip_status(hipGetLastError()))) return rc;
    hipLaunchKernelGGL(k_assign<true>, grid, dim3(256), 0, s, a);
    return hip_status(hipGetLastError());
}
}  // namespace ia

extern "C" int ia_anchor_targets(const ia_head_geom *g, const float *gt_boxes,
                                 const int64_t *gt_labels, const int32_t *num_gt, int batch,
                                 int gmax, const int32_t *valid_hw, float pos_iou_thr,
                                 float neg_iou_thr, float min_pos_iou, float pos_weight,
                                 uint32_t *gt_max_scratch, int64_t *labels, float *label_weights,
                                 float *bbox_targets, float *bbox_weights, int32_t *counts,
                                 void *stream)
{
    if (!gt_boxes || !num_gt || !valid_hw) return IA_E_ARG;
    ia::AssignArgs a;
    memset(&a, 0, sizeof(a));
    a.gt_boxes = gt_boxes; a.gt_labels = gt_labels; a.num_gt = num_gt; a.valid_hw = valid_hw;
    a.by_value = 0;
    return ia::run_assign(a, g, batch, gmax, pos_iou_thr, neg_iou_thr, min_pos_iou, pos_weight,
                          gt_max_scratch, labels, label_weights, bbox_targets, bbox_weights, counts,
                          stream);
}

extern "C" int ia_anchor_targets_ptrs(const ia_head_geom *g, const float *const *gt_boxes,
                                      const int64_t *const *gt_labels, const int32_t *num_gt,
                                      int batch, const int32_t *valid_hw, float pos_iou_thr,
                                      float neg_iou_thr, float min_pos_iou, float pos_weight,
                                      uint32_t *gt_max_scratch, int64_t *labels,
                                      float *label_weights, float *bbox_targets,
                                      float *bbox_weights, int32_t *counts, void *stream)
{
    if (!g || !gt_boxes || !num_gt || !valid_hw || batch < 1 || batch > IA_MAX_TARGET_BATCH)
        return IA_E_ARG;
    if (g->num_levels < 1 || g->num_levels > IA_MAX_LEVELS) return IA_E_ARG;
    ia::AssignArgs a;
    memset(&a, 0, sizeof(a));
    a.by_value = 1;
    int gmax = 1;
    for (int b = 0; b < batch; ++b) {
        if (!gt_boxes[b] || num_gt[b] < 1 || num_gt[b] > ia::kMaxGt) return IA_E_ARG;
        if (((uintptr_t)gt_boxes[b]) & 15u) return IA_E_ARG;          // float4 loads
        a.gt_ptr[b] = gt_boxes[b];
        a.gl_ptr[b] = gt_labels ? gt_labels[b] : nullptr;
        a.num_gt_v[b] = (int16_t)num_gt[b];
        if (num_gt[b] > gmax) gmax = num_gt[b];
        for (int l = 0; l < g->num_levels; ++l)
            for (int k = 0; k < 2; ++k) {
                const int32_t v = valid_hw[((size_t)b * g->num_levels + l) * 2 + k];
                if (v < 0 || v > 32767) return IA_E_ARG;
                a.vhw_v[b][l][k] = (int16_t)v;
            }
    }
    return ia::run_assign(a, g, batch, gmax, pos_iou_thr, neg_iou_thr, min_pos_iou, pos_weight,
                          gt_max_scratch, labels, label_weights, bbox_targets, bbox_weights, counts,
                          stream);
}
