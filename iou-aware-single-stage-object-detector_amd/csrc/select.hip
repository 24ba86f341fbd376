// Per-(image, level) top-k of the row-max scores
// (reference iou_aware_retina_head.py:536-544: `_, topk_inds = max_scores.topk(nms_pre)`).
//
// torch.topk returns indices in descending-score order; the tie order on CPU
// is implementation defined, so the canonical order here is (score
// descending, anchor index ascending) -- identical to the oracle.  One
// workgroup per segment: MSB-first radix select on the 64-bit key
// (ordered(score) << 32 | ~index), wave64 ballot compaction, LDS bitonic sort
// of the k survivors.  Levels with N_l <= nms_pre keep their natural order
// (the reference skips topk there, :537).
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"

namespace ia {

struct SelectArgs {
    LevelTable t;
    const float *rowmax;
    int32_t *cand_idx;
    int32_t anchors_per_img;
    int32_t cands_per_img;
};

constexpr int kSelectThreads = 1024;

__global__ void __launch_bounds__(kSelectThreads) k_select(SelectArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[IA_MAX_NMS_PRE];
    const int l = blockIdx.x, b = blockIdx.y;
    const uint32_t n = (uint32_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    const uint32_t k = (uint32_t)(a.t.cand_off[l + 1] - a.t.cand_off[l]);
    int32_t *out = a.cand_idx + (size_t)b * a.cands_per_img + a.t.cand_off[l];
    if (k == n) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = (int32_t)i;
        return;
    }
    const float *src = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l];
    // storage order is anchor-major (a, p); the reference's anchor index is p*A + a
    const uint32_t HW = (uint32_t)(a.t.H[l] * a.t.W[l]), A = (uint32_t)a.t.A;
    auto key = [src, HW, A](uint32_t i) -> uint64_t {
        const uint32_t an = i / HW, p = i - an * HW;
        return ((uint64_t)ordered_key(src[i]) << 32) | (uint64_t)(0xffffffffu - (p * A + an));
    };
    block_topk_desc(key, n, k, sc, sel);
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x)
        out[i] = (int32_t)(0xffffffffu - (uint32_t)sel[i]);
}

int launch_select(const LevelTable &t, const float *rowmax, int batch, int32_t *cand_idx,
                  hipStream_t s)
{
    if (batch < 1 || !rowmax || !cand_idx) return IA_E_ARG;
    SelectArgs a;
    a.t = t; a.rowmax = rowmax; a.cand_idx = cand_idx;
    a.anchors_per_img = t.anchor_off[t.num_levels];
    a.cands_per_img = t.cand_off[t.num_levels];
    dim3 grid((unsigned)t.num_levels, (unsigned)batch);
    hipLaunchKernelGGL(k_select, grid, dim3(kSelectThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
