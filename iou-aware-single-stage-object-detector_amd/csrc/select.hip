// Per-(image, level) top-k of the row-max scores
// (reference iou_aware_retina_head.py:536-544: `_, topk_inds = max_scores.topk(nms_pre)`).
//
// torch.topk returns indices in descending-score order; the tie order on CPU
// is implementation defined, so the canonical order here is (score
// descending, anchor index ascending) -- identical to the oracle.
//
// Two launches, both built on the workgroup radix select of ia_block.hpp over
// the unique 64-bit key (ordered(score) << 32 | ~anchor_index):
//   k_select_part   every segment is cut into parts of ~10k scores; one
//                   workgroup per part keeps the part's k best keys (sorted).
//                   A single workgroup per 151 200-score segment was
//                   issue-bound inside ONE CU (~60 us per radix pass, 250-360 us
//                   per launch); the parts spread the same work over the chip.
//   k_select_merge  top-k of the union of the parts' survivors == top-k of the
//                   segment; one workgroup per segment, <= 16k keys.
// Levels with N_l <= nms_pre keep their natural order (the reference skips
// topk there, :537).
#include "ia_internal.hpp"
#include "ia_math.hpp"
#include "ia_block.hpp"

namespace ia {

constexpr int kSelectThreads = 1024;
constexpr int kMaxParts = 16;

struct SelectArgs {
    LevelTable t;
    const float *rowmax;
    int32_t *cand_idx;
    uint64_t *part_keys;                       // (B, parts_per_img, kpad)
    int32_t part_off[IA_MAX_LEVELS + 1];       // prefix of parts per level
    int32_t anchors_per_img, cands_per_img, parts_per_img, kpad;
};

__global__ void __launch_bounds__(kSelectThreads) k_select_part(SelectArgs a)
{
    __shared__ TopkScratch sc;
    __shared__ uint64_t sel[IA_MAX_NMS_PRE];
    const int b = blockIdx.y;
    int l = 0;
    while ((int)blockIdx.x >= a.part_off[l + 1]) ++l;
    const uint32_t part = blockIdx.x - a.part_off[l];
    const uint32_t parts = (uint32_t)(a.part_off[l + 1] - a.part_off[l]);
    const uint32_t n = (uint32_t)(a.t.anchor_off[l + 1] - a.t.anchor_off[l]);
    const uint32_t k = (uint32_t)(a.t.cand_off[l + 1] - a.t.cand_off[l]);
    const uint32_t chunk = (n + parts - 1) / parts;
    const uint32_t beg = part * chunk;
    const uint32_t cnt = (beg < n) ? ((n - beg < chunk) ? (n - beg) : chunk) : 0u;
    uint64_t *out = a.part_keys + ((size_t)b * a.parts_per_img + blockIdx.x) * a.kpad;
    const float *src = a.rowmax + (size_t)b * a.anchors_per_img + a.t.anchor_off[l];
    // NCHW heads store the row maxima anchor-major (a, p); the reference's anchor index is
    // p*A + a, which is the storage order itself for channels-last heads
    const uint32_t HW = (uint32_t)(a.t.H[l] * a.t.W[l]), A = (uint32_t)a.t.A;
    const bool natural = a.t.layout == IA_LAYOUT_NHWC;
    auto key = [src, HW, A, beg, natural](uint32_t j) -> uint64_t {
        const uint32_t i = beg + j;
        const uint32_t an = i / HW, p = i - an * HW;
        const uint32_t n = natural ? i : (p * A + an);
        return ((uint64_t)ordered_key(src[i]) << 32) | (uint64_t)(0xffffffffu - n);
    };
    const uint32_t kk = (cnt < k) ? cnt : k;
    if (kk == cnt) {                            // the whole part survives: no selection needed
        for (uint32_t j = threadIdx.x; j < (uint32_t)a.kpad; j += blockDim.x)
            out[j] = (j < cnt) ? key(j) : 0ull;
        return;
    }
    block_topk_desc(key, cnt, kk, sc, sel);
    for (uint32_t j = threadIdx.x; j < (uint32_t)a.kpad; j += blockDim.x)
        out[j] = (j < kk) ? sel[j] : 0ull;
}

__global__ void __launch_bounds__(kSelectThreads) k_select_merge(SelectArgs a)
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
    const uint32_t parts = (uint32_t)(a.part_off[l + 1] - a.part_off[l]);
    const uint64_t *src = a.part_keys + ((size_t)b * a.parts_per_img + a.part_off[l]) * a.kpad;
    // unused slots hold 0, below every real key (ordered(score) of a non-negative score has
    // its top bit set), and there are at least k real keys
    block_topk_desc([src](uint32_t i) -> uint64_t { return src[i]; }, parts * (uint32_t)a.kpad, k,
                    sc, sel);
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x)
        out[i] = (int32_t)(0xffffffffu - (uint32_t)sel[i]);
}

static void plan_parts(const LevelTable &t, SelectArgs &a)
{
    a.part_off[0] = 0;
    int kmax = 1;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        int parts = 0;
        if (l < t.num_levels) {
            const int n = t.anchor_off[l + 1] - t.anchor_off[l];
            const int k = t.cand_off[l + 1] - t.cand_off[l];
            if (k > kmax) kmax = k;
            if (k < n) {
                parts = n / (8 * (k > 1024 ? k : 1024));       // ~8k+ scores per part
                if (parts < 1) parts = 1;
                if (parts > kMaxParts) parts = kMaxParts;
            }
        }
        a.part_off[l + 1] = a.part_off[l] + parts;
    }
    a.parts_per_img = a.part_off[t.num_levels];
    a.kpad = (kmax + 63) / 64 * 64;
}

size_t select_workspace_bytes(const LevelTable &t, int batch)
{
    SelectArgs a;
    plan_parts(t, a);
    return (size_t)batch * (a.parts_per_img > 0 ? a.parts_per_img : 1) * a.kpad * sizeof(uint64_t);
}

int launch_select(const LevelTable &t, const float *rowmax, int batch, int32_t *cand_idx,
                  void *workspace, hipStream_t s)
{
    if (batch < 1 || !rowmax || !cand_idx || !workspace) return IA_E_ARG;
    SelectArgs a;
    a.t = t; a.rowmax = rowmax; a.cand_idx = cand_idx;
    a.part_keys = static_cast<uint64_t *>(workspace);
    a.anchors_per_img = t.anchor_off[t.num_levels];
    a.cands_per_img = t.cand_off[t.num_levels];
    plan_parts(t, a);
    if (a.parts_per_img > 0) {
        hipLaunchKernelGGL(k_select_part, dim3((unsigned)a.parts_per_img, (unsigned)batch),
                           dim3(kSelectThreads), 0, s, a);
        int rc = hip_status(hipGetLastError());
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_select_merge, dim3((unsigned)t.num_levels, (unsigned)batch),
                       dim3(kSelectThreads), 0, s, a);
    return hip_status(hipGetLastError());
}

}  // namespace ia
