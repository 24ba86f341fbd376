// Internal declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/iouaware.h"

namespace ia {

// per-level scalars used by kernels that do not need the base anchors
struct LevelTable {
    int32_t num_levels, A, C, nms_pre, layout;
    int32_t softmax;                         // IA_CLS_SOFTMAX: class tensors carry C + 1 channels per anchor
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS], stride[IA_MAX_LEVELS];
    int32_t anchor_off[IA_MAX_LEVELS + 1];   // prefix of N_l   (anchors per image)
    int32_t cand_off[IA_MAX_LEVELS + 1];     // prefix of k_l   (candidates per image)
    int32_t tile_off[IA_MAX_LEVELS + 1];     // prefix of ceil(HW_l / 256) row-max tiles
};

struct BaseAnchors { float v[IA_MAX_LEVELS][IA_MAX_ANCHORS][4]; };

inline int make_level_table(const ia_head_geom *g, LevelTable &t)
{
    if (!g) return IA_E_ARG;
    if (g->num_levels < 1 || g->num_levels > IA_MAX_LEVELS) return IA_E_ARG;
    if (g->num_anchors < 1 || g->num_anchors > IA_MAX_ANCHORS) return IA_E_ARG;
    if (g->num_classes < 1 || g->num_classes > 4096) return IA_E_ARG;
    if (g->nms_pre > IA_MAX_NMS_PRE) return IA_E_LIMIT_NMS_PRE;
    if (g->layout != IA_LAYOUT_NCHW && g->layout != IA_LAYOUT_NHWC) return IA_E_ARG;
    if (g->cls_activation != IA_CLS_SIGMOID && g->cls_activation != IA_CLS_SOFTMAX) return IA_E_ARG;
    t.layout = g->layout;
    t.softmax = g->cls_activation == IA_CLS_SOFTMAX ? 1 : 0;
    t.num_levels = g->num_levels; t.A = g->num_anchors; t.C = g->num_classes; t.nms_pre = g->nms_pre;
    t.anchor_off[0] = t.cand_off[0] = t.tile_off[0] = 0;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        bool on = l < g->num_levels;
        if (on && (g->H[l] < 1 || g->W[l] < 1 || g->stride[l] < 1)) return IA_E_ARG;
        t.H[l] = on ? g->H[l] : 0; t.W[l] = on ? g->W[l] : 0; t.stride[l] = on ? g->stride[l] : 0;
        int64_t hw = (int64_t)t.H[l] * t.W[l];
        int64_t nl = hw * t.A;
        if (nl > (1 << 30)) return IA_E_ARG;
        int32_t kl = (g->nms_pre > 0 && nl > g->nms_pre) ? g->nms_pre : (int32_t)nl;
        t.anchor_off[l + 1] = t.anchor_off[l] + (int32_t)nl;
        t.cand_off[l + 1] = t.cand_off[l] + kl;
        t.tile_off[l + 1] = t.tile_off[l] + (int32_t)((hw + 255) / 256);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Per-level top-k plan shared by the row-max kernels (which emit group maxima) and select.hip.
//
// Levels with more than kSelDenseMax anchors are FILTERED before the exact selection: the anchors
// of a segment (image, level) are partitioned into groups of g = 64 / 16 / 4 consecutive stored
// scores; with v = the k-th largest group maximum, k DISTINCT scores are >= v, so the segment's
// k-th largest score T is >= v and every top-k member is >= v: everything below v is dropped
// without looking at it again (for independent scores about 1.25 k candidates survive).  Any
// subset of the groups gives a valid (smaller) bound, so groups that straddle two images are
// simply left out.  Smaller levels go to the exact selection whole.
constexpr int kSelDenseMax = 12288;     // also the candidates the final kernel stages in LDS
constexpr int kSelChunk = 4096;         // scores per filter workgroup

struct SelPlan {
    int32_t grp[IA_MAX_LEVELS];          // group size (0: the level is not filtered)
    int32_t goff[IA_MAX_LEVELS + 1];     // prefix of the per-level group-maximum arrays (floats)
    int32_t chunk_off[IA_MAX_LEVELS + 1];// prefix of ceil(N_l / kSelChunk) over filtered levels
};

// group arrays: channels-last heads: PER IMAGE, groups of g consecutive reference anchor indices
// (ceil(N_l / g) words per image: no group, and no 64-row unit of the row-max wavefronts,
// straddles two images -- the state of a segment (image, level) can then be reset by that segment's
// own consumers); NCHW heads: per (image, anchor) plane, groups of g consecutive positions (the
// row-max array is stored anchor-major there)
inline int64_t sel_groups_per_image(const LevelTable &t, int l, int g)
{
    const int64_t n = t.anchor_off[l + 1] - t.anchor_off[l];
    return (n + g - 1) / g;
}
inline int64_t sel_units_per_image(const LevelTable &t, int l)      // 64-row units of the row-max wavefronts
{
    return ((int64_t)(t.anchor_off[l + 1] - t.anchor_off[l]) + 63) / 64;
}
inline int64_t sel_group_count(const LevelTable &t, int l, int g, int batch)
{
    if (t.layout == IA_LAYOUT_NHWC) return (int64_t)batch * sel_groups_per_image(t, l, g);
    const int64_t hw = (int64_t)t.H[l] * t.W[l];
    return (int64_t)batch * t.A * ((hw + g - 1) / g);
}

inline int make_sel_plan(const LevelTable &t, int batch, SelPlan &p)
{
    p.goff[0] = 0; p.chunk_off[0] = 0;
    for (int l = 0; l < IA_MAX_LEVELS; ++l) {
        int g = 0, chunks = 0;
        int64_t groups = 0;
        if (l < t.num_levels) {
            const int n = t.anchor_off[l + 1] - t.anchor_off[l];
            const int k = t.cand_off[l + 1] - t.cand_off[l];
            if (k < n && n > kSelDenseMax) {
                // the coarsest grouping that still leaves about 2k groups per segment
                g = 4;
                for (int c : {64, 16}) if ((int64_t)n / c >= 2 * (int64_t)k + 2) { g = c; break; }
                groups = sel_group_count(t, l, g, batch);
                chunks = (n + kSelChunk - 1) / kSelChunk;
            }
        }
        if (p.goff[l] + groups > 2147483647LL) return IA_E_ARG;
        p.grp[l] = g;
        p.goff[l + 1] = p.goff[l] + (int32_t)groups;
        p.chunk_off[l + 1] = p.chunk_off[l] + chunks;
    }
    return 0;
}

// stage launchers (defined in the .hip files)
// groupmax: when given, the row-max kernel also writes the group maxima of make_sel_plan's
// filtered levels -- the first step of launch_select, folded into the streaming kernel
int launch_rowmax(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype, float *rowmax,
                  hipStream_t s, float *groupmax = nullptr);
size_t select_workspace_bytes(const LevelTable &t, int batch);
// where launch_rowmax has to put the group maxima inside the select workspace
float *select_workspace_groupmax(const LevelTable &t, int batch, void *workspace);
// group maxima of the filtered levels derived from a complete row-max array (what launch_select does
// itself when have_groups is false); for producers that do not emit them (softmax row scores)
int launch_groupmax(const LevelTable &t, const float *rowmax, int batch, void *workspace, hipStream_t s);
// byte offset (inside the select workspace) of the fused launch's status word: 0 = fine,
// 1 = a filter workgroup gave up waiting, 2 = arrival counters found above their maximum
size_t select_workspace_status_offset(const LevelTable &t, int batch);
// have_groups: the row-max kernel already filled the workspace's group maxima; otherwise an
// extra pass over the row-max array derives them first
int launch_select(const LevelTable &t, const float *rowmax, int batch, int32_t *cand_idx,
                  void *workspace, hipStream_t s, bool have_groups = false);
// row-max + top-k the way ia_get_bboxes chains them; channels-last heads: row-max and the top-k
// filter in ONE launch (select.hip, k_rowmax_filter_nhwc).  The select workspace must have been
// zeroed once by its owner.
int launch_rowmax_select(const LevelTable &t, const ia_level_ptrs &p, int batch, int dtype,
                         float *rowmax, int32_t *cand_idx, void *workspace, hipStream_t s);
int launch_gather(const LevelTable &t, const BaseAnchors &ba, const float *means, const float *stds,
                  const ia_level_ptrs &p, int batch, int dtype, const int32_t *cand_idx,
                  const float *img_hw, const float *scale_factor, int rescale, float *boxes,
                  float *scores_t, float *best_score, int Rs, hipStream_t s);
int nms_adj_words(int R);            // 64-bit words per adjacency row (padded)
size_t nms_workspace_bytes(int batch, int R, int C, size_t off[3]);
int launch_nms(const float *boxes, const float *scores_t, const float *best_score, int batch,
               int R, int Rs, int C, float score_thr, float iou_thr, void *workspace,
               int32_t *keep_count, int32_t *keep_rows, hipStream_t s,
               const int32_t *gate = nullptr);       // gate (B): images with gate[b] == 0 are skipped
size_t finalize_workspace_bytes(int batch, int Rs, int C);
int launch_finalize(const float *boxes, const float *scores_t, const int32_t *keep_count,
                    const int32_t *keep_rows, int batch, int R, int Rs, int C, int max_per_img,
                    void *workspace, float *dets, int32_t *labels, int32_t *rows, int32_t *num,
                    hipStream_t s, const int32_t *gate = nullptr);
// lazy NMS (lazynms.hip): the detections straight from the globally best (class, box) pairs
size_t lazy_workspace_bytes(int batch, int Rs, int C);
int launch_lazy_nms(const float *boxes, const float *scores_t, int batch, int R, int Rs, int C,
                    float score_thr, float iou_thr, int max_per_img, int candidates,
                    void *workspace, float *dets, int32_t *labels, int32_t *rows, int32_t *num,
                    int32_t *need_full, hipStream_t s);
size_t nms_single_workspace_bytes(int n);
// more than IA_MAX_CANDIDATES boxes: chunked NMS (bignms.hip), same result
size_t nms_big_workspace_bytes(int n);
int launch_nms_big(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count,
                   void *workspace, size_t workspace_bytes, hipStream_t s);
int launch_nms_single(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count,
                      void *workspace, size_t workspace_bytes, hipStream_t s);

inline int hip_status(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

int launch_soft_nms(const float *boxes, const float *scores_t, int batch, int R, int Rs, int C,
                    float score_thr, float iou_thr, int method, float sigma, float min_score,
                    int32_t *keep_count, int32_t *keep_rows, float *soft_scores, hipStream_t s);
int launch_soft_nms_single(const float *dets, int n, float iou_thr, int method, float sigma,
                           float min_score, float *out_dets, int32_t *out_inds, int32_t *count,
                           hipStream_t s);
}  // namespace ia
