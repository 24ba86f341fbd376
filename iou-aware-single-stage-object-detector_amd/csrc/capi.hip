// C-ABI entry points of the inference half (see include/iouaware.h) and the
// whole-path driver ia_get_bboxes: five launches on the caller's stream, no
// host synchronisation, no allocation.
#include <string.h>
#include <atomic>
#include <mutex>
#include "ia_internal.hpp"
#include "ia_math.hpp"

namespace ia {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void base_from_geom(const ia_head_geom *g, BaseAnchors &ba)
{
    memcpy(ba.v, g->base_anchors, sizeof(ba.v));
}

struct WsLayout { size_t off[11]; size_t total; int32_t N, R, Rs; };

static int ws_layout(const ia_head_geom *g, int batch, WsLayout &w)
{
    LevelTable t;
    int rc = make_level_table(g, t);
    if (rc) return rc;
    if (batch < 1) return IA_E_ARG;
    w.N = t.anchor_off[t.num_levels];
    w.R = t.cand_off[t.num_levels];
    if (w.R > IA_MAX_CANDIDATES) return IA_E_LIMIT_BOXES;
    w.Rs = (w.R + 63) / 64 * 64;
    size_t o = 0;
    const size_t B = (size_t)batch, C = (size_t)t.C;
    w.off[0] = o; o = align_up(o + B * w.N * sizeof(float), 256);            // rowmax
    w.off[1] = o; o = align_up(o + B * w.R * sizeof(int32_t), 256);          // cand_idx
    w.off[2] = o; o = align_up(o + B * w.R * 4 * sizeof(float), 256);        // boxes
    w.off[3] = o; o = align_up(o + B * C * w.Rs * sizeof(float), 256);       // scores_t
    w.off[4] = o; o = align_up(o + B * C * sizeof(int32_t), 256);            // keep_count
    w.off[5] = o; o = align_up(o + B * C * w.Rs * sizeof(int32_t), 256);     // keep_rows
    w.off[6] = o; o = align_up(o + B * w.R * sizeof(float), 256);            // best_score
    size_t noff[3];
    w.off[7] = o; o = align_up(o + nms_workspace_bytes(batch, w.R, t.C, noff), 256);   // NMS stage
    o = align_up(o + finalize_workspace_bytes(batch, w.Rs, t.C), 256);                 // + final keys
    w.off[8] = o; o = align_up(o + select_workspace_bytes(t, batch), 256);             // top-k parts
    w.off[9] = o; o = align_up(o + lazy_workspace_bytes(batch, w.Rs, t.C), 256);       // lazy NMS
    w.off[10] = o; o = align_up(o + B * sizeof(int32_t), 256);                         // need_full
    w.total = o;
    return 0;
}

__global__ void k_test_math(int op, const float *x, const float *y, float *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i], r;
        switch (op) {
        case 0: r = expf_(v); break;
        case 1: r = logf_(v); break;
        case 2: r = sigmoidf_(v); break;
        case 3: r = __builtin_sqrtf(v); break;
        case 4: r = v / y[i]; break;
        case 6: r = (float)exp_f64_((double)(v / y[i])); break;
        default: r = sqrt_sigmoidf_(v); break;
        }
        out[i] = r;
    }
}

}  // namespace ia

extern "C" {

const char *ia_version(void) { return "iouaware-hip 0.1 (gfx950)"; }

int ia_geom_sizes(const ia_head_geom *g, int32_t *N, int32_t *R, int32_t *Rs)
{
    // pure geometry: valid for any head, also when R exceeds what ia_get_bboxes supports
    // (the training losses use a geometry with nms_pre <= 0, i.e. R = N)
    ia::LevelTable t;
    int rc = ia::make_level_table(g, t);
    if (rc) return rc;
    const int32_t r = t.cand_off[t.num_levels];
    if (N) *N = t.anchor_off[t.num_levels];
    if (R) *R = r;
    if (Rs) *Rs = (r + 63) / 64 * 64;
    return 0;
}

int ia_decode_fuse_rowmax(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                          float *rowmax, void *stream)
{
    ia::LevelTable t;
    int rc = ia::make_level_table(g, t);
    if (rc) return rc;
    if (!p) return IA_E_ARG;
    return ia::launch_rowmax(t, *p, batch, dtype, rowmax, (hipStream_t)stream);
}

size_t ia_select_topk_workspace_bytes(const ia_head_geom *g, int batch)
{
    ia::LevelTable t;
    if (ia::make_level_table(g, t) || batch < 1) return 0;
    return ia::select_workspace_bytes(t, batch);
}

int ia_select_topk(const ia_head_geom *g, const float *rowmax, int batch, int32_t *cand_idx,
                   void *workspace, size_t workspace_bytes, void *stream)
{
    ia::LevelTable t;
    int rc = ia::make_level_table(g, t);
    if (rc) return rc;
    if (batch < 1 || !workspace) return IA_E_ARG;
    if (workspace_bytes < ia::select_workspace_bytes(t, batch)) return IA_E_WORKSPACE;
    return ia::launch_select(t, rowmax, batch, cand_idx, workspace, (hipStream_t)stream);
}

int ia_decode_fuse_rowmax_grouped(const ia_head_geom *g, const ia_level_ptrs *p, int batch,
                                  int dtype, float *rowmax, void *select_workspace,
                                  size_t workspace_bytes, void *stream)
{
    ia::LevelTable t;
    int rc = ia::make_level_table(g, t);
    if (rc) return rc;
    if (!p || batch < 1 || !select_workspace) return IA_E_ARG;
    if (workspace_bytes < ia::select_workspace_bytes(t, batch)) return IA_E_WORKSPACE;
    float *groupmax = ia::select_workspace_groupmax(t, batch, select_workspace);
    rc = ia::launch_rowmax(t, *p, batch, dtype, rowmax, (hipStream_t)stream, groupmax);
    if (!rc && t.softmax)       // the softmax row-score kernel does not emit the group maxima
        rc = ia::launch_groupmax(t, rowmax, batch, select_workspace, (hipStream_t)stream);
    return rc;
}

int ia_select_topk_grouped(const ia_head_geom *g, const float *rowmax, int batch,
                           int32_t *cand_idx, void *select_workspace, size_t workspace_bytes,
                           void *stream)
{
    ia::LevelTable t;
    int rc = ia::make_level_table(g, t);
    if (rc) return rc;
    if (batch < 1 || !select_workspace) return IA_E_ARG;
    if (workspace_bytes < ia::select_workspace_bytes(t, batch)) return IA_E_WORKSPACE;
    return ia::launch_select(t, rowmax, batch, cand_idx, select_workspace, (hipStream_t)stream,
                             true);
}

int ia_gather_decode(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                     const int32_t *cand_idx, const float *img_hw, const float *scale_factor,
                     int rescale, float *boxes, float *scores_t, float *best_score, void *stream)
{
    // pure geometry like the row-max and top-k entries: any number of candidates per image (the
    // IA_MAX_CANDIDATES capacity belongs to the batched NMS behind ia_get_bboxes, not to this stage)
    ia::LevelTable t;
    int rc = ia::make_level_table(g, t);
    if (rc) return rc;
    if (!p || batch < 1) return IA_E_ARG;
    const int Rs = (t.cand_off[t.num_levels] + 63) / 64 * 64;
    ia::BaseAnchors ba;
    ia::base_from_geom(g, ba);
    return ia::launch_gather(t, ba, g->means, g->stds, *p, batch, dtype, cand_idx, img_hw,
                             scale_factor, rescale, boxes, scores_t, best_score, Rs,
                             (hipStream_t)stream);
}

size_t ia_multiclass_nms_workspace_bytes(int batch, int R, int C)
{
    if (batch < 1 || R < 1 || R > IA_MAX_CANDIDATES || C < 1) return 0;
    size_t off[3];
    const int Rs = (R + 63) / 64 * 64;
    return (ia::nms_workspace_bytes(batch, R, C, off) + 255) / 256 * 256 +
           ia::finalize_workspace_bytes(batch, Rs, C);
}

int ia_multiclass_nms(const float *boxes, const float *scores_t, const float *best_score,
                      int batch, int R, int C, float score_thr, float iou_thr, int max_per_img,
                      void *workspace, size_t workspace_bytes, float *dets, int32_t *labels,
                      int32_t *rows, int32_t *num, int32_t *keep_count, int32_t *keep_rows,
                      void *stream)
{
    const int Rs = (R + 63) / 64 * 64;
    if (!workspace) return IA_E_ARG;
    if (workspace_bytes < ia_multiclass_nms_workspace_bytes(batch, R, C) || workspace_bytes == 0)
        return IA_E_WORKSPACE;
    int rc = ia::launch_nms(boxes, scores_t, best_score, batch, R, Rs, C, score_thr, iou_thr,
                            workspace, keep_count, keep_rows, (hipStream_t)stream);
    if (rc) return rc;
    size_t off[3];
    char *fin_ws = static_cast<char *>(workspace) +
                   (ia::nms_workspace_bytes(batch, R, C, off) + 255) / 256 * 256;
    return ia::launch_finalize(boxes, scores_t, keep_count, keep_rows, batch, R, Rs, C,
                               max_per_img, fin_ws, dets, labels, rows, num, (hipStream_t)stream);
}

// stage-level lazy NMS: lazy walk first, the complete path (gated) for the images it cannot finish
size_t ia_multiclass_nms_lazy_workspace_bytes(int batch, int R, int C)
{
    const size_t full = ia_multiclass_nms_workspace_bytes(batch, R, C);
    if (full == 0) return 0;
    const int Rs = (R + 63) / 64 * 64;
    return ia::align_up(full, 256) + ia::align_up(ia::lazy_workspace_bytes(batch, Rs, C), 256) +
           ia::align_up((size_t)batch * sizeof(int32_t), 256) +                      // need_full
           ia::align_up((size_t)batch * C * sizeof(int32_t), 256) +                  // keep_count
           ia::align_up((size_t)batch * C * Rs * sizeof(int32_t), 256);              // keep_rows
}

int ia_multiclass_nms_lazy(const float *boxes, const float *scores_t, const float *best_score,
                           int batch, int R, int C, float score_thr, float iou_thr,
                           int max_per_img, int candidates, void *workspace,
                           size_t workspace_bytes, float *dets, int32_t *labels, int32_t *rows,
                           int32_t *num, void *stream)
{
    const int Rs = (R + 63) / 64 * 64;
    if (!workspace || candidates < 0) return IA_E_ARG;
    const size_t need = ia_multiclass_nms_lazy_workspace_bytes(batch, R, C);
    if (need == 0 || workspace_bytes < need) return IA_E_WORKSPACE;
    char *ws = static_cast<char *>(workspace);
    void *full_ws = ws;
    ws += ia::align_up(ia_multiclass_nms_workspace_bytes(batch, R, C), 256);
    void *lazy_ws = ws;
    ws += ia::align_up(ia::lazy_workspace_bytes(batch, Rs, C), 256);
    int32_t *need_full = reinterpret_cast<int32_t *>(ws);
    ws += ia::align_up((size_t)batch * sizeof(int32_t), 256);
    int32_t *kc = reinterpret_cast<int32_t *>(ws);
    ws += ia::align_up((size_t)batch * C * sizeof(int32_t), 256);
    int32_t *kr = reinterpret_cast<int32_t *>(ws);
    hipStream_t s = (hipStream_t)stream;
    int rc = ia::launch_lazy_nms(boxes, scores_t, batch, R, Rs, C, score_thr, iou_thr, max_per_img,
                                 candidates, lazy_ws, dets, labels, rows, num, need_full, s);
    if (rc) return rc;
    if ((rc = ia::launch_nms(boxes, scores_t, best_score, batch, R, Rs, C, score_thr, iou_thr,
                             full_ws, kc, kr, s, need_full)))
        return rc;
    size_t off[3];
    char *fin_ws = static_cast<char *>(full_ws) +
                   (ia::nms_workspace_bytes(batch, R, C, off) + 255) / 256 * 256;
    return ia::launch_finalize(boxes, scores_t, kc, kr, batch, R, Rs, C, max_per_img, fin_ws, dets,
                               labels, rows, num, s, need_full);
}

size_t ia_multiclass_soft_nms_workspace_bytes(int batch, int R, int C)
{
    if (batch < 1 || R < 1 || R > IA_MAX_CANDIDATES || C < 1) return 0;
    const int Rs = (R + 63) / 64 * 64;
    return ia::align_up((size_t)batch * C * Rs * sizeof(float), 256) +
           ia::finalize_workspace_bytes(batch, Rs, C);
}

int ia_multiclass_soft_nms(const float *boxes, const float *scores_t, int batch, int R, int C,
                           float score_thr, float iou_thr, int method, float sigma,
                           float min_score, int max_per_img, void *workspace,
                           size_t workspace_bytes, float *dets, int32_t *labels, int32_t *rows,
                           int32_t *num, int32_t *keep_count, int32_t *keep_rows, void *stream)
{
    const int Rs = (R + 63) / 64 * 64;
    if (!workspace) return IA_E_ARG;
    if (workspace_bytes < ia_multiclass_soft_nms_workspace_bytes(batch, R, C) || workspace_bytes == 0)
        return IA_E_WORKSPACE;
    float *soft = static_cast<float *>(workspace);
    int rc = ia::launch_soft_nms(boxes, scores_t, batch, R, Rs, C, score_thr, iou_thr, method,
                                 sigma, min_score, keep_count, keep_rows, soft,
                                 (hipStream_t)stream);
    if (rc) return rc;
    char *fin_ws = static_cast<char *>(workspace) +
                   ia::align_up((size_t)batch * C * Rs * sizeof(float), 256);
    return ia::launch_finalize(boxes, soft, keep_count, keep_rows, batch, R, Rs, C, max_per_img,
                               fin_ws, dets, labels, rows, num, (hipStream_t)stream);
}

int ia_soft_nms(const float *dets, int n, float iou_thr, int method, float sigma, float min_score,
                float *out_dets, int32_t *out_inds, int32_t *count, void *stream)
{
    return ia::launch_soft_nms_single(dets, n, iou_thr, method, sigma, min_score, out_dets,
                                      out_inds, count, (hipStream_t)stream);
}

size_t ia_get_bboxes_workspace_bytes(const ia_head_geom *g, int batch)
{
    ia::WsLayout w;
    if (ia::ws_layout(g, batch, w)) return 0;
    return w.total;
}

size_t ia_get_bboxes_status_offset(const ia_head_geom *g, int batch)
{
    ia::WsLayout w;
    if (ia::ws_layout(g, batch, w)) return 0;
    ia::LevelTable t;
    ia::make_level_table(g, t);
    return w.off[8] + ia::select_workspace_status_offset(t, batch);
}

int ia_get_bboxes_workspace_layout(const ia_head_geom *g, int batch, size_t offsets[8])
{
    ia::WsLayout w;
    int rc = ia::ws_layout(g, batch, w);
    if (rc) return rc;
    for (int i = 0; i < 8; ++i) offsets[i] = w.off[i];
    return 0;
}

// ia_profile_stage_events: a pair of caller-owned HIP events recorded on the call's stream right
// before the stage's first launch and right behind its last one (bench.py: the decode stage timed
// INSIDE the steps of the timed region).  Both null (the default): nothing is recorded.
// The hook is scoped to ONE stream (the stream current when it was installed is not known to the
// library, so the first stage call after installation binds it; ia_profile_stage_events(NULL,
// NULL) unbinds): calls on other streams / from other models in the process do not record into the
// same events.  The caller clears the hook before releasing the events.
// (events, binding flag and bound stream were four separate atomics -- a caller on another
// stream could compare with a stale stream, or see the old begin event with the new end event.  One small
// mutex around the whole record: the hook is read once per stage call, ~20 ns uncontended.)
static std::mutex g_stage_mu;
static struct { void *ev[2]; void *stream; bool bound; } g_stage = {{nullptr, nullptr}, nullptr, false};

static int decode_stage_launches(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                                 const float *img_hw, const float *scale_factor, int rescale,
                                 char *ws, hipStream_t s, ia::WsLayout &w, ia::LevelTable &t);

static int decode_stage_impl(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                             const float *img_hw, const float *scale_factor, int rescale,
                             void *workspace, size_t workspace_bytes, hipStream_t s,
                             ia::WsLayout &w, ia::LevelTable &t)
{
    int rc = ia::ws_layout(g, batch, w);
    if (rc) return rc;
    if (!p || !workspace) return IA_E_ARG;
    if (workspace_bytes < w.total) return IA_E_WORKSPACE;
    if (((uintptr_t)workspace & 255u) != 0) return IA_E_ARG;
    char *ws = static_cast<char *>(workspace);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        if (g_stage.ev[0]) {
            if (!g_stage.bound) { g_stage.bound = true; g_stage.stream = (void *)s; }
            if (g_stage.stream == (void *)s) { e0 = (hipEvent_t)g_stage.ev[0]; e1 = (hipEvent_t)g_stage.ev[1]; }
        }                                                           // else: another stream's call records nothing
    }
    if (e0 && (rc = ia::hip_status(hipEventRecord(e0, s)))) return rc;
    rc = decode_stage_launches(g, p, batch, dtype, img_hw, scale_factor, rescale, ws, s, w, t);
    if (!rc && e1) rc = ia::hip_status(hipEventRecord(e1, s));
    return rc;
}

static int decode_stage_launches(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                                 const float *img_hw, const float *scale_factor, int rescale,
                                 char *ws, hipStream_t s, ia::WsLayout &w, ia::LevelTable &t)
{
    int rc;
    float *rowmax = reinterpret_cast<float *>(ws + w.off[0]);
    int32_t *cand = reinterpret_cast<int32_t *>(ws + w.off[1]);
    float *boxes = reinterpret_cast<float *>(ws + w.off[2]);
    float *scores_t = reinterpret_cast<float *>(ws + w.off[3]);
    float *best = reinterpret_cast<float *>(ws + w.off[6]);
    ia::make_level_table(g, t);
    ia::BaseAnchors ba;
    ia::base_from_geom(g, ba);
    // row-max (+ group maxima) and the top-k's filter in one launch where the layout allows
    if ((rc = ia::launch_rowmax_select(t, *p, batch, dtype, rowmax, cand, ws + w.off[8], s))) return rc;
    return ia::launch_gather(t, ba, g->means, g->stds, *p, batch, dtype, cand, img_hw, scale_factor,
                             rescale, boxes, scores_t, best, w.Rs, s);
}

// lazy_candidates < 0: the complete NMS for every image (keep_count / keep_rows of all classes
// are produced); >= 0: lazy NMS first (0 = default candidate count), complete path only for the
// images it could not finish
static int get_bboxes_impl(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                           const float *img_hw, const float *scale_factor, int rescale,
                           float score_thr, float iou_thr, int max_per_img, int lazy_candidates,
                           void *workspace, size_t workspace_bytes, float *dets, int32_t *labels,
                           int32_t *rows, int32_t *num, void *stream)
{
    ia::WsLayout w;
    ia::LevelTable t;
    hipStream_t s = (hipStream_t)stream;
    int rc = decode_stage_impl(g, p, batch, dtype, img_hw, scale_factor, rescale, workspace,
                               workspace_bytes, s, w, t);
    if (rc) return rc;
    char *ws = static_cast<char *>(workspace);
    float *boxes = reinterpret_cast<float *>(ws + w.off[2]);
    float *scores_t = reinterpret_cast<float *>(ws + w.off[3]);
    int32_t *kc = reinterpret_cast<int32_t *>(ws + w.off[4]);
    int32_t *kr = reinterpret_cast<int32_t *>(ws + w.off[5]);
    float *best = reinterpret_cast<float *>(ws + w.off[6]);
    void *nms_ws = ws + w.off[7];
    const int32_t *gate = nullptr;
    if (lazy_candidates >= 0) {
        int32_t *need_full = reinterpret_cast<int32_t *>(ws + w.off[10]);
        if ((rc = ia::launch_lazy_nms(boxes, scores_t, batch, w.R, w.Rs, t.C, score_thr, iou_thr,
                                      max_per_img, lazy_candidates, ws + w.off[9], dets, labels,
                                      rows, num, need_full, s)))
            return rc;
        gate = need_full;
    }
    if ((rc = ia::launch_nms(boxes, scores_t, best, batch, w.R, w.Rs, t.C, score_thr, iou_thr,
                             nms_ws, kc, kr, s, gate)))
        return rc;
    size_t noff[3];
    char *fin_ws = static_cast<char *>(nms_ws) +
                   (ia::nms_workspace_bytes(batch, w.R, t.C, noff) + 255) / 256 * 256;
    return ia::launch_finalize(boxes, scores_t, kc, kr, batch, w.R, w.Rs, t.C, max_per_img, fin_ws,
                               dets, labels, rows, num, s, gate);
}

int ia_profile_stage_events(void *begin, void *end)
{
    if ((begin == nullptr) != (end == nullptr)) return IA_E_ARG;
    std::lock_guard<std::mutex> lk(g_stage_mu);
    g_stage.ev[0] = begin; g_stage.ev[1] = end;
    g_stage.bound = false; g_stage.stream = nullptr;                // the next stage call binds its stream
    return 0;
}

int ia_decode_stage(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                    const float *img_hw, const float *scale_factor, int rescale, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    ia::WsLayout w;
    ia::LevelTable t;
    return decode_stage_impl(g, p, batch, dtype, img_hw, scale_factor, rescale, workspace,
                             workspace_bytes, (hipStream_t)stream, w, t);
}

int ia_get_bboxes(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                  const float *img_hw, const float *scale_factor, int rescale, float score_thr,
                  float iou_thr, int max_per_img, void *workspace, size_t workspace_bytes,
                  float *dets, int32_t *labels, int32_t *rows, int32_t *num, void *stream)
{
    return get_bboxes_impl(g, p, batch, dtype, img_hw, scale_factor, rescale, score_thr, iou_thr,
                           max_per_img, -1, workspace, workspace_bytes, dets, labels, rows, num,
                           stream);
}

int ia_get_bboxes_lazy(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                       const float *img_hw, const float *scale_factor, int rescale,
                       float score_thr, float iou_thr, int max_per_img, int candidates,
                       void *workspace, size_t workspace_bytes, float *dets, int32_t *labels,
                       int32_t *rows, int32_t *num, void *stream)
{
    if (candidates < 0) return IA_E_ARG;
    return get_bboxes_impl(g, p, batch, dtype, img_hw, scale_factor, rescale, score_thr, iou_thr,
                           max_per_img, candidates, workspace, workspace_bytes, dets, labels, rows,
                           num, stream);
}

size_t ia_nms_workspace_bytes(int n)
{
    return n > IA_MAX_CANDIDATES ? ia::nms_big_workspace_bytes(n) : ia::nms_single_workspace_bytes(n);
}

int ia_nms(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count, void *workspace,
           size_t workspace_bytes, void *stream)
{
    if (n > IA_MAX_CANDIDATES)              // no size limit, like nms_cpu.cpp: chunked (bignms.hip)
        return ia::launch_nms_big(dets, n, iou_thr, keep, count, workspace, workspace_bytes,
                                  (hipStream_t)stream);
    return ia::launch_nms_single(dets, n, iou_thr, keep, count, workspace, workspace_bytes,
                                 (hipStream_t)stream);
}

int ia_test_math(int op, const float *x, const float *y, float *out, int64_t n, void *stream)
{
    if (n < 0 || op < 0 || op > 6) return IA_E_ARG;
    if (n == 0) return 0;
    if (!x || !out || ((op == 4 || op == 6) && !y)) return IA_E_ARG;
    int64_t blocks = (n + 255) / 256;
    unsigned grid = (unsigned)(blocks > 8192 ? 8192 : blocks);
    hipLaunchKernelGGL(ia::k_test_math, dim3(grid), dim3(256), 0, (hipStream_t)stream, op, x, y,
                       out, n);
    return ia::hip_status(hipGetLastError());
}

}  // extern "C"
