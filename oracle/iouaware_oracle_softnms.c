/*
 * TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/README.md).
 *
 * soft-NMS: restatement of mmdet/ops/nms/src/soft_nms_cpu.pyx:22-127 and of
 * multiclass_nms with nms.type='soft_nms' (mmdet/core/post_processing/bbox_nms.py:29-56,
 * mmdet/ops/nms/nms_wrapper.py:52-78).
 *
 * The .pyx is compiled by Cython into C in which the integer literal `1` next to a C float
 * becomes the DOUBLE constant 1.0, so parts of the arithmetic run in fp64 and are rounded to
 * fp32 on assignment to the `cdef float` variables.  The mixed precision is restated literally
 * (checked against the Cython-generated C and, bit for bit, against the reference module
 * itself built by oracle/build_ref.py into oracle/_ref/soft_nms_cpu.so):
 *   area = fl32( (fl64(x2 - x1) + 1.0) * (fl64(y2 - y1) + 1.0) )            (:85)
 *   iw   = fl32( fl64(min(tx2,x2) - max(tx1,x1)) + 1.0 )                    (:86)
 *   ua   = fl32( (fl64(tx2-tx1)+1.0)*(fl64(ty2-ty1)+1.0) + area - fl32(iw*ih) )   (:90)
 *   ov   = fl32(iw*ih) / ua        (fp32 divide)                            (:91)
 *   linear  : weight = fl32(1.0 - fl64(ov)) if ov > iou_thr else 1          (:93-97)
 *   gaussian: weight = fl32( exp_f64( fl64( fl32(-(ov*ov)) / sigma ) ) )    (:99)
 *   score    = fl32(weight) * score   in fp32 (numpy float32 scalar * weak Python float)
 * Positions are emulated exactly (swap of the maximum to slot i, discard by moving the last
 * box into the slot), so ties (first position wins the strict `<` scan, :52-56) resolve like
 * the reference.                                                                            */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "ia_oracle_math.h"

static inline float sn_max(float a, float b) { return (a >= b) ? a : b; }     /* :15-16 */
static inline float sn_min(float a, float b) { return (a <= b) ? a : b; }     /* :18-19 */

/* dets (n,5) -> out_dets (<=n,5) with decayed scores in selection order, out_inds = input
 * index of each output row.  method: 1 linear, 2 gaussian, anything else hard (:100-105).
 * Returns the number of rows kept.                                                       */
int ia_o_soft_nms(const float *dets, int n, float iou_thr, int method, float sigma,
                  float min_score, float *out_dets, int32_t *out_inds)
{
    if (n <= 0) return 0;
    float *b = (float *)malloc(sizeof(float) * 5 * (size_t)n);
    int32_t *inds = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(b, dets, sizeof(float) * 5 * (size_t)n);
    for (int i = 0; i < n; ++i) inds[i] = i;
    int N = n;
    for (int i = 0; i < n; ++i) {                       /* range(N) is evaluated once (:40) */
        if (i >= N) break;                               /* rows >= N are never returned    */
        float maxscore = b[5 * i + 4];
        int maxpos = i;
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < b[5 * pos + 4]) { maxscore = b[5 * pos + 4]; maxpos = pos; }
        float t[5]; int32_t ti = inds[i];
        memcpy(t, b + 5 * i, 20);
        memcpy(b + 5 * i, b + 5 * maxpos, 20); inds[i] = inds[maxpos];
        memcpy(b + 5 * maxpos, t, 20);          inds[maxpos] = ti;
        const float tx1 = b[5 * i], ty1 = b[5 * i + 1], tx2 = b[5 * i + 2], ty2 = b[5 * i + 3];
        int pos = i + 1;
        while (pos < N) {
            float *q = b + 5 * pos;
            const float x1 = q[0], y1 = q[1], x2 = q[2], y2 = q[3];
            const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
            const float iw = (float)((double)(sn_min(tx2, x2) - sn_max(tx1, x1)) + 1.0);
            if (iw > 0.0f) {
                const float ih = (float)((double)(sn_min(ty2, y2) - sn_max(ty1, y1)) + 1.0);
                if (ih > 0.0f) {
                    const float inter = iw * ih;
                    const float ua = (float)(((((double)(tx2 - tx1) + 1.0) *
                                               ((double)(ty2 - ty1) + 1.0)) + (double)area) -
                                             (double)inter);
                    const float ov = inter / ua;
                    float weight;
                    if (method == 1)      weight = (ov > iou_thr) ? (float)(1.0 - (double)ov) : 1.0f;
                    else if (method == 2) weight = (float)ia_o_exp_f64((double)((-(ov * ov)) / sigma));
                    else                  weight = (ov > iou_thr) ? 0.0f : 1.0f;
                    q[4] = weight * q[4];
                    if (q[4] < min_score) {             /* discard: last box moves here (:113-122) */
                        memcpy(q, b + 5 * (N - 1), 20);
                        inds[pos] = inds[N - 1];
                        --N; --pos;
                    }
                }
            }
            ++pos;
        }
    }
    memcpy(out_dets, b, sizeof(float) * 5 * (size_t)N);
    memcpy(out_inds, inds, sizeof(int32_t) * (size_t)N);
    free(b); free(inds);
    return N;
}

typedef struct { float s; int32_t i; } sn_si;
static int sn_cmp(const void *a, const void *b)
{
    const sn_si *x = (const sn_si *)a, *y = (const sn_si *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* multiclass_nms with a soft-NMS op (bbox_nms.py:33-56) on the candidate matrix of one image:
 * bboxes (R,4), scores (R,C) without background column.  Per class the boxes with
 * score > score_thr enter soft-NMS in candidate order; the surviving rows of all classes are
 * concatenated in class order with their DECAYED scores; more than max_per_img -> sort by
 * score descending (canonical tie order: concatenation position ascending).
 * keep_count (C), keep_rows (C,R) selection order, keep_scores (C,R); det_* like
 * ia_o_get_bboxes_single.  Returns the number of detections.                              */
int ia_o_multiclass_soft_nms(const float *bboxes, const float *scores, int R, int C,
                             float score_thr, float iou_thr, int method, float sigma,
                             float min_score, int max_per_img, int32_t *keep_count,
                             int32_t *keep_rows, float *keep_scores, float *det_bboxes,
                             int32_t *det_labels, int32_t *det_rows)
{
    float *cd = (float *)malloc(sizeof(float) * 5 * (size_t)(R > 0 ? R : 1));
    float *od = (float *)malloc(sizeof(float) * 5 * (size_t)(R > 0 ? R : 1));
    int32_t *cr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
    int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
    int total = 0;
    for (int c = 0; c < C; ++c) {
        int n = 0;
        for (int r = 0; r < R; ++r) {
            const float s = scores[(size_t)r * C + c];
            if (s > score_thr) {
                memcpy(cd + 5 * (size_t)n, bboxes + 4 * (size_t)r, 16);
                cd[5 * (size_t)n + 4] = s; cr[n] = r; ++n;
            }
        }
        const int m = ia_o_soft_nms(cd, n, iou_thr, method, sigma, min_score, od, oi);
        keep_count[c] = m;
        for (int i = 0; i < m; ++i) {
            keep_rows[(size_t)c * R + i] = cr[oi[i]];
            keep_scores[(size_t)c * R + i] = od[5 * (size_t)i + 4];
        }
        total += m;
    }
    free(cd); free(od); free(cr); free(oi);
    sn_si *all = (sn_si *)malloc(sizeof(sn_si) * (size_t)(total > 0 ? total : 1));
    int32_t *arow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    int32_t *alab = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    int t = 0;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < keep_count[c]; ++i) {
            all[t].s = keep_scores[(size_t)c * R + i]; all[t].i = t;
            arow[t] = keep_rows[(size_t)c * R + i]; alab[t] = c; ++t;
        }
    int nd = total;
    if (max_per_img >= 0 && total > max_per_img) {
        qsort(all, (size_t)total, sizeof(sn_si), sn_cmp);
        nd = max_per_img;
    }
    for (int d = 0; d < nd; ++d) {
        const int pos = all[d].i, r = arow[pos];
        memcpy(det_bboxes + 5 * (size_t)d, bboxes + 4 * (size_t)r, 16);
        det_bboxes[5 * (size_t)d + 4] = all[d].s;
        det_labels[d] = alab[pos]; det_rows[d] = r;
    }
    free(all); free(arow); free(alab);
    return nd;
}

void ia_o_vec_exp_f64(const double *x, double *y, long long n)
{
    for (long long i = 0; i < n; ++i) y[i] = ia_o_exp_f64(x[i]);
}
