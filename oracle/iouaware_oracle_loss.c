/*
 * TEST INFRASTRUCTURE ONLY.  CPU oracle, training-loss half (SURVEY 8a rows
 * T3a, T3b, T3c, T4).  Restates the reference's per-level losses of
 * IoUawareRetinaHead.loss_single (mmdet/models/anchor_heads/
 * iou_aware_retina_head.py:221-313) on the NCHW head outputs directly
 * (row n = b*N_l + p*A + a, column c  <->  element (b, a*C+c, p), the
 * permute(0,2,3,1).reshape of :235,:273,:293).
 *
 * Each function returns the SUM of the elementwise loss in double (the
 * reference divides by avg_factor and multiplies by loss_weight afterwards,
 * losses.py:301-303,411,480) and, when grad != NULL, writes
 * gscale * d(sum)/d(input) elementwise in fp32 with a fixed operation order
 * that the HIP kernels restate bit-for-bit.
 */
#include <stdint.h>
#include <stddef.h>
#include <float.h>
#include <math.h>
#include "ia_oracle_math.h"

/* BCE-with-logits for a (possibly soft) target t in [0,1]:
 * max(x,0) - x*t + log(1+exp(-|x|))  ==  (1-t)*x - log_sigmoid(x)
 * (torch binary_cross_entropy_with_logits; call sites losses.py:236,478).    */
static inline float ia_o_bce_logits(float x, float t)
{
    float mx = (x > 0.0f) ? x : 0.0f;
    return (mx - x * t) + ia_o_softplus_negabs(x);
}

/* T3a  py_sigmoid_focal_loss (mmdet/core/loss/losses.py:226-247) with
 * expand_binary_labels (mmdet/core/anchor/anchor_target.py:247-254):
 * t = (label == c+1); weight = label_weight[n] broadcast over classes.       */
double ia_o_focal_loss(const float *cls, const int64_t *labels, const float *label_weights,
                       int B, int A, int C, int HW, float gamma, float alpha_pos,
                       float alpha_neg, float gscale, float *grad)
{
    double total = 0.0;
    const size_t Nl = (size_t)HW * A;
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a)
            for (int c = 0; c < C; ++c)
                for (int p = 0; p < HW; ++p) {
                    size_t e = (((size_t)b * A + a) * C + c) * HW + p;
                    size_t n = (size_t)b * Nl + (size_t)p * A + a;
                    float x = cls[e];
                    float w0 = label_weights[n];
                    int t = (labels[n] == (int64_t)(c + 1));
                    float pr = ia_o_sigmoidf(x);
                    float pt = t ? (1.0f - pr) : pr;                  /* :233 */
                    float at = (t ? alpha_pos : alpha_neg) * w0;      /* :234 */
                    float mod = ia_o_powf_pos(pt, gamma);
                    float W = at * mod;                               /* :235 */
                    float bce = ia_o_bce_logits(x, t ? 1.0f : 0.0f);  /* :236 */
                    float l = bce * W;
                    total += (double)l;
                    if (grad) {
                        float dbce = pr - (t ? 1.0f : 0.0f);
                        float dpt = pr * (1.0f - pr);
                        dpt = t ? -dpt : dpt;
                        float dmod;
                        if (gamma == 2.0f) dmod = 2.0f * pt;
                        else if (gamma == 1.0f) dmod = 1.0f;
                        else if (gamma == 0.0f) dmod = 0.0f;
                        else dmod = gamma * ia_o_powf_pos(pt, gamma - 1.0f);
                        float g = dbce * W + (bce * at) * (dmod * dpt);
                        grad[e] = g * gscale;
                    }
                }
    return total;
}

/* T3b  weighted_smoothl1 / smooth_l1_loss (losses.py:385-411).
 * pred NCHW (B, A*4, H, W); target / weight (B, N_l, 4) row-major.           */
double ia_o_smooth_l1(const float *pred, const float *target, const float *weight, int B, int A,
                      int HW, float beta, float gscale, float *grad)
{
    double total = 0.0;
    const size_t Nl = (size_t)HW * A;
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a)
            for (int k = 0; k < 4; ++k)
                for (int p = 0; p < HW; ++p) {
                    size_t e = (((size_t)b * A + a) * 4 + k) * HW + p;
                    size_t n = ((size_t)b * Nl + (size_t)p * A + a) * 4 + k;
                    float df = pred[e] - target[n];
                    float d = fabsf(df);
                    float l = (d < beta) ? ((0.5f * d) * d) / beta : d - 0.5f * beta;
                    float w = weight[n];
                    total += (double)(l * w);
                    if (grad) {
                        float s = (df > 0.0f) ? 1.0f : ((df < 0.0f) ? -1.0f : 0.0f);
                        float g = (d < beta) ? df / beta : s;
                        grad[e] = (g * w) * gscale;
                    }
                }
    return total;
}

/* decode without image clamp, keeping what the backward needs               */
typedef struct { float x1, y1, x2, y2, gw, gh, pw, ph; int win, hin; } ia_o_dec;
static ia_o_dec ia_o_decode(const float *anc, const float *d, const float *means, const float *stds)
{
    const float max_ratio = 4.135166556742356f;
    ia_o_dec r;
    float dx = d[0] * stds[0] + means[0];
    float dy = d[1] * stds[1] + means[1];
    float dw = d[2] * stds[2] + means[2];
    float dh = d[3] * stds[3] + means[3];
    r.win = (dw >= -max_ratio) && (dw <= max_ratio);
    r.hin = (dh >= -max_ratio) && (dh <= max_ratio);
    dw = (dw < -max_ratio) ? -max_ratio : dw;  dw = (dw > max_ratio) ? max_ratio : dw;
    dh = (dh < -max_ratio) ? -max_ratio : dh;  dh = (dh > max_ratio) ? max_ratio : dh;
    float px = (anc[0] + anc[2]) * 0.5f;
    float py = (anc[1] + anc[3]) * 0.5f;
    r.pw = (anc[2] - anc[0]) + 1.0f;
    r.ph = (anc[3] - anc[1]) + 1.0f;
    r.gw = r.pw * ia_o_expf(dw);
    r.gh = r.ph * ia_o_expf(dh);
    float gx = px + r.pw * dx;
    float gy = py + r.ph * dy;
    r.x1 = (gx - r.gw * 0.5f) + 0.5f;
    r.y1 = (gy - r.gh * 0.5f) + 0.5f;
    r.x2 = (gx + r.gw * 0.5f) - 0.5f;
    r.y2 = (gy + r.gh * 0.5f) - 0.5f;
    return r;
}

/* T3c  IoU target + weighted_iou_regression_loss
 * (iou_aware_retina_head.py:256-259,276-281; losses.py:460-480;
 *  bbox_overlaps(is_aligned=True) mmdet/core/bbox/geometry.py:34-47).
 * bbox_pred NCHW (B,A*4,H,W); iou_pred NCHW (B,A,H,W); bbox_targets,
 * bbox_weights (B,N_l,4); anchors regenerated from base (A,4) + stride/W.
 * iou_out (B*N_l, optional) receives the IoU targets.
 * grad_iou_pred (NCHW (B,A,H,W), optional); grad_bbox_pred (NCHW, optional):
 * gradient THROUGH the (non-detached) IoU target -- reference :256-259 leaves
 * it attached; under torch >= 1.2 BCE-with-logits differentiates w.r.t. the
 * target (d/dt = -x).                                                        */
double ia_o_iou_bce(const float *bbox_pred, const float *iou_pred, const float *bbox_targets,
                    const float *bbox_weights, const float *base, int B, int A, int H, int W,
                    int stride, const float *means, const float *stds, float gscale,
                    float *iou_out, float *grad_iou_pred, float *grad_bbox_pred)
{
    double total = 0.0;
    const int HW = H * W;
    const size_t Nl = (size_t)HW * A;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p)
            for (int a = 0; a < A; ++a) {
                size_t n = (size_t)b * Nl + (size_t)p * A + a;
                int y = p / W, x = p - y * W;
                float sx = (float)(x * stride), sy = (float)(y * stride);
                float anc[4] = { base[4 * a] + sx, base[4 * a + 1] + sy,
                                 base[4 * a + 2] + sx, base[4 * a + 3] + sy };
                float dp[4];
                for (int k = 0; k < 4; ++k)
                    dp[k] = bbox_pred[(((size_t)b * A + a) * 4 + k) * HW + p];
                ia_o_dec pb = ia_o_decode(anc, dp, means, stds);
                ia_o_dec tb = ia_o_decode(anc, bbox_targets + 4 * n, means, stds);
                /* geometry.py:35-47, bboxes1 = target_box, bboxes2 = pred_box */
                float ltx = (tb.x1 < pb.x1) ? pb.x1 : tb.x1;
                float lty = (tb.y1 < pb.y1) ? pb.y1 : tb.y1;
                float rbx = (pb.x2 < tb.x2) ? pb.x2 : tb.x2;
                float rby = (pb.y2 < tb.y2) ? pb.y2 : tb.y2;
                float w0 = (rbx - ltx) + 1.0f, h0 = (rby - lty) + 1.0f;
                float w = (w0 < 0.0f) ? 0.0f : w0, h = (h0 < 0.0f) ? 0.0f : h0;
                float ov = w * h;
                float a1 = ((tb.x2 - tb.x1) + 1.0f) * ((tb.y2 - tb.y1) + 1.0f);
                float a2 = ((pb.x2 - pb.x1) + 1.0f) * ((pb.y2 - pb.y1) + 1.0f);
                float un = (a1 + a2) - ov;
                float t = ov / un;
                if (iou_out) iou_out[n] = t;
                float xl = iou_pred[((size_t)b * A + a) * HW + p];
                float wt = bbox_weights[4 * n];
                float l = ia_o_bce_logits(xl, t) * wt;
                total += (double)l;
                if (grad_iou_pred)
                    grad_iou_pred[((size_t)b * A + a) * HW + p] =
                        ((ia_o_sigmoidf(xl) - t) * wt) * gscale;
                if (grad_bbox_pred) {
                    float gt = ((-xl) * wt) * gscale;                /* dL/dt */
                    float inv_un = 1.0f / un;
                    float g_ov = gt * ((un + ov) * inv_un) * inv_un; /* dt/dov (un depends on ov) */
                    float g_a2 = gt * (-(ov * inv_un) * inv_un);
                    float g_w = (w0 >= 0.0f) ? g_ov * h : 0.0f;
                    float g_h = (h0 >= 0.0f) ? g_ov * w : 0.0f;
                    /* d/d pred box corners */
                    float pw2 = (pb.x2 - pb.x1) + 1.0f, ph2 = (pb.y2 - pb.y1) + 1.0f;
                    float gx1 = -g_a2 * ph2, gx2 = g_a2 * ph2;
                    float gy1 = -g_a2 * pw2, gy2 = g_a2 * pw2;
                    /* lt = max(tb, pb): grad to pb where pb > tb (0.5 on ties); w = rb - lt + 1 */
                    float sx1 = (pb.x1 > tb.x1) ? 1.0f : ((pb.x1 == tb.x1) ? 0.5f : 0.0f);
                    float sy1 = (pb.y1 > tb.y1) ? 1.0f : ((pb.y1 == tb.y1) ? 0.5f : 0.0f);
                    float sx2 = (pb.x2 < tb.x2) ? 1.0f : ((pb.x2 == tb.x2) ? 0.5f : 0.0f);
                    float sy2 = (pb.y2 < tb.y2) ? 1.0f : ((pb.y2 == tb.y2) ? 0.5f : 0.0f);
                    gx1 = gx1 - g_w * sx1;  gx2 = gx2 + g_w * sx2;
                    gy1 = gy1 - g_h * sy1;  gy2 = gy2 + g_h * sy2;
                    /* corners -> deltas: x1 = gx - gw/2 + .5, x2 = gx + gw/2 - .5 */
                    float g_gx = gx1 + gx2, g_gy = gy1 + gy2;
                    float g_gw = (gx2 - gx1) * 0.5f, g_gh = (gy2 - gy1) * 0.5f;
                    float gd[4];
                    gd[0] = (g_gx * pb.pw) * stds[0];
                    gd[1] = (g_gy * pb.ph) * stds[1];
                    gd[2] = pb.win ? (g_gw * pb.gw) * stds[2] : 0.0f;
                    gd[3] = pb.hin ? (g_gh * pb.gh) * stds[3] : 0.0f;
                    for (int k = 0; k < 4; ++k)
                        grad_bbox_pred[(((size_t)b * A + a) * 4 + k) * HW + p] = gd[k];
                }
            }
    return total;
}

/* T4  SigmoidFocalLossForward / Backward of the reference's CUDA op
 * (mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:23-63,65-105):
 * integer targets (0 = background, 1..C classes, negative = ignored by c2),
 * no per-anchor weight, elementwise output (N,C).                            */
void ia_o_focal_loss_op_fwd(const float *logits, const int64_t *targets, int N, int C,
                            float gamma, float alpha, float *losses)
{
    for (int n = 0; n < N; ++n)
        for (int d = 0; d < C; ++d) {
            size_t i = (size_t)n * C + d;
            int t = (int)targets[n];
            float c1 = (t == d + 1) ? 1.0f : 0.0f;
            float c2 = ((t >= 0) & (t != d + 1)) ? 1.0f : 0.0f;
            float zn = 1.0f - alpha, zp = alpha;
            float x = logits[i];
            float p = 1.0f / (1.0f + ia_o_expf(-x));
            float pm = (p > FLT_MIN) ? p : FLT_MIN;
            float term1 = ia_o_powf_pos(1.0f - p, gamma) * ia_o_logf(pm);
            float xs = (x >= 0.0f) ? x : 0.0f;                 /* x * (x >= 0) */
            float term2 = ia_o_powf_pos(p, gamma) *
                          ((-1.0f * xs) - ia_o_logf(1.0f + ia_o_expf(x - 2.0f * xs)));
            float l = 0.0f;
            l += -c1 * term1 * zp;
            l += -c2 * term2 * zn;
            losses[i] = l;
        }
}

void ia_o_focal_loss_op_bwd(const float *logits, const int64_t *targets, const float *d_losses,
                            int N, int C, float gamma, float alpha, float *d_logits)
{
    for (int n = 0; n < N; ++n)
        for (int d = 0; d < C; ++d) {
            size_t i = (size_t)n * C + d;
            int t = (int)targets[n];
            float c1 = (t == d + 1) ? 1.0f : 0.0f;
            float c2 = ((t >= 0) & (t != d + 1)) ? 1.0f : 0.0f;
            float zn = 1.0f - alpha, zp = alpha;
            float x = logits[i];
            float p = 1.0f / (1.0f + ia_o_expf(-x));
            float pm = (p > FLT_MIN) ? p : FLT_MIN;
            float term1 = ia_o_powf_pos(1.0f - p, gamma) *
                          ((1.0f - p) - (p * gamma) * ia_o_logf(pm));
            float xs = (x >= 0.0f) ? x : 0.0f;
            float term2 = ia_o_powf_pos(p, gamma) *
                          ((((-1.0f * xs) - ia_o_logf(1.0f + ia_o_expf(x - 2.0f * xs))) *
                            (1.0f - p)) * gamma - p);
            float g = 0.0f;
            g += -c1 * term1 * zp;
            g += -c2 * term2 * zn;
            d_logits[i] = g * d_losses[i];
        }
}

/* ------------------------------------------------------------------------- */
/* the IoU-balanced loss variants (selected by loss_cls.type =
 * 'IOUbalancedSigmoidFocalLoss' / loss_bbox.type = 'IoUbalancedSmoothL1Loss',
 * mmdet/models/anchor_heads/anchor_head.py:83-84; the target configs keep them in comments).  */

static float ia_o_focal_elem(float x, int t, float w0, float gamma, float alpha_pos,
                             float alpha_neg, float *g_out)
{
    float pr = ia_o_sigmoidf(x);
    float pt = t ? (1.0f - pr) : pr;
    float at = (t ? alpha_pos : alpha_neg) * w0;
    float mod = ia_o_powf_pos(pt, gamma);
    float W = at * mod;
    float bce = ia_o_bce_logits(x, t ? 1.0f : 0.0f);
    if (g_out) {
        float dbce = pr - (t ? 1.0f : 0.0f);
        float dpt = pr * (1.0f - pr);
        dpt = t ? -dpt : dpt;
        float dmod;
        if (gamma == 2.0f) dmod = 2.0f * pt;
        else if (gamma == 1.0f) dmod = 1.0f;
        else if (gamma == 0.0f) dmod = 0.0f;
        else dmod = gamma * ia_o_powf_pos(pt, gamma - 1.0f);
        *g_out = dbce * W + (bce * at) * (dmod * dpt);
    }
    return bce * W;
}

/* iou_balanced_sigmoid_focal_loss (losses.py:309-374, branch IoU_balanced_Cls = True):
 *   loss1 = focal elementwise;  w = (1 - t) + (t * iou)^eta;  normalizer = sum(loss1 t) /
 *   (sum(loss1 w t) + 1e-6);  loss = sum(loss1 * ((1 - t) + (t iou)^eta * normalizer)), the
 *   weights detached (:356).  Only the one positive element of a positive anchor has t = 1.
 * iou: (B*N_l) per anchor (the IoU regression target of loss_single :256-259).
 * sums3 = { sum over t=0 elements, S1 = sum_pos loss1, S2 = sum_pos loss1 iou^eta }.
 * Returns S0 + normalizer * S2.                                                          */
double ia_o_focal_loss_balanced(const float *cls, const int64_t *labels,
                                const float *label_weights, const float *iou, int B, int A,
                                int C, int HW, float gamma, float alpha_pos, float alpha_neg,
                                float eta, float gscale, float *grad, double *sums3)
{
    double S0 = 0.0, S1 = 0.0, S2 = 0.0;
    const size_t Nl = (size_t)HW * A;
    for (int pass = 0; pass < 2; ++pass) {
        float normalizer = 0.0f;
        if (pass == 1) {
            normalizer = (float)S1 / ((float)S2 + 1e-6f);
            if (!grad) break;
        }
        for (int b = 0; b < B; ++b)
            for (int a = 0; a < A; ++a)
                for (int c = 0; c < C; ++c)
                    for (int p = 0; p < HW; ++p) {
                        size_t e = (((size_t)b * A + a) * C + c) * HW + p;
                        size_t n = (size_t)b * Nl + (size_t)p * A + a;
                        int t = (labels[n] == (int64_t)(c + 1));
                        float g;
                        float l = ia_o_focal_elem(cls[e], t, label_weights[n], gamma, alpha_pos,
                                                  alpha_neg, pass ? &g : NULL);
                        if (pass == 0) {
                            if (t) { S1 += (double)l; S2 += (double)(l * ia_o_powf_pos(iou[n], eta)); }
                            else S0 += (double)l;
                        } else {
                            float w = t ? ia_o_powf_pos(iou[n], eta) * normalizer : 1.0f;
                            grad[e] = (g * w) * gscale;
                        }
                    }
    }
    if (sums3) { sums3[0] = S0; sums3[1] = S1; sums3[2] = S2; }
    return S0 + (double)((float)S1 / ((float)S2 + 1e-6f)) * S2;
}

/* weighted_iou_balanced_smoothl1 (losses.py:416-458): smooth-L1 with the per-anchor weight
 * weight * iou^delta (detached).                                                          */
double ia_o_smooth_l1_balanced(const float *pred, const float *target, const float *weight,
                               const float *iou, int B, int A, int HW, float beta, float delta,
                               float gscale, float *grad)
{
    double total = 0.0;
    const size_t Nl = (size_t)HW * A;
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a)
            for (int k = 0; k < 4; ++k)
                for (int p = 0; p < HW; ++p) {
                    size_t e = (((size_t)b * A + a) * 4 + k) * HW + p;
                    size_t r = (size_t)b * Nl + (size_t)p * A + a;
                    size_t n = r * 4 + k;
                    float df = pred[e] - target[n];
                    float d = fabsf(df);
                    float l = (d < beta) ? ((0.5f * d) * d) / beta : d - 0.5f * beta;
                    float w = weight[n] * ia_o_powf_pos(iou[r], delta);           /* :447 */
                    total += (double)(l * w);
                    if (grad) {
                        float s = (df > 0.0f) ? 1.0f : ((df < 0.0f) ? -1.0f : 0.0f);
                        float g = (d < beta) ? df / beta : s;
                        grad[e] = (g * w) * gscale;
                    }
                }
    return total;
}
