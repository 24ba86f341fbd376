/*
 * iouaware.h -- C-ABI of the MI355X-native (gfx950) IoU-aware RetinaNet
 * post-conv hot path.  Shared library: libiouaware_hip.so (built by
 * iou-aware-single-stage-object-detector_amd/csrc/build.py with hipcc).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / ATen types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host or the
 *     comment says "host"; `stream` is a hipStream_t passed as void*.
 *   - nothing is allocated: callers pass a workspace sized by the matching
 *     *_workspace_bytes() query.  Calls are asynchronous on `stream`.
 *   - return value: 0 = ok, negative = IA_E_* argument error, positive =
 *     hipError_t of a failed launch.
 *   - feature maps are (B, ch, H, W) tensors, dtype IA_F32 or IA_BF16, channel
 *     a*C+c (cls), a*4+k (reg), a (iou) -- the reference head's output
 *     (reference mmdet/models/anchor_heads/iou_aware_retina_head.py:171-219) -- stored
 *     contiguously either NCHW or channels-last (memory (B, H, W, ch), what MIOpen's
 *     NHWC convolutions write: exactly the reference's permute(0,2,3,1) view, :502-507);
 *     ia_head_geom.layout says which.  The training kernels take NCHW only.
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the reference repository root).
 */
#ifndef IOUAWARE_H
#define IOUAWARE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IA_MAX_LEVELS 8
#define IA_MAX_ANCHORS 16
#define IA_MAX_NMS_PRE 4096
#define IA_MAX_CANDIDATES 8192     /* sum over levels of min(nms_pre, N_l) */
#define IA_MAX_PER_IMG 1024

#define IA_F32 0
#define IA_BF16 1
#define IA_F16 2      /* the dtype-generic operator entries only (ia_sigmoid_focal_loss_*_dt) */
#define IA_F64 3      /* ia_sigmoid_focal_loss_*_dt, ia_nms_f64 */

#define IA_LAYOUT_NCHW 0
#define IA_LAYOUT_NHWC 1     /* needs C * sizeof(dtype) to be a multiple of 16, <= 512 bytes */

#define IA_CLS_SIGMOID 0
#define IA_CLS_SOFTMAX 1

#define IA_LOSS_SLOTS 64     /* partial sums written by the *_fwd loss kernels */

#define IA_E_ARG (-1)        /* invalid argument / unsupported size */
#define IA_E_WORKSPACE (-2)  /* workspace too small */
/* a size above what the kernels are built for; the reference ops have no such limits
 * (mmdet/ops/nms/src/nms_cpu.cpp:4-59 takes any n) -- the caller can tell them apart: */
#define IA_E_LIMIT_BOXES (-3)    /* more than IA_MAX_CANDIDATES boxes / candidates in one call */
#define IA_E_LIMIT_NMS_PRE (-4)  /* nms_pre above IA_MAX_NMS_PRE */
#define IA_E_LIMIT_PER_IMG (-5)  /* max_per_img above IA_MAX_PER_IMG */

/* Static geometry of one head for one padded input size. */
typedef struct ia_head_geom {
    int32_t num_levels;                 /* L <= IA_MAX_LEVELS */
    int32_t num_anchors;                /* A <= IA_MAX_ANCHORS */
    int32_t num_classes;                /* C, foreground classes (80): columns of scores_t / labels */
    int32_t nms_pre;                    /* <= 0: no per-level top-k */
    int32_t H[IA_MAX_LEVELS];
    int32_t W[IA_MAX_LEVELS];
    int32_t stride[IA_MAX_LEVELS];
    float base_anchors[IA_MAX_LEVELS][IA_MAX_ANCHORS][4];  /* AnchorGenerator.base_anchors */
    float means[4], stds[4];            /* target_means / target_stds */
    int32_t layout;                     /* IA_LAYOUT_*: memory order of the head outputs */
    /* classification activation (iou_aware_retina_head.py:504-507,538-541).
     * IA_CLS_SIGMOID: cls has A*C channels, score_c = sqrt(sigmoid(x_c)) * sqrt(sigmoid(iou)).
     * IA_CLS_SOFTMAX (use_sigmoid_cls=False): cls has A*(C+1) channels, channel 0 of an anchor is
     * the background; score_c = sqrt(softmax(x)_{c+1}) * sqrt(sigmoid(iou)) for c = 0..C-1 and the
     * row maximum runs over the foreground columns only (scores[:, 1:].max).  Everything behind the
     * gather (NMS, labels) is the same.  Covers the inference entries (row-max, top-k, gather,
     * ia_get_bboxes*, ia_decode_stage); the loss entries are sigmoid-only.                       */
    int32_t cls_activation;
} ia_head_geom;

/* Per-level device pointers of the three head outputs, each (B, ch, H, W). */
typedef struct ia_level_ptrs {
    const void *cls[IA_MAX_LEVELS];
    const void *reg[IA_MAX_LEVELS];
    const void *iou[IA_MAX_LEVELS];
} ia_level_ptrs;

/* Derived sizes (host helper): N = anchors per image, R = candidates per image,
 * Rs = R rounded up to 64 (row stride of class-major score / keep arrays).   */
int ia_geom_sizes(const ia_head_geom *g, int32_t *N, int32_t *R, int32_t *Rs);

const char *ia_version(void);

/* ------------------------------------------------------------------ inference
 * Stage entry points.  Together they replace IoUawareRetinaHead.get_bboxes /
 * get_bboxes_single (iou_aware_retina_head.py:390-564) and multiclass_nms
 * (mmdet/core/post_processing/bbox_nms.py:6-67).                             */

/* iou_aware_retina_head.py:502-531,539: per anchor max over classes of
 * sqrt(sigmoid(cls)) * sqrt(sigmoid(iou)).  rowmax: (B, N) fp32; inside an image
 * each level is an (A, H*W) block (anchor-major), i.e. element a*HW + p holds the
 * reference's anchor index p*A + a (IA_LAYOUT_NCHW), or the reference's own order
 * p*A + a (IA_LAYOUT_NHWC).  ia_select_topk reads whichever g->layout implies.   */
int ia_decode_fuse_rowmax(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                          float *rowmax, void *stream);

/* iou_aware_retina_head.py:536-544 (topk per level, descending; ties broken by
 * ascending anchor index).  cand_idx: (B, R) int32 level-local anchor index.  */
size_t ia_select_topk_workspace_bytes(const ia_head_geom *g, int batch);
int ia_select_topk(const ia_head_geom *g, const float *rowmax, int batch, int32_t *cand_idx,
                   void *workspace, size_t workspace_bytes, void *stream);

/* The same two stages the way ia_get_bboxes chains them: the row-max kernel also leaves, inside
 * the top-k workspace (ia_select_topk_workspace_bytes), the maxima of groups of 64 / 16 / 4
 * consecutive scores of the large levels and the cleared candidate counters, which is where
 * ia_select_topk_grouped starts (ia_select_topk derives them from `rowmax` with an extra pass).
 * Both calls take the SAME workspace, back to back on one stream.                              */
int ia_decode_fuse_rowmax_grouped(const ia_head_geom *g, const ia_level_ptrs *p, int batch,
                                  int dtype, float *rowmax, void *select_workspace,
                                  size_t workspace_bytes, void *stream);
int ia_select_topk_grouped(const ia_head_geom *g, const float *rowmax, int batch,
                           int32_t *cand_idx, void *select_workspace, size_t workspace_bytes,
                           void *stream);

/* iou_aware_retina_head.py:545-558 + mmdet/core/bbox/transforms.py:44-78
 * (delta2bbox) + anchor_generator.py:53-70 (anchors regenerated, never read).
 * img_hw: (B,2) fp32 (img_shape h,w); scale_factor: (B,4) fp32.
 * boxes: (B,R,4) fp32; scores_t: (B,C,Rs) fp32 class-major fused scores;
 * best_score: (B,R) fp32 max over classes per candidate (optional, may be NULL).
 * Like the row-max and top-k entries: any R (IA_MAX_CANDIDATES is the capacity of the batched NMS
 * behind ia_get_bboxes / ia_multiclass_nms, not of this stage).                      */
int ia_gather_decode(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                     const int32_t *cand_idx, const float *img_hw, const float *scale_factor,
                     int rescale, float *boxes, float *scores_t, float *best_score, void *stream);

/* bbox_nms.py:33-56 + mmdet/ops/nms/src/nms_cpu.cpp:4-59 (greedy, ">=",
 * ascending kept indices), all (image, class) problems in one launch, then the
 * per-image top max_per_img.  Outputs: dets (B,max_per_img,5) fp32, labels and
 * rows (B,max_per_img) int32 (row = candidate row id), num (B) int32,
 * keep_count (B,C) int32, keep_rows (B,C,Rs) int32 ascending.  best_score (B,R)
 * optional: rows with best_score <= score_thr are skipped when the per-image
 * suppression bit matrix (the workspace) is built.                             */
size_t ia_multiclass_nms_workspace_bytes(int batch, int R, int C);
int ia_multiclass_nms(const float *boxes, const float *scores_t, const float *best_score,
                      int batch, int R, int C, float score_thr, float iou_thr, int max_per_img,
                      void *workspace, size_t workspace_bytes, float *dets, int32_t *labels,
                      int32_t *rows, int32_t *num, int32_t *keep_count, int32_t *keep_rows,
                      void *stream);

/* multiclass_nms with test_cfg.nms.type = 'soft_nms' (bbox_nms.py:29-56 ->
 * mmdet/ops/nms/nms_wrapper.py:52-78 -> src/soft_nms_cpu.pyx:22-127): per class the
 * boxes with score > score_thr go through soft-NMS in candidate order; survivors keep
 * their DECAYED score; then the per-image top max_per_img.  method: IA_SOFT_LINEAR /
 * IA_SOFT_GAUSSIAN (anything else: hard suppression, as in the .pyx).  Outputs as
 * ia_multiclass_nms, except keep_rows lists each class's survivors in selection order
 * and dets[...,4] are decayed scores.  workspace additionally holds the (B,C,Rs)
 * decayed scores.                                                              */
#define IA_SOFT_LINEAR 1
#define IA_SOFT_GAUSSIAN 2
size_t ia_multiclass_soft_nms_workspace_bytes(int batch, int R, int C);
int ia_multiclass_soft_nms(const float *boxes, const float *scores_t, int batch, int R, int C,
                           float score_thr, float iou_thr, int method, float sigma,
                           float min_score, int max_per_img, void *workspace,
                           size_t workspace_bytes, float *dets, int32_t *labels, int32_t *rows,
                           int32_t *num, int32_t *keep_count, int32_t *keep_rows, void *stream);

/* Whole path in one call (what the Python head calls).
 * WORKSPACE CONTRACT of ia_get_bboxes / ia_get_bboxes_lazy / ia_decode_stage: the workspace holds a
 * few words of state (arrival counters of the fused row-max + top-k-filter launch).  Zero-fill it
 * ONCE before its first use; every call leaves it ready for the next one.  Do not let anything else
 * write to it between calls (if something did, zero it again; the entry points do so themselves
 * when they fail between the fused launch and the kernel that resets the state).
 * Bounded waits: a filter workgroup of the fused launch that gives up waiting for the row-max
 * wavefronts (GPU shared with another process, CU masking, a debugger) records the call in the
 * status words, and the next kernel of the same call selects from the complete row maxima instead
 * of the candidate lists -- the detections are the same, the call is slower.
 * ia_get_bboxes_status_offset() = byte offset of two uint32 words: [0] the id of the last call
 * that took this fallback (0: never), [1] the number of such calls.  ia_debug_fused_spin_limit()
 * sets the bound of the waits (tests: 0 forces the fallback; negative: the default, 2^21 polls). */
size_t ia_get_bboxes_workspace_bytes(const ia_head_geom *g, int batch);
size_t ia_get_bboxes_status_offset(const ia_head_geom *g, int batch);
int ia_debug_fused_spin_limit(int64_t limit);
/* Profiling hook (bench.py): two caller-owned hipEvent_t (as void *) that every later
 * ia_get_bboxes / ia_get_bboxes_lazy / ia_decode_stage call records on ITS stream -- `begin` in
 * front of the decode stage's first launch (row-max), `end` behind its last one (gather / decode) --
 * so the stage can be timed inside real steps.  Both NULL (the default) switch it off; one NULL:
 * IA_E_ARG.  The pair belongs to ONE stream: the first stage call after installation binds the
 * hook to its stream and calls on other streams do not record (no cross-stream event records from
 * concurrent callers).  The caller keeps the events alive while they are set and clears the hook
 * (NULL, NULL) before releasing them. */
int ia_profile_stage_events(void *begin, void *end);
int ia_get_bboxes(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                  const float *img_hw, const float *scale_factor, int rescale, float score_thr,
                  float iou_thr, int max_per_img, void *workspace, size_t workspace_bytes,
                  float *dets, int32_t *labels, int32_t *rows, int32_t *num, void *stream);

/* The same result (dets / labels / rows / num) with the NMS evaluated lazily: the (class, box)
 * pairs of an image are walked in the final order (score desc, class asc, row asc) and a pair
 * is kept iff no already kept pair of its class suppresses it; max_per_img survivors end the
 * walk -- greedy per-class NMS decides a box from higher-ranked boxes only, so these are exactly
 * the detections multiclass_nms returns (bbox_nms.py:33-56), without resolving 80 complete class
 * problems.  candidates: pairs walked per image at most (0 = default: 1024, then 4096 for the
 * images that need more); an image that runs
 * out of them with fewer than max_per_img survivors takes the complete path.  The per-class
 * keep lists in the workspace are NOT produced by this entry point.                        */
int ia_get_bboxes_lazy(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                       const float *img_hw, const float *scale_factor, int rescale,
                       float score_thr, float iou_thr, int max_per_img, int candidates,
                       void *workspace, size_t workspace_bytes, float *dets, int32_t *labels,
                       int32_t *rows, int32_t *num, void *stream);

/* The NMS stage of ia_get_bboxes_lazy on its own (inputs as ia_multiclass_nms).             */
size_t ia_multiclass_nms_lazy_workspace_bytes(int batch, int R, int C);
int ia_multiclass_nms_lazy(const float *boxes, const float *scores_t, const float *best_score,
                           int batch, int R, int C, float score_thr, float iou_thr,
                           int max_per_img, int candidates, void *workspace,
                           size_t workspace_bytes, float *dets, int32_t *labels, int32_t *rows,
                           int32_t *num, void *stream);

/* The decode stage of ia_get_bboxes on its own -- row-max (+ group maxima), top-k, gather / decode
 * (iou_aware_retina_head.py:499-558; SURVEY 8(d)'s unit) -- in the ia_get_bboxes workspace, where
 * ia_get_bboxes_workspace_layout finds cand_idx, boxes, scores_t and best_score afterwards.   */
int ia_decode_stage(const ia_head_geom *g, const ia_level_ptrs *p, int batch, int dtype,
                    const float *img_hw, const float *scale_factor, int rescale, void *workspace,
                    size_t workspace_bytes, void *stream);

/* Workspace carve-up of ia_get_bboxes (host helper for stage-level tests):
 * byte offsets of rowmax, cand_idx, boxes, scores_t, keep_count, keep_rows,
 * best_score, NMS stage workspace (bit matrix, sorted rows, counts).          */
int ia_get_bboxes_workspace_layout(const ia_head_geom *g, int batch, size_t offsets[8]);

/* mmdet.ops.nms.nms (mmdet/ops/nms/nms_wrapper.py:8-49 -> nms_cpu.nms /
 * nms_cuda.nms): dets (n,5) fp32 on device, ANY n like the reference op (up to
 * IA_MAX_CANDIDATES boxes: one n x n suppression bit matrix; more: the same greedy NMS in
 * chunks of IA_MAX_CANDIDATES sorted boxes, csrc/bignms.hip).
 * keep (n) int32 ascending input indices, count (1) int32.                      */
size_t ia_nms_workspace_bytes(int n);
int ia_nms(const float *dets, int n, float iou_thr, int32_t *keep, int32_t *count, void *workspace,
           size_t workspace_bytes, void *stream);

/* The same operator on DOUBLE boxes: nms_cpu_kernel<double> (nms_cpu.cpp:63 dispatches float and
 * double; areas, intersections and the quotient in fp64, the float threshold promoted to double --
 * the fp32 entry is not a substitute: float(1/3) as threshold suppresses an IoU of exactly 1/3 in
 * fp32 and does not in fp64).  n <= 16384 (IA_E_LIMIT_BOXES above; csrc/nms64.hip).            */
size_t ia_nms_f64_workspace_bytes(int n);
int ia_nms_f64(const double *dets, int n, float iou_thr, int32_t *keep, int32_t *count,
               void *workspace, size_t workspace_bytes, void *stream);

/* mmdet.ops.nms.soft_nms (nms_wrapper.py:52-78 -> soft_nms_cpu.pyx:22-127): dets (n,5)
 * fp32 on device, n <= IA_MAX_CANDIDATES.  out_dets (n,5): surviving boxes in selection
 * order with decayed scores; out_inds (n) int32 their input indices; count (1) int32.
 * The reference runs this op on the host only (numpy); there is no reference CUDA path. */
int ia_soft_nms(const float *dets, int n, float iou_thr, int method, float sigma, float min_score,
                float *out_dets, int32_t *out_inds, int32_t *count, void *stream);

/* Image pre-processing in front of the backbone: ImageTransform.__call__
 * (mmdet/datasets/transforms.py:31-50): cv2 bilinear resize of the uint8 BGR image to
 * (dst_h, dst_w) (the caller computes the size like mmcv.imrescale / imresize), BGR->RGB,
 * (x - mean) / std, horizontal flip, zero padding to (pad_h, pad_w), HWC->CHW; one launch
 * per <= 16 images.  imgs: HOST array of descriptors whose src pointers are DEVICE uint8
 * (src_h, src_w, 3) images.  out: (B,3,pad_h,pad_w) fp32, or (B,pad_h,pad_w,3) when
 * channels_last (the memory layout of a channels-last (B,3,H,W) tensor).              */
typedef struct {
    const uint8_t *src;
    int32_t src_h, src_w, dst_h, dst_w, flip;
} ia_image_desc;
int ia_image_transform(const ia_image_desc *imgs, int batch, const float *mean, const float *std,
                       int to_rgb, int pad_h, int pad_w, int channels_last, float *out,
                       void *stream);

/* Winograd F(4x4,3x3) transforms around a library batched GEMM, for the 3x3 / stride-1 / pad-1
 * convolutions of the head towers and outputs (iou_aware_retina_head.py:171-219; replaces
 * ConvModule.forward, mmdet/models/utils/conv_module.py:149-163, and the three output
 * nn.Conv2d) and of the FPN outputs (mmdet/models/necks/fpn.py:124-127) at inference.
 * All pyramid levels of a batch form ONE tile list (tiles of 4x4 outputs, level-major, then
 * image, then row-major), so a shared-weight layer is one batched GEMM:
 *   V (groups*36, T, C/groups)  = input transform of channels-last activations (B,H_l,W_l,C)
 *   M[k] = V[k] . U[k]          (caller: rocBLAS / hipBLASLt; U = G g G^T, (36, Cin, Cout))
 *   output transform of M (groups*36, T, C/groups) + bias (+ ReLU) -> channel ranges
 *   ("segments") of channels-last destination tensors.                                   */
typedef struct {
    int32_t num_levels, batch;
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS];
} ia_wino_geom;
typedef struct {
    int32_t c0, n;                  /* output channels [c0, c0+n) of the GEMM result ...   */
    int32_t dst_channels, dst_offset;   /* ... go to channels [dst_offset, +n) of dst     */
    float *dst[IA_MAX_LEVELS];      /* per level (B, H_l, W_l, dst_channels)               */
} ia_wino_seg;
int ia_wino_tiles(const ia_wino_geom *g, int32_t *tiles);
/* pre_shift != NULL: the input is read as relu?(x * pre_scale[c] + pre_shift[c]) (pre_scale
 * NULL = 1): the folded BatchNorm + ReLU of the producing 1x1 convolution, applied on load.  */
int ia_wino_input_transform(const ia_wino_geom *g, const float *const *x, int channels, int groups,
                            const float *pre_scale, const float *pre_shift, int pre_relu, float *V,
                            void *stream);
int ia_wino_output_transform(const ia_wino_geom *g, const float *M, int channels, int groups,
                             const float *bias, int relu, int nseg, const ia_wino_seg *segs,
                             void *stream);

/* Training: gradient of the output transform, dM = A dY A^T per tile (dY (B,H,W,channels)
 * channels-last per level, zero outside the map), written as 36 matrices (36, tiles, channels)
 * like V -- the right-hand operand of the Winograd-domain weight gradient
 * dU[k] = V[k]^T dM[k]  (then dW = G^T dU G on the host side).                          */
int ia_wino_grad_output_transform(const ia_wino_geom *g, const float *const *dy, int channels,
                                  float *dM, void *stream);

/* Training: the per-iteration weight transforms of the Winograd convolution nodes
 * (iouaware/winograd_train.py) and the fused ReLU-backward / bias-gradient pass of the
 * convolution nodes (iouaware/train_fuse.py); they stand where the reference's training step
 * runs torch's convolution / BatchNorm / ReLU backward (mmdet/models/backbones/resnet.py:215-255,
 * mmdet/models/anchor_heads/iou_aware_retina_head.py:171-219 under autograd).
 *   ia_wino_weight_transform: U (36, n_in, n_out), U[6a+b][i][o] = (G w(o,i) G^T)[a][b]; w is
 *     addressed by element strides (any memory format; swap stride_in / stride_out and set
 *     flip=1 for the input-gradient convolution = correlation with w^T rotated by 180 degrees).
 *     G: the 6x3 kernel-transform matrix of F(4x4,3x3), row-major doubles (host memory).
 *   ia_wino_weight_grad: dW (n_out, n_in, 3, 3), addressed by element strides like w above
 *     = G^T dU(.,i,o) G, the adjoint.
 *   ia_bn_fold_fwd / _bwd: eval-mode BatchNorm (`norm_eval=True`, resnet.py:520-527) folded into the
 *     convolution in front of it, differentiably: w_out[o][:] = w[o][:] * s, b_out = beta - mean * s,
 *     s = gamma * inv_std (inv_std = 1/sqrt(running_var + eps), a constant); backward gives dw,
 *     dgamma, dbeta from the gradients w.r.t. w_out / b_out.  Weights are (cout, K) rows, dense in
 *     memory (contiguous or channels-last alike; all four weight-shaped arrays share one layout).
 *   ia_relu_bwd_bias_grad: dy, y, g (rows, n) row-major fp32 (n % 4 == 0, 16-byte aligned):
 *     g = dy where y > 0 else 0 (y == NULL: no mask, nothing written), db[n] = column sums of
 *     the masked gradient (db == NULL: not computed).  The sums go through per-strip partial
 *     rows in the workspace and are added in a fixed order (same bits every run).            */
#define IA_COLSUM_MAX_STRIPS 512
size_t ia_relu_bwd_bias_grad_workspace_bytes(int64_t rows, int n);
int ia_wino_weight_transform(const float *w, int n_in, int n_out, int64_t stride_in,
                             int64_t stride_out, int64_t stride_ky, int64_t stride_kx, int flip,
                             const double *G, float *U, void *stream);
int ia_wino_weight_grad(const float *dU, int n_in, int n_out, const double *G, float *dW,
                        int64_t stride_in, int64_t stride_out, int64_t stride_ky, int64_t stride_kx,
                        void *stream);
int ia_bn_fold_fwd(const float *w, const float *gamma, const float *beta, const float *mean,
                   const float *inv_std, int cout, int K, float *w_out, float *b_out, void *stream);
int ia_bn_fold_bwd(const float *dw_folded, const float *db_folded, const float *w, const float *gamma,
                   const float *mean, const float *inv_std, int cout, int K, float *dw,
                   float *dgamma, float *dbeta, void *stream);
int ia_relu_bwd_bias_grad(const float *dy, const float *y, int64_t rows, int n, float *g, float *db,
                          void *workspace, size_t workspace_bytes, void *stream);

/* 1x1 convolution on a channels-last activation as one library GEMM (hipBLASLt) with the folded
 * BatchNorm bias, the residual and the ReLU in its epilogue (Bottleneck.forward,
 * mmdet/models/backbones/resnet.py:215-255, at inference):
 *   D (rows, n) = act(A (rows, k) . W (k, n) + bias[n] + residual (rows, n)),  rows = B*H*W
 * all row-major fp32; bias / residual optional; residual != D.  The library handle is created on
 * first use; the first call of a shape times the heuristic's candidates on `stream`.         */
int ia_linear_bias_act(const float *A, const float *W, const float *bias, const float *residual,
                       float *D, int64_t rows, int k, int n, int relu, void *workspace,
                       size_t workspace_bytes, void *stream);
/* The same with bf16 operands (A, W, residual, D bf16; bias fp32; fp32 accumulation) for the
 * bf16 configuration (BASELINE config 3).                                                    */
int ia_linear_bias_act_bf16(const void *A, const void *W, const float *bias, const void *residual,
                            void *D, int64_t rows, int k, int n, int relu, void *workspace,
                            size_t workspace_bytes, void *stream);
/* The batched GEMM between the Winograd transforms, D[b] (rows, n) = A[b] (rows, k) . W[b] (k, n),
 * b < batch, contiguous row-major stacks; same library, same per-shape candidate timing.      */
int ia_batched_gemm(const float *A, const float *W, float *D, int batch, int64_t rows, int k, int n,
                    void *workspace, size_t workspace_bytes, void *stream);
/* The same product for the HBM-bound shapes -- (k, n) in {(64, 64), (128, 128), (256, 48)}: the
 * Winograd-domain GEMMs of the 64- / 128-channel bottleneck convolutions and of the reg | iou
 * output -- on this library's streaming MFMA kernel (csrc/conv1x1_stream.hip: one grid row per
 * matrix, its weights in LDS); no workspace.                                                    */
int ia_batched_gemm_stream(const float *A, const float *W, float *D, int batch, int64_t rows, int k,
                           int n, void *stream);

/* Convolutions with a stride as plain contractions with a FIXED reduction order (the library
 * convolution's fast fp32 channels-last kernels for these shapes split the reduction and add the
 * partial sums with atomics: other bits in every run).
 *   ia_im2col3x3_nhwc  3x3 / pad 1 / stride s (1..4) taps of a channels-last (B, H, W, C) tensor as
 *     the row-major matrix col[(b, yo, xo)][tap * C + c] (zeros outside the image), tap = dy * 3 + dx,
 *     Ho = (H - 1) / s + 1; C * sizeof(dtype) must be a multiple of 16.  ia_im2col3x3_bytes: its size.
 *     The convolution is then ia_linear_bias_act[_bf16] with rows = B * Ho * Wo, k = 9 * C and the
 *     weight stored (9 * C, Cout): folded BatchNorm / bias / ReLU in the GEMM epilogue.  Replaces
 *     conv2 (stride 2) of the first bottleneck of ResNet stages 2-4 (resnet.py:147-160) and the
 *     FPN's P6 / P7 (necks/fpn.py:84-99) at inference.
 *   ia_conv1x1_strided  1x1 / stride s on a channels-last (B, H, W, k) tensor -> (B, Ho, Wo, n):
 *     D = act(x[:, ::s, ::s, :] . W_kn + bias + residual) as one strided-batched library GEMM that
 *     reads the input in place (leading dimension s * k; one batch item per output row); the
 *     projection shortcut of those blocks (resnet.py:436-449).  dtype IA_F32 / IA_BF16.          */
size_t ia_im2col3x3_bytes(int B, int H, int W, int C, int stride, int dtype);
int ia_im2col3x3_nhwc(const void *x, void *col, int B, int H, int W, int C, int stride, int dtype,
                      void *stream);
int ia_conv1x1_strided(const void *x, const void *W_kn, const float *bias, const void *residual,
                       void *D, int B, int H, int W, int k, int n, int stride, int relu, int dtype,
                       void *workspace, size_t workspace_bytes, void *stream);

/* How the library kernel of a new GEMM shape is chosen.
 *   2  FROZEN (default): the entry of the tuning table below, else the library heuristic's first
 *      result.  Nothing is timed at run time, so a process computes the same bits in every run
 *      (pick-by-timing made the arithmetic depend on the winner of a race).
 *   0 / 1  OFFLINE tuning (tools/tune_gemm.py writes the table with them): the first call of a
 *      shape times the heuristic's top 16 / every kernel of the library that supports the problem
 *      (~250 for fp32, ~0.3 s per shape; +2 % img/s on the R-50 inference step).
 * The environment variable IA_GEMM_TUNE = heuristic | all selects 0 / 1 at the first GEMM.  Shapes
 * already resolved keep their kernel.  Returns the previous mode; any other argument only queries. */
int ia_gemm_tuning(int mode);

/* The tuning table: (m, n, k, flags, batch, dtype) in the column-major terms of csrc/gemm.hip ->
 * the library's solution index (hipblaslt_ext::getIndexFromAlgo).  The package ships the table
 * measured on MI355X (iouaware/tuning/hipblaslt_gfx950.json, loaded by iouaware.ops at the first
 * GEMM together with the library version it was measured with; a table from another library
 * version is ignored, a solution the library no longer offers for the problem counts as stale
 * and the heuristic's first result is used).
 *   ia_gemm_table_add    one entry (a shape already resolved is resolved again);
 *   ia_gemm_table_clear  drops the table AND every resolved shape;
 *   ia_gemm_table_dump   the solution index in use for every shape resolved so far, rows of
 *                        7 x int64 {m, n, k, flags, batch, dtype, index}; returns the number of
 *                        shapes (rows7 may be NULL to query);
 *   ia_gemm_table_stats  {shapes served by the table, by the heuristic, stale entries};
 *   ia_gemm_library_version  hipblasLtGetVersion of the library mapped into this process.      */
int ia_gemm_table_add(int64_t m, int64_t n, int64_t k, int flags, int batch, int dtype,
                      int solution_index);
int ia_gemm_table_clear(void);
int ia_gemm_table_dump(int64_t *rows7, int capacity);
int ia_gemm_table_stats(int64_t *hits_misses_stale);
int ia_gemm_library_version(void);

/* Training forms of the same library GEMM (convolution autograd nodes, iouaware/train_fuse.py):
 *   ia_linear_bias_act_wt: as ia_linear_bias_act with the weight stored (n, k) row-major -- the
 *     (Cout, Cin) convolution weight as it is, read through the library's transpose flag;
 *   ia_gemm_tn: D[b] (n, k) = G[b]^T (n, rows) . X[b] (rows, k), b < batch: the weight gradient of
 *     a 1x1 convolution (G = output gradient, X = input) and, batch = 36, of the Winograd-domain
 *     product; a reduction over `rows` with a small result, the timed candidates include the
 *     library's split-K kernels (give it a workspace).                                          */
int ia_linear_bias_act_wt(const float *A, const float *W_nk, const float *bias,
                          const float *residual, float *D, int64_t rows, int k, int n, int relu,
                          void *workspace, size_t workspace_bytes, void *stream);
int ia_gemm_tn(const float *G, const float *X, float *D, int batch, int64_t rows, int n, int k,
               void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------- training
 * Per-level losses of IoUawareRetinaHead.loss_single (:221-313), computed on
 * the NCHW head outputs directly.  Each *_fwd ADDS its fp64 partial sums into
 * loss_sum[IA_LOSS_SLOTS] (device, zeroed by the caller; one fp64 atomic per
 * workgroup, spread over the slots to avoid contention); the loss numerator the
 * reference divides by avg_factor (losses.py:301-303,411,480) is the sum of the slots;
 * each *_bwd writes s * d(sum)/d(input) with the input's NCHW shape, where
 * s = gscale (host) * (gscale_dev ? *gscale_dev : 1): the upstream gradient
 * may stay on the device, so backward never synchronises with the host.       */

/* FocalLoss -> weighted_sigmoid_focal_loss -> py_sigmoid_focal_loss
 * (mmdet/core/loss/losses.py:226-247,279-303) fused with expand_binary_labels
 * (mmdet/core/anchor/anchor_target.py:247-254).  labels (B*N_l) int64,
 * label_weights (B*N_l) fp32 in anchor order n = p*A + a.                     */
int ia_focal_loss_fwd(const void *cls, int dtype, const int64_t *labels,
                      const float *label_weights, int B, int A, int C, int HW, float gamma,
                      float alpha, double *loss_sum, void *stream);
int ia_focal_loss_bwd(const void *cls, int dtype, const int64_t *labels,
                      const float *label_weights, int B, int A, int C, int HW, float gamma,
                      float alpha, float gscale, const float *gscale_dev, float *grad_cls,
                      void *stream);

/* SmoothL1Loss -> weighted_smoothl1 (losses.py:385-411).  pred NCHW
 * (B,A*4,H,W); target / weight (B,N_l,4) row-major.                           */
int ia_smooth_l1_fwd(const void *pred, int dtype, const float *target, const float *weight,
                     int B, int A, int HW, float beta, double *loss_sum, void *stream);
int ia_smooth_l1_bwd(const void *pred, int dtype, const float *target, const float *weight,
                     int B, int A, int HW, float beta, float gscale, const float *gscale_dev,
                     float *grad_pred, void *stream);

/* The IoU-balanced variants (loss_cls.type='IOUbalancedSigmoidFocalLoss', losses.py:309-374;
 * loss_bbox.type='IoUbalancedSmoothL1Loss', losses.py:416-458; anchor_head.py:83-84).
 * anchor_iou (B*N_l): the IoU regression target of the level (ia_iou_bce_fwd's iou_target),
 * treated as a constant.
 * focal: the positive element of a positive anchor is weighted by iou^eta * normalizer,
 * normalizer = S1 / (S2 + 1e-6).  fwd ADDS into loss_sums3[3*IA_LOSS_SLOTS]: slots [0,64) the
 * t=0 elements, [64,128) S1 = sum_pos loss, [128,192) S2 = sum_pos loss * iou^eta; the loss is
 * S0 + normalizer*S2.  bwd reads the normalizer from a device scalar.
 * smooth-L1: weight * iou^delta.                                                          */
int ia_focal_loss_balanced_fwd(const void *cls, int dtype, const int64_t *labels,
                               const float *label_weights, const float *anchor_iou, int B, int A,
                               int C, int HW, float gamma, float alpha, float eta,
                               double *loss_sums3, void *stream);
int ia_focal_loss_balanced_bwd(const void *cls, int dtype, const int64_t *labels,
                               const float *label_weights, const float *anchor_iou, int B, int A,
                               int C, int HW, float gamma, float alpha, float eta,
                               const float *normalizer_dev, float gscale, const float *gscale_dev,
                               float *grad_cls, void *stream);
int ia_smooth_l1_balanced_fwd(const void *pred, int dtype, const float *target,
                              const float *weight, const float *anchor_iou, int B, int A, int HW,
                              float beta, float delta, double *loss_sum, void *stream);
int ia_smooth_l1_balanced_bwd(const void *pred, int dtype, const float *target,
                              const float *weight, const float *anchor_iou, int B, int A, int HW,
                              float beta, float delta, float gscale, const float *gscale_dev,
                              float *grad_pred, void *stream);

/* IoU target (delta2bbox x2 + aligned bbox_overlaps,
 * iou_aware_retina_head.py:256-259, mmdet/core/bbox/geometry.py:34-47) fused
 * with weighted_iou_regression_loss (losses.py:460-480).  `level` selects the
 * base anchors / H / W / stride of g.  iou_target (B*N_l, optional).
 * bwd: grad_iou_pred NCHW (B,A,H,W); grad_bbox_pred NCHW (B,A*4,H,W) is the
 * gradient THROUGH the IoU target (pass NULL for the detached variant).       */
int ia_iou_bce_fwd(const ia_head_geom *g, int level, const void *bbox_pred, const void *iou_pred,
                   int dtype, const float *bbox_targets, const float *bbox_weights, int B,
                   float *iou_target, double *loss_sum, void *stream);
int ia_iou_bce_bwd(const ia_head_geom *g, int level, const void *bbox_pred, const void *iou_pred,
                   int dtype, const float *bbox_targets, const float *bbox_weights, int B,
                   float gscale, const float *gscale_dev, float *grad_iou_pred,
                   float *grad_bbox_pred, void *stream);

/* Training targets for a whole batch: anchor_target -> anchor_target_single ->
 * MaxIoUAssigner.assign_wrt_overlaps + PseudoSampler + bbox2delta + unmap
 * (mmdet/core/anchor/anchor_target.py:129-242, assigners/max_iou_assigner.py:98-201,
 * bbox/geometry.py:48-64, bbox/transforms.py:6-41), gt_max_assign_all=True,
 * allowed_border < 0, no ignore regions.  gt_boxes (B,gmax,4) fp32 padded,
 * gt_labels (B,gmax) int64 or NULL (label 1), num_gt (B) >= 1, gmax <= 512;
 * valid_hw (B,L,2) int32 = valid feature rows / cols per level (from pad_shape,
 * anchor_head.py:135-146).  Outputs are level-major: for level l a (B, N_l[,4])
 * block at element offset B * anchor_off_l -- the per-level tensors of
 * images_to_levels.  counts (B,2): positives, negatives per image.
 * gt_max_scratch: (B,gmax) uint32 scratch.                                      */
int ia_anchor_targets(const ia_head_geom *g, const float *gt_boxes, const int64_t *gt_labels,
                      const int32_t *num_gt, int batch, int gmax, const int32_t *valid_hw,
                      float pos_iou_thr, float neg_iou_thr, float min_pos_iou, float pos_weight,
                      uint32_t *gt_max_scratch, int64_t *labels, float *label_weights,
                      float *bbox_targets, float *bbox_weights, int32_t *counts, void *stream);

/* The same assignment with the gt boxes where the data loader left them: HOST arrays of
 * per-image DEVICE pointers gt_boxes[b] (num_gt[b],4) fp32 / gt_labels[b] (num_gt[b]) int64 (or
 * gt_labels == NULL), host num_gt[b] in 1..512 and host valid_hw (B,L,2).  Pointers and sizes
 * travel in the kernel arguments: no padded staging tensors, no host-to-device copies.
 * batch <= IA_MAX_TARGET_BATCH.                                                    */
#define IA_MAX_TARGET_BATCH 16
int ia_anchor_targets_ptrs(const ia_head_geom *g, const float *const *gt_boxes,
                           const int64_t *const *gt_labels, const int32_t *num_gt, int batch,
                           const int32_t *valid_hw, float pos_iou_thr, float neg_iou_thr,
                           float min_pos_iou, float pos_weight, uint32_t *gt_max_scratch,
                           int64_t *labels, float *label_weights, float *bbox_targets,
                           float *bbox_weights, int32_t *counts, void *stream);

/* ------------------------------------------------------------------ IoUawareRetinaHead.loss, all levels
 * One call for the forward and one for the backward of the three losses of every pyramid level
 * (iou_aware_retina_head.py:221-313 x L, :315-387): FocalLoss(gamma = 2) on the class logits,
 * SmoothL1Loss on the deltas, IoU target + BCE on the IoU logits; 4 + 2 kernel launches.
 * Head outputs NCHW (g->layout == IA_LAYOUT_NCHW), per-level targets as ia_anchor_targets
 * writes them.                                                                       */
typedef struct ia_head_targets {
    const int64_t *labels[IA_MAX_LEVELS];        /* (B, N_l) int64 in 0..C                 */
    const float *label_weights[IA_MAX_LEVELS];   /* (B, N_l)                               */
    const float *bbox_targets[IA_MAX_LEVELS];    /* (B, N_l, 4)                            */
    const float *bbox_weights[IA_MAX_LEVELS];    /* (B, N_l, 4)                            */
    /* avg_factor (num_total_samples), first non-NULL / positive of:                       */
    const int32_t *counts;                       /* (B,2) of ia_anchor_targets: sum_b max(pos_b,1) */
    const float *avg_factor_dev;                 /* device scalar                          */
    float avg_factor;                            /* host value                             */
} ia_head_targets;

typedef struct ia_head_loss_cfg {
    float gamma, alpha, loss_weight_cls;         /* FocalLoss; gamma must be 2             */
    float beta, loss_weight_bbox;                /* SmoothL1Loss                           */
    int32_t attach_iou_target;                   /* gradient through the IoU target (reference: yes) */
    int32_t exact_large_logits;                  /* class logits > 60: the loss VALUE of such a negative
                                                    element saturates at 60 unless this is set (one more
                                                    pass over the logits); gradients are exact either way */
    int32_t grad_rows_start_at_reg;              /* ia_head_loss_bwd_nhwc only.  The caller STATES that in every
                                                    level grads->reg[l] is channel 0 of its pixel row of
                                                    grad_strides->reg[l] floats and grads->iou[l] == grads->reg[l]
                                                    + 4 A in the same row; the backward then also writes the zero
                                                    gradient of the row's remaining stride - 5 A (<= 64) channels.
                                                    0 (e.g. a row laid out [X | reg | iou | pad]): only the reg /
                                                    iou slices are written, the caller clears the rest.  Nothing is
                                                    inferred from the pointers alone.  Zero-initialise
                                                    this struct: it has grown since round 4.               */
} ia_head_loss_cfg;

/* workspace (256-byte aligned, ia_head_loss_workspace_bytes): the fp64 partial sums and an
 * anchor-major copy of labels / label_weights that the forward call writes and the backward
 * call reads -- keep it alive and untouched between the two.
 * result: (3L + 4) fp32 = loss_cls[L] | loss_bbox[L] | losses_iou[L] | their sums over the
 * levels [3] | avg_factor -- each loss = loss_weight * (sum / avg_factor).              */
size_t ia_head_loss_workspace_bytes(const ia_head_geom *g, int batch);
int ia_head_loss_fwd(const ia_head_geom *g, const ia_level_ptrs *p, int dtype, int batch,
                     const ia_head_targets *t, const ia_head_loss_cfg *cfg, void *workspace,
                     size_t workspace_bytes, float *result, void *stream);
/* grad_result: (3L + 3) fp32 upstream gradients in result's order (per-level entries and the
 * three totals add up); grads: fp32 NCHW tensors shaped like the head outputs.          */
int ia_head_loss_bwd(const ia_head_geom *g, const ia_level_ptrs *p, int dtype, int batch,
                     const ia_head_targets *t, const ia_head_loss_cfg *cfg, const void *workspace,
                     const float *result, const float *grad_result, const ia_level_ptrs *grads,
                     void *stream);

/* The same losses on CHANNELS-LAST fp32 head outputs (what the training head of
 * iouaware/winograd_train.py produces): element (b, p, a, c) of a class map at
 * (b*HW + p) * pix_stride + a*C + c -- the reference's flattened order
 * cls_score.permute(0, 2, 3, 1).reshape(-1, C) (iou_aware_retina_head.py:236-240) in place, so the
 * anchor-major targets index it directly (no packed copy; the workspace holds only the fp64 sums).
 * pix_stride (elements) may exceed the map's own channel count: reg / iou may be channel slices
 * of one wider tensor.  C % 4 == 0; cls / reg pointers and strides 16-byte aligned.  g->layout is
 * not consulted.  Gradients are written with their own pixel strides (e.g. into the slices of one
 * tensor shaped like the wider one).  With cfg->grad_rows_start_at_reg set -- reg is channel 0 of the
 * gradient's pixel row, iou right behind it (iou == reg + 4 A, equal strides; IA_E_ARG when the pointers
 * contradict the flag) -- the backward also writes the zero gradient of the row's remaining channels
 * (stride - 5 A of them, alignment padding) and the caller need not clear them; without the flag it writes
 * the reg / iou slices and nothing else. */
typedef struct ia_level_pix_strides {
    int64_t cls[IA_MAX_LEVELS], reg[IA_MAX_LEVELS], iou[IA_MAX_LEVELS];
} ia_level_pix_strides;
int ia_head_loss_fwd_nhwc(const ia_head_geom *g, const ia_level_ptrs *p,
                          const ia_level_pix_strides *strides, int batch, const ia_head_targets *t,
                          const ia_head_loss_cfg *cfg, void *workspace, size_t workspace_bytes,
                          float *result, void *stream);
int ia_head_loss_bwd_nhwc(const ia_head_geom *g, const ia_level_ptrs *p,
                          const ia_level_pix_strides *strides, int batch, const ia_head_targets *t,
                          const ia_head_loss_cfg *cfg, const float *result, const float *grad_result,
                          const ia_level_ptrs *grads, const ia_level_pix_strides *grad_strides,
                          void *stream);

/* ------------------------------------------------------------------ ResNeXt grouped 3x3 convolution
 * conv2 of the ResNeXt bottleneck (mmdet/models/backbones/resnext.py:12-91: groups = 32 / 64,
 * 4 / 8 / 16 / 32 channels per group), channels-last fp32, pad 1, stride 1 or 2, + per-channel
 * bias (the folded BatchNorm shift) + optional ReLU, on v_mfma_f32_16x16x4_f32.
 * ia_grouped_conv3x3_pack is HOST code: weight (C, C/groups, 3, 3) and optional per-output-channel
 * scale (folded BatchNorm scale) -> wpack, C/16 * 9 * 16 * 64 floats for <= 16 channels per
 * group, C/32 * 9 * 64 * 64 for 32 (host pointers); copy wpack to the device once.        */
int ia_grouped_conv3x3_pack(const float *weight, const float *scale, int channels, int groups,
                            float *wpack);
int ia_grouped_conv3x3_nhwc(const float *x, const float *wpack, const float *bias, float *y,
                            int batch, int H, int W, int channels, int groups, int stride, int relu,
                            void *stream);

/* mmdet.ops.sigmoid_focal_loss: sigmoid_focal_loss_cuda.forward / .backward
 * (mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:23-63,65-105;
 * binding sigmoid_focal_loss.cpp:17-43).  logits (N,C) fp32, targets (N) int64,
 * losses / d_losses / d_logits (N,C) fp32.                                    */
int ia_sigmoid_focal_loss_fwd(const float *logits, const int64_t *targets, int N, int C,
                              float gamma, float alpha, float *losses, void *stream);
int ia_sigmoid_focal_loss_bwd(const float *logits, const int64_t *targets, const float *d_losses,
                              int N, int C, float gamma, float alpha, float *d_logits,
                              void *stream);
/* The same for every storage type the reference op is instantiated for
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF: float, double, half -- sigmoid_focal_loss_cuda.cu:128,166)
 * plus bf16: dtype = IA_F32 / IA_F64 / IA_F16 / IA_BF16 is the type of logits, losses, d_losses and
 * d_logits.  Like the reference kernel (expf / powf / logf) the arithmetic is single precision
 * whatever the storage type; the result is rounded once at the store.                          */
int ia_sigmoid_focal_loss_fwd_dt(const void *logits, int dtype, const int64_t *targets, int N, int C,
                                 float gamma, float alpha, void *losses, void *stream);
int ia_sigmoid_focal_loss_bwd_dt(const void *logits, int dtype, const int64_t *targets,
                                 const void *d_losses, int N, int C, float gamma, float alpha,
                                 void *d_logits, void *stream);

/* ------------------------------------------------------------------ conv epilogues
 * In place on an NCHW tensor x (N, C, HW contiguous):
 *   x = act(x * scale[c] + shift[c] [+ residual * res_scale[c] + res_shift[c]]).
 * Any of scale / shift / res_scale / res_shift may be NULL (1 / 0).  Fuses the
 * eval-mode BatchNorm -> (+identity) -> ReLU chains of the reference Bottleneck
 * (mmdet/models/backbones/resnet.py:215-255) and ConvModule's bias -> ReLU
 * (mmdet/models/utils/conv_module.py:149-163) into one pass per convolution.   */
int ia_channel_affine_act(void *x, int dtype, const float *scale, const float *shift,
                          const void *residual, const float *res_scale, const float *res_shift,
                          int relu, int N, int C, int64_t HW, void *stream);
/* the same on a channels-last tensor: x is (N, H, W, C) in memory (NHW = N*H*W);
 * C must be a multiple of 4 (fp32) / 8 (bf16).                                 */
int ia_channel_affine_act_nhwc(void *x, int dtype, const float *scale, const float *shift,
                               const void *residual, const float *res_scale,
                               const float *res_shift, int relu, int64_t NHW, int C, void *stream);

/* (N, H*W, C) channels-last memory -> (N, C, H*W) NCHW memory (LDS-tiled transpose): lets the
 * convolutions run in MIOpen's NHWC kernels while the head kernels read NCHW.    */
int ia_nhwc_to_nchw(const void *src, void *dst, int dtype, int N, int C, int64_t HW, void *stream);

/* ResNet stem at inference (mmdet/models/backbones/resnet.py:506-512): folded BatchNorm + ReLU +
 * MaxPool2d(kernel 3, stride 2, padding 1) in one pass.  x (B,H,W,C) channels-last fp32, C % 4 == 0;
 * out (B, (H-1)/2+1, (W-1)/2+1, C).                                                          */
int ia_affine_relu_maxpool_nhwc(const float *x, const float *scale, const float *shift, int B, int H,
                                int W, int C, float *out, void *stream);
/* the same for dtype IA_F32 / IA_BF16 (C % 4 / C % 8 == 0): fp32 arithmetic, a bf16 result is
 * rounded once -- equal to eager's affine -> ReLU -> max-pool on bf16 values                      */
int ia_affine_relu_maxpool_nhwc_dt(const void *x, int dtype, const float *scale, const float *shift,
                                   int B, int H, int W, int C, void *out, void *stream);

/* FPN top-down step, in place (mmdet/models/necks/fpn.py:118-120): fine += nearest-x2(coarse);
 * channels-last fp32, H = 2*Hc, W = 2*Wc, C % 4 == 0.                                          */
int ia_upsample2x_add_nhwc(float *fine, const float *coarse, int B, int H, int W, int Hc, int Wc,
                           int C, void *stream);
/* dtype IA_F32 / IA_BF16 (C % 4 / C % 8 == 0); bf16: one rounding of the fp32 sum, like eager's add */
int ia_upsample2x_add_nhwc_dt(void *fine, const void *coarse, int dtype, int B, int H, int W, int Hc,
                              int Wc, int C, void *stream);

/* The ResNet stem convolution (resnet.py:403-414 `conv1`: 7x7 / stride 2 / pad 3, 3 -> 64, no bias;
 * forward :506-512) on a channels-last fp32 image batch as an implicit GEMM on fp32 MFMA
 * (csrc/stem.hip): x (B, H, W, 3), w_packed (148, 64) with row ky * 21 + kx * 3 + c = w[:, c, ky, kx]
 * and a zero row 147, y (B, Ho, Wo, 64) = the RAW convolution, Ho = (H - 1) / 2 + 1.  The folded
 * BatchNorm + ReLU + max-pool behind it: ia_affine_relu_maxpool_nhwc.                            */
int ia_stem_conv7x7s2(const float *x, const float *w_packed, float *y, int B, int H, int W,
                      void *stream);
/* The same convolution on bf16 (BASELINE config 3) on v_mfma_f32_16x16x16_bf16: x (B, H, W, 3) bf16,
 * y (B, Ho, Wo, 64) bf16 (fp32 accumulation, one rounding), w_packed = 11 x 4 x 64 x 4 bf16 in
 * fragment order: element [s][nb][lane][e] = w[nb * 16 + (lane & 15)][k = 16 s + 4 (lane >> 4) + e]
 * with k = ky * 24 + 1 + kx * 3 + c (zero at the other positions and for ky = 7).                                    */
int ia_stem_conv7x7s2_bf16(const void *x, const void *w_packed, void *y, int B, int H, int W,
                           void *stream);

/* 1x1 convolutions of ResNet stage 1 (resnet.py:215-255, folded BatchNorm) as a streaming MFMA
 * kernel with the weights resident in LDS (csrc/conv1x1_stream.hip): y = relu?(x . w + bias
 * (+ residual)), fp32, x (rows, k), w (k, n) row-major, (k, n) in {(64, 256), (256, 64), (64, 64)}. */
int ia_conv1x1_stream(const float *x, const float *w, const float *bias, const float *residual,
                      float *y, int64_t rows, int k, int n, int relu, void *stream);
/* The boundary between two stage-1 bottlenecks in one pass (resnet.py:215-255: conv3 + bn3 + add +
 * ReLU of a block, then conv1 + bn1 + ReLU of the next): y = relu(x . w + bias + residual) (rows, n)
 * is stored and, from the accumulators, h = relu(y . w2 + bias2) (rows, n2) -- the second product
 * does not read y back.  fp32, (k, n, n2) = (64, 256, 64), residual / biases may be NULL.      */
int ia_conv1x1_chain(const float *x, const float *w, const float *bias, const float *residual,
                     const float *w2, const float *bias2, float *y, float *h, int64_t rows, int k,
                     int n, int n2, void *stream);
/* The same streaming product for a wide output row: (k, n) = (128, 512), column blocks of 256
 * channels per workgroup (ResNet stage 2: conv3 + bn3 + identity + ReLU, resnet.py:215-255).   */
int ia_conv1x1_wide(const float *x, const float *w, const float *bias, const float *residual,
                    float *y, int64_t rows, int k, int n, int relu, void *stream);

/* ------------------------------------------------------------------ bf16 3x3 convolution
 * 3x3 / stride 1 / pad 1 convolution + bias (+ReLU) on bf16 channels-last tensors, fp32
 * accumulation: the tower / output / FPN ConvModules of BASELINE config 3 (reference
 * mmdet/models/utils/conv_module.py:149-163, iou_aware_retina_head.py:171-219) as an implicit GEMM
 * on v_mfma_f32_32x32x16_bf16 (csrc/conv3x3_bf16.hip).  One launch covers a list of feature maps
 * (the pyramid levels the head's weights are shared over) and up to two groups (the cls and the
 * reg tower: own inputs, weights, outputs).  cin % 32 == 0, cout even (per group).
 * ia_conv3x3_bf16_pack: weights (groups * cout, 3, 3, cin) bf16 (= a channels-last
 * (groups * cout, cin, 3, 3) tensor) -> the kernel's layout, ia_conv3x3_bf16_packed_bytes bytes,
 * once per model.  bias (groups * cout) fp32 or NULL.                                             */
typedef struct ia_conv3x3_desc {
    int32_t num_levels, batch, groups;
    int32_t cin, cout;                    /* channels per group                                   */
    int32_t x_stride, y_stride;           /* elements between pixels of x / y (>= cin / cout)     */
    int32_t H[IA_MAX_LEVELS], W[IA_MAX_LEVELS];
    const void *x[2][IA_MAX_LEVELS];      /* [group][level] -> (batch, H, W, x_stride) bf16, 16-byte aligned */
    void *y[2][IA_MAX_LEVELS];            /* [group][level] -> (batch, H, W, y_stride) bf16       */
} ia_conv3x3_desc;
size_t ia_conv3x3_bf16_packed_bytes(int cin, int cout, int groups);
int ia_conv3x3_bf16_pack(const void *w, int cin, int cout, int groups, void *wp, void *stream);
int ia_conv3x3_bf16_levels(const ia_conv3x3_desc *d, const void *wp, const float *bias, int relu, void *stream);

/* ------------------------------------------------------------------ self-test
 * Elementwise fp32 math used by the kernels, exposed so tests can pin the
 * device implementation bit-for-bit: op 0 exp, 1 log, 2 sigmoid, 3 sqrt,
 * 4 x/y (y = second input), 5 sqrt(sigmoid(x)).                               */
int ia_test_math(int op, const float *x, const float *y, float *out, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IOUAWARE_H */
